/*
 * ghmm_init.c — initial model construction on the host.
 *
 * Same algorithm as the reference's creating_initial_model (TF:732-1317), written
 * for utterances that are already in memory (the reference re-reads every
 * feature file on each pass):
 *   - A: one-step left-to-right, uniform over the allowed band (TF:774-806)
 *   - every utterance is cut into N equal runs of frames, the first T%N runs one
 *     frame longer (TF:1005-1013); state i owns run i of every utterance
 *   - per state: the mean of its frames, then LBG splitting (x1.005 / x0.995,
 *     TF:1120-1160) up to M cells with three nearest-mean passes after each
 *     split (TF:1043-1093); empty cells are re-seeded from the cell with the
 *     largest distortion (TF:1236-1270)
 *   - per cell: variance around the cell mean, floored at 1e-5 (TF:883-905);
 *     det = product of variances, inverse variances stored (TF:907-911);
 *     weights = cell share of the state's frames, floored at 1e-5 and
 *     renormalised (TF:918-933)
 *
 * Host code by design: it runs once per training job; the per-iteration path
 * (E-step / M-step) is the HIP part.
 */
#include "ghmm.h"

#include <stdlib.h>
#include <string.h>

#define INIT_DELTA 1       /* TF:38 */
#define INIT_FLOOR 1.0e-5  /* TF:39 */
#define SPLIT_UP 1.005     /* TF:1138 */
#define SPLIT_DOWN 0.995
#define KMEANS_PASSES 3    /* TF:1043 */

/* nearest cell by squared Euclidean distance; strict '<' so ties keep the lowest
   cell, start value 1e20 as in TF:1179-1215 (a frame farther than that from
   every cell keeps the previous frame's cell) */
static double nearest(const double *x, const double *cells, int n_cells, int D, int *cell)
{
    double best = 1.0e20;
    for (int i = 0; i < n_cells; i++) {
        double dist = 0.0;
        for (int j = 0; j < D; j++) {
            double a = cells[(size_t)i * D + j] - x[j];
            dist += a * a;
        }
        if (dist < best) {
            best = dist;
            *cell = i;
        }
    }
    return best;
}

/* indices by decreasing key, adjacent-swap passes with strict '<' (TF:1289-1315) */
static void order_desc(const double *key, int *idx, int n)
{
    int done = 0;
    for (int i = 0; i < n; i++) idx[i] = i;
    while (!done) {
        done = 1;
        for (int i = 0; i < n - 1; i++) {
            int j = idx[i], k = idx[i + 1];
            if (key[j] < key[k]) {
                idx[i] = k;
                idx[i + 1] = j;
                done = 0;
            }
        }
    }
}

static void split_cell(double *cells, int from, int to, int D)
{
    for (int l = 0; l < D; l++) cells[(size_t)to * D + l] = cells[(size_t)from * D + l] * SPLIT_UP;
    for (int l = 0; l < D; l++) cells[(size_t)from * D + l] = cells[(size_t)from * D + l] * SPLIT_DOWN;
}

/* run [begin,end) of state k in an utterance of T frames */
static void run_of(int T, int N, int k, int *begin, int *end)
{
    int q = T / N, r = T % N;
    *begin = k * q + (k < r ? k : r);
    *end = *begin + q + (k < r ? 1 : 0);
}

int ghmm_init_model(const double *X, const int32_t *len, int n_utt, int N, int M, int D,
                    ghmm_host_model *hm)
{
    if (!X || !len || n_utt <= 0 || N <= 0 || M <= 0 || D <= 0 || !hm) return GHMM_ERR_ARG;
    int rc = ghmm_host_model_alloc(hm, N, M, D);
    if (rc) return rc;

    /* transition matrix */
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            double a;
            if (j > INIT_DELTA + i || j < i) a = 0.0;
            else if (INIT_DELTA + 1 > N - i) a = 1.0 / (double)(N - i);
            else a = 1.0 / (double)(INIT_DELTA + 1);
            hm->A[(size_t)i * N + j] = a;
        }

    size_t cellsz = (size_t)M * D;
    double *cells = (double *)calloc((size_t)N * cellsz, sizeof(double)); /* [N][M][D] */
    double *sum = (double *)calloc((size_t)N * cellsz, sizeof(double));
    double *dist = (double *)calloc((size_t)N * M, sizeof(double));
    int *count = (int *)calloc((size_t)N * M, sizeof(int));
    int *idx = (int *)calloc((size_t)M, sizeof(int));
    int *dur = (int *)calloc((size_t)N, sizeof(int));
    if (!cells || !sum || !dist || !count || !idx || !dur) {
        free(cells); free(sum); free(dist); free(count); free(idx); free(dur);
        ghmm_host_model_free(hm);
        return GHMM_ERR_ALLOC;
    }

    /* one cell per state: mean of the state's frames */
    {
        size_t f0 = 0;
        for (int u = 0; u < n_utt; u++) {
            for (int k = 0; k < N; k++) {
                int b, e;
                run_of(len[u], N, k, &b, &e);
                for (int j = b; j < e; j++) {
                    const double *x = X + (f0 + (size_t)j) * D;
                    for (int l = 0; l < D; l++) cells[(size_t)k * cellsz + l] += x[l];
                    count[k * M]++;
                }
            }
            f0 += (size_t)len[u];
        }
        for (int k = 0; k < N; k++)
            for (int l = 0; l < D; l++) cells[(size_t)k * cellsz + l] /= (double)count[k * M];
    }

    int n_cells = 1;
    while (n_cells < M) {
        /* split: double while 2n < M, otherwise split the M-n cells with the
           largest distortion */
        int next;
        for (int k = 0; k < N; k++) {
            double *ck = cells + (size_t)k * cellsz;
            if (2 * n_cells < M) {
                for (int i = 0; i < n_cells; i++) split_cell(ck, i, n_cells + i, D);
            } else {
                order_desc(dist + (size_t)k * M, idx, n_cells);
                for (int i = 0; i < M - n_cells; i++) split_cell(ck, idx[i], n_cells + i, D);
            }
        }
        next = (2 * n_cells < M) ? 2 * n_cells : M;
        n_cells = next;

        for (int pass = 0; pass < KMEANS_PASSES; pass++) {
            for (int k = 0; k < N; k++)
                for (int i = 0; i < n_cells; i++) {
                    count[k * M + i] = 0;
                    dist[(size_t)k * M + i] = 0.0;
                    memset(sum + (size_t)k * cellsz + (size_t)i * D, 0, sizeof(double) * (size_t)D);
                }
            size_t f0 = 0;
            int cell = 0; /* carried across frames like the reference's `index` */
            for (int u = 0; u < n_utt; u++) {
                for (int k = 0; k < N; k++) {
                    int b, e;
                    run_of(len[u], N, k, &b, &e);
                    for (int j = b; j < e; j++) {
                        const double *x = X + (f0 + (size_t)j) * D;
                        /* TF:1076 `distortion[k][index] += classifying(..,&index)` is
                           unsequenced in C; gcc calls first and indexes with the
                           new cell, which is what is done here */
                        double d = nearest(x, cells + (size_t)k * cellsz, n_cells, D, &cell);
                        dist[(size_t)k * M + cell] += d;
                        count[k * M + cell]++;
                        for (int l = 0; l < D; l++) sum[(size_t)k * cellsz + (size_t)cell * D + l] += x[l];
                    }
                }
                f0 += (size_t)len[u];
            }
            for (int k = 0; k < N; k++) {
                double *ck = cells + (size_t)k * cellsz;
                for (int j = 0; j < n_cells; j++)
                    for (int l = 0; l < D; l++)
                        ck[(size_t)j * D + l] = sum[(size_t)k * cellsz + (size_t)j * D + l] / (double)count[k * M + j];
                order_desc(dist + (size_t)k * M, idx, n_cells);
                int i = 0;
                for (int j = 0; j < n_cells; j++)
                    if (count[k * M + j] == 0) split_cell(ck, idx[i++], j, D);
            }
        }
    }

    /* per-cell variance and weight */
    memset(count, 0, sizeof(int) * (size_t)N * M);
    {
        size_t f0 = 0;
        int cell = 0;
        for (int u = 0; u < n_utt; u++) {
            for (int k = 0; k < N; k++) {
                int b, e;
                run_of(len[u], N, k, &b, &e);
                for (int j = b; j < e; j++) {
                    const double *x = X + (f0 + (size_t)j) * D;
                    nearest(x, cells + (size_t)k * cellsz, M, D, &cell);
                    size_t g = (size_t)k * M + cell;
                    for (int l = 0; l < D; l++) {
                        double a = x[l] - cells[g * D + l];
                        hm->inv_var[g * D + l] += a * a;
                    }
                    count[g]++;
                }
                dur[k] += e - b;
            }
            f0 += (size_t)len[u];
        }
    }
    for (size_t g = 0; g < (size_t)N * M; g++) {
        double det = 1.0;
        for (int l = 0; l < D; l++) {
            double v = hm->inv_var[g * D + l] / (double)count[g];
            if (v < INIT_FLOOR) v = INIT_FLOOR;
            hm->inv_var[g * D + l] = v;
        }
        for (int l = 0; l < D; l++) det *= hm->inv_var[g * D + l];
        hm->det[g] = det;
        for (int l = 0; l < D; l++) hm->inv_var[g * D + l] = 1.0 / hm->inv_var[g * D + l];
        for (int l = 0; l < D; l++) hm->mean[g * D + l] = cells[g * D + l];
    }
    for (int k = 0; k < N; k++) {
        double *c = hm->c + (size_t)k * M, s = 0.0;
        for (int j = 0; j < M; j++) c[j] = (double)count[k * M + j] / (double)dur[k];
        /* weights below the floor are raised to it, then renormalised (TF:1338-1359) */
        for (int j = 0; j < M; j++) {
            if (c[j] < INIT_FLOOR) c[j] = INIT_FLOOR;
            s += c[j];
        }
        for (int j = 0; j < M; j++) c[j] /= s;
    }

    free(cells); free(sum); free(dist); free(count); free(idx); free(dur);
    return GHMM_OK;
}
