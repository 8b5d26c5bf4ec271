/*
 * hmm-continuous-train-fs — the reference trainer's command line on the MI355X core.
 *
 * Same argv, list/.perfil/.hmm formats and report file as the reference's
 * main() (TF:101-391):
 *     word states_number param_number mix_number1..N input_file1..N output_file [initial_model]
 * The work of the EM loop (TF:238-358) is done by the C ABI of include/ghmm.h on
 * the GPU: one ghmm_estep over all utterances per iteration (the reference
 * streams every file twice per iteration) and a device-side ghmm_mstep.
 * Messages go to stdout and failures exit(1), like the reference (TF:419-422).
 *
 * Differences, on purpose:
 *   - [initial_model] works (the reference reads argv[argc], a NULL, TF:218);
 *   - no MAX_* capacity limits (up to GHMM_MAX_STREAMS = 8 feature streams).
 *
 * Several GPUs: start one process per GPU with
 *     GHMM_WORLD=<ranks> GHMM_RANK=<0..ranks-1> GHMM_COMM_ID=<a path unique to the job>
 *     [GHMM_DEVICE=<gpu index, default = rank>]
 * and the same argv.  Every rank reads the list, takes its length-balanced share of the
 * utterances (ghmm_shard_balanced), runs the E-step on it, and the ranks sum the statistics
 * with ONE RCCL all-reduce per iteration (ghmm_stats_allreduce); every rank applies the same
 * M-step; rank 0 writes the model and the report.  (GHMM_WORLD=1 with GHMM_COMM_ID set runs
 * the same code path over a one-rank communicator.)
 */
#include "ghmm.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/times.h>
#include <time.h>

#define THRESHOLD 1.0e-3 /* TF:37 */

static void die(const char *what, int rc)
{
    const char *d = ghmm_last_error();
    printf("%s: %s \n", what, (d && *d) ? d : ghmm_strerror(rc));
    exit(1);
}

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

static void usage(void)
{
    puts("Usage: hmm_continuous_fs word states_number param_number mix_number1 ... mix_numberN  input_file1 ... input_fileN output_file [initial_model]");
    puts("word: word that will be represented by the model");
    puts("states_number: number of states");
    puts("param_number: number of parameters to train the model");
    puts("mix_number1: number of mixtures per state (parameter 1)");
    puts("mix_numberN: number of mixtures per state (parameter N)");
    puts("input_file1: name of file with names of files with parameters 1 ");
    puts("input_fileN: name of file with names of files with parameters N");
    puts("output_file: output file name");
    puts("initial_model: name of initial model, if there is one");
    exit(1);
}

/* the reference derives the report name with strtok(name, ".") + ".txt" (TF:205-207):
   leading dots are skipped, the name is cut at the next dot */
static void report_name(const char *model, char *out, size_t n)
{
    snprintf(out, n, "%s", model);
    size_t i = 0;
    while (out[i] == '.') i++;
    while (out[i] && out[i] != '.') i++;
    out[i] = 0;
    strncat(out, ".txt", n - strlen(out) - 1);
}

int main(int argc, char **argv)
{
    char t_start[64], t_end[64], t_cpu[64], text_file[4096];
    time_t now;
    time(&now);
    strftime(t_start, sizeof t_start, "%d-%h-%Y %X", localtime(&now));

    if (argc < 7) usage();
    const char *word = argv[1];
    int N = atoi(argv[2]);
    const int P = atoi(argv[3]);
    if (P < 1 || P > GHMM_MAX_STREAMS) {
        printf("param_number = %d: between 1 and %d feature streams are supported \n", P, GHMM_MAX_STREAMS);
        exit(1);
    }
    if (argc < 2 * P + 5) usage();
    int M[GHMM_MAX_STREAMS], D[GHMM_MAX_STREAMS];
    const char *list[GHMM_MAX_STREAMS];
    for (int p = 0; p < P; p++) {
        M[p] = atoi(argv[4 + p]);
        list[p] = argv[4 + P + p];
        D[p] = 0;
        if (M[p] <= 0) {
            printf("states_number and mix_number must be positive \n");
            exit(1);
        }
    }
    const char *output = argv[2 * P + 4];
    const char *initial = (argc == 2 * P + 6) ? argv[argc - 1] : NULL;
    if (N <= 0) {
        printf("states_number and mix_number must be positive \n");
        exit(1);
    }
    report_name(output, text_file, sizeof text_file);

    const int world = env_int("GHMM_WORLD", 1), rank = env_int("GHMM_RANK", 0);
    const char *comm_id = getenv("GHMM_COMM_ID");
    if (world < 1 || rank < 0 || rank >= world) {
        printf("GHMM_RANK=%d GHMM_WORLD=%d: bad rank layout \n", rank, world);
        exit(1);
    }
    if (world > 1 && !(comm_id && *comm_id)) {
        printf("GHMM_WORLD=%d needs GHMM_COMM_ID=<path> for the rendezvous \n", world);
        exit(1);
    }
    const int device = env_int("GHMM_DEVICE", world > 1 ? rank : 0);

    /* the list files: one path per utterance and stream (TF:482-502) */
    char path[4096];
    char **files[GHMM_MAX_STREAMS];
    int n_files = 0, rc;
    for (int p = 0; p < P; p++) {
        FILE *fl = fopen(list[p], "r");
        if (!fl) {
            printf("file %s not found \n", list[p]);
            exit(1);
        }
        int n = 0, cap_f = 0;
        files[p] = NULL;
        while (fscanf(fl, "%4095s", path) == 1) {
            if (n == cap_f) {
                cap_f = cap_f ? cap_f * 2 : 64;
                files[p] = (char **)realloc(files[p], (size_t)cap_f * sizeof(char *));
                if (!files[p]) die("memory", GHMM_ERR_ALLOC);
            }
            files[p][n] = strdup(path);
            if (!files[p][n++]) die("memory", GHMM_ERR_ALLOC);
        }
        fclose(fl);
        if (p == 0) n_files = n;
        if (n != n_files) {
            printf("%s lists %d files, %s lists %d \n", list[p], n, list[0], n_files);
            exit(1);
        }
    }
    if (n_files == 0) {
        printf("no training utterances in %s \n", list[0]);
        exit(1);
    }
    /* this rank's share: all files, or the length-balanced shard (lengths from the file sizes) */
    int32_t *mine = (int32_t *)malloc((size_t)n_files * sizeof(int32_t));
    int n_mine = n_files;
    if (!mine) die("memory", GHMM_ERR_ALLOC);
    if (world > 1) {
        int32_t *all_len = (int32_t *)malloc((size_t)n_files * sizeof(int32_t));
        if (!all_len) die("memory", GHMM_ERR_ALLOC);
        for (int k = 0; k < n_files; k++) {
            int d, T;
            if ((rc = ghmm_perfil_stat(files[0][k], &d, &T))) die("reading", rc);
            all_len[k] = T;
        }
        if ((rc = ghmm_shard_balanced(all_len, n_files, rank, world, mine, &n_mine))) die("sharding", rc);
        free(all_len);
        if (n_mine == 0) {
            printf("rank %d of %d has no utterances (%d in %s) \n", rank, world, n_files, list[0]);
            exit(1);
        }
    } else {
        for (int k = 0; k < n_files; k++) mine[k] = k;
    }
    /* every utterance is read once and stays resident */
    double *X[GHMM_MAX_STREAMS];
    int32_t *len = (int32_t *)malloc((size_t)n_mine * sizeof(int32_t));
    if (!len) die("memory", GHMM_ERR_ALLOC);
    size_t frames = 0;
    int n_utt = n_mine;
    for (int p = 0; p < P; p++) {
        size_t fr = 0, cap = 0;
        X[p] = NULL;
        for (int k = 0; k < n_mine; k++) {
            int d, T;
            double *x;
            snprintf(path, sizeof path, "%s", files[p][mine[k]]);
            if ((rc = ghmm_perfil_read(path, &d, &T, &x))) die("reading", rc);
            if (k == 0) D[p] = d;
            if (d != D[p]) {
                printf("file %s has %d coefficients per frame, expected %d \n", path, d, D[p]);
                exit(1);
            }
            if (p == 0) len[k] = T;
            if (T != len[k]) {
                printf("file %s has %d frames, parameter 1 of the same utterance has %d \n", path, T, len[k]);
                exit(1);
            }
            if (fr + (size_t)T > cap) {
                cap = (fr + (size_t)T) * 2;
                X[p] = (double *)realloc(X[p], cap * (size_t)D[p] * sizeof(double));
                if (!X[p]) die("memory", GHMM_ERR_ALLOC);
            }
            memcpy(X[p] + fr * (size_t)D[p], x, (size_t)T * (size_t)D[p] * sizeof(double));
            ghmm_free(x);
            fr += (size_t)T;
        }
        frames = fr;
        for (int k = 0; k < n_files; k++) free(files[p][k]);
        free(files[p]);
    }
    free(mine);

    ghmm_host_model hm[GHMM_MAX_STREAMS];
    memset(hm, 0, sizeof hm);
    ghmm_ctx *ctx;
    ghmm_model *model[GHMM_MAX_STREAMS];
    ghmm_corpus *corpus[GHMM_MAX_STREAMS];
    ghmm_stats *stats[GHMM_MAX_STREAMS];
    ghmm_comm *comm = NULL;
    if ((rc = ghmm_ctx_create(device, NULL, &ctx))) die("GPU context", rc);
    if (comm_id && *comm_id && (rc = ghmm_comm_create_file(ctx, comm_id, rank, world, 300.0, &comm)))
        die("communicator", rc);
    for (int p = 0; p < P; p++)
        if ((rc = ghmm_corpus_create(ctx, X[p], len, n_utt, D[p], &corpus[p]))) die("corpus", rc);
    if (initial) {
        int Pf = 0;
        if ((rc = ghmm_hmm_read_streams(initial, hm, P, &Pf))) die("initial model", rc);
        if (Pf != P) {
            printf("initial model %s has %d parameters, the command line has %d \n", initial, Pf, P);
            exit(1);
        }
        N = hm[0].N;
        for (int p = 0; p < P; p++) {
            if (hm[p].D != D[p]) {
                printf("initial model %s has %d coefficients, data has %d \n", initial, hm[p].D, D[p]);
                exit(1);
            }
            M[p] = hm[p].M;
            if ((rc = ghmm_model_create(ctx, N, M[p], D[p], &model[p]))) die("model", rc);
            if ((rc = ghmm_model_set(ctx, model[p], hm[p].A, hm[p].c, hm[p].mean, hm[p].inv_var, hm[p].det)))
                die("model", rc);
        }
    } else {
        /* creating_initial_model (TF:226; init_mix_param once per stream, TF:814): on the GPU from the
           resident corpus, or with GHMM_HOST_INIT=1 by the host implementation (bit-exact with the
           reference) */
        const char *hi = getenv("GHMM_HOST_INIT");
        for (int p = 0; p < P; p++) {
            if ((rc = ghmm_model_create(ctx, N, M[p], D[p], &model[p]))) die("model", rc);
            if (hi && *hi == '1') {
                if (world > 1) {
                    printf("GHMM_HOST_INIT=1 needs the whole corpus on one rank \n");
                    exit(1);
                }
                if ((rc = ghmm_init_model(X[p], len, n_utt, N, M[p], D[p], &hm[p]))) die("creating initial model", rc);
                if ((rc = ghmm_model_set(ctx, model[p], hm[p].A, hm[p].c, hm[p].mean, hm[p].inv_var, hm[p].det)))
                    die("model", rc);
            } else {
                if ((rc = ghmm_host_model_alloc(&hm[p], N, M[p], D[p]))) die("memory", rc);
                if ((rc = ghmm_model_init_comm(ctx, model[p], corpus[p], comm))) die("creating initial model", rc);
            }
        }
    }
    for (int p = 0; p < P; p++) {
        snprintf(hm[p].word, sizeof hm[p].word, "%s", word);
        if ((rc = ghmm_stats_create(ctx, N, M[p], D[p], &stats[p]))) die("statistics", rc);
    }
    double lp[2] = {0.0, 0.0}; /* probab and exemplar_number of the pass (TF:318-320) */

    printf("\r\nCreating HMM using Forward-Backward algorithm (Baum-Welch)");
    double probab, old_probab = 1.0, variation; /* TF:151 */
    int iteration = 0;
    do {
        iteration++;
        printf("\r\nStarting training sequence (%d utterances, %zu frames)", n_utt, frames);
        if ((rc = ghmm_estep_streams(ctx, model, corpus, P, stats))) die("E-step", rc);
        /* the one exchange of the iteration: sum of the accumulators over ranks (one vector per stream) */
        for (int p = 0; comm && p < P; p++)
            if ((rc = ghmm_stats_allreduce(ctx, stats[p], comm))) die("all-reduce", rc);
        /* the stopping rule reads two numbers: 16 bytes come back, not the whole vector */
        if ((rc = ghmm_stats_loglik(ctx, stats[0], lp))) die("E-step", rc);
        probab = lp[0];
        printf("\r\nEnding training sequence");
        variation = fabs((old_probab - probab) / old_probab);
        printf("\r\nVerifying Probability: %f > Threshold: %f", variation, THRESHOLD);
        if (variation > THRESHOLD) {
            /* the statistics of the converging pass are discarded (TF:328) */
            old_probab = probab;
            for (int p = 0; p < P; p++)
                if ((rc = ghmm_mstep(ctx, model[p], stats[p]))) die("M-step", rc);
        }
    } while (variation > THRESHOLD);
    printf("\r\nFinal Probability = %f\r\n\r\n", variation);
    if (comm) n_utt = (int)lp[1]; /* exemplars of all ranks (TF:320, summed) */
    probab /= (double)n_utt;

    for (int p = 0; p < P; p++)
        if ((rc = ghmm_model_get(ctx, model[p], hm[p].A, hm[p].c, hm[p].mean, hm[p].inv_var, hm[p].det)))
            die("model", rc);

    /* cpu time exactly as the reference formats it (TF:364-369; its /60 assumes 60 ticks/s) */
    struct tms tm_cpu;
    times(&tm_cpu);
    time_t cpu = (time_t)(int)(tm_cpu.tms_utime / 60.0);
    struct tm *g = gmtime(&cpu);
    g->tm_mday -= 1;
    strftime(t_cpu, sizeof t_cpu, "%d %X", g);
    time(&now);
    strftime(t_end, sizeof t_end, "%d-%h-%Y %X", localtime(&now));

    if (rank != 0) goto done; /* every rank holds the same model; rank 0 writes it */
    if ((rc = ghmm_hmm_write_streams(output, hm, P, 8))) die("writing model", rc);

    FILE *ft = fopen(text_file, "w");
    if (!ft) {
        printf("can't open file %s \n", text_file);
        exit(1);
    }
    /* report lines of writing_text, TF:2189-2259 */
    fprintf(ft, "Continuous HMM created using forward backward algorithm (diagonal covariance matrix). It is considered a final state.\n");
    fprintf(ft, "model file: %s \n", output);
    fprintf(ft, "word: %s \n", word);
    fprintf(ft, "number of states: %d \n", N);
    fprintf(ft, "number of parameters: %d \n", P);
    for (int p = 0; p < P; p++) fprintf(ft, "number of mixtures %d: %d \n", p + 1, M[p]);
    for (int p = 0; p < P; p++) fprintf(ft, "parameter %d: %s \n", p + 1, list[p]);
    fprintf(ft, "threshould to finish training: %f \n", THRESHOLD);
    fprintf(ft, "number of exemplars in training sequence: %d \n", n_utt);
    fprintf(ft, "mean probability: %f \n", probab);
    fprintf(ft, "number of iterations: %d \n", iteration);
    fprintf(ft, "starting time: %s \n", t_start);
    fprintf(ft, "ending time: %s \n", t_end);
    fprintf(ft, "cpu time: %s \n", t_cpu);
    if (ferror(ft) || fclose(ft) != 0) {
        printf("writing error on file %s \n", text_file);
        exit(1);
    }

done:
    ghmm_comm_destroy(comm);
    for (int p = 0; p < P; p++) {
        ghmm_stats_destroy(ctx, stats[p]);
        ghmm_corpus_destroy(ctx, corpus[p]);
        ghmm_model_destroy(ctx, model[p]);
        ghmm_host_model_free(&hm[p]);
        free(X[p]);
    }
    ghmm_ctx_destroy(ctx);
    free(len);
    return 0;
}
