/*
 * ghmm_io.c — host side of the drop-in boundary: the reference's on-disk formats
 * (SURVEY.md §2.1).  Plain C, no device code.
 *
 *   .perfil  int32 D, then T*D little-endian doubles; T implied by EOF
 *            (reader TF:527-581: one fread of D doubles per frame)
 *   .hmm     size_t len | word[len] | int N | int P | int M[P] | int D[P] |
 *            double A[N][N] | per stream, per state: double c[M], then per
 *            mixture: double mean[D], double det, double inv_var[D]
 *            (writer TF:2043-2146, readers TF:604-711 / RF:595-715)
 *
 * The shipped .hmm files were written by a 32-bit build (4-byte size_t), a
 * 64-bit build writes 8 bytes: the reader accepts both by checking which header
 * width makes the file length come out exactly.
 */
#include "ghmm.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static _Thread_local char g_err[512];

void ghmm_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

const char *ghmm_last_error(void) { return g_err; }

const char *ghmm_strerror(int code)
{
    switch (code) {
    case GHMM_OK: return "ok";
    case GHMM_ERR_ARG: return "bad argument";
    case GHMM_ERR_ALLOC: return "allocation failed";
    case GHMM_ERR_HIP: return "HIP runtime error";
    case GHMM_ERR_NODEVICE: return "no gfx950 device";
    case GHMM_ERR_UNSUPPORTED: return "unsupported configuration";
    case GHMM_ERR_IO: return "file i/o error";
    case GHMM_ERR_FORMAT: return "bad file format";
    default: return "unknown error";
    }
}

int ghmm_version(void) { return GHMM_VERSION; }

void ghmm_free(void *p) { free(p); }

static long file_size(FILE *f)
{
    long cur = ftell(f), end;
    if (fseek(f, 0, SEEK_END) != 0) return -1;
    end = ftell(f);
    fseek(f, cur, SEEK_SET);
    return end;
}

int ghmm_perfil_read(const char *path, int *D, int *T, double **X)
{
    if (!path || !D || !T || !X) return GHMM_ERR_ARG;
    FILE *f = fopen(path, "rb");
    if (!f) {
        ghmm_set_error("file %s not found", path);
        return GHMM_ERR_IO;
    }
    int32_t d = 0;
    long size = file_size(f);
    if (size < 4 || fread(&d, sizeof d, 1, f) != 1 || d <= 0 || d > (1 << 20)) {
        fclose(f);
        ghmm_set_error("%s: not a .perfil file", path);
        return GHMM_ERR_FORMAT;
    }
    /* whole frames only; the reference would also consume a ragged tail
       (TF:537 returns the item count), which no writer produces */
    long frames = (size - 4) / (long)(sizeof(double) * (size_t)d);
    double *buf = (double *)malloc(sizeof(double) * (size_t)(frames > 0 ? frames : 1) * (size_t)d);
    if (!buf) {
        fclose(f);
        return GHMM_ERR_ALLOC;
    }
    size_t want = (size_t)frames * (size_t)d;
    if (fread(buf, sizeof(double), want, f) != want) {
        fclose(f);
        free(buf);
        ghmm_set_error("reading error on file %s", path);
        return GHMM_ERR_IO;
    }
    fclose(f);
    *D = d;
    *T = (int)frames;
    *X = buf;
    return GHMM_OK;
}

int ghmm_perfil_stat(const char *path, int *D, int *T)
{
    if (!path || !D || !T) return GHMM_ERR_ARG;
    FILE *f = fopen(path, "rb");
    if (!f) {
        ghmm_set_error("file %s not found", path);
        return GHMM_ERR_IO;
    }
    int32_t d = 0;
    long size = file_size(f);
    int bad = size < 4 || fread(&d, sizeof d, 1, f) != 1 || d <= 0 || d > (1 << 20);
    fclose(f);
    if (bad) {
        ghmm_set_error("%s: not a .perfil file", path);
        return GHMM_ERR_FORMAT;
    }
    *D = d;
    *T = (int)((size - 4) / (long)(sizeof(double) * (size_t)d));
    return GHMM_OK;
}

/* Length-balanced utterance shards: utterances ordered by decreasing
 * length (ties: lower index first), dealt to the ranks in turn; each rank's list is then
 * put back in index order so that a rank reads its files in list order. */
int ghmm_shard_balanced(const int32_t *len, int n_utt, int rank, int world, int32_t *index,
                        int *n_out)
{
    if (n_utt < 0 || (!len && n_utt > 0) || world <= 0 || rank < 0 || rank >= world || !index || !n_out)
        return GHMM_ERR_ARG;
    int32_t *ord = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_utt > 0 ? n_utt : 1));
    if (!ord) return GHMM_ERR_ALLOC;
    /* stable counting order over the distinct lengths would need their range: a plain
       merge sort on (-len, index) keeps it O(n log n) and stable */
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_utt > 0 ? n_utt : 1));
    if (!tmp) {
        free(ord);
        return GHMM_ERR_ALLOC;
    }
    for (int i = 0; i < n_utt; i++) ord[i] = i;
    for (int w = 1; w < n_utt; w *= 2) {
        for (int lo = 0; lo < n_utt; lo += 2 * w) {
            int mid = lo + w < n_utt ? lo + w : n_utt, hi = lo + 2 * w < n_utt ? lo + 2 * w : n_utt;
            int a = lo, b = mid, k = lo;
            while (a < mid && b < hi) tmp[k++] = len[ord[b]] > len[ord[a]] ? ord[b++] : ord[a++];
            while (a < mid) tmp[k++] = ord[a++];
            while (b < hi) tmp[k++] = ord[b++];
        }
        int32_t *sw = ord;
        ord = tmp;
        tmp = sw;
    }
    int n = 0;
    for (int k = rank; k < n_utt; k += world) index[n++] = ord[k];
    /* back into index order (insertion sort is quadratic: merge again) */
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int a = lo, b = mid, k = lo;
            while (a < mid && b < hi) tmp[k++] = index[b] < index[a] ? index[b++] : index[a++];
            while (a < mid) tmp[k++] = index[a++];
            while (b < hi) tmp[k++] = index[b++];
        }
        memcpy(index, tmp, sizeof(int32_t) * (size_t)n);
    }
    *n_out = n;
    free(ord);
    free(tmp);
    return GHMM_OK;
}

int ghmm_perfil_write(const char *path, int D, int T, const double *X)
{
    if (!path || D <= 0 || T < 0 || (!X && T > 0)) return GHMM_ERR_ARG;
    FILE *f = fopen(path, "wb");
    if (!f) {
        ghmm_set_error("can't open file %s", path);
        return GHMM_ERR_IO;
    }
    int32_t d = D;
    size_t n = (size_t)T * (size_t)D;
    int ok = fwrite(&d, sizeof d, 1, f) == 1 && fwrite(X, sizeof(double), n, f) == n;
    if (fclose(f) != 0) ok = 0;
    if (!ok) {
        ghmm_set_error("writing error on file %s", path);
        return GHMM_ERR_IO;
    }
    return GHMM_OK;
}

int ghmm_host_model_alloc(ghmm_host_model *hm, int N, int M, int D)
{
    if (!hm || N <= 0 || M <= 0 || D <= 0) return GHMM_ERR_ARG;
    size_t G = (size_t)N * M;
    hm->N = N; hm->M = M; hm->D = D;
    hm->A = (double *)calloc((size_t)N * N, sizeof(double));
    hm->c = (double *)calloc(G, sizeof(double));
    hm->mean = (double *)calloc(G * D, sizeof(double));
    hm->inv_var = (double *)calloc(G * D, sizeof(double));
    hm->det = (double *)calloc(G, sizeof(double));
    if (!hm->A || !hm->c || !hm->mean || !hm->inv_var || !hm->det) {
        ghmm_host_model_free(hm);
        return GHMM_ERR_ALLOC;
    }
    return GHMM_OK;
}

void ghmm_host_model_free(ghmm_host_model *hm)
{
    if (!hm) return;
    free(hm->A); free(hm->c); free(hm->mean); free(hm->inv_var); free(hm->det);
    hm->A = hm->c = hm->mean = hm->inv_var = hm->det = NULL;
}

/* try to parse with a `lb`-byte length prefix; 0 = fits the file exactly.  Up to `max_p` feature
   streams (the reference's param_number, TF:2084-2099): hm[p] = stream p, each with its own copy
   of the word, N and A; *P_out = the file's stream count. */
static int hmm_try(FILE *f, long size, int lb, ghmm_host_model *hm, int max_p, int *P_out, const char *path)
{
    unsigned char raw[8] = {0};
    rewind(f);
    if (fread(raw, 1, (size_t)lb, f) != (size_t)lb) return GHMM_ERR_FORMAT;
    uint64_t len = 0;
    for (int i = lb - 1; i >= 0; i--) len = (len << 8) | raw[i];
    if (len >= GHMM_MAX_WORD) return GHMM_ERR_FORMAT;
    char word[GHMM_MAX_WORD] = {0};
    int32_t N = 0, P = 0, M[GHMM_MAX_STREAMS], D[GHMM_MAX_STREAMS];
    if (fread(word, 1, (size_t)len, f) != (size_t)len) return GHMM_ERR_FORMAT;
    if (fread(&N, 4, 1, f) != 1 || fread(&P, 4, 1, f) != 1) return GHMM_ERR_FORMAT;
    if (N <= 0 || N > 65536 || P <= 0 || P > GHMM_MAX_STREAMS) return GHMM_ERR_FORMAT;
    if (fread(M, 4, (size_t)P, f) != (size_t)P || fread(D, 4, (size_t)P, f) != (size_t)P) return GHMM_ERR_FORMAT;
    long expect = lb + (long)len + 8 + 8L * P + 8L * (long)N * N;
    for (int p = 0; p < P; p++) {
        if (M[p] <= 0 || M[p] > 65536 || D[p] <= 0 || D[p] > 65536) return GHMM_ERR_FORMAT;
        expect += 8L * (long)N * ((long)M[p] + (long)M[p] * (2L * D[p] + 1));
    }
    if (expect != size) return GHMM_ERR_FORMAT;
    if (P > max_p) {
        ghmm_set_error("%s: %d feature streams, the caller takes %d", path, P, max_p);
        return GHMM_ERR_UNSUPPORTED;
    }
    int rc = GHMM_OK, ok = 1;
    for (int p = 0; p < P; p++) memset(&hm[p], 0, sizeof hm[p]);
    for (int p = 0; p < P && !rc; p++) {
        rc = ghmm_host_model_alloc(&hm[p], N, M[p], D[p]);
        if (!rc) memcpy(hm[p].word, word, GHMM_MAX_WORD);
    }
    if (!rc) {
        ok = fread(hm[0].A, 8, (size_t)N * N, f) == (size_t)N * N;
        for (int p = 1; p < P; p++) memcpy(hm[p].A, hm[0].A, sizeof(double) * (size_t)N * N);
        for (int p = 0; ok && p < P; p++)
            for (int i = 0; ok && i < N; i++) {
                ok = fread(hm[p].c + (size_t)i * M[p], 8, (size_t)M[p], f) == (size_t)M[p];
                for (int k = 0; ok && k < M[p]; k++) {
                    size_t g = (size_t)i * M[p] + k;
                    ok = fread(hm[p].mean + g * D[p], 8, (size_t)D[p], f) == (size_t)D[p] &&
                         fread(hm[p].det + g, 8, 1, f) == 1 &&
                         fread(hm[p].inv_var + g * D[p], 8, (size_t)D[p], f) == (size_t)D[p];
                }
            }
    }
    if (rc || !ok) {
        for (int p = 0; p < P; p++) ghmm_host_model_free(&hm[p]);
        return rc ? rc : GHMM_ERR_IO;
    }
    *P_out = P;
    return GHMM_OK;
}

int ghmm_hmm_read_streams(const char *path, ghmm_host_model *hm, int max_streams, int *n_streams)
{
    if (!path || !hm || !n_streams || max_streams <= 0) return GHMM_ERR_ARG;
    FILE *f = fopen(path, "rb");
    if (!f) {
        ghmm_set_error("file %s not found", path);
        return GHMM_ERR_IO;
    }
    long size = file_size(f);
    int rc = hmm_try(f, size, 8, hm, max_streams, n_streams, path);
    if (rc == GHMM_ERR_FORMAT) rc = hmm_try(f, size, 4, hm, max_streams, n_streams, path);
    fclose(f);
    if (rc == GHMM_ERR_FORMAT)
        ghmm_set_error("%s: not a diagonal-covariance .hmm file (4- or 8-byte header)", path);
    else if (rc == GHMM_ERR_IO)
        ghmm_set_error("reading error on file %s", path);
    return rc;
}

int ghmm_hmm_read(const char *path, ghmm_host_model *hm)
{
    if (!path || !hm) return GHMM_ERR_ARG;
    memset(hm, 0, sizeof *hm);
    int P = 0;
    int rc = ghmm_hmm_read_streams(path, hm, 1, &P);
    if (rc == GHMM_ERR_UNSUPPORTED)
        ghmm_set_error("%s holds several feature streams: read it with ghmm_hmm_read_streams", path);
    return rc;
}

int ghmm_hmm_write_streams(const char *path, const ghmm_host_model *hm, int n_streams, int len_bytes)
{
    if (!path || !hm || n_streams <= 0 || n_streams > GHMM_MAX_STREAMS || (len_bytes != 4 && len_bytes != 8))
        return GHMM_ERR_ARG;
    const int N = hm[0].N, P = n_streams;
    for (int p = 1; p < P; p++)
        if (hm[p].N != N) {
            ghmm_set_error("ghmm_hmm_write_streams: the streams differ in their number of states");
            return GHMM_ERR_ARG;
        }
    FILE *f = fopen(path, "wb");
    if (!f) {
        ghmm_set_error("can't open file %s", path);
        return GHMM_ERR_IO;
    }
    int32_t hdr[2] = {N, P}, MD[2 * GHMM_MAX_STREAMS];
    for (int p = 0; p < P; p++) {
        MD[p] = hm[p].M;
        MD[P + p] = hm[p].D;
    }
    uint64_t len = strnlen(hm[0].word, GHMM_MAX_WORD - 1);
    int ok = fwrite(&len, 1, (size_t)len_bytes, f) == (size_t)len_bytes; /* little-endian */
    ok = ok && fwrite(hm[0].word, 1, (size_t)len, f) == (size_t)len;
    ok = ok && fwrite(hdr, 4, 2, f) == 2 && fwrite(MD, 4, 2 * (size_t)P, f) == 2 * (size_t)P;
    ok = ok && fwrite(hm[0].A, 8, (size_t)N * N, f) == (size_t)N * N;
    for (int p = 0; ok && p < P; p++) {
        const int M = hm[p].M, D = hm[p].D;
        for (int i = 0; ok && i < N; i++) {
            ok = fwrite(hm[p].c + (size_t)i * M, 8, (size_t)M, f) == (size_t)M;
            for (int k = 0; ok && k < M; k++) {
                size_t g = (size_t)i * M + k;
                ok = fwrite(hm[p].mean + g * D, 8, (size_t)D, f) == (size_t)D &&
                     fwrite(hm[p].det + g, 8, 1, f) == 1 &&
                     fwrite(hm[p].inv_var + g * D, 8, (size_t)D, f) == (size_t)D;
            }
        }
    }
    if (fclose(f) != 0) ok = 0;
    if (!ok) {
        ghmm_set_error("writing error on file %s", path);
        return GHMM_ERR_IO;
    }
    return GHMM_OK;
}

int ghmm_hmm_write(const char *path, const ghmm_host_model *hm, int len_bytes)
{
    return ghmm_hmm_write_streams(path, hm, 1, len_bytes);
}
