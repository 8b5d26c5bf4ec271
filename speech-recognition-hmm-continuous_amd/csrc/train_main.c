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
 *   - param_number must be 1 (every BASELINE configuration);
 *   - no MAX_* capacity limits.
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
    int P = atoi(argv[3]);
    if (P != 1) {
        printf("param_number = %d: only one feature stream is supported \n", P);
        exit(1);
    }
    if (argc < 2 * P + 5) usage();
    int M = atoi(argv[4]);
    const char *list = argv[P + 4];
    const char *output = argv[2 * P + 4];
    const char *initial = (argc == 2 * P + 6) ? argv[argc - 1] : NULL;
    if (N <= 0 || M <= 0) {
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

    /* every utterance is read once and stays resident */
    FILE *fl = fopen(list, "r");
    if (!fl) {
        printf("file %s not found \n", list);
        exit(1);
    }
    char path[4096];
    char **files = NULL;
    int n_files = 0, cap_f = 0, rc;
    while (fscanf(fl, "%4095s", path) == 1) {
        if (n_files == cap_f) {
            cap_f = cap_f ? cap_f * 2 : 64;
            files = (char **)realloc(files, (size_t)cap_f * sizeof(char *));
            if (!files) die("memory", GHMM_ERR_ALLOC);
        }
        files[n_files] = strdup(path);
        if (!files[n_files++]) die("memory", GHMM_ERR_ALLOC);
    }
    fclose(fl);
    if (n_files == 0) {
        printf("no training utterances in %s \n", list);
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
            if ((rc = ghmm_perfil_stat(files[k], &d, &T))) die("reading", rc);
            all_len[k] = T;
        }
        if ((rc = ghmm_shard_balanced(all_len, n_files, rank, world, mine, &n_mine))) die("sharding", rc);
        free(all_len);
        if (n_mine == 0) {
            printf("rank %d of %d has no utterances (%d in %s) \n", rank, world, n_files, list);
            exit(1);
        }
    } else {
        for (int k = 0; k < n_files; k++) mine[k] = k;
    }
    double *X = NULL;
    int32_t *len = NULL;
    size_t frames = 0, cap = 0;
    int n_utt = 0, cap_u = 0, D = 0;
    for (int k = 0; k < n_mine; k++) {
        int d, T;
        double *x;
        snprintf(path, sizeof path, "%s", files[mine[k]]);
        if ((rc = ghmm_perfil_read(path, &d, &T, &x))) die("reading", rc);
        if (n_utt == 0) D = d;
        if (d != D) {
            printf("file %s has %d coefficients per frame, expected %d \n", path, d, D);
            exit(1);
        }
        if (frames + (size_t)T > cap) {
            cap = (frames + (size_t)T) * 2;
            X = (double *)realloc(X, cap * (size_t)D * sizeof(double));
        }
        if (n_utt == cap_u) {
            cap_u = cap_u ? cap_u * 2 : 64;
            len = (int32_t *)realloc(len, (size_t)cap_u * sizeof(int32_t));
        }
        if (!X || !len) die("memory", GHMM_ERR_ALLOC);
        memcpy(X + frames * (size_t)D, x, (size_t)T * (size_t)D * sizeof(double));
        ghmm_free(x);
        len[n_utt++] = T;
        frames += (size_t)T;
    }
    for (int k = 0; k < n_files; k++) free(files[k]);
    free(files);
    free(mine);

    ghmm_host_model hm;
    memset(&hm, 0, sizeof hm);
    ghmm_ctx *ctx;
    ghmm_model *model;
    ghmm_corpus *corpus;
    ghmm_stats *stats;
    ghmm_comm *comm = NULL;
    if ((rc = ghmm_ctx_create(device, NULL, &ctx))) die("GPU context", rc);
    if (comm_id && *comm_id && (rc = ghmm_comm_create_file(ctx, comm_id, rank, world, 300.0, &comm)))
        die("communicator", rc);
    if ((rc = ghmm_corpus_create(ctx, X, len, n_utt, D, &corpus))) die("corpus", rc);
    if (initial) {
        if ((rc = ghmm_hmm_read(initial, &hm))) die("initial model", rc);
        if (hm.D != D) {
            printf("initial model %s has %d coefficients, data has %d \n", initial, hm.D, D);
            exit(1);
        }
        N = hm.N;
        M = hm.M;
        if ((rc = ghmm_model_create(ctx, N, M, D, &model))) die("model", rc);
        if ((rc = ghmm_model_set(ctx, model, hm.A, hm.c, hm.mean, hm.inv_var, hm.det))) die("model", rc);
    } else {
        /* creating_initial_model (TF:226): on the GPU from the resident corpus, or with
           GHMM_HOST_INIT=1 by the host implementation (bit-exact with the reference) */
        if ((rc = ghmm_host_model_alloc(&hm, N, M, D))) die("memory", rc);
        if ((rc = ghmm_model_create(ctx, N, M, D, &model))) die("model", rc);
        const char *hi = getenv("GHMM_HOST_INIT");
        if (hi && *hi == '1') {
            if (world > 1) {
                printf("GHMM_HOST_INIT=1 needs the whole corpus on one rank \n");
                exit(1);
            }
            ghmm_host_model_free(&hm);
            if ((rc = ghmm_init_model(X, len, n_utt, N, M, D, &hm))) die("creating initial model", rc);
            if ((rc = ghmm_model_set(ctx, model, hm.A, hm.c, hm.mean, hm.inv_var, hm.det))) die("model", rc);
        } else if ((rc = ghmm_model_init_comm(ctx, model, corpus, comm))) {
            die("creating initial model", rc);
        }
    }
    snprintf(hm.word, sizeof hm.word, "%s", word);
    if ((rc = ghmm_stats_create(ctx, N, M, D, &stats))) die("statistics", rc);
    size_t ns = ghmm_stats_len(N, M, D);
    double *sv = (double *)malloc(ns * sizeof(double));
    if (!sv) die("memory", GHMM_ERR_ALLOC);

    printf("\r\nCreating HMM using Forward-Backward algorithm (Baum-Welch)");
    double probab, old_probab = 1.0, variation; /* TF:151 */
    int iteration = 0;
    do {
        iteration++;
        printf("\r\nStarting training sequence (%d utterances, %zu frames)", n_utt, frames);
        if ((rc = ghmm_estep(ctx, model, corpus, stats))) die("E-step", rc);
        /* the one exchange of the iteration: sum of the accumulators over ranks */
        if (comm && (rc = ghmm_stats_allreduce(ctx, stats, comm))) die("all-reduce", rc);
        if ((rc = ghmm_stats_download(ctx, stats, sv))) die("E-step", rc);
        probab = sv[ns - 2];
        printf("\r\nEnding training sequence");
        variation = fabs((old_probab - probab) / old_probab);
        printf("\r\nVerifying Probability: %f > Threshold: %f", variation, THRESHOLD);
        if (variation > THRESHOLD) {
            /* the statistics of the converging pass are discarded (TF:328) */
            old_probab = probab;
            if ((rc = ghmm_mstep(ctx, model, stats))) die("M-step", rc);
        }
    } while (variation > THRESHOLD);
    printf("\r\nFinal Probability = %f\r\n\r\n", variation);
    if (comm) n_utt = (int)sv[ns - 1]; /* exemplars of all ranks (TF:320, summed) */
    probab /= (double)n_utt;

    if ((rc = ghmm_model_get(ctx, model, hm.A, hm.c, hm.mean, hm.inv_var, hm.det))) die("model", rc);

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
    if ((rc = ghmm_hmm_write(output, &hm, 8))) die("writing model", rc);

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
    fprintf(ft, "number of mixtures %d: %d \n", 1, M);
    fprintf(ft, "parameter %d: %s \n", 1, list);
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
    ghmm_stats_destroy(ctx, stats);
    ghmm_corpus_destroy(ctx, corpus);
    ghmm_model_destroy(ctx, model);
    ghmm_ctx_destroy(ctx);
    ghmm_host_model_free(&hm);
    free(sv);
    free(X);
    free(len);
    return 0;
}
