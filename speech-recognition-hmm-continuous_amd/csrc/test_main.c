/*
 * recognition-continuous-test-fs — the reference recogniser's command line on the
 * MI355X core.
 *
 * Same argv, list/.perfil/.hmm formats and report file as the reference's main()
 * (RF:87-428):
 *     models_number model1..N coef_model1..N input_file1..N word_file output_file
 * The reference re-reads every utterance file once per word model and runs
 * emission + forward per (utterance, model) (RF:326-374).  Here all utterances are
 * read once, kept in HBM, and every word model scores the whole batch with one
 * ghmm_score call; the ranking, the correct / error / second-candidate bookkeeping
 * and the report text follow RF:374-412 and RF:1014-1194 line by line, including
 * the NaN-blind bubble sort (RF:968-995).
 *
 * A model set may have several feature streams (param_number P > 1): its P feature lists follow
 * each other on the command line (RF:253-262) and the score runs on the product of the streams'
 * emission densities (ghmm_score_streams, RF:349-366).
 */
#include "ghmm.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/times.h>
#include <time.h>

#define MAX_SETS 16

static void die(const char *what, int rc)
{
    const char *d = ghmm_last_error();
    printf("%s: %s \n", what, (d && *d) ? d : ghmm_strerror(rc));
    exit(1);
}

static void usage(void)
{
    puts("Usage: recognition_continuous_fs models_number  model1 ... modelN coef_model1 ... coef_modelN input_file1 ... input_fileM  word_file output_file");
    puts("models_number: number of model");
    puts("model1: name of file with the name of model 1");
    puts("modelN: name of file with the name of model N");
    puts("coef_model1: weighting coefficient of model 1");
    puts("coef_modelN: weighting coefficient of model N");
    puts("input_file1: name of file with name of files with parameters 1 ");
    puts("input_fileM: name of file with name of files with parameters M ");
    puts("word_file: name of file with the spoken words ");
    puts("output_file: name of output file ");
    exit(1);
}

static FILE *open_read(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) {
        printf("file %s not found \n", path);
        exit(1);
    }
    return f;
}

static double cpu_units(void)
{
    struct tms t;
    times(&t);
    return t.tms_utime / 60.0; /* RF:278 */
}

/* sorting_probab, RF:968-995 */
static void rank_scores(const double *probab, int *index, int n)
{
    int done = 0;
    for (int i = 0; i < n; i++) index[i] = i;
    while (!done) {
        done = 1;
        for (int i = 0; i < n - 1; i++)
            if (probab[index[i]] < probab[index[i + 1]]) {
                int aux = index[i];
                index[i] = index[i + 1];
                index[i + 1] = aux;
                done = 0;
            }
    }
}

static FILE *f_out;

/* writing_result_word, RF:1111-1150 */
static void report_word(int correct, int error, int second, int word_number, const char *spoken,
                        const int *wrong_word, char **word, double cpu_time, int word_frames)
{
    int sum = correct + error;
    double per = (double)correct / (double)sum;
    cpu_time /= sum;
    word_frames /= sum;
    fprintf(f_out, "\nResults: \n");
    fprintf(f_out, "Spoken word: %s\n", spoken);
    fprintf(f_out, "Correct words: %d\n", correct);
    fprintf(f_out, "Errors: %d\n", error);
    fprintf(f_out, "Percentagen correct : %.2f%%\n", per * 100.0);
    fprintf(f_out, "Second candidate: %d\n", second);
    if (error != 0) {
        fprintf(f_out, "Wrong words: \n");
        for (int i = 0; i < word_number; i++)
            if (wrong_word[i] != 0)
                fprintf(f_out, "%s: %d time%s\n", word[i], wrong_word[i], wrong_word[i] == 1 ? "" : "s");
    }
    fprintf(f_out, "Average recognition time: %.2f sec. \n", cpu_time);
    fprintf(f_out, "Average word length: %d frames \n", word_frames);
}

int main(int argc, char **argv)
{
    char date_time[64];
    time_t now;
    time(&now);
    strftime(date_time, sizeof date_time, "%d-%h-%Y %X", localtime(&now));
    if (argc < 7) usage();
    int K = atoi(argv[1]);
    if (K < 1 || K > MAX_SETS || argc < 3 * K + 4) usage();
    double coef_model[MAX_SETS];
    for (int i = 0; i < K; i++) coef_model[i] = atof(argv[K + 2 + i]);
    const char *output_file = argv[argc - 1], *word_file = argv[argc - 2];
    int rc;

    /* models: one list per set, the same vocabulary in every set */
    ghmm_host_model *hm[MAX_SETS]; /* hm[j][k * GHMM_MAX_STREAMS + p]: set j, word k, stream p */
    int Pj[MAX_SETS];              /* feature streams of set j (param_number of its models) */
    int word_number = 0;
    printf("\r\nLoading Models\r\n");
    for (int j = 0; j < K; j++) {
        FILE *fl = open_read(argv[2 + j]);
        char name[4096];
        int n = 0, cap = 0;
        hm[j] = NULL;
        while (fscanf(fl, "%4095s", name) == 1) {
            printf("Model: %s\r\n", name);
            if (n == cap) {
                cap = cap ? 2 * cap : 32;
                hm[j] = (ghmm_host_model *)realloc(hm[j], (size_t)cap * GHMM_MAX_STREAMS * sizeof(ghmm_host_model));
                if (!hm[j]) die("memory", GHMM_ERR_ALLOC);
            }
            int pn = 0;
            if ((rc = ghmm_hmm_read_streams(name, &hm[j][(size_t)n * GHMM_MAX_STREAMS], GHMM_MAX_STREAMS, &pn)))
                die("reading model", rc);
            if (n == 0) Pj[j] = pn;
            if (pn != Pj[j]) {
                printf("model %s has %d parameters, the first model of %s has %d \n", name, pn, argv[2 + j], Pj[j]);
                exit(1);
            }
            n++;
        }
        fclose(fl);
        if (j > 0 && n != word_number) {
            printf("model list %s holds %d models, expected %d \n", argv[2 + j], n, word_number);
            exit(1);
        }
        word_number = n;
    }
    if (word_number == 0) {
        printf("no models in %s \n", argv[2]);
        exit(1);
    }
    char **word = (char **)malloc((size_t)word_number * sizeof(char *));
    for (int k = 0; k < word_number; k++) word[k] = hm[K - 1][(size_t)k * GHMM_MAX_STREAMS].word; /* RF:229 */
    int n_lists = 0;
    for (int j = 0; j < K; j++) n_lists += Pj[j];
    if (argc != 2 * K + n_lists + 4) usage();

    /* spoken words and their feature files, read once */
    FILE *fw = open_read(word_file);
    /* one feature list per (set, stream), in command-line order (RF:253-262) */
    FILE *ff[MAX_SETS][GHMM_MAX_STREAMS];
    const char *ffname[MAX_SETS][GHMM_MAX_STREAMS];
    for (int j = 0, q = 0; j < K; j++)
        for (int p = 0; p < Pj[j]; p++, q++) {
            ffname[j][p] = argv[2 + 2 * K + q];
            ff[j][p] = open_read(ffname[j][p]);
        }
    char (*spoken)[256] = NULL;
    int n_utt = 0, cap_u = 0;
    double *X[MAX_SETS][GHMM_MAX_STREAMS] = {{0}};
    size_t frames[MAX_SETS][GHMM_MAX_STREAMS] = {{0}}, capx[MAX_SETS][GHMM_MAX_STREAMS] = {{0}};
    int32_t *len[MAX_SETS] = {0};
    int D[MAX_SETS][GHMM_MAX_STREAMS] = {{0}};
    char w[4096], path[4096];
    while (fscanf(fw, "%4095s", w) == 1) {
        if (n_utt == cap_u) {
            cap_u = cap_u ? 2 * cap_u : 64;
            spoken = realloc(spoken, (size_t)cap_u * sizeof *spoken);
            for (int j = 0; j < K; j++) len[j] = (int32_t *)realloc(len[j], (size_t)cap_u * sizeof(int32_t));
        }
        snprintf(spoken[n_utt], sizeof spoken[n_utt], "%s", w);
        for (int j = 0; j < K; j++)
            for (int p = 0; p < Pj[j]; p++) {
                if (fscanf(ff[j][p], "%4095s", path) != 1) {
                    printf("reading error on file %s \n", ffname[j][p]);
                    exit(1);
                }
                int d, T;
                double *x;
                if ((rc = ghmm_perfil_read(path, &d, &T, &x))) die("reading", rc);
                if (n_utt == 0) D[j][p] = d;
                if (d != D[j][p]) {
                    printf("file %s has %d coefficients per frame, expected %d \n", path, d, D[j][p]);
                    exit(1);
                }
                if (p == 0) len[j][n_utt] = T;
                if (T != len[j][n_utt]) {
                    printf("file %s has %d frames, parameter 1 of the same utterance has %d \n", path, T, len[j][n_utt]);
                    exit(1);
                }
                if (frames[j][p] + (size_t)T > capx[j][p]) {
                    capx[j][p] = (frames[j][p] + (size_t)T) * 2;
                    X[j][p] = (double *)realloc(X[j][p], capx[j][p] * (size_t)d * sizeof(double));
                    if (!X[j][p]) die("memory", GHMM_ERR_ALLOC);
                }
                memcpy(X[j][p] + frames[j][p] * (size_t)d, x, (size_t)T * (size_t)d * sizeof(double));
                ghmm_free(x);
                frames[j][p] += (size_t)T;
            }
        n_utt++;
    }
    fclose(fw);
    for (int j = 0; j < K; j++)
        for (int p = 0; p < Pj[j]; p++) fclose(ff[j][p]);

    f_out = fopen(output_file, "w");
    if (!f_out) {
        printf("can't open file %s \n", output_file);
        exit(1);
    }
    /* writing_header, RF:1014-1031.  The reference declares coef_model as int* there
       and prints it with %.2d: the integer words of the double array are shown. */
    fprintf(f_out, "Isolated word recognition using Continuous HMM (diagonal covariance matrix). It is considered a final state. \n");
    fprintf(f_out, "Algorithm used for recognition: Forward \n");
    fprintf(f_out, "Number of models: %d  \n", K);
    for (int i = 0; i < K; i++) {
        int as_int;
        memcpy(&as_int, (const char *)coef_model + sizeof(int) * (size_t)i, sizeof as_int);
        fprintf(f_out, "Model name %d: %s\n", i + 1, argv[2 + i]);
        fprintf(f_out, "Weighting coefficient of model %d:%.2d\n", i + 1, as_int);
    }
    fprintf(f_out, "Date and time: %s \n\n", date_time);

    /* score[k][u] = sum_j w_j log P(utterance u | model k of set j), RF:366 */
    double *score = (double *)calloc((size_t)word_number * (size_t)(n_utt ? n_utt : 1), sizeof(double));
    double *part = (double *)malloc((size_t)(n_utt ? n_utt : 1) * sizeof(double));
    double old_aux = cpu_units();
    printf("\r\nStarting Tests\r\n");
    if (n_utt > 0) {
        ghmm_ctx *ctx;
        if ((rc = ghmm_ctx_create(0, NULL, &ctx))) die("GPU context", rc);
        for (int j = 0; j < K; j++) {
            const int P = Pj[j];
            ghmm_corpus *corpus[GHMM_MAX_STREAMS];
            for (int p = 0; p < P; p++)
                if ((rc = ghmm_corpus_create(ctx, X[j][p], len[j], n_utt, D[j][p], &corpus[p]))) die("corpus", rc);
            /* one stream: the whole vocabulary in one batched call when the models share M and D
               (ghmm_score_batch), model by model otherwise; several streams: model by model on
               the product of the streams' densities (ghmm_score_streams) */
            ghmm_model **dm = (ghmm_model **)calloc((size_t)word_number * GHMM_MAX_STREAMS, sizeof(ghmm_model *));
            double *all = (double *)malloc((size_t)word_number * (size_t)n_utt * sizeof(double));
            if (!dm || !all) die("memory", GHMM_ERR_ALLOC);
            int same = 1;
            for (int k = 0; k < word_number; k++)
                for (int p = 0; p < P; p++) {
                    ghmm_host_model *m = &hm[j][(size_t)k * GHMM_MAX_STREAMS + p];
                    if (m->D != D[j][p]) {
                        printf("model %s has %d coefficients, data has %d \n", m->word, m->D, D[j][p]);
                        exit(1);
                    }
                    if (m->M != hm[j][p].M) same = 0;
                    ghmm_model **slot = &dm[(size_t)k * GHMM_MAX_STREAMS + p];
                    if ((rc = ghmm_model_create(ctx, m->N, m->M, m->D, slot))) die("model", rc);
                    if ((rc = ghmm_model_set(ctx, *slot, m->A, m->c, m->mean, m->inv_var, m->det))) die("model", rc);
                }
            if (P == 1 && same) {
                ghmm_model **flat = (ghmm_model **)malloc((size_t)word_number * sizeof(ghmm_model *));
                if (!flat) die("memory", GHMM_ERR_ALLOC);
                for (int k = 0; k < word_number; k++) flat[k] = dm[(size_t)k * GHMM_MAX_STREAMS];
                if ((rc = ghmm_score_batch(ctx, flat, word_number, corpus[0], all))) die("scoring", rc);
                free(flat);
            } else {
                for (int k = 0; k < word_number; k++)
                    if ((rc = ghmm_score_streams(ctx, &dm[(size_t)k * GHMM_MAX_STREAMS], corpus, P,
                                                 all + (size_t)k * n_utt)))
                        die("scoring", rc);
            }
            for (int k = 0; k < word_number; k++) {
                for (int u = 0; u < n_utt; u++)
                    score[(size_t)k * n_utt + u] += coef_model[j] * all[(size_t)k * n_utt + u];
                for (int p = 0; p < P; p++) ghmm_model_destroy(ctx, dm[(size_t)k * GHMM_MAX_STREAMS + p]);
            }
            free(dm);
            free(all);
            for (int p = 0; p < P; p++) ghmm_corpus_destroy(ctx, corpus[p]);
        }
        ghmm_ctx_destroy(ctx);
    }

    /* bookkeeping and report, RF:283-412 */
    int correct = 0, error = 0, second = 0, sum_correct = 0, sum_error = 0, sum_second = 0;
    int word_frames = 0, total_frames = 0;
    double cpu_time, sum_cpu_time = 0.0, aux;
    int *index = (int *)malloc((size_t)word_number * sizeof(int));
    int *wrong_word = (int *)calloc((size_t)word_number, sizeof(int));
    double *probab = (double *)malloc((size_t)word_number * sizeof(double));
    char last_word[256] = " ";
    for (int u = 0; u < n_utt; u++) {
        printf("\r\nSpoken word: %s", spoken[u]);
        if (strcmp(last_word, spoken[u]) != 0) {
            if (strcmp(last_word, " ") != 0) {
                aux = cpu_units();
                cpu_time = aux - old_aux;
                old_aux = aux;
                sum_cpu_time += cpu_time;
                report_word(correct, error, second, word_number, last_word, wrong_word, word, cpu_time,
                            word_frames);
                sum_correct += correct;
                sum_error += error;
                sum_second += second;
                total_frames += word_frames;
                word_frames = 0;
                correct = error = second = 0;
                for (int i = 0; i < word_number; i++) wrong_word[i] = 0;
            }
            fprintf(f_out, "\nSpoken word: %s\n", spoken[u]);
        }
        for (int k = 0; k < word_number; k++) probab[k] = score[(size_t)k * n_utt + u];
        word_frames += len[K - 1][u];
        rank_scores(probab, index, word_number);
        printf("\r\nWriting result\r\n");
        for (int i = 0; i < word_number; i++) printf("%s :  %f \n", word[index[i]], probab[index[i]]);
        printf("\n");
        if (strcmp(spoken[u], word[index[0]]) == 0) {
            correct++;
        } else {
            error++;
            wrong_word[index[0]]++;
            if (word_number > 1 && strcmp(spoken[u], word[index[1]]) == 0) second++;
        }
        snprintf(last_word, sizeof last_word, "%s", spoken[u]);
    }
    printf("\r\nEnding Tests\r\n");
    aux = cpu_units();
    cpu_time = aux - old_aux;
    sum_cpu_time += cpu_time;
    if (n_utt > 0) {
        /* the reference passes models_number where word_number belongs (RF:400) */
        report_word(correct, error, second, K, last_word, wrong_word, word, cpu_time, word_frames);
        sum_correct += correct;
        sum_error += error;
        sum_second += second;
        total_frames += word_frames;
        /* writing_total_result, RF:1164-1194 */
        int sum = sum_correct + sum_error;
        double per = (double)sum_correct / (double)sum;
        fprintf(f_out, "\nConsidering all the words: \n");
        fprintf(f_out, "Results: \n");
        fprintf(f_out, "Correct words: %d\n", sum_correct);
        fprintf(f_out, "Errors: %d\n", sum_error);
        fprintf(f_out, "Percentagen correct : %.2f%%\n", per * 100.0);
        fprintf(f_out, "Second candidate: %d\n", sum_second);
        fprintf(f_out, "Average recognition time: %.2f sec. \n", sum_cpu_time / sum);
        fprintf(f_out, "Average word length: %d frames \n", total_frames / sum);
    }
    if (ferror(f_out) || fclose(f_out) != 0) {
        printf("writing error on file %s \n", output_file);
        exit(1);
    }
    for (int j = 0; j < K; j++) {
        for (int k = 0; k < word_number; k++)
            for (int p = 0; p < Pj[j]; p++) ghmm_host_model_free(&hm[j][(size_t)k * GHMM_MAX_STREAMS + p]);
        free(hm[j]);
        for (int p = 0; p < Pj[j]; p++) free(X[j][p]);
        free(len[j]);
    }
    free(word); free(spoken); free(score); free(part); free(index); free(wrong_word); free(probab);
    return 0;
}
