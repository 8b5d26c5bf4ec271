/*
 * ref_harness.c — TEST INFRASTRUCTURE.  Function-level driver around the REAL
 * reference trainer.  It contains no reference code: build_ref.sh pipes
 * /root/reference/train/source/hmm-fs/hmm_continuous_fs.c through sed (raised
 * MAX_* capacities only) into a scratch file outside the repo and passes its
 * path as -DREF_TF_SOURCE; this file #includes it with main() renamed and then
 * calls the reference's own functions in the order its main() does (TF:226-346),
 * dumping every intermediate as tagged binary arrays.
 *
 * usage: ref_harness <list.txt> <N> <M> <init.hmm|-> <out.bin>
 *   init "-"  : model from the reference's creating_initial_model (TF:732)
 *   otherwise : model read with the reference's reading_model (TF:604)
 *
 * Dump record: char name[32]; int32 ndim; int64 dim[4]; double data[prod(dim)].
 */
#define main ref_tf_main
#include REF_TF_SOURCE
#undef main

static FILE *g_out;

static void dump(const char *name, int ndim, long d0, long d1, long d2, const double *data)
{
    char nm[32];
    long long dims[4] = {d0, d1, d2, 1};
    int nd = ndim;
    size_t n = 1;
    int i;
    memset(nm, 0, sizeof nm);
    strncpy(nm, name, 31);
    for (i = 0; i < ndim; i++) n *= (size_t)dims[i];
    fwrite(nm, 1, 32, g_out);
    fwrite(&nd, sizeof(int), 1, g_out);
    fwrite(dims, sizeof(long long), 4, g_out);
    fwrite(data, sizeof(double), n, g_out);
}

/* storage with the reference's own (patched) static shapes */
static double transition_probab[MAX_STATES_NUMBER][MAX_STATES_NUMBER];
static double symbol_probab[MAX_PARAMETERS_NUMBER][MAX_STATES_NUMBER][MAX_TIME];
static double gaus_probab_dens[MAX_TIME][MAX_STATES_NUMBER][MAX_MIXTURE_NUMBER];
static double alpha[MAX_STATES_NUMBER][MAX_TIME], beta[MAX_STATES_NUMBER][MAX_TIME];
static double scaling_factor[MAX_TIME];
static double num_trans_probab[MAX_STATES_NUMBER][MAX_STATES_NUMBER];
static double den_trans_probab[MAX_STATES_NUMBER], den_mixture_coef[MAX_STATES_NUMBER];
static struct state state_mix[MAX_PARAMETERS_NUMBER][MAX_STATES_NUMBER];
static struct state num_mix_param[MAX_PARAMETERS_NUMBER][MAX_STATES_NUMBER];
static double frames[MAX_TIME][MAX_COEF_NUMBER];
static double flat[MAX_TIME * MAX_STATES_NUMBER * MAX_MIXTURE_NUMBER + 1024];

static void dump_model(const char *tag, int N, int M, int D)
{
    char nm[32];
    int i, j, k, q;
    q = 0;
    for (i = 0; i < N; i++) for (j = 0; j < N; j++) flat[q++] = transition_probab[i][j];
    sprintf(nm, "%s.A", tag); dump(nm, 2, N, N, 1, flat);
    q = 0;
    for (i = 0; i < N; i++) for (j = 0; j < M; j++) flat[q++] = state_mix[0][i].mix_coef[j];
    sprintf(nm, "%s.c", tag); dump(nm, 2, N, M, 1, flat);
    q = 0;
    for (i = 0; i < N; i++) for (j = 0; j < M; j++) for (k = 0; k < D; k++)
        flat[q++] = state_mix[0][i].mix[j].mean[k];
    sprintf(nm, "%s.mean", tag); dump(nm, 3, N, M, D, flat);
    q = 0;
    for (i = 0; i < N; i++) for (j = 0; j < M; j++) for (k = 0; k < D; k++)
        flat[q++] = state_mix[0][i].mix[j].cov_matrix[k];
    sprintf(nm, "%s.inv_var", tag); dump(nm, 3, N, M, D, flat);
    q = 0;
    for (i = 0; i < N; i++) for (j = 0; j < M; j++) flat[q++] = state_mix[0][i].mix[j].det;
    sprintf(nm, "%s.det", tag); dump(nm, 2, N, M, 1, flat);
}

int main(int argc, char **argv)
{
    int N, M, D = 0, T, u = 0, i, j, k, t, q;
    int mixture_number[MAX_PARAMETERS_NUMBER], coef_number[MAX_PARAMETERS_NUMBER];
    int pi[MAX_STATES_NUMBER];
    char data_file[MAX_PARAMETERS_NUMBER][MAX_NAME_SIZE], path[MAX_NAME_SIZE], nm[32];
    char word[MAX_WORD_SIZE];
    double probab = 0.0, p, dd;
    FILE *flist, *f;

    if (argc != 6) {
        fprintf(stderr, "usage: %s list N M init.hmm|- out.bin\n", argv[0]);
        return 2;
    }
    N = atoi(argv[2]);
    M = atoi(argv[3]);
    if (N > MAX_STATES_NUMBER || M > MAX_MIXTURE_NUMBER) {
        fprintf(stderr, "capacity: N<=%d M<=%d\n", MAX_STATES_NUMBER, MAX_MIXTURE_NUMBER);
        return 2;
    }
    mixture_number[0] = M;
    strncpy(data_file[0], argv[1], MAX_NAME_SIZE);
    g_out = fopen(argv[5], "wb");
    if (!g_out) return 2;

    if (strcmp(argv[4], "-") == 0) {
        creating_initial_model(1, data_file, N, mixture_number, coef_number, transition_probab,
                               state_mix);
    } else {
        int P = 1;
        reading_model(argv[4], &P, &N, mixture_number, coef_number, transition_probab, state_mix,
                      word);
        M = mixture_number[0];
    }
    D = coef_number[0];
    dump_model("model0", N, M, D);

    pi[0] = 1;
    for (i = 1; i < N; i++) pi[i] = 0;

    /* zeroing, TF:244-270 */
    for (i = 0; i < N; i++) {
        for (j = 0; j < N; j++) num_trans_probab[i][j] = 0.0;
        den_trans_probab[i] = 0.0;
        den_mixture_coef[i] = 0.0;
        for (j = 0; j < M; j++) {
            for (k = 0; k < D; k++) {
                num_mix_param[0][i].mix[j].mean[k] = 0.0;
                num_mix_param[0][i].mix[j].cov_matrix[k] = 0.0;
            }
            num_mix_param[0][i].mix_coef[j] = 0.0;
        }
    }

    flist = fopen(argv[1], "r");
    if (!flist) return 2;
    while (fscanf(flist, "%s", path) != EOF) {
        f = opening_file_read(path, "rb");
        reading_coef_number(f, path);
        T = 0;
        while (reading_coef(f, path, D, frames[T]) != 0) {
            calc_symbol_probab(N, M, D, frames[T], state_mix[0], gaus_probab_dens[T],
                               symbol_probab[0], T);
            T++;
        }
        fclose(f);
        calc_alpha(N, T, 1, alpha, scaling_factor, transition_probab, symbol_probab, pi);
        calc_beta(N, T, 1, beta, scaling_factor, transition_probab, symbol_probab);
        calc_transition_probab(N, T, 1, alpha, beta, scaling_factor, transition_probab,
                               symbol_probab, num_trans_probab, den_trans_probab);
        calc_den_mix_coef(T, N, alpha, beta, scaling_factor, den_mixture_coef);
        for (t = 0; t < T; t++)
            calc_mix_param(t, N, M, D, frames[t], alpha, beta, scaling_factor,
                           gaus_probab_dens[t], num_mix_param[0], state_mix[0]);
        p = calc_probability(T, scaling_factor, alpha[N - 1][T - 1]);
        probab += p;

        q = 0;
        for (t = 0; t < T; t++) for (k = 0; k < D; k++) flat[q++] = frames[t][k];
        sprintf(nm, "u%d.X", u); dump(nm, 2, T, D, 1, flat);
        q = 0;
        for (t = 0; t < T; t++) for (i = 0; i < N; i++) flat[q++] = symbol_probab[0][i][t];
        sprintf(nm, "u%d.b", u); dump(nm, 2, T, N, 1, flat);
        q = 0;
        for (t = 0; t < T; t++) for (i = 0; i < N; i++) for (j = 0; j < M; j++)
            flat[q++] = gaus_probab_dens[t][i][j];
        sprintf(nm, "u%d.post", u); dump(nm, 3, T, N, M, flat);
        q = 0;
        for (t = 0; t < T; t++) for (i = 0; i < N; i++) flat[q++] = alpha[i][t];
        sprintf(nm, "u%d.alpha", u); dump(nm, 2, T, N, 1, flat);
        q = 0;
        for (t = 0; t < T; t++) for (i = 0; i < N; i++) flat[q++] = beta[i][t];
        sprintf(nm, "u%d.beta", u); dump(nm, 2, T, N, 1, flat);
        sprintf(nm, "u%d.scale", u); dump(nm, 1, T, 1, 1, scaling_factor);
        sprintf(nm, "u%d.loglik", u); dump(nm, 1, 1, 1, 1, &p);
        u++;
    }
    fclose(flist);

    /* accumulators, flattened in the layout of include/ghmm.h */
    q = 0;
    for (i = 0; i < N; i++) for (j = 0; j < N; j++) flat[q++] = num_trans_probab[i][j];
    dump("stats.num_a", 2, N, N, 1, flat);
    dump("stats.den_a", 1, N, 1, 1, den_trans_probab);
    dump("stats.den_c", 1, N, 1, 1, den_mixture_coef);
    q = 0;
    for (i = 0; i < N; i++) for (j = 0; j < M; j++) flat[q++] = num_mix_param[0][i].mix_coef[j];
    dump("stats.num_c", 2, N, M, 1, flat);
    q = 0;
    for (i = 0; i < N; i++) for (j = 0; j < M; j++) for (k = 0; k < D; k++)
        flat[q++] = num_mix_param[0][i].mix[j].mean[k];
    dump("stats.num_mu", 3, N, M, D, flat);
    q = 0;
    for (i = 0; i < N; i++) for (j = 0; j < M; j++) for (k = 0; k < D; k++)
        flat[q++] = num_mix_param[0][i].mix[j].cov_matrix[k];
    dump("stats.num_var", 3, N, M, D, flat);
    dump("stats.loglik", 1, 1, 1, 1, &probab);
    dd = (double)u;
    dump("stats.n_utt", 1, 1, 1, 1, &dd);

    /* M-step exactly as main() chains it, TF:332-346 */
    updating_transition_probab(N, num_trans_probab, den_trans_probab, transition_probab);
    updating_mix_param(N, M, D, den_mixture_coef, num_mix_param[0], state_mix[0]);
    for (j = 0; j < N; j++)
        for (k = 0; k < M; k++) {
            state_mix[0][j].mix[k].det = calc_det(D, state_mix[0][j].mix[k].cov_matrix);
            inv_matrix(D, state_mix[0][j].mix[k].cov_matrix);
        }
    dump_model("model1", N, M, D);
    fclose(g_out);
    return 0;
}
