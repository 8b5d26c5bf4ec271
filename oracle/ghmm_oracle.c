/*
 * ghmm_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded, double-precision restatement of the reference's
 * diagonal-covariance hot path, function by function and in
 * the reference's operation order, on flat frame-major arrays.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (libghmm_hip.so and the CLIs) never links or calls it.
 *
 * Parity pin: every function below is checked against outputs of the REAL
 * reference compiled from /root/reference (oracle/build_ref.sh -> oracle/_ref,
 * fixtures in tests/golden/, generator tests/golden/make_golden.py).
 * Exception: orc_viterbi — the reference contains no Viterbi,
 * so its parity is UNPINNED by the reference; it is pinned by brute-force path
 * enumeration in tests/test_oracle.py instead.
 *
 * Aliases: TF = train/source/hmm-fs/hmm_continuous_fs.c
 *          RF = test/source/recognition-fs/recognition_continuous_fs.c
 * Build with -ffp-contract=off: the reference is compiled without FMA.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_FINITE_PROBAB 1.0e-5 /* TF:39 */

/* calc_gaus, TF:1804-1841 (= RF:910-947).  det == 0 leaves the reference's
 * return value uninitialised; the oracle returns NaN there (outside parity). */
double orc_gauss(int D, const double *x, const double *mean, const double *inv_var, double det)
{
    double aux = 0.0, aux1, aux2, dif;
    aux1 = 2.0 * M_PI;
    aux2 = D / 2.0;
    aux1 = pow(aux1, aux2);
    if (det != 0) {
        aux2 = fabs(det);
        aux2 = pow(aux2, 0.5);
        for (int i = 0; i < D; i++) {
            dif = x[i] - mean[i];
            aux += dif * inv_var[i] * dif;
        }
        aux *= (-0.5);
        aux = exp(aux);
        return aux / (aux1 * aux2);
    }
    return NAN;
}

/* calc_symbol_probab for one frame, TF:1749-1783 (post != NULL) and RF:860-889
 * (post == NULL).  b[N], post[N*M]. */
void orc_emission_frame(int N, int M, int D, const double *x, const double *c,
                        const double *mean, const double *inv_var, const double *det, double *b,
                        double *post)
{
    for (int i = 0; i < N; i++) {
        double bi = 0.0;
        for (int j = 0; j < M; j++) {
            int g = i * M + j;
            double v = orc_gauss(D, x, mean + (size_t)g * D, inv_var + (size_t)g * D, det[g]);
            v *= c[g];
            if (post) post[g] = v;
            bi += v;
        }
        b[i] = bi;
        if (post) {
            if (bi != 0.0)
                for (int j = 0; j < M; j++) post[i * M + j] /= bi;
            else
                for (int j = 0; j < M; j++) post[i * M + j] = 0.0;
        }
    }
}

/* emission over T frames: b[T*N], post[T*N*M] or NULL */
void orc_emission(int N, int M, int D, int T, const double *X, const double *c,
                  const double *mean, const double *inv_var, const double *det, double *b,
                  double *post)
{
    for (int t = 0; t < T; t++)
        orc_emission_frame(N, M, D, X + (size_t)t * D, c, mean, inv_var, det, b + (size_t)t * N,
                           post ? post + (size_t)t * N * M : NULL);
}

/* calc_alpha, TF:1380-1443 (= RF:739-799), one stream, pi = one-hot at state 0
 * (TF:232-234).  alpha[T*N], scale[T]. */
void orc_forward(int N, int T, const double *A, const double *b, double *alpha, double *scale)
{
    double sum = 0.0;
    for (int i = 0; i < N; i++) {
        double product = 1.0;
        product *= b[i];
        alpha[i] = (i == 0 ? 1 : 0) * product;
        sum += alpha[i];
    }
    scale[0] = 1.0 / sum;
    for (int i = 0; i < N; i++) alpha[i] *= scale[0];
    for (int k = 1; k < T; k++) {
        const double *ap = alpha + (size_t)(k - 1) * N;
        double *ak = alpha + (size_t)k * N;
        sum = 0.0;
        for (int i = 0; i < N; i++) {
            double aux = 0.0;
            for (int j = 0; j < N; j++) aux += ap[j] * A[j * N + i];
            double product = 1.0;
            product *= b[(size_t)k * N + i];
            ak[i] = aux * product;
            sum += ak[i];
        }
        scale[k] = 1.0 / sum;
        for (int i = 0; i < N; i++) ak[i] *= scale[k];
    }
}

/* calc_beta, TF:1463-1516: final-state constraint, scaled by the forward scales */
void orc_backward(int N, int T, const double *A, const double *b, const double *scale,
                  double *beta)
{
    double *bl = beta + (size_t)(T - 1) * N;
    for (int i = 0; i < N - 1; i++) bl[i] = 0.0;
    bl[N - 1] = 1.0;
    bl[N - 1] *= scale[T - 1];
    for (int k = T - 2; k >= 0; k--) {
        const double *bn = beta + (size_t)(k + 1) * N;
        double *bk = beta + (size_t)k * N;
        for (int i = 0; i < N; i++) {
            double aux = 0.0;
            for (int j = 0; j < N; j++) {
                double product = 1.0;
                product *= b[(size_t)(k + 1) * N + j];
                aux += bn[j] * A[i * N + j] * product;
            }
            bk[i] = aux;
        }
        for (int i = 0; i < N; i++) bk[i] *= scale[k];
    }
}

/* calc_probability, TF:1536-1553 (= RF:820-836) */
double orc_loglik(int T, const double *scale, double alpha_last)
{
    double p = 0.0;
    for (int i = 0; i < T; i++) p -= log(scale[i]);
    p += log(alpha_last);
    return p;
}

/* calc_transition_probab, TF:1577-1620; num_a[N*N] += , den_a[N] += */
void orc_acc_trans(int N, int T, int delta, const double *A, const double *b,
                   const double *alpha, const double *beta, const double *scale, double *num_a,
                   double *den_a)
{
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < N; j++) {
            if (j >= i && j < (i + delta + 1)) {
                double aux = 0.0;
                for (int k = 0; k < T - 1; k++) {
                    double product = 1.0;
                    product *= b[(size_t)(k + 1) * N + j];
                    aux += alpha[(size_t)k * N + i] * A[i * N + j] * product *
                           beta[(size_t)(k + 1) * N + j];
                }
                num_a[i * N + j] += aux;
            }
        }
        for (int k = 0; k < T - 1; k++)
            den_a[i] += alpha[(size_t)k * N + i] * beta[(size_t)k * N + i] / scale[k];
    }
}

/* calc_den_mix_coef, TF:1642-1664 */
void orc_acc_den_mix(int N, int T, const double *alpha, const double *beta, const double *scale,
                     double *den_c)
{
    for (int i = 0; i < N; i++)
        for (int j = 0; j < T; j++) {
            double aux = alpha[(size_t)j * N + i] * beta[(size_t)j * N + i] / scale[j];
            den_c[i] += aux;
        }
}

/* calc_mix_param for frame t, TF:1691-1727; the variance statistic is taken
 * around the CURRENT (old) mean */
void orc_acc_mix_frame(int N, int M, int D, const double *x, const double *alpha_t,
                       const double *beta_t, double scale_t, const double *post_t,
                       const double *mean, double *num_c, double *num_mu, double *num_var)
{
    for (int i = 0; i < N; i++) {
        double aux = alpha_t[i] * beta_t[i] / scale_t;
        for (int j = 0; j < M; j++) {
            int g = i * M + j;
            double aux1 = aux * post_t[g];
            num_c[g] += aux1;
            for (int k = 0; k < D; k++) {
                num_mu[(size_t)g * D + k] += aux1 * x[k];
                double dif = x[k] - mean[(size_t)g * D + k];
                dif *= dif;
                num_var[(size_t)g * D + k] += aux1 * dif;
            }
        }
    }
}

/* updating_transition_probab, TF:1862-1889 (the range warning is not restated) */
void orc_update_trans(int N, const double *num_a, const double *den_a, double *A)
{
    for (int i = 0; i < N; i++)
        if (den_a[i] != 0.0)
            for (int j = 0; j < N; j++) A[i * N + j] = num_a[i * N + j] / den_a[i];
}

/* changing_zero_coef, TF:1338-1359 */
static void orc_floor_weights(int M, double *c)
{
    double sum = 0.0;
    for (int k = 0; k < M; k++) {
        if (c[k] < ORC_FINITE_PROBAB) c[k] = ORC_FINITE_PROBAB;
        sum += c[k];
    }
    for (int k = 0; k < M; k++) c[k] /= sum;
}

/* updating_mix_param TF:1911-1955, then calc_det TF:1976 + inv_matrix TF:2012
 * exactly as main() chains them (TF:337-346).  On return inv_var holds
 * 1/sigma^2 and det the product of the floored variances. */
void orc_update_mix(int N, int M, int D, const double *den_c, const double *num_c,
                    const double *num_mu, const double *num_var, double *c, double *mean,
                    double *inv_var, double *det)
{
    /* between TF:337 and TF:343 `cov_matrix` holds variances for updated states
       and still the inverse variances for states whose den_c is 0; calc_det and
       inv_matrix then run on whatever is there.  Restated literally. */
    for (int i = 0; i < N; i++) {
        if (den_c[i] != 0.0) {
            for (int j = 0; j < M; j++) {
                int g = i * M + j;
                c[g] = num_c[g] / den_c[i];
                for (int k = 0; k < D; k++) {
                    size_t q = (size_t)g * D + k;
                    mean[q] = num_mu[q] / num_c[g];
                    inv_var[q] = num_var[q] / num_c[g];
                    if (inv_var[q] < ORC_FINITE_PROBAB) inv_var[q] = ORC_FINITE_PROBAB;
                }
            }
        }
    }
    for (int i = 0; i < N; i++) orc_floor_weights(M, c + i * M);
    for (int g = 0; g < N * M; g++) {
        double d = 1.0;
        for (int k = 0; k < D; k++) d *= inv_var[(size_t)g * D + k];
        det[g] = d;
        for (int k = 0; k < D; k++) inv_var[(size_t)g * D + k] = 1.0 / inv_var[(size_t)g * D + k];
    }
}

/* ---------------------------------------------------------------- batched */

size_t orc_stats_len(int N, int M, int D)
{
    return (size_t)N * N + 2 * (size_t)N + (size_t)N * M * (2 * (size_t)D + 1) + 2;
}

/* One E-step over a corpus, TF:244-321.  stats layout = include/ghmm.h.
 * Optional per-frame dumps (any may be NULL): b[F*N], post[F*N*M], alpha[F*N],
 * beta[F*N], scale[F], loglik[U].  Returns 0, or 1 on allocation failure. */
int orc_estep(int N, int M, int D, int delta, const double *A, const double *c,
              const double *mean, const double *inv_var, const double *det, const double *X,
              const int32_t *len, int n_utt, double *stats, double *o_b, double *o_post,
              double *o_alpha, double *o_beta, double *o_scale, double *o_loglik)
{
    int G = N * M, Tmax = 0;
    for (int u = 0; u < n_utt; u++)
        if (len[u] > Tmax) Tmax = len[u];
    double *b = malloc(sizeof(double) * (size_t)Tmax * N);
    double *post = malloc(sizeof(double) * (size_t)Tmax * G);
    double *alpha = malloc(sizeof(double) * (size_t)Tmax * N);
    double *beta = malloc(sizeof(double) * (size_t)Tmax * N);
    double *scale = malloc(sizeof(double) * (size_t)Tmax);
    if (!b || !post || !alpha || !beta || !scale) {
        free(b); free(post); free(alpha); free(beta); free(scale);
        return 1;
    }
    memset(stats, 0, sizeof(double) * orc_stats_len(N, M, D));
    double *num_a = stats, *den_a = num_a + (size_t)N * N, *den_c = den_a + N;
    double *num_c = den_c + N, *num_mu = num_c + G, *num_var = num_mu + (size_t)G * D;
    double *ll = num_var + (size_t)G * D, *nu = ll + 1;
    size_t f0 = 0;
    for (int u = 0; u < n_utt; u++) {
        int T = len[u];
        const double *Xu = X + f0 * D;
        if (T > 0) {
            orc_emission(N, M, D, T, Xu, c, mean, inv_var, det, b, post);
            orc_forward(N, T, A, b, alpha, scale);
            orc_backward(N, T, A, b, scale, beta);
            orc_acc_trans(N, T, delta, A, b, alpha, beta, scale, num_a, den_a);
            orc_acc_den_mix(N, T, alpha, beta, scale, den_c);
            for (int t = 0; t < T; t++)
                orc_acc_mix_frame(N, M, D, Xu + (size_t)t * D, alpha + (size_t)t * N,
                                  beta + (size_t)t * N, scale[t], post + (size_t)t * G, mean,
                                  num_c, num_mu, num_var);
            double p = orc_loglik(T, scale, alpha[(size_t)(T - 1) * N + (N - 1)]);
            *ll += p;
            if (o_loglik) o_loglik[u] = p;
            if (o_b) memcpy(o_b + f0 * N, b, sizeof(double) * (size_t)T * N);
            if (o_post) memcpy(o_post + f0 * G, post, sizeof(double) * (size_t)T * G);
            if (o_alpha) memcpy(o_alpha + f0 * N, alpha, sizeof(double) * (size_t)T * N);
            if (o_beta) memcpy(o_beta + f0 * N, beta, sizeof(double) * (size_t)T * N);
            if (o_scale) memcpy(o_scale + f0, scale, sizeof(double) * (size_t)T);
        }
        *nu += 1.0;
        f0 += (size_t)T;
    }
    free(b); free(post); free(alpha); free(beta); free(scale);
    return 0;
}

/* The same E-step with the utterances dealt to `threads` host threads in contiguous blocks and
 * the threads' accumulator vectors added in thread order (the optional all-core CPU
 * figure beside the single-thread one; the reference itself is single-threaded).  No per-frame
 * dumps.  Returns 0, or 1 on failure. */
#include <pthread.h>
struct orc_mt_job {
    int N, M, D, delta, n_utt, rc;
    const double *A, *c, *mean, *inv_var, *det, *X;
    const int32_t *len;
    double *stats;
};
static void *orc_mt_run(void *arg)
{
    struct orc_mt_job *j = (struct orc_mt_job *)arg;
    j->rc = orc_estep(j->N, j->M, j->D, j->delta, j->A, j->c, j->mean, j->inv_var, j->det, j->X, j->len,
                      j->n_utt, j->stats, NULL, NULL, NULL, NULL, NULL, NULL);
    return NULL;
}
int orc_estep_mt(int threads, int N, int M, int D, int delta, const double *A, const double *c,
                 const double *mean, const double *inv_var, const double *det, const double *X,
                 const int32_t *len, int n_utt, double *stats)
{
    if (threads < 1) threads = 1;
    if (threads > n_utt) threads = n_utt > 0 ? n_utt : 1;
    size_t ns = orc_stats_len(N, M, D);
    struct orc_mt_job *jobs = calloc((size_t)threads, sizeof *jobs);
    pthread_t *tid = calloc((size_t)threads, sizeof *tid);
    double *all = calloc((size_t)threads * ns, sizeof(double));
    if (!jobs || !tid || !all) return 1;
    size_t f0 = 0;
    int u0 = 0, bad = 0;
    for (int t = 0; t < threads; t++) {
        int u1 = (int)((long long)n_utt * (t + 1) / threads);
        struct orc_mt_job *j = &jobs[t];
        j->N = N; j->M = M; j->D = D; j->delta = delta; j->A = A; j->c = c; j->mean = mean;
        j->inv_var = inv_var; j->det = det; j->X = X + f0 * D; j->len = len + u0; j->n_utt = u1 - u0;
        j->stats = all + (size_t)t * ns;
        for (int u = u0; u < u1; u++) f0 += (size_t)len[u];
        u0 = u1;
        if (pthread_create(&tid[t], NULL, orc_mt_run, j)) bad = 1, j->n_utt = -1;
    }
    memset(stats, 0, sizeof(double) * ns);
    for (int t = 0; t < threads; t++) {
        if (jobs[t].n_utt >= 0) pthread_join(tid[t], NULL);
        if (jobs[t].n_utt < 0 || jobs[t].rc) bad = 1;
        for (size_t k = 0; k < ns; k++) stats[k] += all[(size_t)t * ns + k];
    }
    free(jobs); free(tid); free(all);
    return bad;
}

/* M-step from a stats vector, TF:332-346; model updated in place */
void orc_mstep(int N, int M, int D, const double *stats, double *A, double *c, double *mean,
               double *inv_var, double *det)
{
    int G = N * M;
    const double *num_a = stats, *den_a = num_a + (size_t)N * N, *den_c = den_a + N;
    const double *num_c = den_c + N, *num_mu = num_c + G, *num_var = num_mu + (size_t)G * D;
    orc_update_trans(N, num_a, den_a, A);
    orc_update_mix(N, M, D, den_c, num_c, num_mu, num_var, c, mean, inv_var, det);
}

/* The EM driver, TF:238-358: old = 1.0 (TF:151); an M-step only when the
 * relative change exceeds `threshold` (1e-3, TF:37); the statistics of the
 * converging pass are discarded.  max_iter <= 0 means "until converged";
 * fixed_iter != 0 runs exactly max_iter E+M steps regardless of the test
 * (benchmark mode).  Returns the iteration count; *mean_loglik = probab/U. */
int orc_train(int N, int M, int D, int delta, double threshold, int max_iter, int fixed_iter,
              double *A, double *c, double *mean, double *inv_var, double *det, const double *X,
              const int32_t *len, int n_utt, double *mean_loglik, double *loglik_trace)
{
    double *stats = malloc(sizeof(double) * orc_stats_len(N, M, D));
    if (!stats) return -1;
    double old = 1.0, probab = 0.0, var;
    int it = 0, more;
    do {
        it++;
        if (orc_estep(N, M, D, delta, A, c, mean, inv_var, det, X, len, n_utt, stats, NULL,
                      NULL, NULL, NULL, NULL, NULL)) {
            free(stats);
            return -1;
        }
        probab = stats[orc_stats_len(N, M, D) - 2];
        if (loglik_trace) loglik_trace[it - 1] = probab;
        var = fabs((old - probab) / old);
        if (fixed_iter || var > threshold) {
            old = probab;
            orc_mstep(N, M, D, stats, A, c, mean, inv_var, det);
        }
        if (fixed_iter)
            more = it < max_iter;
        else
            more = var > threshold && (max_iter <= 0 || it < max_iter);
    } while (more);
    if (mean_loglik) *mean_loglik = probab / (double)n_utt;
    free(stats);
    return it;
}

/* ------------------------------------------------- several feature streams
 * param_number P > 1: every recursion takes the product over streams of the emission
 * densities, `product = 1.0; for (l < P) product *= symbol_probab[l][i][t]` (calc_alpha
 * TF:1406-1409 / 1429-1432, calc_beta TF:1501-1504, calc_transition_probab TF:1607-1610), i.e.
 * b_i(t) = ((b^0 b^1) b^2) ... in stream order; calc_symbol_probab (TF:278-288) and calc_mix_param
 * (TF:306-315) run once per stream with that stream's own mixtures and posteriors; the
 * transition sums, den_c (TF:300), log P and the exemplar count are common.  stats[p] has the
 * single-stream layout for (N, M[p], D[p]); the common entries are written into every one. */
int orc_estep_streams(int P, int N, const int *M, const int *D, int delta, const double *A,
                      const double *const *c, const double *const *mean,
                      const double *const *inv_var, const double *const *det,
                      const double *const *X, const int32_t *len, int n_utt, double *const *stats,
                      double *o_b, double *o_loglik)
{
    int Tmax = 0, Gmax = 0;
    for (int u = 0; u < n_utt; u++)
        if (len[u] > Tmax) Tmax = len[u];
    for (int p = 0; p < P; p++)
        if (N * M[p] > Gmax) Gmax = N * M[p];
    double *b = malloc(sizeof(double) * (size_t)Tmax * N);
    double *bp = malloc(sizeof(double) * (size_t)Tmax * N);
    double **post = malloc(sizeof(double *) * (size_t)P);
    double *alpha = malloc(sizeof(double) * (size_t)Tmax * N);
    double *beta = malloc(sizeof(double) * (size_t)Tmax * N);
    double *scale = malloc(sizeof(double) * (size_t)Tmax);
    int bad = !b || !bp || !post || !alpha || !beta || !scale;
    for (int p = 0; post && p < P; p++) {
        post[p] = malloc(sizeof(double) * (size_t)Tmax * N * M[p]);
        if (!post[p]) bad = 1;
    }
    if (bad) return 1; /* (test infrastructure: the leak on this path is accepted) */
    for (int p = 0; p < P; p++) memset(stats[p], 0, sizeof(double) * orc_stats_len(N, M[p], D[p]));
    /* the common sums are accumulated in stats[0] and copied at the end */
    double *num_a = stats[0], *den_a = num_a + (size_t)N * N, *den_c = den_a + N;
    double ll = 0.0, nu = 0.0;
    size_t f0 = 0;
    for (int u = 0; u < n_utt; u++) {
        int T = len[u];
        if (T > 0) {
            for (int p = 0; p < P; p++) {
                orc_emission(N, M[p], D[p], T, X[p] + f0 * D[p], c[p], mean[p], inv_var[p], det[p],
                             p == 0 ? b : bp, post[p]);
                if (p > 0)
                    for (size_t k = 0; k < (size_t)T * N; k++) b[k] *= bp[k];
            }
            orc_forward(N, T, A, b, alpha, scale);
            orc_backward(N, T, A, b, scale, beta);
            orc_acc_trans(N, T, delta, A, b, alpha, beta, scale, num_a, den_a);
            orc_acc_den_mix(N, T, alpha, beta, scale, den_c);
            for (int p = 0; p < P; p++) {
                int G = N * M[p];
                double *num_c = stats[p] + (size_t)N * N + 2 * (size_t)N, *num_mu = num_c + G;
                double *num_var = num_mu + (size_t)G * D[p];
                for (int t = 0; t < T; t++)
                    orc_acc_mix_frame(N, M[p], D[p], X[p] + (f0 + (size_t)t) * D[p],
                                      alpha + (size_t)t * N, beta + (size_t)t * N, scale[t],
                                      post[p] + (size_t)t * G, mean[p], num_c, num_mu, num_var);
            }
            double pr = orc_loglik(T, scale, alpha[(size_t)(T - 1) * N + (N - 1)]);
            ll += pr;
            if (o_loglik) o_loglik[u] = pr;
            if (o_b) memcpy(o_b + f0 * N, b, sizeof(double) * (size_t)T * N);
        }
        nu += 1.0;
        f0 += (size_t)T;
    }
    for (int p = 0; p < P; p++) {
        size_t n = orc_stats_len(N, M[p], D[p]);
        if (p > 0) memcpy(stats[p], stats[0], sizeof(double) * ((size_t)N * N + 2 * (size_t)N));
        stats[p][n - 2] = ll;
        stats[p][n - 1] = nu;
    }
    for (int p = 0; p < P; p++) free(post[p]);
    free(b); free(bp); free(post); free(alpha); free(beta); free(scale);
    return 0;
}

/* EM driver for P streams, TF:238-358: one transition update, one updating_mix_param + calc_det +
 * inv_matrix per stream (TF:332-346).  Model arrays updated in place; A is common. */
int orc_train_streams(int P, int N, const int *M, const int *D, int delta, double threshold,
                      int max_iter, int fixed_iter, double *A, double *const *c, double *const *mean,
                      double *const *inv_var, double *const *det, const double *const *X,
                      const int32_t *len, int n_utt, double *mean_loglik, double *loglik_trace)
{
    double **stats = malloc(sizeof(double *) * (size_t)P);
    if (!stats) return -1;
    for (int p = 0; p < P; p++) {
        stats[p] = malloc(sizeof(double) * orc_stats_len(N, M[p], D[p]));
        if (!stats[p]) return -1;
    }
    double old = 1.0, probab = 0.0, var;
    int it = 0, more;
    do {
        it++;
        if (orc_estep_streams(P, N, M, D, delta, A, (const double *const *)c, (const double *const *)mean,
                              (const double *const *)inv_var, (const double *const *)det, X, len, n_utt,
                              stats, NULL, NULL))
            return -1;
        probab = stats[0][orc_stats_len(N, M[0], D[0]) - 2];
        if (loglik_trace) loglik_trace[it - 1] = probab;
        var = fabs((old - probab) / old);
        if (fixed_iter || var > threshold) {
            old = probab;
            const double *num_a = stats[0], *den_a = num_a + (size_t)N * N, *den_c = den_a + N;
            orc_update_trans(N, num_a, den_a, A);
            for (int p = 0; p < P; p++) {
                int G = N * M[p];
                const double *num_c = stats[p] + (size_t)N * N + 2 * (size_t)N, *num_mu = num_c + G;
                const double *num_var = num_mu + (size_t)G * D[p];
                orc_update_mix(N, M[p], D[p], den_c, num_c, num_mu, num_var, c[p], mean[p],
                               inv_var[p], det[p]);
            }
        }
        if (fixed_iter)
            more = it < max_iter;
        else
            more = var > threshold && (max_iter <= 0 || it < max_iter);
    } while (more);
    if (mean_loglik) *mean_loglik = probab / (double)n_utt;
    for (int p = 0; p < P; p++) free(stats[p]);
    free(stats);
    return it;
}

/* Recogniser score of one utterance under one P-stream model, RF:349-366 */
double orc_score_streams(int P, int N, const int *M, const int *D, int T, const double *A,
                         const double *const *c, const double *const *mean,
                         const double *const *inv_var, const double *const *det,
                         const double *const *X)
{
    double *b = malloc(sizeof(double) * (size_t)T * N);
    double *bp = malloc(sizeof(double) * (size_t)T * N);
    double *alpha = malloc(sizeof(double) * (size_t)T * N);
    double *scale = malloc(sizeof(double) * (size_t)T);
    double pr = NAN;
    if (b && bp && alpha && scale) {
        for (int p = 0; p < P; p++) {
            orc_emission(N, M[p], D[p], T, X[p], c[p], mean[p], inv_var[p], det[p], p == 0 ? b : bp, NULL);
            if (p > 0)
                for (size_t k = 0; k < (size_t)T * N; k++) b[k] *= bp[k];
        }
        orc_forward(N, T, A, b, alpha, scale);
        pr = orc_loglik(T, scale, alpha[(size_t)(T - 1) * N + (N - 1)]);
    }
    free(b); free(bp); free(alpha); free(scale);
    return pr;
}

/* Recogniser score of one utterance under one model, RF:354-366 */
double orc_score(int N, int M, int D, int T, const double *A, const double *c,
                 const double *mean, const double *inv_var, const double *det, const double *X)
{
    double *b = malloc(sizeof(double) * (size_t)T * N);
    double *alpha = malloc(sizeof(double) * (size_t)T * N);
    double *scale = malloc(sizeof(double) * (size_t)T);
    double p = NAN;
    if (b && alpha && scale) {
        orc_emission(N, M, D, T, X, c, mean, inv_var, det, b, NULL);
        orc_forward(N, T, A, b, alpha, scale);
        p = orc_loglik(T, scale, alpha[(size_t)(T - 1) * N + (N - 1)]);
    }
    free(b); free(alpha); free(scale);
    return p;
}

/* sorting_probab, RF:968-995: bubble sort of indices, descending, `<` only
 * (NaNs never move) */
void orc_sort_scores(int n, const double *score, int *index)
{
    int done = 0;
    for (int i = 0; i < n; i++) index[i] = i;
    while (!done) {
        done = 1;
        for (int i = 0; i < n - 1; i++)
            if (score[index[i]] < score[index[i + 1]]) {
                int aux = index[i];
                index[i] = index[i + 1];
                index[i + 1] = aux;
                done = 0;
            }
    }
}

/* ------------------------------------------------------------------ Viterbi
 * NOT in the reference (parity unpinned by it).  Definition: the max-plus
 * analogue of calc_alpha with the same one-hot start (RF:249-251) and the same
 * final-state termination as calc_beta / calc_probability (TF:1487, TF:1549):
 *   logb_j(t) = log sum_k c_jk N_jk(x_t), evaluated as m + log(sum exp(e_k - m))
 *   delta_0(j) = (j == 0 ? 0 : -inf) + logb_j(0)
 *   delta_t(j) = max_i (delta_{t-1}(i) + log a_ij) + logb_j(t), ties -> lowest i
 *   score = delta_{T-1}(N-1), path by back-pointers from state N-1.
 * log a_ij = -inf where a_ij == 0.
 */
void orc_log_emission(int N, int M, int D, int T, const double *X, const double *c,
                      const double *mean, const double *inv_var, const double *det, double *logb)
{
    double lognorm0 = 0.5 * D * log(2.0 * M_PI);
    for (int t = 0; t < T; t++) {
        const double *x = X + (size_t)t * D;
        for (int i = 0; i < N; i++) {
            double e[1024], m = -INFINITY;
            for (int j = 0; j < M; j++) {
                int g = i * M + j;
                double aux = 0.0;
                for (int k = 0; k < D; k++) {
                    double dif = x[k] - mean[(size_t)g * D + k];
                    aux += dif * inv_var[(size_t)g * D + k] * dif;
                }
                e[j] = log(c[g]) - lognorm0 - 0.5 * log(fabs(det[g])) - 0.5 * aux;
                if (e[j] > m) m = e[j];
            }
            double s = 0.0;
            for (int j = 0; j < M; j++) s += exp(e[j] - m);
            logb[(size_t)t * N + i] = (m == -INFINITY) ? -INFINITY : m + log(s);
        }
    }
}

double orc_viterbi_lattice(int N, int T, const double *A, const double *logb, int32_t *path)
{
    double *dl = malloc(sizeof(double) * 2 * (size_t)N);
    double *la = malloc(sizeof(double) * (size_t)N * N);
    int32_t *psi = malloc(sizeof(int32_t) * (size_t)T * N);
    double score = NAN;
    if (dl && la && psi) {
        for (int i = 0; i < N * N; i++) la[i] = A[i] > 0.0 ? log(A[i]) : -INFINITY;
        double *prev = dl, *cur = dl + N;
        for (int j = 0; j < N; j++) {
            prev[j] = (j == 0 ? 0.0 : -INFINITY) + logb[j];
            psi[j] = 0;
        }
        for (int t = 1; t < T; t++) {
            for (int j = 0; j < N; j++) {
                double best = -INFINITY;
                int arg = 0;
                for (int i = 0; i < N; i++) {
                    double v = prev[i] + la[i * N + j];
                    if (v > best) { best = v; arg = i; }
                }
                cur[j] = best + logb[(size_t)t * N + j];
                psi[(size_t)t * N + j] = arg;
            }
            double *tmp = prev; prev = cur; cur = tmp;
        }
        score = prev[N - 1];
        int s = N - 1;
        for (int t = T - 1; t >= 0; t--) {
            path[t] = s;
            s = psi[(size_t)t * N + s];
        }
    }
    free(dl); free(la); free(psi);
    return score;
}

double orc_viterbi(int N, int M, int D, int T, const double *A, const double *c,
                   const double *mean, const double *inv_var, const double *det, const double *X,
                   int32_t *path)
{
    double *logb = malloc(sizeof(double) * (size_t)T * N);
    if (!logb) return NAN;
    orc_log_emission(N, M, D, T, X, c, mean, inv_var, det, logb);
    double s = orc_viterbi_lattice(N, T, A, logb, path);
    free(logb);
    return s;
}
