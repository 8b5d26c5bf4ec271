/*
 * ghmm_synth.c — deterministic synthetic corpus generator (host C, no libm-free tricks).
 *
 * The reference ships thirteen 9-d utterances and nothing at BASELINE sizes
 * (SURVEY.md §8(d) "synthetic generator"), so benchmark and parity inputs come from
 * this generator: a ground-truth left-to-right HMM whose state/mixture means are
 * N(0, 2^2) per dimension and whose standard deviations are U[0.5, 1.5]; every
 * utterance visits states 0..N-1 in order with random segment lengths (>= 5 frames
 * when T allows) and picks a mixture uniformly per frame.
 *
 * Utterance u of a corpus depends only on (seed, u): shards of one conceptual corpus
 * can be generated independently on every rank.
 *
 * RNG: splitmix64 -> xoshiro256**, Box-Muller (cos branch).  Default seed 20260104.
 */
#include "ghmm.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { uint64_t s[4]; } rng_t;

static uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static void rng_seed(rng_t *r, uint64_t seed)
{
    uint64_t x = seed;
    for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&x);
}

static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static uint64_t rng_next(rng_t *r)
{
    uint64_t *s = r->s;
    uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}

/* uniform in (0,1): 53 random bits, never exactly 0 */
static double rng_uniform(rng_t *r)
{
    return ((double)(rng_next(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

static double rng_normal(rng_t *r)
{
    double u1 = rng_uniform(r), u2 = rng_uniform(r);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}

int ghmm_synth_truth(uint64_t seed, int N, int M, int D, double *mean, double *stddev)
{
    if (N <= 0 || M <= 0 || D <= 0 || !mean || !stddev) return GHMM_ERR_ARG;
    rng_t r;
    rng_seed(&r, seed ^ 0x7472757468ull /* "truth" */);
    size_t n = (size_t)N * M * D;
    for (size_t i = 0; i < n; i++) mean[i] = 2.0 * rng_normal(&r);
    for (size_t i = 0; i < n; i++) stddev[i] = 0.5 + rng_uniform(&r);
    return GHMM_OK;
}

int ghmm_synth_utterances(uint64_t seed, int N, int M, int D, const double *mean,
                          const double *stddev, int64_t first_utt, int n_utt,
                          const int32_t *len, double *X)
{
    if (N <= 0 || M <= 0 || D <= 0 || n_utt < 0 || !mean || !stddev || !len || !X)
        return GHMM_ERR_ARG;
    int *seg = (int *)malloc(sizeof(int) * (size_t)N);
    if (!seg) return GHMM_ERR_ALLOC;
    double *x = X;
    for (int u = 0; u < n_utt; u++) {
        int T = len[u];
        if (T < 0) { free(seg); return GHMM_ERR_ARG; }
        rng_t r;
        rng_seed(&r, seed + 0x9E3779B97F4A7C15ull * (uint64_t)(first_utt + u + 1));
        /* segment lengths: a floor for every state, the rest dealt at random */
        int base = T / N < 5 ? T / N : 5;
        int rest = T - base * N;
        for (int i = 0; i < N; i++) seg[i] = base;
        for (int k = 0; k < rest; k++) seg[rng_next(&r) % (uint64_t)N]++;
        for (int i = 0; i < N; i++) {
            for (int k = 0; k < seg[i]; k++) {
                int m = (int)(rng_next(&r) % (uint64_t)M);
                const double *mu = mean + ((size_t)i * M + m) * D;
                const double *sd = stddev + ((size_t)i * M + m) * D;
                for (int d = 0; d < D; d++) x[d] = mu[d] + sd[d] * rng_normal(&r);
                x += D;
            }
        }
    }
    free(seg);
    return GHMM_OK;
}

int ghmm_synth_start_model(uint64_t seed, int N, int M, int D, const double *mean,
                           const double *stddev, double perturb, double *A, double *c,
                           double *mu0, double *inv_var0, double *det0)
{
    if (N <= 0 || M <= 0 || D <= 0 || !mean || !stddev || !A || !c || !mu0 || !inv_var0 || !det0)
        return GHMM_ERR_ARG;
    rng_t r;
    rng_seed(&r, seed ^ 0x7374617274ull /* "start" */);
    /* left-to-right, one-step topology, same shape as the reference's own
       init_transition_probab (TF:774): a_ii = a_i,i+1 = 1/2, a_NN = 1 */
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++)
            A[(size_t)i * N + j] = (j == i || j == i + 1) ? (i == N - 1 ? 1.0 : 0.5) : 0.0;
    for (int g = 0; g < N * M; g++) {
        c[g] = 1.0 / M;
        double det = 1.0;
        for (int d = 0; d < D; d++) {
            size_t k = (size_t)g * D + d;
            double sd = stddev[k] * (1.0 + perturb * (2.0 * rng_uniform(&r) - 1.0));
            mu0[k] = mean[k] + perturb * stddev[k] * (2.0 * rng_uniform(&r) - 1.0);
            double var = sd * sd;
            det *= var;              /* plain product, like calc_det TF:1976 */
            inv_var0[k] = 1.0 / var; /* like inv_matrix TF:2012 */
        }
        det0[g] = det;
    }
    return GHMM_OK;
}
