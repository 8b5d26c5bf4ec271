// ghmm_kernels.hpp — CDNA4 (gfx950) kernels of the GMM-HMM hot path, vector-ALU f64 tier.
//
// Every kernel cites the reference loop it replaces (TF = train/source/hmm-fs/
// hmm_continuous_fs.c, RF = test/source/recognition-fs/recognition_continuous_fs.c).
// Wavefront = 64 lanes everywhere; nothing here is warp-32 shaped.
//
// Layout in HBM (all f64, frame-major; F = total frames, G = N*M):
//   X[F][D]  b[F][N]  post[F][G]  alpha[F][N]  beta[F][N]  scale[F]  gamma[F][N]
//   off[U+1] (int64 frame offsets)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ghmm {

constexpr int WAVE = 64;
constexpr int MAX_DELTA = 7;       // widest transition band that receives statistics
constexpr double FLOOR = 1.0e-5;   // FINITE_PROBAB, TF:39

// ------------------------------------------------------------------ prepare
// Derived per-Gaussian constants, rebuilt whenever the model changes:
//   wk[g]    = c_g / (pow(2 pi, D/2) * sqrt(|det_g|))     (TF:1821-1836: gaus = e/(aux1*aux2), then *c)
//   logwk[g] = log(wk[g]); logA = log(A) (-inf where 0)   (robust emission / Viterbi only)
__global__ void k_prepare(int N, int M, const double *__restrict__ A, const double *__restrict__ c,
                          const double *__restrict__ det, double norm2pi, double *__restrict__ wk,
                          double *__restrict__ logwk, double *__restrict__ logA)
{
    int G = N * M;
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x) {
        double den = norm2pi * sqrt(fabs(det[g]));
        double w = c[g] / den;
        wk[g] = w;
        logwk[g] = log(c[g]) - log(den);
    }
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < N * N; k += gridDim.x * blockDim.x)
        logA[k] = A[k] > 0.0 ? log(A[k]) : -INFINITY;
}

// ----------------------------------------------------------------- emission
// calc_symbol_probab + calc_gaus (TF:1749-1841, RF:860-947).
// One wave = 64 consecutive frames (lane = frame), all N states.  The 64xD frame
// tile is read from HBM once with coalesced loads into LDS (row stride odd ->
// conflict-free ds_read_b64 per lane); mean / inverse variance are wave-uniform
// and come through the scalar cache.  MODE 0: linear densities exactly like the
// reference (exp underflows where the reference's does).  MODE 1 (robust): densities
// relative to the frame's largest exponent, lognorm[f] holds that exponent.
// MODE 2: log b (for the Viterbi lattice), evaluated as m + log(sum exp(e - m)).
template <int MODE>
__global__ void __launch_bounds__(WAVE)
k_emission(int N, int M, int D, long long F, const double *__restrict__ X,
           const double *__restrict__ mean, const double *__restrict__ inv_var,
           const double *__restrict__ wk, const double *__restrict__ logwk,
           double *__restrict__ b, double *__restrict__ post, double *__restrict__ lognorm)
{
    extern __shared__ double lds[];
    const int DS = D | 1;
    const int lane = threadIdx.x;
    const long long f0 = (long long)blockIdx.x * WAVE;
    const int nf = (int)((F - f0) < WAVE ? (F - f0) : WAVE);
    // coalesced tile load: the tile is nf*D contiguous doubles
    for (int k = lane; k < nf * D; k += WAVE) {
        int r = k / D, cidx = k - r * D;
        lds[r * DS + cidx] = X[f0 * D + k];
    }
    __syncthreads();
    if (lane >= nf) return;
    const double *x = lds + lane * DS;
    const long long f = f0 + lane;
    const int G = N * M;
    double *pf = post ? post + f * G : nullptr;

    if (MODE == 0) {
        for (int i = 0; i < N; i++) {
            double bi = 0.0;
            for (int j = 0; j < M; j++) {
                const int g = i * M + j;
                const double *mu = mean + (size_t)g * D, *iv = inv_var + (size_t)g * D;
                double aux = 0.0;
                for (int d = 0; d < D; d++) {
                    double dif = x[d] - mu[d];
                    aux += dif * iv[d] * dif;
                }
                double v = exp(-0.5 * aux) * wk[g];
                if (pf) pf[g] = v;
                bi += v;
            }
            b[f * N + i] = bi;
            if (pf) {
                // gauss[i][j] /= b_i, or 0 when b_i == 0 (TF:1773-1778)
                // a true division: b_i can be subnormal, where 1/b_i overflows
                for (int j = 0; j < M; j++) {
                    double v = pf[i * M + j];
                    pf[i * M + j] = bi != 0.0 ? v / bi : 0.0;
                }
            }
        }
    } else {
        // pass 1: exponents e_g = log(wk_g) - maha/2, parked in post[] (or recomputed)
        double mx = -INFINITY;
        for (int i = 0; i < N; i++) {
            double mi = -INFINITY;
            for (int j = 0; j < M; j++) {
                const int g = i * M + j;
                const double *mu = mean + (size_t)g * D, *iv = inv_var + (size_t)g * D;
                double aux = 0.0;
                for (int d = 0; d < D; d++) {
                    double dif = x[d] - mu[d];
                    aux += dif * iv[d] * dif;
                }
                double e = logwk[g] - 0.5 * aux;
                if (pf) pf[g] = e;
                mi = e > mi ? e : mi;
            }
            b[f * N + i] = mi; // per-state max, replaced below
            mx = mi > mx ? mi : mx;
        }
        if (MODE == 1 && lognorm) lognorm[f] = mx;
        for (int i = 0; i < N; i++) {
            const double mi = b[f * N + i];
            const double ref = (MODE == 1) ? mx : mi;
            double s = 0.0;
            for (int j = 0; j < M; j++) {
                const int g = i * M + j;
                double e;
                if (pf) e = pf[g];
                else {
                    const double *mu = mean + (size_t)g * D, *iv = inv_var + (size_t)g * D;
                    double aux = 0.0;
                    for (int d = 0; d < D; d++) {
                        double dif = x[d] - mu[d];
                        aux += dif * iv[d] * dif;
                    }
                    e = logwk[g] - 0.5 * aux;
                }
                double v = (ref == -INFINITY) ? 0.0 : exp(e - ref);
                if (pf) pf[g] = v;
                s += v;
            }
            if (MODE == 1) {
                b[f * N + i] = s;
            } else {
                b[f * N + i] = (mi == -INFINITY) ? -INFINITY : mi + log(s);
            }
            if (pf) {
                for (int j = 0; j < M; j++) pf[i * M + j] = s != 0.0 ? pf[i * M + j] / s : 0.0;
            }
        }
    }
}

// ------------------------------------------------------------ group helpers
// A "group" is L consecutive lanes (L = 16 or 64) that own one utterance, lane i =
// state i.  Cross-lane traffic stays inside the group.
template <int L> __device__ inline double group_sum(double v)
{
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, L);
    return v;
}

constexpr int PF = 8; // frames of b / alpha prefetched ahead of the serial recursion

// ------------------------------------------------------------------ forward
// calc_alpha (TF:1380-1443 = RF:739-799) with pi = one-hot at state 0 (TF:232-234)
// and calc_probability (TF:1536-1553).  One group per utterance; the time loop is
// serial, utterances are the parallel axis (SURVEY.md §5 "long-context").
//   alpha_t(i) = (sum_j alpha^_{t-1}(j) a_ji) b_i(t);  c_t = 1/sum_i alpha_t(i);  alpha^ = alpha c_t
//   log P = -sum_t log c_t + log alpha^_{T-1}(N-1)   [+ sum_t lognorm_t in robust mode]
template <int L>
__global__ void __launch_bounds__(WAVE)
k_forward(int N, int U, const double *__restrict__ A, const double *__restrict__ b,
          const long long *__restrict__ off, double *__restrict__ alpha,
          double *__restrict__ scale, const double *__restrict__ lognorm,
          double *__restrict__ loglik)
{
    const int u = blockIdx.x * (WAVE / L) + threadIdx.x / L;
    const int i = threadIdx.x % L;
    if (u >= U) return;
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    if (T <= 0) {
        if (i == 0) loglik[u] = 0.0;
        return;
    }
    const bool act = i < N;
    double acol[L]; // column i of A: acol[j] = a_ji
#pragma unroll
    for (int j = 0; j < L; j++) acol[j] = (act && j < N) ? A[j * N + i] : 0.0;
    bool offband = false;
#pragma unroll
    for (int j = 0; j < L; j++) offband |= (acol[j] != 0.0 && j != i && j != i - 1);
    const bool banded = !__any(offband);
    const double a_self = act ? A[i * N + i] : 0.0;
    const double a_prev = (act && i > 0) ? A[(i - 1) * N + i] : 0.0;

    const double *bu = b + f0 * N;
    double *au = alpha + f0 * N;
    double *su = scale + f0;

    double bq[PF], bn[PF];
    double a = ((i == 0) ? 1.0 : 0.0) * (act ? bu[i] : 0.0);
    {
        double s = group_sum<L>(a);
        double c = 1.0 / s;
        a *= c;
        if (act) au[i] = a;
        if (i == 0) su[0] = c;
    }
#pragma unroll
    for (int k = 0; k < PF; k++) bq[k] = (act && 1 + k < T) ? bu[(size_t)(1 + k) * N + i] : 0.0;
    for (int tb = 1; tb < T; tb += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++)
            bn[k] = (act && tb + PF + k < T) ? bu[(size_t)(tb + PF + k) * N + i] : 0.0;
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int t = tb + k;
            if (t < T) {
                double aux = 0.0;
                if (banded) {
                    double up = __shfl_up(a, 1, L);
                    aux = (i > 0 ? up * a_prev : 0.0) + a * a_self;
                } else {
#pragma unroll
                    for (int j = 0; j < L; j++)
                        if (j < N) aux += __shfl(a, j, L) * acol[j];
                }
                double v = aux * bq[k];
                double s = group_sum<L>(v);
                double c = 1.0 / s;
                a = v * c;
                if (act) au[(size_t)t * N + i] = a;
                if (i == 0) su[t] = c;
            }
        }
#pragma unroll
        for (int k = 0; k < PF; k++) bq[k] = bn[k];
    }
    // log P: the T logs are spread over the group's lanes instead of a serial loop
    __threadfence_block();
    double lp = 0.0;
    for (int t = i; t < T; t += L) {
        lp -= log(su[t]);
        if (lognorm) lp += lognorm[f0 + t];
    }
    lp = group_sum<L>(lp);
    double last = __shfl(a, N - 1, L);
    if (i == 0) loglik[u] = lp + log(last);
}

// ----------------------------------------------------------------- backward
// calc_beta (TF:1463-1516) fused with the per-utterance sums of
// calc_transition_probab (TF:1577-1620) and calc_den_mix_coef (TF:1642-1664):
//   beta^_{T-1}(i) = [i == N-1] c_{T-1};  beta^_t(i) = c_t sum_j beta^_{t+1}(j) a_ij b_j(t+1)
//   gamma_t(i) = alpha^_t(i) beta^_t(i) / c_t                                  -> gamma[F][N]
//   xi[u][i][o] = sum_{t<T-1} alpha^_t(i) a_{i,i+o} b_{i+o}(t+1) beta^_{t+1}(i+o),  o = 0..delta
//   dena[u][i] = sum_{t<T-1} gamma_t(i);   denc[u][i] = sum_{t<T} gamma_t(i)
// Per-utterance partial sums are written out and added in utterance order by
// k_reduce (bitwise reproducible; no atomics).
template <int L>
__global__ void __launch_bounds__(WAVE)
k_backward(int N, int U, int delta, const double *__restrict__ A, const double *__restrict__ b,
           const long long *__restrict__ off, const double *__restrict__ alpha,
           const double *__restrict__ scale, double *__restrict__ beta,
           double *__restrict__ gamma, double *__restrict__ part_xi,
           double *__restrict__ part_dena, double *__restrict__ part_denc)
{
    const int u = blockIdx.x * (WAVE / L) + threadIdx.x / L;
    const int i = threadIdx.x % L;
    if (u >= U) return;
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    const bool act = i < N;
    double xi[MAX_DELTA + 1];
#pragma unroll
    for (int o = 0; o <= MAX_DELTA; o++) xi[o] = 0.0;
    double dena = 0.0, denc = 0.0;
    if (T > 0) {
        double arow[L]; // row i of A
#pragma unroll
        for (int j = 0; j < L; j++) arow[j] = (act && j < N) ? A[i * N + j] : 0.0;
        bool offband = false;
#pragma unroll
        for (int j = 0; j < L; j++) offband |= (arow[j] != 0.0 && j != i && j != i + 1);
        const bool banded = !__any(offband);
        const double a_self = act ? A[i * N + i] : 0.0;
        const double a_next = (act && i + 1 < N) ? A[i * N + i + 1] : 0.0;
        double aband[MAX_DELTA + 1]; // a_{i,i+o}
#pragma unroll
        for (int o = 0; o <= MAX_DELTA; o++)
            aband[o] = (act && i + o < N && o <= delta) ? A[i * N + i + o] : 0.0;

        const double *bu = b + f0 * N;
        const double *au = alpha + f0 * N;
        const double *su = scale + f0;
        double *beu = beta + f0 * N;
        double *gu = gamma + f0 * N;

        // t = T-1
        double cT = su[T - 1];
        double be = (i == N - 1) ? 1.0 * cT : 0.0;
        {
            double al = act ? au[(size_t)(T - 1) * N + i] : 0.0;
            double g = al * be / cT;
            if (act) {
                beu[(size_t)(T - 1) * N + i] = be;
                gu[(size_t)(T - 1) * N + i] = g;
            }
            denc += g;
        }
        // queues for step t (descending): b[t+1][i], alpha[t][i], c[t]
        double qb[PF], qa[PF], qc[PF], nb[PF], na[PF], nc[PF];
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int t = T - 2 - k;
            const bool ok = t >= 0;
            qb[k] = (act && ok) ? bu[(size_t)(t + 1) * N + i] : 0.0;
            qa[k] = (act && ok) ? au[(size_t)t * N + i] : 0.0;
            qc[k] = ok ? su[t] : 1.0;
        }
        for (int tb = T - 2; tb >= 0; tb -= PF) {
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const int t = tb - PF - k;
                const bool ok = t >= 0;
                nb[k] = (act && ok) ? bu[(size_t)(t + 1) * N + i] : 0.0;
                na[k] = (act && ok) ? au[(size_t)t * N + i] : 0.0;
                nc[k] = ok ? su[t] : 1.0;
            }
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const int t = tb - k;
                if (t >= 0) {
                    const double w = be * qb[k]; // beta^_{t+1}(i) b_i(t+1), lane i
                    double aux = 0.0;
                    if (banded) {
                        // upper-bidiagonal A: only j = i and j = i+1 contribute
                        double dn = __shfl_down(w, 1, L);
                        aux = a_self * w + (i + 1 < N ? a_next * dn : 0.0);
                    } else {
                        // general form: sum_j a_ij w_j, ascending j
#pragma unroll
                        for (int j = 0; j < L; j++)
                            if (j < N) aux += arow[j] * __shfl(w, j, L);
                    }
                    // xi band sums
#pragma unroll
                    for (int o = 0; o <= MAX_DELTA; o++)
                        if (o <= delta) {
                            double wj = (o == 0) ? w : __shfl_down(w, o, L);
                            if (i + o < N) xi[o] += qa[k] * aband[o] * wj;
                        }
                    be = aux * qc[k];
                    double g = qa[k] * be / qc[k];
                    if (act) {
                        beu[(size_t)t * N + i] = be;
                        gu[(size_t)t * N + i] = g;
                    }
                    dena += g;
                    denc += g;
                }
            }
#pragma unroll
            for (int k = 0; k < PF; k++) {
                qb[k] = nb[k];
                qa[k] = na[k];
                qc[k] = nc[k];
            }
        }
    }
    if (act) {
        for (int o = 0; o <= delta; o++) part_xi[((size_t)u * N + i) * (MAX_DELTA + 1) + o] = xi[o];
        part_dena[(size_t)u * N + i] = dena;
        part_denc[(size_t)u * N + i] = denc;
    }
}


// ---------------------------------------------------------------- mixstats
// calc_mix_param (TF:1691-1727) over a block of frames.  Element space E = G*(D+1):
// element (g, d<D) accumulates num_mu and num_var, element (g, D) (x := 1)
// accumulates num_c.  blockIdx.x = frame range, blockIdx.y = batch of 256*EPT
// elements; frames are staged through LDS FS at a time together with the weights
//   w_t(g) = gamma_t(state(g)) * post_t(g)                      (TF:1706-1711)
// and every thread keeps its EPT pairs of sums in registers.  The variance
// statistic is taken around the CURRENT mean (TF:1720-1722).  Each block writes its
// partial sums; k_reduce adds them in block order (bitwise reproducible).
constexpr int MS_THREADS = 256;
constexpr int MS_EPT = 8;
constexpr int MS_FS = 32; // frames staged per pass (fewer when a wide tile would not fit LDS)

__global__ void __launch_bounds__(MS_THREADS)
k_mixstats(int N, int M, int D, long long F, long long frames_per_block, int FS,
           const double *__restrict__ X, const double *__restrict__ gamma,
           const double *__restrict__ post, const double *__restrict__ mean,
           double *__restrict__ part_mu, double *__restrict__ part_var,
           const int *__restrict__ only_if)
{
    extern __shared__ double lds[];
    if (only_if && only_if[0] == 0) return; // matrix-core tier: nothing is ill-conditioned
    const int G = N * M, D1 = D + 1;
    const long long E = (long long)G * D1;
    const int tid = threadIdx.x;
    const long long e0 = (long long)blockIdx.y * (MS_THREADS * MS_EPT);
    const long long e1 = (e0 + MS_THREADS * MS_EPT < E) ? e0 + MS_THREADS * MS_EPT : E;
    const int g0 = (int)(e0 / D1);
    const int g1 = (int)((e1 - 1) / D1); // inclusive
    const int GW = g1 - g0 + 1;
    double *xs = lds;                // [FS][D1]
    double *ws = lds + FS * D1;      // [FS][GW]

    int gx[MS_EPT], dx[MS_EPT];
    double mu[MS_EPT], acc_mu[MS_EPT], acc_var[MS_EPT];
#pragma unroll
    for (int k = 0; k < MS_EPT; k++) {
        long long e = e0 + tid + (long long)k * MS_THREADS;
        bool ok = e < e1;
        int g = ok ? (int)(e / D1) : g0;
        int d = ok ? (int)(e - (long long)g * D1) : D;
        gx[k] = g - g0;
        dx[k] = d;
        mu[k] = (ok && d < D) ? mean[(size_t)g * D + d] : 0.0;
        acc_mu[k] = 0.0;
        acc_var[k] = 0.0;
    }
    const long long fb0 = (long long)blockIdx.x * frames_per_block;
    const long long fb1 = (fb0 + frames_per_block < F) ? fb0 + frames_per_block : F;
    for (long long fs = fb0; fs < fb1; fs += FS) {
        const int nf = (int)((fb1 - fs) < FS ? (fb1 - fs) : FS);
        __syncthreads();
        for (int k = tid; k < nf * D1; k += MS_THREADS) {
            int r = k / D1, d = k - r * D1;
            xs[k] = d < D ? X[(fs + r) * D + d] : 1.0;
        }
        for (int k = tid; k < nf * GW; k += MS_THREADS) {
            int r = k / GW, gl = k - r * GW;
            int g = g0 + gl;
            ws[k] = gamma[(fs + r) * N + g / M] * post[(fs + r) * G + g];
        }
        __syncthreads();
        for (int r = 0; r < nf; r++) {
#pragma unroll
            for (int k = 0; k < MS_EPT; k++) {
                double w = ws[r * GW + gx[k]];
                double x = xs[r * D1 + dx[k]];
                acc_mu[k] += w * x;
                double dif = x - mu[k];
                acc_var[k] += w * (dif * dif);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < MS_EPT; k++) {
        long long e = e0 + tid + (long long)k * MS_THREADS;
        if (e < e1) {
            part_mu[(size_t)blockIdx.x * E + e] = acc_mu[k];
            part_var[(size_t)blockIdx.x * E + e] = acc_var[k];
        }
    }
}

// ------------------------------------------------------------------- reduce
// Ordered sums of all partials into the flat statistics vector (layout:
// include/ghmm.h).  One block per output quantity group; every sum is taken in a
// fixed order (thread-strided ascending, then a fixed LDS tree), so the result
// does not depend on scheduling.
constexpr int RD_THREADS = 256;

__device__ inline double block_sum_fixed(double v, double *sh)
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = RD_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    double r = sh[0];
    __syncthreads();
    return r;
}

// grid.x = N*N (num_a) + N (den_a) + N (den_c) + 1 (loglik, n_utt)
__global__ void __launch_bounds__(RD_THREADS)
k_reduce_utt(int N, int U, int delta, const double *__restrict__ part_xi,
             const double *__restrict__ part_dena, const double *__restrict__ part_denc,
             const double *__restrict__ loglik, double *__restrict__ stats, size_t off_loglik)
{
    __shared__ double sh[RD_THREADS];
    const int q = blockIdx.x;
    double v = 0.0;
    if (q < N * N) {
        int i = q / N, j = q % N, o = j - i;
        if (o >= 0 && o <= delta)
            for (int u = threadIdx.x; u < U; u += RD_THREADS)
                v += part_xi[((size_t)u * N + i) * (MAX_DELTA + 1) + o];
        v = block_sum_fixed(v, sh);
        if (threadIdx.x == 0) stats[q] = v;
    } else if (q < N * N + N) {
        int i = q - N * N;
        for (int u = threadIdx.x; u < U; u += RD_THREADS) v += part_dena[(size_t)u * N + i];
        v = block_sum_fixed(v, sh);
        if (threadIdx.x == 0) stats[q] = v;
    } else if (q < N * N + 2 * N) {
        int i = q - N * N - N;
        for (int u = threadIdx.x; u < U; u += RD_THREADS) v += part_denc[(size_t)u * N + i];
        v = block_sum_fixed(v, sh);
        if (threadIdx.x == 0) stats[q] = v;
    } else {
        for (int u = threadIdx.x; u < U; u += RD_THREADS) v += loglik[u];
        v = block_sum_fixed(v, sh);
        if (threadIdx.x == 0) {
            stats[off_loglik] = v;
            stats[off_loglik + 1] = (double)U;
        }
    }
}

// one thread per element (g, d in 0..D); sums the P frame-block partials in order
__global__ void __launch_bounds__(RD_THREADS)
k_reduce_mix(int N, int M, int D, int P, const double *__restrict__ part_mu,
             const double *__restrict__ part_var, double *__restrict__ num_c,
             double *__restrict__ num_mu, double *__restrict__ num_var,
             const int *__restrict__ only_if)
{
    if (only_if && only_if[0] == 0) return;
    const int G = N * M, D1 = D + 1;
    const long long E = (long long)G * D1;
    long long e = (long long)blockIdx.x * RD_THREADS + threadIdx.x;
    if (e >= E) return;
    double sm = 0.0, sv = 0.0;
    for (int p = 0; p < P; p++) {
        sm += part_mu[(size_t)p * E + e];
        sv += part_var[(size_t)p * E + e];
    }
    int g = (int)(e / D1), d = (int)(e - (long long)g * D1);
    if (d < D) {
        num_mu[(size_t)g * D + d] = sm;
        num_var[(size_t)g * D + d] = sv;
    } else {
        num_c[g] = sm;
    }
}

// -------------------------------------------------------------------- mstep
// updating_transition_probab (TF:1862-1889), updating_mix_param (TF:1911-1955) with
// changing_zero_coef (TF:1338-1359), then calc_det (TF:1976) and inv_matrix
// (TF:2012) as main() chains them (TF:332-346) — including the reference's
// behaviour for a state whose den_c is 0 (its stored inverse variances go through
// det/inverse as if they were variances).  Single block: the model is tiny.
__global__ void __launch_bounds__(256)
k_mstep(int N, int M, int D, const double *__restrict__ stats, double *__restrict__ A,
        double *__restrict__ c, double *__restrict__ mean, double *__restrict__ inv_var,
        double *__restrict__ det)
{
    const int G = N * M;
    const double *num_a = stats, *den_a = num_a + (size_t)N * N, *den_c = den_a + N;
    const double *num_c = den_c + N, *num_mu = num_c + G, *num_var = num_mu + (size_t)G * D;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int k = tid; k < N * N; k += nt) {
        int i = k / N;
        if (den_a[i] != 0.0) A[k] = num_a[k] / den_a[i];
    }
    for (long long k = tid; k < (long long)G * D; k += nt) {
        int g = (int)(k / D), i = g / M;
        if (den_c[i] != 0.0) {
            mean[k] = num_mu[k] / num_c[g];
            double v = num_var[k] / num_c[g];
            if (v < FLOOR) v = FLOOR;
            inv_var[k] = v;
        }
    }
    for (int g = tid; g < G; g += nt) {
        int i = g / M;
        if (den_c[i] != 0.0) c[g] = num_c[g] / den_c[i];
    }
    __syncthreads();
    for (int i = tid; i < N; i += nt) {
        double sum = 0.0;
        for (int k = 0; k < M; k++) {
            double v = c[i * M + k];
            if (v < FLOOR) v = FLOOR;
            c[i * M + k] = v;
            sum += v;
        }
        for (int k = 0; k < M; k++) c[i * M + k] /= sum;
    }
    for (int g = tid; g < G; g += nt) {
        double d = 1.0;
        for (int k = 0; k < D; k++) d *= inv_var[(size_t)g * D + k];
        det[g] = d;
        for (int k = 0; k < D; k++) inv_var[(size_t)g * D + k] = 1.0 / inv_var[(size_t)g * D + k];
    }
}

// ------------------------------------------------------------------ viterbi
// Max-plus lattice (absent from the reference; definition in oracle/ghmm_oracle.c):
//   delta_0(j) = (j == 0 ? 0 : -inf) + logb_j(0)
//   delta_t(j) = max_i (delta_{t-1}(i) + log a_ij) + logb_j(t), ties -> lowest i
//   score = delta_{T-1}(N-1); path by back-pointers from state N-1.
template <int L>
__global__ void __launch_bounds__(WAVE)
k_viterbi(int N, int U, const double *__restrict__ logA, const double *__restrict__ logb,
          const long long *__restrict__ off, unsigned char *__restrict__ psi,
          int *__restrict__ path, double *__restrict__ score)
{
    const int u = blockIdx.x * (WAVE / L) + threadIdx.x / L;
    const int j = threadIdx.x % L;
    if (u >= U) return;
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    if (T <= 0) {
        if (j == 0) score[u] = 0.0;
        return;
    }
    const bool act = j < N;
    double lacol[L];
#pragma unroll
    for (int i = 0; i < L; i++) lacol[i] = (act && i < N) ? logA[i * N + j] : -INFINITY;
    const double *lb = logb + f0 * N;
    unsigned char *ps = psi + f0 * N;
    double d = ((j == 0) ? 0.0 : -INFINITY) + (act ? lb[j] : -INFINITY);
    if (act) ps[j] = 0;
    double q[PF], qn[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) q[k] = (act && 1 + k < T) ? lb[(size_t)(1 + k) * N + j] : 0.0;
    for (int tb = 1; tb < T; tb += PF) {
#pragma unroll
        for (int k = 0; k < PF; k++)
            qn[k] = (act && tb + PF + k < T) ? lb[(size_t)(tb + PF + k) * N + j] : 0.0;
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const int t = tb + k;
            if (t < T) {
                double best = -INFINITY;
                int arg = 0;
#pragma unroll
                for (int i = 0; i < L; i++)
                    if (i < N) {
                        double v = __shfl(d, i, L) + lacol[i];
                        if (v > best) {
                            best = v;
                            arg = i;
                        }
                    }
                d = best + q[k];
                if (act) ps[(size_t)t * N + j] = (unsigned char)arg;
            }
        }
#pragma unroll
        for (int k = 0; k < PF; k++) q[k] = qn[k];
    }
    double sc = __shfl(d, N - 1, L);
    __threadfence_block();
    if (j == 0) {
        score[u] = sc;
        int s = N - 1;
        for (int t = T - 1; t >= 0; t--) {
            path[f0 + t] = s;
            s = ps[(size_t)t * N + s];
        }
    }
}

} // namespace ghmm
