// ghmm_wide.hpp — forward / backward / Viterbi for models of more than 64 states (up to WIDE_MAX).
//
// The kernels of ghmm_kernels.hpp / ghmm_pair.hpp hold one state per lane; here one wave owns an
// utterance and every lane owns the states i = lane, lane + 64, ...  The state vectors of the
// previous step live in LDS (a wave's LDS operations complete in order, the barrier of a one-wave
// block is free), the operations and their order are the reference's own: calc_alpha (TF:1380-1443),
// calc_beta scaled by the forward pass's c_t (TF:1463-1516) with gamma and the xi / den sums on its
// chain (TF:1577-1664), exact divisions.  A band-diagonal A (a_ij = 0 unless j = i or i + 1: what
// the reference's trainer produces) takes the 2-term update; whether A is one is decided once per
// pass by k_wide_band.  As in k_backward, the band-only update next to an overflowed beta^ does not
// make the NaN row the reference's dense loop makes (inf * 0): such an utterance is taken again,
// dense, by the same wave.
#pragma once
#include "ghmm_kernels.hpp"

namespace ghmm {

constexpr int WIDE_MAX = 512; // states; LDS of the backward pass: (8 + MAX_DELTA + 1) N doubles = 64 KB (raised limit)

// flag[0] = 1 when some entry outside {j = i, j = i + 1} differs from `zero` (0 for A, -inf for log A)
__global__ void __launch_bounds__(256)
k_wide_band(int N, const double *__restrict__ A, double zero, int *__restrict__ flag)
{
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= (long long)N * N) return;
    const int i = (int)(k / N), j = (int)(k - (long long)i * N);
    if (j != i && j != i + 1 && A[k] != zero) flag[0] = 1; // (everybody writes the same value)
}

__device__ inline double wave_sum(double v) { return group_sum<WAVE>(v); }

// calc_alpha + calc_probability, one wave per utterance (longest first)
__global__ void __launch_bounds__(WAVE)
k_forward_wide(int N, int U, const double *__restrict__ A, const double *__restrict__ b,
               const long long *__restrict__ off, double *__restrict__ alpha, double *__restrict__ scale,
               double *__restrict__ sinv, const double *__restrict__ lognorm, double *__restrict__ loglik,
               const int *__restrict__ order, const int *__restrict__ offband, int bstride,
               const double *__restrict__ bbase)
{
    extern __shared__ double lds[]; // la[2][N] | aself[N] | aprev[N]
    const int l = threadIdx.x;
    const int u = order[blockIdx.x];
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    if (T <= 0) {
        if (l == 0) loglik[u] = 0.0;
        return;
    }
    const bool banded = offband[0] == 0;
    double *la = lds, *aself = lds + 2 * N, *aprev = lds + 3 * N;
    for (int i = l; i < N; i += WAVE) {
        aself[i] = A[(size_t)i * N + i];
        aprev[i] = i > 0 ? A[(size_t)(i - 1) * N + i] : 0.0;
    }
    const double *bu = bbase ? bbase + f0 * bstride : b + f0 * N;
    double *au = alpha ? alpha + f0 * N : nullptr, *su = scale + f0, *si = sinv ? sinv + f0 : nullptr;
    double lp = 0.0, last = 0.0;
    for (int t = 0; t < T; t++) {
        const double *prev = la + (size_t)((t + 1) & 1) * N;
        double *cur = la + (size_t)(t & 1) * N;
        double part = 0.0;
        for (int i = l; i < N; i += WAVE) {
            double aux;
            if (t == 0) {
                aux = i == 0 ? 1.0 : 0.0; // pi = one-hot at state 0 (TF:232-234)
            } else if (banded) {
                aux = prev[i] * aself[i] + (i > 0 ? prev[i - 1] : 0.0) * aprev[i];
            } else {
                aux = 0.0;
                for (int j = 0; j < N; j++) aux += prev[j] * A[(size_t)j * N + i];
            }
            const double v = aux * bu[(size_t)t * bstride + i];
            cur[i] = v;
            part += v;
        }
        const double s = wave_sum(part);
        const double c = 1.0 / s;
        __syncthreads();
        for (int i = l; i < N; i += WAVE) {
            const double a = cur[i] * c;
            cur[i] = a;
            if (au) au[(size_t)t * N + i] = a;
            if (i == N - 1) last = a;
        }
        if (l == 0) {
            su[t] = c;
            if (si) si[t] = s;
        }
        lp -= log(c);
        if (lognorm) lp += lognorm[f0 + t];
        __syncthreads();
    }
    // (every lane carries the same lp; the last state's owner holds alpha^_{T-1}(N-1))
    last = __shfl(last, (N - 1) % WAVE, WAVE);
    if (l == 0) loglik[u] = lp + log(last);
}

// calc_beta with gamma and the utterance's xi / den sums; partial-sum slot = utterance (S = U)
__global__ void __launch_bounds__(WAVE)
k_backward_wide(int N, int U, int delta, const double *__restrict__ A, const double *__restrict__ b,
                const long long *__restrict__ off, const double *__restrict__ alpha,
                const double *__restrict__ scale, double *__restrict__ beta, double *__restrict__ gamma,
                double *__restrict__ part_xi, double *__restrict__ part_dena,
                double *__restrict__ part_denc, const int *__restrict__ order,
                const int *__restrict__ offband)
{
    extern __shared__ double lds[]; // be[N] | w[N] | dena[N] | denc[N] | aself[N] | anext[N] | xi[MAX_DELTA + 1][N] | bo[N] | bn[N]
    const int l = threadIdx.x;
    const int u = order[blockIdx.x];
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    if (T <= 0) {
        for (int i = l; i < N; i += WAVE) {
            for (int o = 0; o <= MAX_DELTA; o++) part_xi[pxi_at(u, i, o, U)] = 0.0;
            part_dena[pden_at(u, i, U)] = 0.0;
            part_denc[pden_at(u, i, U)] = 0.0;
        }
        return;
    }
    double *be = lds, *w = lds + N, *dena = lds + 2 * N, *denc = lds + 3 * N, *aself = lds + 4 * N,
           *anext = lds + 5 * N, *xi = lds + 6 * N, *bo = xi + (size_t)(MAX_DELTA + 1) * N, *bn = bo + N;
    for (int i = l; i < N; i += WAVE) {
        aself[i] = A[(size_t)i * N + i];
        anext[i] = i + 1 < N ? A[(size_t)i * N + i + 1] : 0.0;
    }
    const double *bu = b + f0 * N, *au = alpha + f0 * N, *su = scale + f0;
    double *beu = beta + f0 * N, *gu = gamma + f0 * N;
    bool dense = offband[0] != 0;
    for (;;) {
        bool wild = false;
        {
            const double cT = su[T - 1];
            for (int i = l; i < N; i += WAVE) {
                const double b0 = (i == N - 1) ? 1.0 * cT : 0.0;
                const double g = au[(size_t)(T - 1) * N + i] * b0 * (1.0 / cT);
                be[i] = b0;
                beu[(size_t)(T - 1) * N + i] = b0;
                gu[(size_t)(T - 1) * N + i] = g;
                denc[i] = g; // gamma_{T-1}: in den_c (t < T, TF:1660) but not in den_a (t < T-1, TF:1618)
                dena[i] = 0.0;
                for (int o = 0; o <= MAX_DELTA; o++) xi[(size_t)o * N + i] = 0.0;
            }
        }
        __syncthreads();
        for (int t = T - 2; t >= 0; t--) {
            for (int i = l; i < N; i += WAVE) {
                const double bj = bu[(size_t)(t + 1) * N + i];
                bo[i] = be[i]; // beta^_{t+1} and b(t+1) apart: the recursion multiplies them in the
                bn[i] = bj;    // reference's association, (beta^ a) b (TF:1505)
                w[i] = be[i] * bj;
            }
            __syncthreads();
            const double c = su[t], sv = 1.0 / c;
            for (int i = l; i < N; i += WAVE) {
                double aux;
                if (!dense) {
                    aux = (bo[i] * aself[i]) * bn[i];
                    if (i + 1 < N) aux += (bo[i + 1] * anext[i]) * bn[i + 1];
                } else {
                    aux = 0.0;
                    for (int j = 0; j < N; j++) aux += (bo[j] * A[(size_t)i * N + j]) * bn[j];
                }
                const double al = au[(size_t)t * N + i];
                for (int o = 0; o <= delta; o++)
                    if (i + o < N) xi[(size_t)o * N + i] = fma(al, w[i + o], xi[(size_t)o * N + i]);
                const double bt = aux * c;
                if (!dense) wild |= !(fabs(bt) < INFINITY);
                const double g = (al * sv) * bt;
                be[i] = bt; // (be[i] of step t + 1 has gone into w: only this lane reads be[i])
                beu[(size_t)t * N + i] = bt;
                gu[(size_t)t * N + i] = g;
                dena[i] += g;
            }
            __syncthreads();
        }
        if (!dense && __any(wild)) { // the reference's dense loop turns such rows NaN (TF:1493-1510)
            dense = true;
            __syncthreads();
            continue;
        }
        break;
    }
    for (int i = l; i < N; i += WAVE) {
        for (int o = 0; o <= MAX_DELTA; o++)
            part_xi[pxi_at(u, i, o, U)] =
                (o <= delta && i + o < N) ? A[(size_t)i * N + i + o] * xi[(size_t)o * N + i] : 0.0;
        part_dena[pden_at(u, i, U)] = dena[i];
        part_denc[pden_at(u, i, U)] = dena[i] + denc[i];
    }
}

// Viterbi in the log domain; back-pointers one byte, rows of N bytes (N <= 255)
__global__ void __launch_bounds__(WAVE)
k_viterbi_wide(int N, int U, const double *__restrict__ logA, const double *__restrict__ logb,
               const long long *__restrict__ off, unsigned char *__restrict__ psi,
               unsigned char *__restrict__ path, double *__restrict__ score,
               const int *__restrict__ order, const int *__restrict__ offband)
{
    extern __shared__ double lds[]; // d[2][N]
    const int l = threadIdx.x;
    const int u = order[blockIdx.x];
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    if (T <= 0) {
        if (l == 0) score[u] = 0.0;
        return;
    }
    const bool banded = offband[0] == 0;
    const double *lb = logb + f0 * N;
    unsigned char *ps = psi + (size_t)f0 * N;
    for (int j = l; j < N; j += WAVE) {
        lds[j] = ((j == 0) ? 0.0 : -INFINITY) + lb[j];
        ps[j] = 0;
    }
    __syncthreads();
    for (int t = 1; t < T; t++) {
        const double *prev = lds + (size_t)((t + 1) & 1) * N;
        double *cur = lds + (size_t)(t & 1) * N;
        for (int j = l; j < N; j += WAVE) {
            double best = -INFINITY;
            int arg = 0;
            if (banded) {
                const double c1 = (j > 0 ? prev[j - 1] : 0.0) + (j > 0 ? logA[(size_t)(j - 1) * N + j] : -INFINITY);
                const double c0 = prev[j] + logA[(size_t)j * N + j];
                if (c1 > best) { // predecessor j - 1 (the lower index first)
                    best = c1;
                    arg = j - 1;
                }
                if (c0 > best) {
                    best = c0;
                    arg = j;
                }
            } else {
                for (int i = 0; i < N; i++) {
                    const double v = prev[i] + logA[(size_t)i * N + j];
                    if (v > best) {
                        best = v;
                        arg = i;
                    }
                }
            }
            cur[j] = best + lb[(size_t)t * N + j];
            ps[(size_t)t * N + j] = (unsigned char)arg;
        }
        __syncthreads();
    }
    __threadfence_block();
    if (l != 0) return;
    score[u] = lds[(size_t)((T - 1) & 1) * N + (N - 1)];
    int s = N - 1;
    unsigned char *pu = path + f0;
    for (int t = T - 1; t >= 0; t--) {
        pu[t] = (unsigned char)s;
        s = ps[(size_t)t * N + s];
    }
}

} // namespace ghmm
