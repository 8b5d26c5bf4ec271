// ghmm_mfma.hpp — CDNA4 matrix-core (v_mfma_f64_16x16x4_f64) tier of the hot path.
//
// The diagonal Mahalanobis exponent is a bilinear form once the frame is extended
// to [x', 1, x'^2] (x' = x - offset):
//     -1/2 sum_d inv_d (x'_d - mu'_d)^2 = sum_d x'_d (mu'_d inv_d) + sum_d x'_d^2 (-inv_d/2)
//                                        + (-1/2 sum_d mu'_d^2 inv_d)
// so a tile of 16 frames x 16 Gaussians is a [16 x K] . [K x 16] product with
// K = 2*DP (DP = D+1 rounded up to a multiple of 4), i.e. K/4 f64 MFMAs.  f64 MFMA
// runs at the same 78.6 TFLOP/s as the f64 vector ALU on MI355X, but it leaves the
// vector ALU free for exp()/normalisation and keeps operands out of VGPR traffic.
//
// Cancellation: the expanded form loses eps * sum_d inv_d mu'_d^2 absolutely, so
// (1) every tile of 16 Gaussians gets its own offset (the mean of its means) and
// (2) tiles that still hold an ill-conditioned Gaussian (cond > COND_MAX, e.g. the
// variance-floored "needle" components of SURVEY.md §7) are evaluated in the
// reference's direct form (x-mu)*inv*(x-mu) by the same kernel.
//
// Fragment maps (cdna_hip_programming.md §3): lane l, A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15], C/D reg r -> row (l>>4) + 4r, col l&15.
#pragma once
#include <hip/hip_runtime.h>

#include "ghmm_kernels.hpp"

namespace ghmm {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr double COND_MAX = 1.0e4; // expanded-form error ~ 4*eps*cond  (<= ~5e-12)
constexpr int EM_WAVES = 4;        // waves per emission block

// One thread per padded Gaussian gp = 16*c + j (state gp / Mp, mixture gp % Mp).
//   Wm[c][s][lane]  B fragments in lane order: row kk = 4s + (lane>>4), col lane&15
//                   rows 0..DP-1 multiply [x'_0..x'_{D-1}, 1, 0..], rows DP.. multiply x'^2
//   offs[c][DP]     tile offset (0 beyond D);  wkp[gp], gmap[gp] (-1 = padding), condp[gp]
__global__ void k_prepare_mfma(int N, int M, int D, int Mp, int NT, int DP,
                               const double *__restrict__ mean, const double *__restrict__ inv_var,
                               const double *__restrict__ wk, double *__restrict__ Wm,
                               double *__restrict__ offs, double *__restrict__ wkp,
                               int *__restrict__ gmap, double *__restrict__ condp)
{
    const int gp = blockIdx.x * blockDim.x + threadIdx.x;
    if (gp >= NT * 16) return;
    const int c = gp >> 4, j = gp & 15, KS = DP / 2;
    const int i = gp / Mp, m = gp % Mp;
    const bool real = (i < N) && (m < M);
    const int g = real ? i * M + m : -1;
    gmap[gp] = g;
    wkp[gp] = real ? wk[g] : 0.0;
    double *Wc = Wm + (size_t)c * KS * 64;
    double c0 = 0.0;
    for (int d = 0; d < DP; d++) {
        double o = 0.0;
        if (d < D) {
            int cnt = 0;
            for (int jj = 0; jj < 16; jj++) {
                int gq = c * 16 + jj, ii = gq / Mp, mm = gq % Mp;
                if (ii < N && mm < M) {
                    o += mean[((size_t)ii * M + mm) * D + d];
                    cnt++;
                }
            }
            o = cnt ? o / cnt : 0.0;
        }
        if (j == 0) offs[(size_t)c * DP + d] = o;
        double bc = 0.0, ac = 0.0;
        if (real && d < D) {
            double mu = mean[(size_t)g * D + d] - o, iv = inv_var[(size_t)g * D + d];
            bc = mu * iv;
            ac = -0.5 * iv;
            c0 += mu * mu * iv;
        }
        if (d != D) Wc[(d >> 2) * 64 + (d & 3) * 16 + j] = bc;
        const int k2 = DP + d;
        Wc[(k2 >> 2) * 64 + (k2 & 3) * 16 + j] = ac;
    }
    Wc[(D >> 2) * 64 + (D & 3) * 16 + j] = real ? -0.5 * c0 : 0.0; // multiplies the constant 1
    condp[gp] = real ? c0 : 0.0;
}

// calc_symbol_probab + calc_gaus (TF:1749-1841) for 16 frames x TC Gaussian tiles per
// wave iteration, linear domain (the reference's: exp(q) * c / (norm), summed over the
// state's mixtures, posteriors = share of the sum).  blockIdx.y picks the chunk of TC
// tiles whose B fragments are resident in LDS; blocks stride over frame tiles.
//   Mp <= 16: Mp is a power of two, a state's mixtures sit in Mp adjacent lanes;
//   Mp  > 16: Mp is a multiple of 16, a state spans Mp/16 consecutive tiles of the chunk.
__global__ void __launch_bounds__(EM_WAVES *WAVE, 2)
k_emission_mfma(int N, int M, int Mp, int D, int DP, int NT, int TC, long long F,
                const double *__restrict__ X, const double *__restrict__ Wm,
                const double *__restrict__ offs, const double *__restrict__ wkp,
                const int *__restrict__ gmap, const double *__restrict__ condp,
                const double *__restrict__ mean, const double *__restrict__ inv_var,
                double *__restrict__ b, double *__restrict__ post)
{
    extern __shared__ double lds[];
    const int KS = DP / 2, Q = DP / 4, XS = DP + 1, G = N * M;
    double *Wl = lds;                                // [TC][KS][64]
    double *ol = Wl + (size_t)TC * KS * 64;          // [TC][DP]
    double *xl = ol + (size_t)TC * DP;               // [EM_WAVES][16][XS]
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, j = l & 15, kq = l >> 4;
    const int c0 = blockIdx.y * TC;
    const int tc = (NT - c0) < TC ? (NT - c0) : TC;
    for (int k = tid; k < tc * KS * 64; k += EM_WAVES * WAVE) Wl[k] = Wm[(size_t)c0 * KS * 64 + k];
    for (int k = tid; k < tc * DP; k += EM_WAVES * WAVE) ol[k] = offs[(size_t)c0 * DP + k];
    __syncthreads();
    double *xw = xl + w * 16 * XS;
    const int tps = Mp > 16 ? Mp / 16 : 1;
    const long long ntf = (F + 15) / 16;
    for (long long tf = (long long)blockIdx.x * EM_WAVES + w; tf < ntf;
         tf += (long long)gridDim.x * EM_WAVES) {
        const long long f0 = tf * 16;
        const int nf = (int)((F - f0) < 16 ? (F - f0) : 16);
        // the wave's 16 x D frame tile: contiguous in HBM, read once
        for (int k = l; k < 16 * DP; k += WAVE) {
            int r = k / DP, d = k - r * DP;
            xw[r * XS + d] = (r < nf && d < D) ? X[(f0 + r) * D + d] : 0.0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // LDS is in order per wave
        const double *xr = xw + j * XS; // A rows: frame l&15
        double run[4] = {0.0, 0.0, 0.0, 0.0};
        for (int ct = 0; ct < tc; ct++) {
            const int gp = (c0 + ct) * 16 + j;
            const double wkj = wkp[gp];
            const int gm = gmap[gp];
            const bool flagged = __any(condp[gp] > COND_MAX);
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            if (!flagged) {
                const double *Wt = Wl + (size_t)ct * KS * 64 + l;
                const double *ot = ol + ct * DP;
                for (int s = 0; s < Q; s++) {
                    const int d = 4 * s + kq;
                    const double xo = xr[d] - ot[d];
                    const double a1 = d < D ? xo : (d == D ? 1.0 : 0.0);
                    const double a2 = d < D ? xo * xo : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, Wt[s * 64], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, Wt[(Q + s) * 64], acc, 0, 0, 0);
                }
            } else {
                // ill-conditioned tile: the reference's own form, TF:1829-1832
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    double q = 0.0;
                    if (gm >= 0) {
                        const double *xf = xw + (kq + 4 * r) * XS;
                        const double *mu = mean + (size_t)gm * D, *iv = inv_var + (size_t)gm * D;
                        for (int d = 0; d < D; d++) {
                            double dif = xf[d] - mu[d];
                            q += dif * iv[d] * dif;
                        }
                    }
                    acc[r] = -0.5 * q;
                }
            }
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = exp(acc[r]) * wkj;
            if (Mp <= 16) {
                const int st = gp / Mp;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    double s = v[r];
                    for (int o = 1; o < Mp; o <<= 1) s += __shfl_xor(s, o, 16);
                    const long long fr = f0 + kq + 4 * r;
                    if (fr < F) {
                        if ((j & (Mp - 1)) == 0 && st < N) b[fr * N + st] = s;
                        // gauss[i][j] /= b_i, 0 when b_i == 0 (TF:1773-1778)
                        if (post && gm >= 0) post[fr * G + gm] = s != 0.0 ? v[r] / s : 0.0;
                    }
                }
            } else {
                // a state spans `tps` tiles: park the raw terms, close the state on its
                // last tile and normalise what this lane parked
                const int st = gp / Mp, tin = (c0 + ct) % tps;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if (tin == 0) run[r] = 0.0;
                    run[r] += v[r];
                    const long long fr = f0 + kq + 4 * r;
                    if (post && gm >= 0 && fr < F) post[fr * G + gm] = v[r];
                }
                if (tin == tps - 1) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        double s = run[r];
                        for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 16);
                        const long long fr = f0 + kq + 4 * r;
                        if (fr < F && st < N) {
                            if (j == 0) b[fr * N + st] = s;
                            if (post)
                                for (int tt = 0; tt < tps; tt++) {
                                    const int gq = gmap[(c0 + ct - tt) * 16 + j];
                                    if (gq >= 0) {
                                        double raw = post[fr * G + gq];
                                        post[fr * G + gq] = s != 0.0 ? raw / s : 0.0;
                                    }
                                }
                        }
                    }
                }
            }
        }
    }
}

} // namespace ghmm
