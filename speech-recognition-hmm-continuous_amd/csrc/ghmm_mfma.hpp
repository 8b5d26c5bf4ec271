// ghmm_mfma.hpp — CDNA4 matrix-core (v_mfma_f64_16x16x4_f64) tier of the hot path.
//
// The diagonal Mahalanobis exponent is a bilinear form once the frame is extended
// to [x', 1, x'^2] (x' = x - offset):
//     -1/2 sum_d inv_d (x'_d - mu'_d)^2 = sum_d x'_d (mu'_d inv_d) + sum_d x'_d^2 (-inv_d/2)
//                                        + (-1/2 sum_d mu'_d^2 inv_d)
// so a tile of 16 frames x 16 Gaussians is a [16 x K] . [K x 16] product with
// K = 2*DP (DP = D+1 rounded up to a multiple of 4), i.e. K/4 f64 MFMAs.  f64 MFMA
// runs at the same 78.6 TFLOP/s as the f64 vector ALU on MI355X and never overlaps it on
// a SIMD (measured): what it buys is issue slots (2 048 flops per
// instruction) and operands straight from LDS.
//
// Cancellation: the expanded form loses eps * sum_d inv_d mu'_d^2 absolutely, so
// (1) frames and means are taken relative to an offset near the data's centre (oglob),
// (2) a tile of 16 Gaussians that holds a variance-floored "needle" component (SURVEY.md §7;
// cond ~ 1e7 around the centre) takes that component's mean as ITS offset (dtile / tshift) and
// (3) a tile that is still ill-conditioned then (cond > COND_MAX: two needles in one tile) is
// evaluated in the reference's direct form (x-mu)*inv*(x-mu) by k_emission_mfma.
//
// Fragment maps (cdna_hip_programming.md §3): lane l, A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15], C/D reg r -> row (l>>4) + 4r, col l&15.
#pragma once
#include <hip/hip_runtime.h>

#include "ghmm_kernels.hpp"

namespace ghmm {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr double COND_MAX = 1.0e4; // expanded-form error ~ 4*eps*cond  (<= ~5e-12)
constexpr int EM_WAVES = 8;        // waves per block of the generic emission kernel
constexpr int EM_XR = 16;          // frame-tile doubles per lane: 16*D/64, D <= 64

// Offsets for the expanded forms: offs[c][DP] = mean of the means of tile c's real
// Gaussians (0 beyond D); oglob[d] = mean of all means (grid = NT + D blocks).
__global__ void __launch_bounds__(64)
k_prepare_offsets(int N, int M, int D, int Mp, int NT, int DP, const double *__restrict__ mean,
                  double *__restrict__ offs, double *__restrict__ oglob)
{
    const int q = blockIdx.x, t = threadIdx.x;
    if (q < NT) {
        for (int d = t; d < DP; d += 64) {
            double o = 0.0;
            int cnt = 0;
            if (d < D)
                for (int jj = 0; jj < 16; jj++) {
                    int gq = q * 16 + jj, ii = gq / Mp, mm = gq % Mp;
                    if (ii < N && mm < M) {
                        o += mean[((size_t)ii * M + mm) * D + d];
                        cnt++;
                    }
                }
            offs[(size_t)q * DP + d] = cnt ? o / cnt : 0.0;
        }
    } else {
        __shared__ double sh[64];
        const int d = q - NT, G = N * M;
        double o = 0.0;
        for (int g = t; g < G; g += 64) o += mean[(size_t)g * D + d];
        sh[t] = o;
        __syncthreads();
        for (int k = 32; k > 0; k >>= 1) {
            if (t < k) sh[t] += sh[t + k];
            __syncthreads();
        }
        if (t == 0) oglob[d] = sh[0] / (double)G;
    }
}

// Offset of the expanded form per tile.  A variance-floored component (EM from the reference's
// initial model produces a few within two iterations: every variance at 1e-5, cond ~ 1e7
// around the data's centre) is harmless around ITS OWN mean, and so are its 15 neighbours
// (inverse variances ~ 1): a tile that holds a Gaussian with cond > COND_MAX takes the mean of its
// worst Gaussian as offset.  tile_offset_choice: one wave; cond[16] of the tile's Gaussians
// (padding: 0) -> index of the worst one, or -1 to keep the global offset.  First maximum wins,
// so that every caller agrees.
__device__ inline int tile_offset_choice(const double *cond16)
{
    int w = -1;
    double best = COND_MAX;
    for (int k = 0; k < 16; k++)
        if (cond16[k] > best) {
            best = cond16[k];
            w = k;
        }
    return w;
}

// ghmm_model_set path: the choice from the model itself.  tnext / otile: what the next
// preparation will use (k_prepare_mfma right behind this kernel).
__global__ void __launch_bounds__(64)
k_prepare_tiles(int N, int M, int D, int Mp, int NT, int DP, const double *__restrict__ mean,
                const double *__restrict__ inv_var, const double *__restrict__ oglob,
                double *__restrict__ otile, int *__restrict__ tnext)
{
    __shared__ double cond16[16];
    const int c = blockIdx.x, t = threadIdx.x;
    for (int k = 0; k < 16; k++) {
        const int gp = c * 16 + k, i = gp / Mp, m = gp % Mp;
        double v = 0.0;
        if (i < N && m < M)
            for (int d = t; d < D; d += 64) {
                const size_t q = ((size_t)i * M + m) * D + d;
                const double mu = mean[q] - oglob[d];
                v += mu * mu * inv_var[q];
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (t == 0) cond16[k] = v;
    }
    __syncthreads();
    const int w = tile_offset_choice(cond16);
    if (t == 0) tnext[c] = w >= 0 ? 1 : 0;
    if (w >= 0) {
        const int gp = c * 16 + w, g = (gp / Mp) * M + gp % Mp;
        for (int d = t; d < DP; d += 64) otile[(size_t)c * DP + d] = d < D ? mean[(size_t)g * D + d] : 0.0;
    }
}

// One 64-thread block per padded Gaussian gp = 16*c + j (state gp / Mp, mixture gp % Mp),
// threads over the coefficient index.
//   Wm[c][s][lane]  B fragments in lane order: row kk = 4s + (lane>>4), col lane&15
//                   rows 0..DP-1 multiply [x'_0..x'_{D-1}, 1, 0..], rows DP.. multiply x'^2
//   wkp[gp], gmap[gp] (-1 = padding); condg = sum_d inv_d mu'_d^2 with mu' = mu - oglob,
//   the cancellation measure of the expanded forms (condp: the same around the tile's
//   own offset, kept for diagnostics)
__global__ void __launch_bounds__(64)
k_prepare_mfma(int N, int M, int D, int Mp, int NT, int DP, const double *__restrict__ mean,
               const double *__restrict__ inv_var, const double *__restrict__ wk,
               const double *__restrict__ logwk, const double *__restrict__ otile,
               const int *__restrict__ tnext, const double *__restrict__ oglob,
               double *__restrict__ Wm, double *__restrict__ wkp, double *__restrict__ logwkp,
               int *__restrict__ gmap, double *__restrict__ condt, double *__restrict__ condg,
               int *__restrict__ anyflag, int *__restrict__ sflag, int epoch,
               double *__restrict__ dtile, int *__restrict__ tshift)
{
    __shared__ double sh0[64], sh1[64];
    const int gp = blockIdx.x, t = threadIdx.x;
    const int c = gp >> 4, j = gp & 15, KS = DP / 2;
    const int i = gp / Mp, m = gp % Mp;
    const bool real = (i < N) && (m < M);
    const int g = real ? i * M + m : -1;
    double *Wc = Wm + (size_t)c * KS * 64;
    const bool shifted = tnext[c] != 0; // this tile has its own offset (k_prepare_tiles / k_reduce_all)
    double c0 = 0.0, cg = 0.0;
    for (int d = t; d < DP; d += 64) {
        double bc = 0.0, ac = 0.0;
        if (real && d < D) {
            const double mraw = mean[(size_t)g * D + d], iv = inv_var[(size_t)g * D + d];
            const double mu = mraw - oglob[d];
            const double mt = mraw - (shifted ? otile[(size_t)c * DP + d] : oglob[d]);
            bc = mt * iv;
            ac = -0.5 * iv;
            cg += mu * mu * iv;
            c0 += mt * mt * iv;
        }
        if (j == 0) dtile[(size_t)c * DP + d] = (shifted && d < D) ? otile[(size_t)c * DP + d] - oglob[d] : 0.0;
        if (d != D) Wc[(d >> 2) * 64 + (d & 3) * 16 + j] = bc;
        const int k2 = DP + d;
        Wc[(k2 >> 2) * 64 + (k2 & 3) * 16 + j] = ac;
    }
    sh0[t] = cg;
    sh1[t] = c0;
    __syncthreads();
    for (int k = 32; k > 0; k >>= 1) {
        if (t < k) {
            sh0[t] += sh0[t + k];
            sh1[t] += sh1[t + k];
        }
        __syncthreads();
    }
    if (t == 0) {
        cg = sh0[0];
        c0 = sh1[0];
        gmap[gp] = g;
        wkp[gp] = real ? wk[g] : 0.0;
        logwkp[gp] = real ? logwk[g] : -1.0e300; // padding never wins a max, exp() of it is 0
        Wc[(D >> 2) * 64 + (D & 3) * 16 + j] = real ? -0.5 * c0 : 0.0; // multiplies the constant 1
        condt[gp] = real ? c0 : 0.0; // around the tile's offset: what the emission kernels go by
        condg[gp] = real ? cg : 0.0; // around the global offset: what the statistics go by
        if (j == 0) tshift[c] = shifted ? 1 : 0;
        // some Gaussian too ill-conditioned for the expanded forms: k_emission_mfma's direct
        // form (anyflag) / the vector-ALU k_mixstats (sflag) then take over for it
        // (flag[0] == the model's preparation count means "flagged now": nothing has to
        // clear it, every writer of one preparation stores the same value)
        if (real && c0 > COND_MAX) anyflag[0] = epoch;
        if (real && cg > COND_MAX) sflag[0] = epoch;
    }
}

// The M-step (k_mstep's arithmetic, unchanged) and the matrix-core preparation of the new
// model in one launch: block = state.  After the state's parameters are final, each wave of
// the block builds the B-fragment columns of the state's padded Gaussians exactly as
// k_prepare_mfma does (the last block also fills the padding columns behind state N-1).
// The offset of the expanded forms is the data's centre taken from the statistics themselves,
// oglob_d = sum_g num_mu[g][d] / sum_g num_c[g], which every block computes for itself (same
// sums in the same order): no kernel boundary is needed for it.
constexpr int MSF_THREADS = 512; // one wave per padded Gaussian of an 8-mixture state
__global__ void __launch_bounds__(MSF_THREADS)
k_mstep_mfma(int N, int M, int D, const double *__restrict__ stats, double norm2pi,
             double *__restrict__ A, double *__restrict__ c, double *__restrict__ mean,
             double *__restrict__ inv_var, double *__restrict__ det, double *__restrict__ wk,
             double *__restrict__ logwk, double *__restrict__ logA, int lds_doubles, int Mp, int NT,
             int DP, double *__restrict__ oglob, double *__restrict__ Wm, double *__restrict__ wkp,
             double *__restrict__ logwkp, int *__restrict__ gmap, double *__restrict__ condg,
             int *__restrict__ anyflag, int epoch, const double *__restrict__ otile,
             const int *__restrict__ tnext, double *__restrict__ condt, double *__restrict__ dtile,
             int *__restrict__ tshift, int *__restrict__ sflag, int delta)
{
    extern __shared__ double vs[]; // [lds_doubles] staging of mstep_state | og[DP] | red[MSF_THREADS]
    double *og = vs + lds_doubles, *red = og + DP;
    const int G = N * M, i = blockIdx.x, tid = threadIdx.x, KS = DP / 2;
    {
        // thread (d, part) adds every np-th Gaussian of column d; parts are added in order
        const double *num_c = stats + (size_t)N * N + 2 * (size_t)N, *num_mu = num_c + G;
        const int np = MSF_THREADS / DP, d = tid % DP, part = tid / DP;
        double o = 0.0, cn = 0.0;
        if (part < np)
            for (int g = part; g < G; g += np) {
                cn += num_c[g];
                if (d < D) o += num_mu[(size_t)g * D + d];
            }
        red[tid] = o;
        __syncthreads();
        if (tid < DP) {
            double tot = 0.0;
            for (int q = 0; q < np; q++) tot += red[q * DP + tid];
            og[tid] = tot;
        }
        __syncthreads();
        red[tid] = cn; // every column's threads hold the same partial counts: column 0's are used
        __syncthreads();
        if (tid < DP) {
            double cnt = 0.0;
            for (int q = 0; q < np; q++) cnt += red[q * DP];
            double o2 = 0.0;
            if (tid < D) {
                o2 = (cnt > 0.0 && cnt < INFINITY) ? og[tid] / cnt : oglob[tid];
                if (!(fabs(o2) < INFINITY)) o2 = 0.0;
                if (i == 0) oglob[tid] = o2;
            }
            og[tid] = o2;
        }
        __syncthreads();
    }
    mstep_state(N, M, D, stats, norm2pi, A, c, mean, inv_var, det, wk, logwk, logA, lds_doubles, vs, delta);
    __syncthreads(); // the state's new parameters (global) and og (LDS) are visible to the block
    const int w = tid >> 6, l = tid & 63;
    const int gp0 = i * Mp, gp1 = (i == N - 1) ? NT * 16 : (i + 1) * Mp;
    for (int gp = gp0 + w; gp < gp1; gp += MSF_THREADS / 64) {
        const int ct = gp >> 4, j = gp & 15;
        const int ii = gp / Mp, m = gp % Mp;
        const bool real = (ii < N) && (m < M);
        const int g = real ? ii * M + m : -1;
        double *Wc = Wm + (size_t)ct * KS * 64;
        // the tile's own offset (chosen by k_reduce_all from the model before this M-step: a
        // collapsed component does not move any more) or the data's centre
        const bool shifted = tnext[ct] != 0;
        double cg = 0.0, c0 = 0.0;
        for (int d = l; d < DP; d += 64) {
            double bc = 0.0, ac = 0.0, dt = 0.0;
            if (d < D) dt = shifted ? otile[(size_t)ct * DP + d] - og[d] : 0.0;
            if (real && d < D) {
                const double mu = mean[(size_t)g * D + d] - og[d], iv = inv_var[(size_t)g * D + d];
                const double mt = mu - dt;
                bc = mt * iv;
                ac = -0.5 * iv;
                cg += mu * mu * iv;
                c0 += mt * mt * iv;
            }
            if (d != D) Wc[(d >> 2) * 64 + (d & 3) * 16 + j] = bc;
            const int k2 = DP + d;
            Wc[(k2 >> 2) * 64 + (k2 & 3) * 16 + j] = ac;
            dtile[(size_t)ct * DP + d] = dt; // every Gaussian of the tile writes the same values
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            cg += __shfl_xor(cg, o, 64);
            c0 += __shfl_xor(c0, o, 64);
        }
        if (l == 0) {
            gmap[gp] = g;
            wkp[gp] = real ? wk[g] : 0.0;
            logwkp[gp] = real ? logwk[g] : -1.0e300;
            Wc[(D >> 2) * 64 + (D & 3) * 16 + j] = real ? -0.5 * c0 : 0.0;
            condg[gp] = real ? cg : 0.0;
            condt[gp] = real ? c0 : 0.0;
            tshift[ct] = shifted ? 1 : 0;
            if (real && c0 > COND_MAX) anyflag[0] = epoch;
            if (real && cg > COND_MAX) sflag[0] = epoch;
        }
    }
}

// calc_symbol_probab + calc_gaus (TF:1749-1841) for 16 frames x TC Gaussian tiles per
// wave iteration, linear domain (the reference's: exp(q) * c / (norm), summed over the
// state's mixtures, posteriors = share of the sum).  One block of EM_WAVES waves per
// CU; blockIdx.y picks the chunk of TC tiles whose B fragments the block keeps in LDS;
// waves stride over frame tiles.  Per frame tile a wave builds the extended frames
// Fext[16][2*DP] = [x', 1, 0.., x'^2, 0..] (x' = x - oglob) in its own LDS slab ONCE; every
// MFMA then takes both operands from LDS with no vector-ALU work in between.
//   Mp <= 16: Mp is a power of two, a state's mixtures sit in Mp adjacent lanes;
//   Mp  > 16: Mp is a multiple of 16, a state spans Mp/16 consecutive tiles of the chunk.
__global__ void __launch_bounds__(EM_WAVES *WAVE, 2)
k_emission_mfma(int N, int M, int Mp, int D, int DP, int NT, int TC, long long F,
                const double *__restrict__ X, const double *__restrict__ Wm,
                const double *__restrict__ oglob, const double *__restrict__ wkp,
                const int *__restrict__ gmap, const double *__restrict__ condg,
                const double *__restrict__ mean, const double *__restrict__ inv_var,
                double *__restrict__ b, double *__restrict__ post, const int *__restrict__ only_if,
                int epoch, const int *__restrict__ tshift, const double *__restrict__ dtile)
{
    extern __shared__ double lds[];
    if (only_if && only_if[0] != epoch) return; // k_emission_sched has done the job
    const int KS = DP / 2, XS = 2 * DP + 1, G = N * M;
    double *Wl = lds;                                // [TC][KS][64]
    double *xl = Wl + (size_t)TC * KS * 64;          // [EM_WAVES][16][XS]
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, j = l & 15, kq = l >> 4;
    const int c0 = blockIdx.y * TC;
    const int tc = (NT - c0) < TC ? (NT - c0) : TC;
    for (int k = tid; k < tc * KS * 64; k += EM_WAVES * WAVE) Wl[k] = Wm[(size_t)c0 * KS * 64 + k];
    __syncthreads();
    double *xw = xl + w * 16 * XS;
    const int tps = Mp > 16 ? Mp / 16 : 1;
    const long long ntf = (F + 15) / 16;
    const long long FD = F * D;
    // the wave's 16 x D frame tile is contiguous in HBM; lane l moves elements l + 64u.
    // The NEXT tile is fetched into registers while the current one is computed.
    int loff[EM_XR], roff[EM_XR];
    double oo[EM_XR];
#pragma unroll
    for (int u = 0; u < EM_XR; u++) {
        const int k = l + 64 * u, r = k / D, d = k - r * D;
        const bool in = k < 16 * D;
        loff[u] = in ? r * XS + d : -1;
        roff[u] = in ? k : 0;
        oo[u] = in ? oglob[d] : 0.0;
    }
    double xn[EM_XR];
    auto fetch = [&](long long tf) {
        const long long base = tf * 16 * D;
#pragma unroll
        for (int u = 0; u < EM_XR; u++) {
            long long q = base + roff[u];
            q = q < FD ? q : FD - 1; // clamped, never predicated (keeps vmcnt countable)
            xn[u] = X[q];
        }
    };
    // columns that never change: the constant 1 and the zero padding of both halves
    for (int k = l; k < 16 * (2 * DP - 2 * D); k += WAVE) {
        const int r = k / (2 * DP - 2 * D), e = k - r * (2 * DP - 2 * D);
        const int col = e < DP - D ? D + e : DP + D + (e - (DP - D));
        xw[r * XS + col] = (col == D) ? 1.0 : 0.0;
    }
    const long long tstride = (long long)gridDim.x * EM_WAVES;
    long long tf = (long long)blockIdx.x * EM_WAVES + w;
    fetch(tf < ntf ? tf : 0);
    for (; tf < ntf; tf += tstride) {
        const long long f0 = tf * 16;
#pragma unroll
        for (int u = 0; u < EM_XR; u++)
            if (loff[u] >= 0) {
                const double xo = xn[u] - oo[u];
                xw[loff[u]] = xo;
                xw[loff[u] + DP] = xo * xo;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // LDS is in order per wave
        fetch(tf + tstride < ntf ? tf + tstride : tf);
        const double *xr = xw + j * XS + kq; // A operand: frame l&15, k = 4s + (l>>4)
        double run[4] = {0.0, 0.0, 0.0, 0.0};
        for (int ct = 0; ct < tc; ct++) {
            const int gp = (c0 + ct) * 16 + j;
            const double wkj = wkp[gp];
            const int gm = gmap[gp];
            // condg here = conditioning around the TILE's offset (the global one, or the mean of
            // the tile's variance-floored component: dtile holds the difference)
            const bool flagged = __any(condg[gp] > COND_MAX);
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            if (!flagged && tshift[c0 + ct] == 0) {
                const double *Wt = Wl + (size_t)ct * KS * 64 + l;
#pragma unroll 4
                for (int s = 0; s < KS; s++)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[4 * s], Wt[s * 64], acc, 0, 0, 0);
            } else if (!flagged) {
                // the tile's own offset: x'' = x' - (offset - oglob), squares formed here
                const double *Wt = Wl + (size_t)ct * KS * 64 + l;
                const double *dq = dtile + (size_t)(c0 + ct) * DP + kq;
                const int Q = KS / 2;
                for (int s = 0; s < Q; s++) {
                    const double x1 = xr[4 * s] - dq[4 * s];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, Wt[s * 64], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x1 * x1, Wt[(Q + s) * 64], acc, 0, 0, 0);
                }
            } else {
                // ill-conditioned tile: the reference's own form, TF:1829-1832 (summed over d in
                // its order; the Gaussian's mean and inverse variance are read once per d for
                // the lane's four frames)
                double q4[4] = {0.0, 0.0, 0.0, 0.0};
                const int gq = gm >= 0 ? gm : 0;
                const double *mu = mean + (size_t)gq * D, *iv = inv_var + (size_t)gq * D;
                const double *xf = xw + kq * XS;
                for (int d = 0; d < D; d++) {
                    const double md = mu[d] - oglob[d], id = iv[d];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const double dif = xf[4 * r * XS + d] - md;
                        q4[r] += dif * id * dif;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = gm >= 0 ? -0.5 * q4[r] : 0.0;
            }
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = exp_emis(acc[r]) * wkj;
            if (Mp <= 16) {
                const int st = gp / Mp;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double s = segment_sum(v[r], Mp);
                    const long long fr = f0 + kq + 4 * r;
                    if (fr < F) {
                        if ((j & (Mp - 1)) == 0 && st < N) b[fr * N + st] = s;
                        // gauss[i][j] /= b_i, 0 when b_i == 0 (TF:1773-1778); a reciprocal
                        // for normal b_i, the true quotient when b_i is tiny
                        if (post && gm >= 0) {
                            const bool nrm = s >= 1.0e-290 && s <= 1.0e290;
                            post[fr * G + gm] = nrm ? v[r] * recip_scale(s) : (s != 0.0 ? v[r] / s : 0.0);
                        }
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
                        const double s = segment_sum(run[r], 16);
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

// ------------------------------------------------------- emission, occupancy
// Same computation as k_emission_mfma for the common, well-conditioned case (no
// ill-conditioned Gaussian anywhere, Mp a power of two <= 64), built for 4 waves per SIMD: sixteen
// waves per block (one block per CU) share the chunk's B fragments in LDS, every wave
// keeps x' = x - oglob of its 16 frames in a 5 KB slab and, where registers allow, as A
// fragments (x', x'^2) for all of the frame tile's Gaussian tiles; the kernel stays under 128
// VGPRs.  An f64 MFMA and vector-ALU work never overlap on a SIMD; the four
// waves cover each other's LDS and HBM waits.
//   - K steps, mixture padding and "posteriors wanted" are compile-time
//   - the epilogue is branch-free: lanes without an output store to a sink
//   - v / b_i is a reciprocal (hardware seed + two Newton steps) after an exact
//     power-of-two rescale of tiny or huge b_i; 0 when b_i == 0 (TF:1773-1778)
constexpr int EMS_WAVES = 16;

template <int MP> __device__ inline double segment_sum_t(double v)
{
    if (MP >= 2) v += dpp_f64<DPP_QUAD_XOR1>(v);
    if (MP >= 4) v += dpp_f64<DPP_QUAD_XOR2>(v);
    if (MP >= 8) v += dpp_f64<DPP_ROW_HALF_MIRROR>(v);
    if (MP >= 16) v += dpp_f64<DPP_ROW_MIRROR>(v);
    return v;
}

// exp_emis on four values at once: four independent dependency chains
__device__ inline void exp_emis4(const v4d &x, double (&out)[4])
{
    double k[4], r[4], p[4];
    int ki[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        // clamp at -750 on the high word (negative doubles order like unsigned integers):
        // one 32-bit v_min_u32, where fmax() costs two v_max_f64 (it canonicalises first).
        // Anything in (-751, -750] underflows to exp = 0 through v_ldexp_f64 like libm;
        // -inf and sign-bit NaNs become 0 too, a positive NaN stays NaN: either way the
        // frame's scale is 1/0 or NaN and the utterance's log-likelihood ends up NaN like
        // the reference's
        const unsigned hi = (unsigned)__double2hiint(x[q]);
        const double xc = __hiloint2double((int)(hi < 0xC0877000u ? hi : 0xC0877000u), __double2loint(x[q]));
        // round to nearest by adding 1.5 * 2^52: the integer lands in the low mantissa bits
        const double t = fma(xc, 1.4426950408889634074, 0x1.8p52);
        ki[q] = __double2loint(t);
        k[q] = t - 0x1.8p52;
        r[q] = fma(-k[q], 6.93147180369123816490e-01, xc);
        r[q] = fma(-k[q], 1.90821492927058770002e-10, r[q]);
        p[q] = 2.08767569878680989792e-09; // 1/12!: |r| <= ln2/2, the next term is below 1.7e-16
    }
    const double cf[11] = {2.50521083854417187751e-08, 2.75573192239858906526e-07,
                           2.75573192239858906526e-06, 2.48015873015873015873e-05,
                           1.98412698412698412698e-04, 1.38888888888888888889e-03,
                           8.33333333333333333333e-03, 4.16666666666666666667e-02,
                           1.66666666666666666667e-01, 0.5,
                           1.0};
#pragma unroll
    for (int t = 0; t < 11; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) p[q] = fma(p[q], r[q], cf[t]);
#pragma unroll
    for (int q = 0; q < 4; q++) out[q] = ldexp(fma(p[q], r[q], 1.0), ki[q]);
}

template <int MP> __device__ inline double segment_max_t(double v)
{
    if (MP >= 2) v = fmax(v, dpp_f64<DPP_QUAD_XOR1>(v));
    if (MP >= 4) v = fmax(v, dpp_f64<DPP_QUAD_XOR2>(v));
    if (MP >= 8) v = fmax(v, dpp_f64<DPP_ROW_HALF_MIRROR>(v));
    if (MP >= 16) v = fmax(v, dpp_f64<DPP_ROW_MIRROR>(v));
    return v;
}

// State sums by a transposing butterfly: at the xor-1 and xor-2 levels each lane keeps only the
// rows whose index matches its lane bits and sends the others, so that after the sums lane j holds
// the sum of ONE row, r = j & 3 (MPL >= 4), or of two rows r = (j & 1) and 2 + (j & 1)
// (MPL == 2).  The additions pair the same lanes as a plain butterfly: bit-identical sums.
template <int MPL, int NV>
__device__ inline void transposed_sums(const double (&tot)[4], int j, double (&sv)[NV])
{
    if (MPL == 1) {
#pragma unroll
        for (int r = 0; r < NV; r++) sv[r] = tot[r % 4];
    } else {
        const bool o0 = (j & 1) != 0;
        double k0 = o0 ? tot[1] : tot[0], k1 = o0 ? tot[3] : tot[2];
        const double s0 = o0 ? tot[0] : tot[1], s1 = o0 ? tot[2] : tot[3];
        k0 += dpp_f64<DPP_QUAD_XOR1>(s0);
        k1 += dpp_f64<DPP_QUAD_XOR1>(s1);
        if (MPL == 2) {
            sv[0] = k0;
            sv[NV - 1] = k1;
        } else {
            const bool o1 = (j & 2) != 0;
            double kk = o1 ? k1 : k0;
            const double ss = o1 ? k0 : k1;
            kk += dpp_f64<DPP_QUAD_XOR2>(ss);
            if (MPL >= 8) kk += dpp_xor4_f64(kk);
            if (MPL >= 16) kk += dpp_f64<DPP_ROW_ROR8>(kk);
            sv[0] = kk;
        }
    }
}

// OUT 0: b (recogniser, RF:860-889); 1: b and posteriors (trainer, TF:1749-1783);
// 2: log b for the Viterbi lattice, m + log(sum exp(e - m)) like the oracle's definition
template <int KS, int MP, int OUT>
__global__ void __launch_bounds__(EMS_WAVES *WAVE)
k_emission_sched(int N, int M, int D, int NT, int TC, long long F, const double *__restrict__ X,
                 const double *__restrict__ Wm, const double *__restrict__ oglob,
                 const double *__restrict__ wkp, const int *__restrict__ gmap,
                 double *__restrict__ b, double *__restrict__ post, double *__restrict__ sink,
                 const int *__restrict__ anyflag, int epoch, const double *__restrict__ dtile,
                 const int *__restrict__ tshift)
{
    extern __shared__ double lds[];
    if (anyflag[0] == epoch) return; // an ill-conditioned Gaussian somewhere: k_emission_mfma does the job
    constexpr int DP = 2 * KS, Q = KS / 2, XS = DP + 1;
    constexpr int LOGMP = MP == 1 ? 0 : MP == 2 ? 1 : MP == 4 ? 2 : MP == 8 ? 3 : MP == 16 ? 4 : MP == 32 ? 5 : 6;
    constexpr int MPL = MP < 16 ? MP : 16, TPS = MP <= 16 ? 1 : MP / 16;
    const int G = N * M;
    double *Wl = lds;                                // [TC][KS][64]
    double *xl = Wl + (size_t)TC * KS * 64;          // [EMS_WAVES][16][XS]
    double *ol = xl + (size_t)EMS_WAVES * 16 * XS;   // [DP]
    double *wkl = ol + DP;                           // [TC][16]
    // output cursors per (tile, lane & 15): element offset {posterior, b} inside a frame row,
    // or the distance to the lane's sink slot, and the row stride {G, N} or 0 for the sink
    long long *offl = (long long *)(wkl + (size_t)TC * 16); // [TC][16][2]
    unsigned *strl = (unsigned *)(offl + (size_t)TC * 32);  // [TC][16][2]
    // tiles that hold a variance-floored component take that component's mean as their offset:
    // dl = offset - oglob (subtracted from the A fragments of that tile), tsl = "shifted"
    double *dl = (double *)(strl + (size_t)TC * 32);        // [TC][DP]
    int *tsl = (int *)(dl + (size_t)TC * DP);               // [TC]
    const int tid = threadIdx.x, l = tid & 63, j = l & 15, kq = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index, in a scalar register
    const int c0 = blockIdx.y * TC;
    const int tc = (NT - c0) < TC ? (NT - c0) : TC;
    for (int k = tid; k < tc * KS * 64; k += EMS_WAVES * WAVE) Wl[k] = Wm[(size_t)c0 * KS * 64 + k];
    for (int k = tid; k < tc * 16; k += EMS_WAVES * WAVE) {
        wkl[k] = wkp[c0 * 16 + k];
        const int gm = gmap[c0 * 16 + k], jj = k & 15;
        const int stt = (c0 * 16 + k) >> LOGMP;
        const bool hold = (MPL >= 4 ? (jj & (MPL - 1)) < 4 : true) && stt < N;
        // (sink slots of this block: its first wave's region)
        double *bsink = sink + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) * EMS_WAVES % SINK_WAVES) * 2 * WAVE;
        offl[2 * k] = OUT != 1 ? 0 : (gm >= 0 ? (long long)gm : (bsink + jj) - post);
        offl[2 * k + 1] = hold ? (long long)stt : (bsink + 16 + jj) - b;
        strl[2 * k] = gm >= 0 ? (unsigned)G : 0u;
        strl[2 * k + 1] = hold ? (unsigned)N : 0u;
    }
    for (int k = tid; k < DP; k += EMS_WAVES * WAVE) ol[k] = k < D ? oglob[k] : 0.0;
    for (int k = tid; k < tc * DP; k += EMS_WAVES * WAVE) dl[k] = dtile[(size_t)c0 * DP + k];
    for (int k = tid; k < tc; k += EMS_WAVES * WAVE) tsl[k] = tshift[c0 + k];
    __syncthreads();
    double *xw = xl + w * 16 * XS;
    const long long ntf = (F + 15) / 16;
    const long long FD = F * D;
    // constant columns of the slab: the 1 at column D, zeros beyond
    for (int k = l; k < 16 * (DP - D); k += WAVE) {
        const int r = k / (DP - D), e = k - r * (DP - D);
        xw[r * XS + D + e] = e == 0 ? 1.0 : 0.0;
    }
    const int q64 = 64 / D, r64 = 64 - q64 * D; // element index step 64 in (row, column) form
    double *snk = wave_sink(sink);
    const double *xr = xw + j * XS + kq; // A operand: frame l&15, k = 4s + (l>>4)
    // Work units = (frame tile, group of TPS Gaussian tiles), dealt to the grid's waves in
    // equal contiguous shares: with whole frame tiles per wave, 18 750 tiles on 4 096 waves
    // left 42 % of the CUs idle during the last of five rounds.  A wave that starts or ends
    // inside a frame tile loads that tile's slab like any other.
    const int ng = (tc + TPS - 1) / TPS;
    const long long U = ntf * ng, GWV = (long long)gridDim.x * EMS_WAVES;
    const long long gwv = (long long)blockIdx.x * EMS_WAVES + w;
    long long u = U * gwv / GWV;
    const long long u1 = U * (gwv + 1) / GWV;
    long long tf = u / ng;
    int g0 = (int)(u - tf * ng);
    for (; u < u1; tf++, g0 = 0) {
        const long long f0 = tf * 16;
        const int g1 = (u1 - u) < (long long)(ng - g0) ? g0 + (int)(u1 - u) : ng;
        u += g1 - g0;
        {
            // the wave's 16 x D frame tile is contiguous in HBM; lane l moves elements
            // l + 64u (clamped addresses, never predicated).  No register prefetch: with
            // four waves per SIMD the other waves cover this latency.
            const long long base = f0 * D + l;
            double xn[EM_XR];
#pragma unroll
            for (int u = 0; u < EM_XR; u++) {
                long long q = base + 64 * u;
                q = q < FD ? q : FD - 1;
                xn[u] = X[q];
            }
            int r = l / D, d = l - r * D;
#pragma unroll
            for (int u = 0; u < EM_XR; u++) {
                if (l + 64 * u < 16 * D) xw[r * XS + d] = xn[u] - ol[d];
                r += q64;
                d += r64;
                if (d >= D) {
                    d -= D;
                    r++;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // LDS is in order per wave
        // the A fragments of this frame tile (x' and x'^2 of frame l & 15, k = 4s + (l >> 4))
        // stay in registers for all of its Gaussian tiles where the epilogue leaves room
        // (128 VGPRs = 4 waves per SIMD); otherwise they are re-read from the slab per tile
        constexpr bool AREG = OUT != 2 && TPS == 1;
        double a1[AREG ? Q : 1], a2[AREG ? Q : 1];
        if (AREG) {
#pragma unroll
            for (int s = 0; s < Q; s++) {
                a1[s] = xr[4 * s];
                a2[s] = a1[s] * a1[s];
            }
        }
        // a state's mixtures fill MPL adjacent lanes of TPS consecutive tiles
        for (int ct = g0 * TPS; ct < g1 * TPS; ct += TPS) {
            double e[TPS][4];
#pragma unroll
            for (int tt = 0; tt < TPS; tt++) {
                v4d acc = {0.0, 0.0, 0.0, 0.0};
                const double *Wt = Wl + (size_t)(ct + tt) * KS * 64 + l;
                if (__builtin_amdgcn_readfirstlane(tsl[ct + tt]) != 0) {
                    // this tile's own offset: x'' = x' - (offset - oglob), from the slab
                    const double *dq = dl + (ct + tt) * DP + kq;
#pragma unroll 5
                    for (int s = 0; s < Q; s++) {
                        const double x1 = xr[4 * s] - dq[4 * s];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, Wt[s * 64], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x1 * x1, Wt[(Q + s) * 64], acc, 0, 0, 0);
                    }
                } else if (AREG) {
#pragma unroll
                    for (int s = 0; s < Q; s++) {
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[AREG ? s : 0], Wt[s * 64], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[AREG ? s : 0], Wt[(Q + s) * 64], acc, 0, 0, 0);
                    }
                } else {
#pragma unroll 5
                    for (int s = 0; s < Q; s++) {
                        const double x1 = xr[4 * s];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, Wt[s * 64], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x1 * x1, Wt[(Q + s) * 64], acc, 0, 0, 0);
                    }
                }
                const double wkj = wkl[(ct + tt) * 16 + j];
                if (OUT == 2) {
                    // wkl holds log(wk) here: keep the exponents, exponentiate after the max
#pragma unroll
                    for (int r = 0; r < 4; r++) e[tt][r] = acc[r] + wkj;
                } else {
                    exp_emis4(acc, e[tt]);
#pragma unroll
                    for (int r = 0; r < 4; r++) e[tt][r] *= wkj;
                }
            }
            if (OUT == 2) {
                double mx[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    double m = e[0][r];
#pragma unroll
                    for (int tt = 1; tt < TPS; tt++) m = fmax(m, e[tt][r]);
                    mx[r] = segment_max_t<MPL>(m);
                }
                v4d sum4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int tt = 0; tt < TPS; tt++) {
                    v4d dlt;
                    double ex[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) dlt[r] = e[tt][r] - mx[r];
                    exp_emis4(dlt, ex);
#pragma unroll
                    for (int r = 0; r < 4; r++) sum4[r] += ex[r];
                }
                // sum over the state's lanes transposed: one (frame, state) and one log per lane
                constexpr int NV2 = MPL >= 4 ? 1 : (MPL == 2 ? 2 : 4);
                double s4[4] = {sum4[0], sum4[1], sum4[2], sum4[3]}, sv2[NV2];
                transposed_sums<MPL, NV2>(s4, j, sv2);
                const long long boff2 = offl[2 * (ct * 16 + j) + 1];
                const unsigned bstr2 = strl[2 * (ct * 16 + j) + 1];
#pragma unroll
                for (int v = 0; v < NV2; v++) {
                    const int rw = MPL >= 4 ? (j & 3) : (MPL == 2 ? 2 * v + (j & 1) : v);
                    double mxr;
                    if (MPL >= 4) mxr = (j & 2) ? ((j & 1) ? mx[3] : mx[2]) : ((j & 1) ? mx[1] : mx[0]);
                    else if (MPL == 2) mxr = (j & 1) ? mx[(2 * v + 1) % 4] : mx[(2 * v) % 4];
                    else mxr = mx[v % 4];
                    const double lb = mxr < -1.0e299 ? -INFINITY : mxr + log(sv2[v]);
                    double *pb = b + ((unsigned long long)((unsigned)(f0 + kq) + 4 * rw) * bstr2 + boff2);
                    pb = f0 + kq + 4 * rw < F ? pb : snk;
                    *pb = lb;
                }
                continue;
            }
            // state sums, transposed: one (frame, state) per lane (transposed_sums above).  The
            // reciprocal for the posteriors (TF:1773-1778) is then formed once per (frame, state)
            // instead of once per lane and row, and handed back to the state's lanes by quad
            // broadcasts.
            constexpr int NV = MPL >= 4 ? 1 : (MPL == 2 ? 2 : 4);
            double sv[NV];
            {
                double tot[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    tot[r] = e[0][r];
#pragma unroll
                    for (int tt = 1; tt < TPS; tt++) tot[r] += e[tt][r];
                }
                transposed_sums<MPL, NV>(tot, j, sv);
            }
            // rows held by this lane: rw = v-th held row; b is stored by the first lanes of the
            // state's group (one per row), everything else goes to the sink (cursor tables)
            const unsigned frow = (unsigned)(f0 + kq);
            const bool full = f0 + 16 <= F; // wave-uniform
            const long long boff = offl[2 * (ct * 16 + j) + 1];
            const unsigned bstr = strl[2 * (ct * 16 + j) + 1];
            double rrv[NV];
#pragma unroll
            for (int v = 0; v < NV; v++) {
                const int rw = MPL >= 4 ? (j & 3) : (MPL == 2 ? 2 * v + (j & 1) : v);
                double *pb = b + ((unsigned long long)(frow + 4 * rw) * bstr + boff);
                if (!full) pb = (long long)frow + 4 * rw < F ? pb : snk;
                const double sm = sv[v];
                *pb = sm;
                if (OUT == 1) {
                    // gauss[i][j] /= b_i, 0 when b_i == 0: reciprocal (hardware seed + two
                    // Newton steps = the IEEE quotient in every case measured) after an exact
                    // power-of-two rescale of a tiny or huge b_i
                    const double sc = sm < 1.0e-290 ? 0x1p600 : (sm > 1.0e290 ? 0x1p-600 : 1.0);
                    const double s2 = sm * sc;
                    double rr = __builtin_amdgcn_rcp(s2);
                    rr = fma(rr, fma(-s2, rr, 1.0), rr);
                    rr = fma(rr, fma(-s2, rr, 1.0), rr);
                    rrv[v] = sm != 0.0 ? rr * sc : 0.0;
                }
            }
            if (OUT == 1) {
                double rr4[4];
                if (MPL >= 4) {
                    rr4[0] = dpp_f64<0x00>(rrv[0]); // quad_perm:[r,r,r,r]
                    rr4[1] = dpp_f64<0x55>(rrv[0]);
                    rr4[2] = dpp_f64<0xAA>(rrv[0]);
                    rr4[3] = dpp_f64<0xFF>(rrv[0]);
                } else if (MPL == 2) {
                    rr4[0] = dpp_f64<0xA0>(rrv[0]);      // quad_perm:[0,0,2,2]: the pair's even lane
                    rr4[1] = dpp_f64<0xF5>(rrv[0]);      // quad_perm:[1,1,3,3]: the pair's odd lane
                    rr4[2] = dpp_f64<0xA0>(rrv[NV - 1]);
                    rr4[3] = dpp_f64<0xF5>(rrv[NV - 1]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) rr4[r] = rrv[r % NV];
                }
                // posterior cursor of this lane for row 0 (frame f0 + kq), 4 frames per r; a
                // lane without an output has cursor = sink and stride 0, rows past the end of
                // a ragged last tile go to the sink too: branches around the stores were
                // measured slower than these selects
#pragma unroll
                for (int tt = 0; tt < TPS; tt++) {
                    const long long goff = offl[2 * ((ct + tt) * 16 + j)];
                    const unsigned gstr = strl[2 * ((ct + tt) * 16 + j)];
                    double *pp0 = post + ((unsigned long long)frow * gstr + goff);
                    const unsigned long long stp = (unsigned long long)(4u * gstr);
                    if (full) {
#pragma unroll
                        for (int r = 0; r < 4; r++) pp0[r * stp] = e[tt][r] * rr4[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            double *pp = (long long)frow + 4 * r < F ? pp0 + r * stp : snk;
                            *pp = e[tt][r] * rr4[r];
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------- mixstats
// calc_mix_param (TF:1691-1727) as a matrix product over the time axis:
//     S[g][e] = sum_t w_t(g) * Fext_t[e],  w_t(g) = gamma_t(state(g)) * post_t(g) (TF:1706-1711)
// with Fext = [x'_0..x'_{D-1}, 1, 0.., x'^2_0..x'^2_{D-1}, 0..] (x' = x - oglob), so that
//     num_c = S[.][D],  num_mu_d = S_d + oglob_d num_c,
//     num_var_d = sum w (x_d - mu_d)^2 = S_{DP+d} - 2 mu'_d S_d + mu'_d^2 num_c   (old mean, TF:1720)
// A = w^T (16 Gaussians x 4 frames), B = Fext (4 frames x 16 features).  One wave owns
// CT x NE accumulator tiles and a contiguous range of frames; operands are staged through
// LDS (STAGED) or come straight from HBM/L2, prefetched a few k-steps ahead.  The four waves
// of a block fold their tiles through LDS in wave order and the block writes ONE partial;
// k_reduce_all adds the partials in a fixed order (bitwise reproducible).
// a 64-bit value that is the same in every lane, moved to scalar registers
__device__ inline long long uniform64(long long v)
{
    const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
    const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned)lo;
}

// e / d and e % d for small non-negative e (< 2^20) without the ~40-instruction integer
// division: float reciprocal, then one correction either way
__device__ inline void divmod_small(int e, int d, float rd, int &q, int &r)
{
    q = (int)((float)e * rd);
    r = e - q * d;
    if (r < 0) {
        r += d;
        q--;
    }
    if (r >= d) {
        r -= d;
        q++;
    }
}

constexpr int MSM_WAVES = 4;
constexpr int MSM_PD = 4; // k-steps of operands in flight per wave

typedef double v2d __attribute__((ext_vector_type(2)));

// STAGED = true: frames go HBM -> registers -> LDS in 16-frame stages with fully
// coalesced 16-byte-per-lane loads (one stage ahead), MFMA operands come from LDS.
// Needs G, gmin, GW even (16-byte alignment of the posterior rows) and N <= 16.
// STAGED = false: operands straight from HBM, 8 bytes per lane (any shape).
template <int CT, int NE, bool STAGED>
__global__ void __launch_bounds__(MSM_WAVES *WAVE, 1)
k_mixstats_mfma(int N, int M, int Mp, int D, int DP, int NT, long long F, int gmin, int GW,
                const double *__restrict__ X, const double *__restrict__ gamma,
                const double *__restrict__ post, const int *__restrict__ gmap,
                const double *__restrict__ oglob, double *__restrict__ part)
{
    extern __shared__ double lds[]; // fold: [CT*NE*4][64]; STAGED: per-wave frame stages
    const int tid = threadIdx.x, l = tid & 63, j = l & 15, kq = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index, in a scalar register
    const int G = N * M, ES = NE * 16;
    const int c0 = STAGED ? gmin / 16 : blockIdx.y * CT; // staged: one launch per chunk
    int gmA[CT], stA[CT];
#pragma unroll
    for (int c = 0; c < CT; c++) {
        const int gp = (c0 + c) * 16 + j;
        gmA[c] = (c0 + c < NT) ? gmap[gp] : -1;
        stA[c] = gp / Mp;
    }
    // per-lane feature constants of the direct (non-staged) operand path; rebuilt where
    // they are used so that they do not stay live across the staged main loop
    int dn[NE];
    double on[NE], k0[NE], k1[NE], k2[NE];
    auto feature_setup = [&]() {
#pragma unroll
        for (int n = 0; n < NE; n++) {
            const int e = 16 * n + j, k = e / DP, d = e - k * DP;
            // 0: x', 1: x'^2, 2: the constant one, 3: zero padding
            const int kind = (k < 2 && d < D) ? k : ((k == 0 && d == D) ? 2 : 3);
            dn[n] = (k < 2 && d < D) ? d : 0;
            on[n] = (k < 2 && d < D) ? oglob[d] : 0.0;
            k0[n] = kind == 2 ? 1.0 : 0.0;
            k1[n] = kind == 0 ? 1.0 : 0.0;
            k2[n] = kind == 1 ? 1.0 : 0.0;
        }
    };
    v4d acc[CT][NE];
#pragma unroll
    for (int c = 0; c < CT; c++)
#pragma unroll
        for (int n = 0; n < NE; n++) acc[c][n] = (v4d){0.0, 0.0, 0.0, 0.0};

    // this wave's frames, dealt evenly in whole units (STAGED: 16-frame stages, else
    // 4-frame k-steps); the corpus' ragged end is done apart (masked) by the last wave
    const long long nwaves = (long long)gridDim.x * MSM_WAVES;
    const long long wi = (long long)blockIdx.x * MSM_WAVES + w;
    constexpr int UNIT = STAGED ? 16 : 4;
    const long long steps = F / UNIT;
    const long long s0 = steps * wi / nwaves, s1 = steps * (wi + 1) / nwaves;

    int gmC[CT];
    double mk[CT]; // 1 for a real Gaussian, 0 for padding: masks by multiplication, so
                   // that hipcc cannot sink the loads under a branch
#pragma unroll
    for (int c = 0; c < CT; c++) {
        gmC[c] = gmA[c] >= 0 ? gmA[c] : 0;
        stA[c] = stA[c] < N ? stA[c] : 0;
        mk[c] = gmA[c] >= 0 ? 1.0 : 0.0;
    }
    auto mfmas = [&](const double *wv, const double *xv) {
        // feature = k0 + k1 x' + k2 x'^2 with (k0,k1,k2) fixed per lane: no branches
        double ft[NE];
#pragma unroll
        for (int n = 0; n < NE; n++) {
            const double xo = xv[n] - on[n];
            ft[n] = fma(xo, fma(xo, k2[n], k1[n]), k0[n]);
        }
#pragma unroll
        for (int c = 0; c < CT; c++)
#pragma unroll
            for (int n = 0; n < NE; n++)
                acc[c][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[c], ft[n], acc[c][n], 0, 0, 0);
    };
    if (STAGED) {
        // per-wave LDS stage: fx[16][XS] extended frames | ps[16][GW] | gs[16][N] | ol[DP]
        // The stage writer turns raw frames into [x', 1, 0.., x'^2, 0..] once, so the B
        // operand of every MFMA is a plain LDS read (no per-k-step feature arithmetic).
        const int XS = NE * 16; // >= 2*DP; XS*2 dwords = 32 (mod 64) for odd NE: conflict-free rows
        const int SZ = 16 * (XS + GW + N) + DP;
        double *fx = lds + (size_t)w * SZ, *ps = fx + 16 * XS, *gs = ps + 16 * GW, *ol = gs + 16 * N;
        // (the offset vector is read now and parked in LDS after the first stage's loads have
        // been issued, so that the kernel's set-up runs under their HBM latency)
        const double og_l = l < D ? oglob[l] : 0.0, og_h = l + WAVE < D ? oglob[l + WAVE < D ? l + WAVE : 0] : 0.0;
        // 16-byte pieces moved per stage: X 8*D, posteriors 8*GW, gamma 8*N; lane l takes
        // pieces l + 64u.  Bounds: 8*D <= 64*NE, 8*GW <= 64*2*CT, 8*N <= 64*2.  Surplus lanes
        // repeat the last piece, load and store alike (same value to the same place): no
        // predicates, no branches in the stage writer.
        constexpr int NXL = NE, NPL = 2 * CT, NGL = 2;
        const int nxp = 8 * D, npp = 8 * GW, ngp = 8 * N, ppr = GW / 2;
        const float rD = 1.0f / (float)D, rppr = 1.0f / (float)ppr;
        v2d rx[NXL], rp[NPL], rg[NGL];
        unsigned offp[NPL]; // posterior piece -> element offset inside a stage (rows strided by G)
        unsigned pcp[NPL], pcx[NXL], pcg[NGL]; // clamped piece indices
#pragma unroll
        for (int u = 0; u < NPL; u++) {
            int pc = l + 64 * u;
            pc = pc < npp ? pc : npp - 1;
            pcp[u] = (unsigned)pc;
            int row, c2;
            divmod_small(pc, ppr, rppr, row, c2);
            offp[u] = (unsigned)(row * G + gmin + 2 * c2);
        }
#pragma unroll
        for (int u = 0; u < NGL; u++) {
            const int pc = l + 64 * u;
            pcg[u] = (unsigned)(pc < ngp ? pc : ngp - 1);
        }
        // where the two doubles of X piece u land: (slab offset << 8) | coefficient index
        unsigned xa[NXL], xb[NXL];
#pragma unroll
        for (int u = 0; u < NXL; u++) {
            int pc = l + 64 * u;
            pc = pc < nxp ? pc : nxp - 1;
            pcx[u] = (unsigned)pc;
            const int e0 = 2 * pc;
            int r0, d0;
            divmod_small(e0, D, rD, r0, d0);
            const int r1 = d0 + 1 < D ? r0 : r0 + 1, d1 = d0 + 1 < D ? d0 + 1 : 0;
            xa[u] = ((unsigned)(r0 * XS + d0) << 8) | (unsigned)d0;
            xb[u] = ((unsigned)(r1 * XS + d1) << 8) | (unsigned)d1;
        }
        auto fetch = [&](long long stg) {
            const long long f = stg * 16;
            const v2d *xsrc = (const v2d *)(X + uniform64(f * D));
            const v2d *gsrc = (const v2d *)(gamma + uniform64(f * N));
            const double *psrc = post + uniform64(f * G);
#pragma unroll
            for (int u = 0; u < NXL; u++) rx[u] = xsrc[pcx[u]];
#pragma unroll
            for (int u = 0; u < NPL; u++) rp[u] = *(const v2d *)(psrc + offp[u]);
#pragma unroll
            for (int u = 0; u < NGL; u++) rg[u] = gsrc[pcg[u]];
        };
        if (s0 < s1) fetch(s0);
        if (l < DP) ol[l] = og_l;
        if (l + WAVE < DP) ol[l + WAVE] = og_h;
        for (int r = 0; r < 16; r++)               // constant columns: the 1 and the zeros
            for (int col = l; col < XS; col += WAVE) fx[r * XS + col] = col == D ? 1.0 : 0.0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (long long stg = s0; stg < s1; stg++) {
#pragma unroll
            for (int u = 0; u < NXL; u++) {
                const double x0 = rx[u][0] - ol[xa[u] & 255u], x1 = rx[u][1] - ol[xb[u] & 255u];
                fx[xa[u] >> 8] = x0;
                fx[(xa[u] >> 8) + DP] = x0 * x0;
                fx[xb[u] >> 8] = x1;
                fx[(xb[u] >> 8) + DP] = x1 * x1;
            }
#pragma unroll
            for (int u = 0; u < NPL; u++) ((v2d *)ps)[pcp[u]] = rp[u];
#pragma unroll
            for (int u = 0; u < NGL; u++) ((v2d *)gs)[pcg[u]] = rg[u];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // LDS is in order per wave
            fetch(stg + 1 < s1 ? stg + 1 : stg); // in flight under this stage's MFMAs
            // four k-steps per stage, two per iteration of a rolled loop (unrolling all four
            // makes hipcc shuffle the accumulators between AGPRs and VGPRs).  Two operand
            // sets alternate: the operands of the next step are read from LDS before the
            // MFMAs of the current one are issued, since a lone wave has nothing else to hide
            // the LDS latency with.
            double gA[CT], pA[CT], fA[NE], gB[CT], pB[CT], fB[NE];
            const double *gsl[CT], *psl[CT];
#pragma unroll
            for (int c = 0; c < CT; c++) {
                gsl[c] = gs + kq * N + stA[c];
                psl[c] = ps + kq * GW + (gmC[c] - gmin);
            }
            const double *fxl = fx + kq * XS + j;
            auto rd = [&](int q, double (&g)[CT], double (&p)[CT], double (&f)[NE]) {
                const int og = 4 * q * N, op = 4 * q * GW, of = 4 * q * XS; // wave-uniform
#pragma unroll
                for (int c = 0; c < CT; c++) {
                    g[c] = gsl[c][og];
                    p[c] = psl[c][op];
                }
#pragma unroll
                for (int n = 0; n < NE; n++) f[n] = fxl[of + 16 * n];
            };
            auto run = [&](const double (&g)[CT], const double (&p)[CT], const double (&f)[NE]) {
                double wv[CT];
#pragma unroll
                for (int c = 0; c < CT; c++) wv[c] = g[c] * p[c] * mk[c];
#pragma unroll
                for (int c = 0; c < CT; c++)
#pragma unroll
                    for (int n = 0; n < NE; n++)
                        acc[c][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[c], f[n], acc[c][n], 0, 0, 0);
            };
            rd(0, gA, pA, fA);
#pragma unroll 1
            for (int h = 0; h < 2; h++) {
                __builtin_amdgcn_sched_barrier(0);
                rd(2 * h + 1, gB, pB, fB);
                run(gA, pA, fA);
                __builtin_amdgcn_sched_barrier(0);
                rd(h == 0 ? 2 : 3, gA, pA, fA); // the last read is a repeat, unused
                run(gB, pB, fB);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads(); // the stages alias the fold buffer below
    } else
    // Software pipeline over k-steps: operands are fetched MSM_PD k-steps ahead.  Addresses
    // are a wave-uniform base (scalar registers) plus a 32-bit per-lane element offset that
    // advances by a constant per step: one 32-bit add per load, no address rebuild, no
    // predication (hipcc then counts vmcnt instead of draining it).  Past the wave's last
    // step the offsets stop advancing, so the run-ahead re-reads valid data.  With one
    // wave per SIMD nothing else hides the HBM latency.
    if (s0 < s1) {
        feature_setup();
        const long long fw = s0 * 4; // first frame of this wave (wave-uniform)
        const double *gw = gamma + uniform64(fw * N);
        const double *pw = post + uniform64(fw * G);
        const double *xw = X + uniform64(fw * D);
        unsigned og[CT], op[CT], ox[NE];
#pragma unroll
        for (int c = 0; c < CT; c++) {
            og[c] = (unsigned)(kq * N + stA[c]);
            op[c] = (unsigned)(kq * G + gmC[c]);
        }
#pragma unroll
        for (int n = 0; n < NE; n++) ox[n] = (unsigned)(kq * D + dn[n]);
        const long long nst = s1 - s0;
        long long ld = 0; // stages loaded so far
        double wq[MSM_PD][CT], xq[MSM_PD][NE];
        auto load = [&](double *wo, double *xo) {
#pragma unroll
            for (int c = 0; c < CT; c++) wo[c] = gw[og[c]] * pw[op[c]] * mk[c];
#pragma unroll
            for (int n = 0; n < NE; n++) xo[n] = xw[ox[n]];
            ld++;
            const unsigned adv = ld < nst ? 4u : 0u; // wave-uniform
#pragma unroll
            for (int c = 0; c < CT; c++) {
                og[c] += adv * (unsigned)N;
                op[c] += adv * (unsigned)G;
            }
#pragma unroll
            for (int n = 0; n < NE; n++) ox[n] += adv * (unsigned)D;
        };
#pragma unroll
        for (int u = 0; u < MSM_PD; u++) load(wq[u], xq[u]);
        for (long long st = 0; st < nst; st += MSM_PD) {
#pragma unroll
            for (int u = 0; u < MSM_PD; u++) {
                if (st + u < nst) mfmas(wq[u], xq[u]);
                load(wq[u], xq[u]);
            }
        }
    }
    // the corpus' ragged end (F mod UNIT frames): masked k-steps by the last wave; lanes of
    // frames past the end contribute zeros
    {
        const bool tail = (F % UNIT) != 0 && wi == nwaves - 1; // uniform per wave
        if (tail) {
            feature_setup();
            for (long long tb = steps * UNIT; tb < F; tb += 4) {
                long long t = tb + kq;
                const double okf = t < F ? 1.0 : 0.0;
                t = t < F ? t : F - 1;
                double wv[CT], xv[NE];
#pragma unroll
                for (int c = 0; c < CT; c++)
                    wv[c] = gamma[t * N + stA[c]] * post[t * G + gmC[c]] * (mk[c] * okf);
#pragma unroll
                for (int n = 0; n < NE; n++) xv[n] = X[t * D + dn[n]];
                mfmas(wv, xv);
            }
        }
    }
    // Fold the block's waves in wave order and write the block's partial, all four waves at
    // work: in two rounds of half the tiles, every wave parks its tiles in LDS, then each wave
    // adds up (wave 0 + wave 1 + wave 2 + wave 3, in that order) a quarter of the round's
    // tile rows and stores them.
    constexpr int TILES = CT * NE, TPR = (TILES + 1) / 2; // tiles per round
#pragma unroll
    for (int rnd = 0; rnd < 2; rnd++) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CT; c++)
#pragma unroll
            for (int n = 0; n < NE; n++) {
                const int tl = c * NE + n - rnd * TPR; // tile index inside this round
                if (tl >= 0 && tl < TPR) {
#pragma unroll
                    for (int r = 0; r < 4; r++) lds[((size_t)(w * TPR + tl) * 4 + r) * 64 + l] = acc[c][n][r];
                }
            }
        __syncthreads();
        const int nt = (TILES - rnd * TPR) < TPR ? (TILES - rnd * TPR) : TPR;
        for (int q = w; q < nt * 4; q += MSM_WAVES) { // q = (tile, register row)
            const int tl = q >> 2, r = q & 3, t = tl + rnd * TPR, c = t / NE, n = t - c * NE;
            double v = lds[((size_t)(0 * TPR + tl) * 4 + r) * 64 + l];
#pragma unroll
            for (int ww = 1; ww < MSM_WAVES; ww++) v += lds[((size_t)(ww * TPR + tl) * 4 + r) * 64 + l];
            const int gp = (c0 + c) * 16 + kq + 4 * r;
            if (c0 + c < NT) part[((size_t)blockIdx.x * NT * 16 + gp) * ES + 16 * n + j] = v;
        }
    }
}

} // namespace ghmm
