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
#include <type_traits>
#include <hip/hip_runtime.h>

#include "ghmm_kernels.hpp"

namespace ghmm {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// a 64-bit value that is the same in every lane, moved to scalar registers
__device__ inline long long uniform64(long long v)
{
    const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
    const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned)lo;
}

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

// How a Gaussian's statistics are taken (k_mixstats_mfma / k_mixstats / k_reduce_all):
//   0  well-conditioned around the global offset: the expanded sums of the matrix-core kernel
//   1  ill-conditioned only because EVERY variance sits at the 1e-5 floor (a component that EM
//      has collapsed onto one frame, or onto a few identical ones): still the expanded sums —
//      its variance statistic is far below the floor, where any value gives the same M-step —
//      checked by k_reduce_all, which recomputes the Gaussian exactly if the check fails
//   2  ill-conditioned otherwise: the direct-form sums of the vector-ALU kernel
__device__ inline int stats_class(bool real, double cond_global, double wide_coefficients)
{
    if (!real || !(cond_global > COND_MAX)) return 0;
    return wide_coefficients == 0.0 ? 1 : 2;
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
               double *__restrict__ dtile, int *__restrict__ tshift, int *__restrict__ scls,
               int *__restrict__ hflag)
{
    __shared__ double sh0[64], sh1[64], sh2[64];
    const int gp = blockIdx.x, t = threadIdx.x;
    const int c = gp >> 4, j = gp & 15, KS = DP / 2;
    const int i = gp / Mp, m = gp % Mp;
    const bool real = (i < N) && (m < M);
    const int g = real ? i * M + m : -1;
    double *Wc = Wm + (size_t)c * KS * 64;
    const bool shifted = tnext[c] != 0; // this tile has its own offset (k_prepare_tiles / k_reduce_all)
    double c0 = 0.0, cg = 0.0, wide = 0.0; // wide: coefficients whose variance is above the floor
    for (int d = t; d < DP; d += 64) {
        double bc = 0.0, ac = 0.0;
        if (real && d < D) {
            const double mraw = mean[(size_t)g * D + d], iv = inv_var[(size_t)g * D + d];
            wide += iv < 0.99 / FLOOR ? 1.0 : 0.0;
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
    sh2[t] = wide;
    __syncthreads();
    for (int k = 32; k > 0; k >>= 1) {
        if (t < k) {
            sh0[t] += sh0[t + k];
            sh1[t] += sh1[t + k];
            sh2[t] += sh2[t + k];
        }
        __syncthreads();
    }
    if (t == 0) {
        cg = sh0[0];
        c0 = sh1[0];
        scls[gp] = stats_class(real, cg, sh2[0]);
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
        if (scls[gp] == 2) {
            sflag[0] = epoch;
            __hip_atomic_store(hflag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // (host memory: tells the host, some time later, to launch k_mixstats)
        }
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
             int *__restrict__ tshift, int *__restrict__ sflag, int delta, int *__restrict__ scls,
             int *__restrict__ hflag)
{
    extern __shared__ double vs[]; // [lds_doubles] staging of mstep_state | og[DP] | red[MSF_THREADS]
    double *og = vs + lds_doubles, *red = og + DP;
    const int G = N * M, i = blockIdx.x, tid = threadIdx.x, KS = DP / 2;
    {
        // thread (d, part) adds every np-th Gaussian of column d; parts are added in order
        const double *num_c = stats + (size_t)N * N + 2 * (size_t)N, *num_mu = num_c + G;
        const int np = MSF_THREADS / DP, d = tid % DP, part = tid / DP;
        double o = 0.0, cn = 0.0;
        if (part < np) {
            // eight Gaussians' loads in flight at a time (clamped indices, nothing predicated), then
            // the additions in the same order as before: the rolled loop ran one L2 round trip per
            // Gaussian — seven in a row at 10x8, more than half of this kernel's 11 us
            for (int g0 = part; g0 < G; g0 += 8 * np) {
                double vc[8], vm[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int g = g0 + k * np, gg = g < G ? g : G - 1;
                    vc[k] = num_c[gg];
                    vm[k] = num_mu[(size_t)gg * D + (d < D ? d : 0)];
                }
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (g0 + k * np < G) {
                        cn += vc[k];
                        if (d < D) o += vm[k];
                    }
            }
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
        double cg = 0.0, c0 = 0.0, wide = 0.0;
        for (int d = l; d < DP; d += 64) {
            double bc = 0.0, ac = 0.0, dt = 0.0;
            if (d < D) dt = shifted ? otile[(size_t)ct * DP + d] - og[d] : 0.0;
            if (real && d < D) {
                const double mu = mean[(size_t)g * D + d] - og[d], iv = inv_var[(size_t)g * D + d];
                wide += iv < 0.99 / FLOOR ? 1.0 : 0.0;
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
            wide += __shfl_xor(wide, o, 64);
        }
        if (l == 0) {
            gmap[gp] = g;
            wkp[gp] = real ? wk[g] : 0.0;
            logwkp[gp] = real ? logwk[g] : -1.0e300;
            Wc[(D >> 2) * 64 + (D & 3) * 16 + j] = real ? -0.5 * c0 : 0.0;
            condg[gp] = real ? cg : 0.0;
            condt[gp] = real ? c0 : 0.0;
            tshift[ct] = shifted ? 1 : 0;
            const int cls = stats_class(real, cg, wide);
            scls[gp] = cls;
            if (real && c0 > COND_MAX) anyflag[0] = epoch;
            if (cls == 2) {
                sflag[0] = epoch;
                __hip_atomic_store(hflag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // (host memory, see k_prepare_mfma)
            }
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
// keeps x' = x - oglob of its 16 frames in a 5 KB slab and, where registers allow, as MFMA
// operands (x', x'^2) for all of the frame tile's Gaussian tiles; the kernel stays under 128
// VGPRs.  An f64 MFMA and vector-ALU work never overlap on a SIMD: the
// kernel's time is MFMA time + vector time, so the epilogue is built to need few vector
// instructions:
//   - the product is taken TRANSPOSED (A = Gaussians x k, B = k x frames): a lane then holds
//     FOUR GAUSSIANS OF ONE FRAME (rows (l>>4) + 4r of column l&15), and the Gaussians of a
//     tile are placed on the MFMA rows so that those four are neighbours (slot p = 4(l>>4) + r):
//     the sum over a state's mixtures is 3 additions in the lane (+ one v_permlane16_swap
//     level for 8 mixtures, + one v_permlane32_swap level for 16), every lane ends up with
//     its own state's b_i, and a lane's four posteriors are 32 contiguous bytes of the
//     posterior row: two 16-byte stores
//   - exp: 32-entry table of 2^(j/32) in LDS + degree-6 polynomial on |r| <= ln2/64
//     (truncation 3.5e-18; measured <= 1 ulp), magic-number rounding, integer clamp, v_ldexp_f64
//   - v / b_i is a reciprocal (hardware seed + two Newton steps); tiny, huge or zero b_i
//     (exact power-of-two rescale; 0 when b_i == 0, TF:1773-1778) take a wave-uniform side path
//   - K steps, mixture padding and "posteriors wanted" are compile-time
constexpr int EMS_WAVES = 16;
// Measurement builds only (profiles/tools/lab.sh compiles the library with -DGHMM_LAB=<bits> into
// separate .so files; the product build leaves it 0): parts of k_emission_sched switched off so
// that the rest can be timed on the hardware.  1: no exp, 2: no stores, 4: no MFMA chain,
// 8: no state sums / reciprocal; posterior stores 16: as whole lines from consecutive lanes (data
// misplaced), 512: whole lines from lanes 8 apart (data misplaced), 128: into one 4 MB window (no HBM
// writes).  Results: profiles/r3_lab_stores.txt.  8192: cycle stamps (s_memtime) around the phases of a
// wave of the statistics kernel, printed by a few waves (profiles/tools/stamp_run.py, profiles/r3_lab_mixstats.txt).
#ifndef GHMM_LAB
#define GHMM_LAB 0
#endif
#ifndef GHMM_EMS_W
#define GHMM_EMS_W 12 // waves per block: 3 per SIMD, 168 registers each (measured against 16 and 8)
#endif
#ifndef GHMM_EMS_PF
#define GHMM_EMS_PF 1 // next frame tile fetched into registers while the current one computes
#endif
// waves per block: 12 (3 per SIMD, 168 registers each) hold the x operands of a frame tile
// (x', x'^2: 40 registers), the next frame tile on its way from HBM (20) and, where a state
// spans 2 or 4 tiles, the densities of those tiles until the state's sum is known
#ifndef GHMM_EMS_W32
#define GHMM_EMS_W32 12 // the same for states of 32 / 64 mixtures
#endif
// How a frame tile's operands reach the MFMA lanes, per output mode (measured in one session,
// profiles/r2_lab_emission_direct.txt): through a per-wave LDS slab (16-byte coalesced loads,
// 12 waves of 168 registers) for the trainer's and the log-b variants; for the b-only variant
// (recogniser, 2 000-state configuration) straight from HBM into the lanes' registers — ten
// 8-byte loads per lane from one row pointer — which needs no slab (14 Gaussian tiles per chunk
// instead of 8) and fits 128 registers, i.e. 16 waves.  States of 32 / 64 mixtures (12 waves either
// way) gain 2.7 % from the larger chunks in the trainer's variant too.  Bit mask: bit OUT for
// <= 16 mixtures, bit 3 + OUT for 32 / 64.
#ifndef GHMM_EMS_DIRECT
#define GHMM_EMS_DIRECT (1 | (1 << 3) | (1 << 4))
#endif
__host__ __device__ constexpr bool ems_direct(int MP, int OUT)
{
    return ((GHMM_EMS_DIRECT >> (OUT + (MP >= 32 ? 3 : 0))) & 1) != 0;
}
#ifndef GHMM_EMS_TC
#define GHMM_EMS_TC 8 // most Gaussian tiles per chunk with the slab (each chunk reads the frames again)
#endif
#ifndef GHMM_EMS_TCD
#define GHMM_EMS_TCD 14 // the same without it
#endif
__host__ __device__ constexpr int ems_tc_cap(int MP, int OUT) { return ems_direct(MP, OUT) ? GHMM_EMS_TCD : GHMM_EMS_TC; }
__host__ __device__ constexpr int ems_waves(int MP, int OUT)
{
    return MP >= 32 ? GHMM_EMS_W32 : (ems_direct(MP, OUT) ? 16 : GHMM_EMS_W);
}

// 2^(j/32), j = 0..31, correctly rounded
__device__ const double EXP2_32[32] = {
    0x1.0000000000000p+0, 0x1.059b0d3158574p+0, 0x1.0b5586cf9890fp+0, 0x1.11301d0125b51p+0,
    0x1.172b83c7d517bp+0, 0x1.1d4873168b9aap+0, 0x1.2387a6e756238p+0, 0x1.29e9df51fdee1p+0,
    0x1.306fe0a31b715p+0, 0x1.371a7373aa9cbp+0, 0x1.3dea64c123422p+0, 0x1.44e086061892dp+0,
    0x1.4bfdad5362a27p+0, 0x1.5342b569d4f82p+0, 0x1.5ab07dd485429p+0, 0x1.6247eb03a5585p+0,
    0x1.6a09e667f3bcdp+0, 0x1.71f75e8ec5f74p+0, 0x1.7a11473eb0187p+0, 0x1.82589994cce13p+0,
    0x1.8ace5422aa0dbp+0, 0x1.93737b0cdc5e5p+0, 0x1.9c49182a3f090p+0, 0x1.a5503b23e255dp+0,
    0x1.ae89f995ad3adp+0, 0x1.b7f76f2fb5e47p+0, 0x1.c199bdd85529cp+0, 0x1.cb720dcef9069p+0,
    0x1.d5818dcfba487p+0, 0x1.dfc97337b9b5fp+0, 0x1.ea4afa2a490dap+0, 0x1.f50765b6e4540p+0};

// 2^(j/256), j = 0..255, correctly rounded (60-digit decimal arithmetic)
__device__ const double EXP2_256[256] = {
    0x1.0000000000000p+0, 0x1.00b1afa5abcbfp+0, 0x1.0163da9fb3335p+0, 0x1.02168143b0281p+0,
    0x1.02c9a3e778061p+0, 0x1.037d42e11bbccp+0, 0x1.04315e86e7f85p+0, 0x1.04e5f72f654b1p+0,
    0x1.059b0d3158574p+0, 0x1.0650a0e3c1f89p+0, 0x1.0706b29ddf6dep+0, 0x1.07bd42b72a836p+0,
    0x1.0874518759bc8p+0, 0x1.092bdf66607e0p+0, 0x1.09e3ecac6f383p+0, 0x1.0a9c79b1f3919p+0,
    0x1.0b5586cf9890fp+0, 0x1.0c0f145e46c85p+0, 0x1.0cc922b7247f7p+0, 0x1.0d83b23395decp+0,
    0x1.0e3ec32d3d1a2p+0, 0x1.0efa55fdfa9c5p+0, 0x1.0fb66affed31bp+0, 0x1.1073028d7233ep+0,
    0x1.11301d0125b51p+0, 0x1.11edbab5e2ab6p+0, 0x1.12abdc06c31ccp+0, 0x1.136a814f204abp+0,
    0x1.1429aaea92de0p+0, 0x1.14e95934f312ep+0, 0x1.15a98c8a58e51p+0, 0x1.166a45471c3c2p+0,
    0x1.172b83c7d517bp+0, 0x1.17ed48695bbc0p+0, 0x1.18af9388c8deap+0, 0x1.1972658375d2fp+0,
    0x1.1a35beb6fcb75p+0, 0x1.1af99f8138a1cp+0, 0x1.1bbe084045cd4p+0, 0x1.1c82f95281c6bp+0,
    0x1.1d4873168b9aap+0, 0x1.1e0e75eb44027p+0, 0x1.1ed5022fcd91dp+0, 0x1.1f9c18438ce4dp+0,
    0x1.2063b88628cd6p+0, 0x1.212be3578a819p+0, 0x1.21f49917ddc96p+0, 0x1.22bdda27912d1p+0,
    0x1.2387a6e756238p+0, 0x1.2451ffb82140ap+0, 0x1.251ce4fb2a63fp+0, 0x1.25e85711ece75p+0,
    0x1.26b4565e27cddp+0, 0x1.2780e341ddf29p+0, 0x1.284dfe1f56381p+0, 0x1.291ba7591bb70p+0,
    0x1.29e9df51fdee1p+0, 0x1.2ab8a66d10f13p+0, 0x1.2b87fd0dad990p+0, 0x1.2c57e39771b2fp+0,
    0x1.2d285a6e4030bp+0, 0x1.2df961f641589p+0, 0x1.2ecafa93e2f56p+0, 0x1.2f9d24abd886bp+0,
    0x1.306fe0a31b715p+0, 0x1.31432edeeb2fdp+0, 0x1.32170fc4cd831p+0, 0x1.32eb83ba8ea32p+0,
    0x1.33c08b26416ffp+0, 0x1.3496266e3fa2dp+0, 0x1.356c55f929ff1p+0, 0x1.36431a2de883bp+0,
    0x1.371a7373aa9cbp+0, 0x1.37f26231e754ap+0, 0x1.38cae6d05d866p+0, 0x1.39a401b7140efp+0,
    0x1.3a7db34e59ff7p+0, 0x1.3b57fbfec6cf4p+0, 0x1.3c32dc313a8e5p+0, 0x1.3d0e544ede173p+0,
    0x1.3dea64c123422p+0, 0x1.3ec70df1c5175p+0, 0x1.3fa4504ac801cp+0, 0x1.40822c367a024p+0,
    0x1.4160a21f72e2ap+0, 0x1.423fb2709468ap+0, 0x1.431f5d950a897p+0, 0x1.43ffa3f84b9d4p+0,
    0x1.44e086061892dp+0, 0x1.45c2042a7d232p+0, 0x1.46a41ed1d0057p+0, 0x1.4786d668b3237p+0,
    0x1.486a2b5c13cd0p+0, 0x1.494e1e192aed2p+0, 0x1.4a32af0d7d3dep+0, 0x1.4b17dea6db7d7p+0,
    0x1.4bfdad5362a27p+0, 0x1.4ce41b817c114p+0, 0x1.4dcb299fddd0dp+0, 0x1.4eb2d81d8abffp+0,
    0x1.4f9b2769d2ca7p+0, 0x1.508417f4531eep+0, 0x1.516daa2cf6642p+0, 0x1.5257de83f4eefp+0,
    0x1.5342b569d4f82p+0, 0x1.542e2f4f6ad27p+0, 0x1.551a4ca5d920fp+0, 0x1.56070dde910d2p+0,
    0x1.56f4736b527dap+0, 0x1.57e27dbe2c4cfp+0, 0x1.58d12d497c7fdp+0, 0x1.59c0827ff07ccp+0,
    0x1.5ab07dd485429p+0, 0x1.5ba11fba87a03p+0, 0x1.5c9268a5946b7p+0, 0x1.5d84590998b93p+0,
    0x1.5e76f15ad2148p+0, 0x1.5f6a320dceb71p+0, 0x1.605e1b976dc09p+0, 0x1.6152ae6cdf6f4p+0,
    0x1.6247eb03a5585p+0, 0x1.633dd1d1929fdp+0, 0x1.6434634ccc320p+0, 0x1.652b9febc8fb7p+0,
    0x1.6623882552225p+0, 0x1.671c1c70833f6p+0, 0x1.68155d44ca973p+0, 0x1.690f4b19e9538p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6b052fa75173ep+0, 0x1.6c012750bdabfp+0, 0x1.6cfdcddd47645p+0,
    0x1.6dfb23c651a2fp+0, 0x1.6ef9298593ae5p+0, 0x1.6ff7df9519484p+0, 0x1.70f7466f42e87p+0,
    0x1.71f75e8ec5f74p+0, 0x1.72f8286ead08ap+0, 0x1.73f9a48a58174p+0, 0x1.74fbd35d7cbfdp+0,
    0x1.75feb564267c9p+0, 0x1.77024b1ab6e09p+0, 0x1.780694fde5d3fp+0, 0x1.790b938ac1cf6p+0,
    0x1.7a11473eb0187p+0, 0x1.7b17b0976cfdbp+0, 0x1.7c1ed0130c132p+0, 0x1.7d26a62ff86f0p+0,
    0x1.7e2f336cf4e62p+0, 0x1.7f3878491c491p+0, 0x1.80427543e1a12p+0, 0x1.814d2add106d9p+0,
    0x1.82589994cce13p+0, 0x1.8364c1eb941f7p+0, 0x1.8471a4623c7adp+0, 0x1.857f4179f5b21p+0,
    0x1.868d99b4492edp+0, 0x1.879cad931a436p+0, 0x1.88ac7d98a6699p+0, 0x1.89bd0a478580fp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8be05bad61778p+0, 0x1.8cf3216b5448cp+0, 0x1.8e06a5e0866d9p+0,
    0x1.8f1ae99157736p+0, 0x1.902fed0282c8ap+0, 0x1.9145b0b91ffc6p+0, 0x1.925c353aa2fe2p+0,
    0x1.93737b0cdc5e5p+0, 0x1.948b82b5f98e5p+0, 0x1.95a44cbc8520fp+0, 0x1.96bdd9a7670b3p+0,
    0x1.97d829fde4e50p+0, 0x1.98f33e47a22a2p+0, 0x1.9a0f170ca07bap+0, 0x1.9b2bb4d53fe0dp+0,
    0x1.9c49182a3f090p+0, 0x1.9d674194bb8d5p+0, 0x1.9e86319e32323p+0, 0x1.9fa5e8d07f29ep+0,
    0x1.a0c667b5de565p+0, 0x1.a1e7aed8eb8bbp+0, 0x1.a309bec4a2d33p+0, 0x1.a42c980460ad8p+0,
    0x1.a5503b23e255dp+0, 0x1.a674a8af46052p+0, 0x1.a799e1330b358p+0, 0x1.a8bfe53c12e59p+0,
    0x1.a9e6b5579fdbfp+0, 0x1.ab0e521356ebap+0, 0x1.ac36bbfd3f37ap+0, 0x1.ad5ff3a3c2774p+0,
    0x1.ae89f995ad3adp+0, 0x1.afb4ce622f2ffp+0, 0x1.b0e07298db666p+0, 0x1.b20ce6c9a8952p+0,
    0x1.b33a2b84f15fbp+0, 0x1.b468415b749b1p+0, 0x1.b59728de5593ap+0, 0x1.b6c6e29f1c52ap+0,
    0x1.b7f76f2fb5e47p+0, 0x1.b928cf22749e4p+0, 0x1.ba5b030a1064ap+0, 0x1.bb8e0b79a6f1fp+0,
    0x1.bcc1e904bc1d2p+0, 0x1.bdf69c3f3a207p+0, 0x1.bf2c25bd71e09p+0, 0x1.c06286141b33dp+0,
    0x1.c199bdd85529cp+0, 0x1.c2d1cd9fa652cp+0, 0x1.c40ab5fffd07ap+0, 0x1.c544778fafb22p+0,
    0x1.c67f12e57d14bp+0, 0x1.c7ba88988c933p+0, 0x1.c8f6d9406e7b5p+0, 0x1.ca3405751c4dbp+0,
    0x1.cb720dcef9069p+0, 0x1.ccb0f2e6d1675p+0, 0x1.cdf0b555dc3fap+0, 0x1.cf3155b5bab74p+0,
    0x1.d072d4a07897cp+0, 0x1.d1b532b08c968p+0, 0x1.d2f87080d89f2p+0, 0x1.d43c8eacaa1d6p+0,
    0x1.d5818dcfba487p+0, 0x1.d6c76e862e6d3p+0, 0x1.d80e316c98398p+0, 0x1.d955d71ff6075p+0,
    0x1.da9e603db3285p+0, 0x1.dbe7cd63a8315p+0, 0x1.dd321f301b460p+0, 0x1.de7d5641c0658p+0,
    0x1.dfc97337b9b5fp+0, 0x1.e11676b197d17p+0, 0x1.e264614f5a129p+0, 0x1.e3b333b16ee12p+0,
    0x1.e502ee78b3ff6p+0, 0x1.e653924676d76p+0, 0x1.e7a51fbc74c83p+0, 0x1.e8f7977cdb740p+0,
    0x1.ea4afa2a490dap+0, 0x1.eb9f4867cca6ep+0, 0x1.ecf482d8e67f1p+0, 0x1.ee4aaa2188510p+0,
    0x1.efa1bee615a27p+0, 0x1.f0f9c1cb6412ap+0, 0x1.f252b376bba97p+0, 0x1.f3ac948dd7274p+0,
    0x1.f50765b6e4540p+0, 0x1.f6632798844f8p+0, 0x1.f7bfdad9cbe14p+0, 0x1.f91d802243c89p+0,
    0x1.fa7c1819e90d8p+0, 0x1.fbdba3692d514p+0, 0x1.fd3c22b8f71f1p+0, 0x1.fe9d96b2a23d9p+0};

// Table of the emission kernels' exp: 256 entries of 2^(j/256) (the argument reduced to |r| <=
// ln2/512, a cubic for exp(r) - 1 = r (1 + r/2 + r^2/6 + r^3/24), truncation r^5/120 = 4e-17) or,
// GHMM_EXP_TAB = 32, round 2's 32 entries with a quintic.  With 32 entries every lane of a
// ds_read_b64 group reads its own bank pair (or a broadcast); 256 entries conflict ~3-way on four
// reads per tile and save two of twelve f64 instructions per value — the f64 pipe, which MFMA and
// vector instructions share without overlap, is what the kernel is bound by.
#ifndef GHMM_MIX_TILEMAJOR
#define GHMM_MIX_TILEMAJOR 1 // statistics kernel: a stage's active tiles one after the other (0: k-step-major, round 2)
#endif
#ifndef GHMM_EXP_TAB
#define GHMM_EXP_TAB 256
#endif
constexpr int EXP_TAB = GHMM_EXP_TAB;

// exp on four values at once (four independent chains): x = n ln2/EXP_TAB + r, exp(x) =
// 2^(n / EXP_TAB) * 2^((n % EXP_TAB)/EXP_TAB) * exp(r).  `etab` = the table in LDS.
__device__ inline void exp_emis4(const v4d &x, const double *__restrict__ etab, double (&out)[4])
{
    double r[4], p[4];
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
        if (EXP_TAB == 256) {
            const double t = fma(xc, 0x1.71547652b82fep+8, 0x1.8p52); // 256 / ln2
            ki[q] = __double2loint(t);
            const double k = t - 0x1.8p52;
            r[q] = fma(-k, 0x1.62e42fe000000p-9, xc);  // ln2/256, high part (k * hi is exact)
            r[q] = fma(-k, 0x1.f473de6af278fp-38, r[q]);
            p[q] = 4.16666666666666666667e-02; // 1/24
        } else {
            const double t = fma(xc, 0x1.71547652b82fep+5, 0x1.8p52); // 32 / ln2
            ki[q] = __double2loint(t);
            const double k = t - 0x1.8p52;
            r[q] = fma(-k, 0x1.62e42fee00000p-6, xc);  // ln2/32, high part (k * hi is exact)
            r[q] = fma(-k, 0x1.a39ef35793c76p-38, r[q]);
            p[q] = 1.38888888888888888889e-03; // 1/720
        }
    }
    const double cf[5] = {8.33333333333333333333e-03, 4.16666666666666666667e-02,
                          1.66666666666666666667e-01, 0.5, 1.0};
#pragma unroll
    for (int t = (EXP_TAB == 256 ? 2 : 0); t < 5; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) p[q] = fma(p[q], r[q], cf[t]);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        // (the table is read here, not ahead of the polynomial: four waves per SIMD cover the
        // LDS latency, and the kernel sits at its 128-register limit)
        const double tj = etab[ki[q] & (EXP_TAB - 1)];
        out[q] = ldexp(fma(tj, p[q] * r[q], tj), ki[q] >> (EXP_TAB == 256 ? 8 : 5));
    }
}

// v_permlane16_swap / v_permlane32_swap with both operands the same value: the two results
// are the value of the lane's partner half and its own, whichever way round — their sum
// (max) is the sum (max) over the pair of 16-lane rows {0,1} / {2,3}, resp. over the two
// halves of the wave, in every lane.
__device__ inline void swap16_pair(double v, double &a, double &b)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    a = __hiloint2double((int)rh[0], (int)rl[0]);
    b = __hiloint2double((int)rh[1], (int)rl[1]);
}
__device__ inline void swap32_pair(double v, double &a, double &b)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    a = __hiloint2double((int)rh[0], (int)rl[0]);
    b = __hiloint2double((int)rh[1], (int)rl[1]);
}
// sum / max over the lanes l>>4 = kq that share a state of MPL mixtures (slots 4kq .. 4kq+3)
template <int MPL> __device__ inline double kq_sum(double v)
{
    double a, b;
    if (MPL >= 8) {
        swap16_pair(v, a, b);
        v = a + b;
    }
    if (MPL >= 16) {
        swap32_pair(v, a, b);
        v = a + b;
    }
    return v;
}
template <int MPL> __device__ inline double kq_max(double v)
{
    double a, b;
    if (MPL >= 8) {
        swap16_pair(v, a, b);
        v = fmax(a, b);
    }
    if (MPL >= 16) {
        swap32_pair(v, a, b);
        v = fmax(a, b);
    }
    return v;
}

// 1 / s for the posteriors; 0 when s == 0 (TF:1773-1778).  Normal-range s: hardware seed + two
// Newton steps (= the IEEE quotient's reciprocal in every case measured).  Tiny, huge, zero or
// NaN s anywhere in the wave: the exact power-of-two rescale on a side path (wave-uniform branch).
__device__ inline double recip_post(double s, double &pre)
{
    double rr = __builtin_amdgcn_rcp(s);
    rr = fma(rr, fma(-s, rr, 1.0), rr);
    rr = fma(rr, fma(-s, rr, 1.0), rr);
    pre = 1.0;
    const bool odd = !(s >= 1.0e-290 && s <= 1.0e290);
    if (__any(odd)) {
        // (the empty asm keeps this a branch: folded into selects, the side path would cost every
        // tile a second reciprocal and a dozen selects)
        asm volatile("" ::: "memory");
        const double sc = s < 1.0e-290 ? 0x1p600 : (s > 1.0e290 ? 0x1p-600 : 1.0);
        const double s2 = s * sc;
        double r2 = __builtin_amdgcn_rcp(s2);
        r2 = fma(r2, fma(-s2, r2, 1.0), r2);
        r2 = fma(r2, fma(-s2, r2, 1.0), r2);
        // a subnormal s: 1/s is beyond the largest double (the reference divides, TF:1776, and
        // gets a finite share): the caller scales the density by `pre` first, then by 1/(s pre)
        const bool sub = s < 0x1p-1020 && s > 0.0;
        rr = odd ? (s != 0.0 ? (sub ? r2 : r2 * sc) : 0.0) : rr;
        pre = sub ? sc : 1.0;
    }
    return rr;
}

// MFMA row <-> slot of a Gaussian inside its tile of 16: the four accumulator rows of a lane,
// (l>>4) + 4r, carry slots 4(l>>4) + r
__device__ __host__ inline int slot_row(int p) { return (p >> 2) + 4 * (p & 3); }

// LDS bytes of k_emission_sched for a chunk of TC tiles
__host__ __device__ inline size_t ems_lds_bytes(int TC, int DP, int waves, int MP, int OUT)
{
    const int KS = DP / 2, XS = DP + 2;
    return (size_t)TC * KS * 64 * 8 + (ems_direct(MP, OUT) ? 0 : (size_t)waves * 16 * XS * 8) + (size_t)DP * 8 + // (see the kernel)
           (size_t)TC * 16 * 8 + (size_t)TC * DP * 8 + EXP_TAB * 8 + (size_t)TC * 16 * 4 + (size_t)TC * 4 * 2 + 64;
}

// OUT 0: b (recogniser, RF:860-889); 1: b and posteriors (trainer, TF:1749-1783);
// 2: log b for the Viterbi lattice, m + log(sum exp(e - m)) like the oracle's definition
// (wkp then holds log wk).  condt: conditioning of every padded Gaussian around its tile's offset
// (slots beyond COND_MAX are re-evaluated in direct form where their density is not 0).  tfull[tile]: the tile's 16 slots are 16 consecutive real Gaussians
// starting at an even index and G is even (its posteriors go out as aligned 16-byte stores).
template <int KS, int MP, int OUT>
__global__ void __launch_bounds__(ems_waves(MP, OUT) * WAVE)
k_emission_sched(int N, int M, int D, int NT, int TC, long long F, const double *__restrict__ X,
                 const double *__restrict__ Wm, const double *__restrict__ oglob,
                 const double *__restrict__ wkp, const int *__restrict__ gmap,
                 double *__restrict__ b, double *__restrict__ post,
                 const double *__restrict__ dtile, const int *__restrict__ tshift,
                 const int *__restrict__ tfull, const double *__restrict__ condt,
                 const double *__restrict__ mean, const double *__restrict__ inv_var, int ntp)
{   // ntp: posteriors leave with non-temporal stores.  Written once and mostly never read (gamma is 0
    // for most states of a frame), they otherwise flush the 256 MB Infinity Cache of the frames, alpha^
    // and W that the iteration's other kernels read again: 10x8 step 0.245 -> 0.232 ms; with 19 GB of
    // them (64 mixtures) the host leaves it off (profiles/r3_lab_stores.txt, which also records what did
    // NOT pay: whole 128-byte lines per store instruction need the lanes' data turned through LDS, and
    // that costs what it gains)
    extern __shared__ double lds[];
    // slab row stride 2 * odd doubles: the 32 lanes of a ds_read_b64 group (16 frames x 2
    // k-columns) then fall on 32 different bank pairs
    constexpr int DP = 2 * KS, Q = KS / 2, XS = DP + 2, WV = ems_waves(MP, OUT);
    constexpr bool DIRECT = ems_direct(MP, OUT);
    constexpr int LOGMP = MP == 1 ? 0 : MP == 2 ? 1 : MP == 4 ? 2 : MP == 8 ? 3 : MP == 16 ? 4 : MP == 32 ? 5 : 6;
    constexpr int MPL = MP < 16 ? MP : 16, TPS = MP <= 16 ? 1 : MP / 16;
    constexpr int NS = MPL == 1 ? 4 : (MPL == 2 ? 2 : 1); // states per lane
    constexpr int GS = 4 / NS;                            // a state's registers in the lane
    const int G = N * M;
    double *Wl = lds;                                // [TC][KS][64], Gaussians on their MFMA rows
    double *xl = Wl + (size_t)TC * KS * 64;          // [WV][16][XS] (slab variant only)
    double *ol = xl + (DIRECT ? 0 : (size_t)WV * 16 * XS); // [DP]
    double *wkl = ol + DP;                           // [TC][16] by slot
    // tiles that hold a variance-floored component take that component's mean as their offset:
    // dl = offset - oglob (subtracted from the x operand of that tile), tsl = "shifted"
    double *dl = wkl + (size_t)TC * 16;              // [TC][DP]
    double *etab = dl + (size_t)TC * DP;             // [EXP_TAB]
    int *gml = (int *)(etab + EXP_TAB);                   // [TC][16] real Gaussian of a slot, -1 = padding
    int *tsl = gml + (size_t)TC * 16;                // [TC]
    int *tfl = tsl + TC;                             // [TC] all 16 slots real, consecutive, even start
    const int tid = threadIdx.x, l = tid & 63, j = l & 15, kq = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index, in a scalar register
    const int c0 = blockIdx.y * TC;
    const int tc = (NT - c0) < TC ? (NT - c0) : TC;
    // this wave's share of the work units and its first frame tile: the load is issued before
    // the block's tables are filled, so that its HBM latency runs under the set-up.
    // Work units = (frame tile, group of TPS Gaussian tiles), dealt to the grid's waves in
    // equal contiguous shares: with whole frame tiles per wave, 18 750 tiles on 4 096 waves
    // left 42 % of the CUs idle during the last of five rounds.  A wave that starts or ends
    // inside a frame tile loads that tile's slab like any other.
    const long long ntf = (F + 15) / 16;
    const long long FD = F * D;
    const bool x16 = (((unsigned long long)X) & 15ull) == 0;
    const int ng = (tc + TPS - 1) / TPS;
    const long long U = ntf * ng, GWV = (long long)gridDim.x * WV;
    const long long gwv = (long long)blockIdx.x * WV + w;
    long long u = U * gwv / GWV;
    const long long u1 = U * (gwv + 1) / GWV;
    long long tf = u / ng;
    int g0 = (int)(u - tf * ng);
    // 8 D pieces of 16 bytes per frame tile, lane l moves pieces l + 64u; surplus lanes repeat
    // the last piece, load and store alike
    constexpr int NP = (8 * (DP - 1) + 63) / 64; // D <= DP - 1
    v2d xn[NP];
    const int np = 8 * D;
    auto loadx = [&](long long tfx) {
        const v2d *src = (const v2d *)(X + uniform64(tfx * 16 * D));
#pragma unroll
        for (int u2 = 0; u2 < NP; u2++) {
            int pc = l + 64 * u2;
            pc = pc < np ? pc : np - 1;
            xn[u2] = src[pc];
        }
    };
    // DIRECT: lane (frame j = l & 15, kq = l >> 4) needs x[f0 + j][4 s + kq], s = 0 .. Q-1: ten
    // 8-byte loads from one per-lane row pointer with immediate offsets (rows past the corpus
    // clamp to its last frame; the last k-step, whose columns can lie beyond D, reads a clamped
    // column).  No slab, no alignment condition, one path for whole and ragged tiles.
    double xd[Q];
    const int kl = 4 * (Q - 1) + kq < D ? 4 * (Q - 1) + kq : D - 1; // column of the last k-step, clamped
    auto loadd = [&](long long tfx) {
        long long fr2 = tfx * 16 + j;
        fr2 = fr2 < F ? fr2 : F - 1;
        const double *row = X + fr2 * D;
#pragma unroll
        for (int s2 = 0; s2 < Q - 1; s2++) xd[s2] = row[4 * s2 + kq];
        xd[Q - 1] = row[kl];
    };
    bool have = GHMM_EMS_PF && u < u1 && (DIRECT || (x16 && tf * 16 + 16 <= F)); // xn / xd hold the frames of tile tf (wave-uniform)
    if (have) {
        if (DIRECT) loadd(tf);
        else loadx(tf);
    }
    {
        // the chunk's B fragments, 16 bytes per lane, ALL loads in flight before the first store
        // (a load-store loop of unknown trip count runs one L2 round trip per iteration while
        // the whole chip waits for its first MFMA; asking for them before the work-share
        // arithmetic above was measured and is not better)
        constexpr int WPL = (ems_tc_cap(MP, OUT) * KS * 32 + WV * WAVE - 1) / (WV * WAVE); // pairs per lane
        const int npair = tc * KS * 32;
        const v2d *wsrc = (const v2d *)(Wm + (size_t)c0 * KS * 64);
        v2d wq[WPL];
#pragma unroll
        for (int q = 0; q < WPL; q++) {
            const int pk = tid + q * WV * WAVE;
            wq[q] = wsrc[pk < npair ? pk : npair - 1];
        }
#pragma unroll
        for (int q = 0; q < WPL; q++) {
            const int pk = tid + q * WV * WAVE, k = 2 * pk;
            if (pk < npair) {
                Wl[(k & ~15) | slot_row(k & 15)] = wq[q][0];
                Wl[(k & ~15) | slot_row((k & 15) + 1)] = wq[q][1];
            }
        }
    }
    {
        // the chunk's small tables the same way: unconditional loads at clamped indices first
        // (tc * 16, tc, DP, tc * DP are all below the block's 768 threads), stores afterwards
        const int n16 = tc * 16, nd = tc * DP;
        const int k16 = tid < n16 ? tid : n16 - 1, kt = tid < tc ? tid : tc - 1;
        const int kd = tid < D ? tid : D - 1, kdl = tid < nd ? tid : nd - 1;
        const double wk_v = wkp[c0 * 16 + k16];
        const int gm_v = gmap[c0 * 16 + k16];
        const int ts_v = tshift[c0 + kt], tf_v = tfull[c0 + kt];
        double cd_v[16];
#pragma unroll
        for (int pp = 0; pp < 16; pp++) cd_v[pp] = condt[(c0 + kt) * 16 + pp];
        const double ol_v = oglob[kd];
        const double dl_v = dtile[(size_t)c0 * DP + kdl];
        if (tid < n16) {
            wkl[tid] = wk_v;
            gml[tid] = gm_v;
        }
        if (tid < tc) {
            tsl[tid] = ts_v;
            // low bit: whole tile of consecutive Gaussians; bits 16..31: slots whose Gaussian is too
            // ill-conditioned for the expanded form even around the tile's offset (condt)
            unsigned bm = 0;
#pragma unroll
            for (int pp = 0; pp < 16; pp++) bm |= (cd_v[pp] > COND_MAX ? 1u : 0u) << (16 + pp);
            tfl[tid] = (int)(bm | (tf_v != 0 ? 1u : 0u));
        }
        if (tid < DP) ol[tid] = tid < D ? ol_v : 0.0;
        if (tid < nd) dl[tid] = dl_v;
    }
    if (tid < EXP_TAB) etab[tid] = EXP_TAB == 256 ? EXP2_256[tid] : EXP2_32[tid & 31];
    __syncthreads();
    // per-tile flags of the chunk as bit masks in scalar registers (TC <= 32): no LDS round trip
    // per tile for them
    unsigned shm = 0, fum = 0, bdm = 0;
    for (int k = 0; k < tc; k++) {
        shm |= (tsl[k] != 0 ? 1u : 0u) << k;
        fum |= ((unsigned)tfl[k] & 1u) << k;
        bdm |= (((unsigned)tfl[k] >> 16) != 0 ? 1u : 0u) << k;
    }
    shm = (unsigned)__builtin_amdgcn_readfirstlane((int)shm);
    fum = (unsigned)__builtin_amdgcn_readfirstlane((int)fum);
    bdm = (unsigned)__builtin_amdgcn_readfirstlane((int)bdm); // tiles that hold such a slot
    double *xw = xl + w * 16 * XS;
    if (!DIRECT) {
        // constant columns of the slab: the 1 at column D, zeros beyond
        for (int k = l; k < 16 * (XS - D); k += WAVE) {
            const int r = k / (XS - D), e = k - r * (XS - D);
            xw[r * XS + D + e] = e == 0 ? 1.0 : 0.0;
        }
    }
    const double *xr = xw + j * XS + kq; // x operand: frame l&15, k = 4s + (l>>4)
    // DIRECT: the last k-step's column 4 (Q-1) + kq is a coefficient, the constant 1 (column D) or padding
    const int klast = 4 * (Q - 1) + kq;
    const bool last_real = klast < D;
    const double last_const = klast == D ? 1.0 : 0.0;
    if (GHMM_LAB & 32) return; // (lab: block set-up only)
    for (; u < u1; tf++, g0 = 0) {
        const long long f0 = tf * 16;
        const int g1 = (u1 - u) < (long long)(ng - g0) ? g0 + (int)(u1 - u) : ng;
        u += g1 - g0;
        const bool full = f0 + 16 <= F; // wave-uniform
        double a1[Q], a2[Q];
        if (DIRECT) {
            if (!have) loadd(tf);
#pragma unroll
            for (int s2 = 0; s2 < Q - 1; s2++) a1[s2] = xd[s2] - ol[4 * s2 + kq];
            a1[Q - 1] = last_real ? xd[Q - 1] - ol[kl] : last_const;
#pragma unroll
            for (int s2 = 0; s2 < Q; s2++) a2[s2] = a1[s2] * a1[s2];
            // the wave's next frame tile, under this tile's matrix and vector work
            have = GHMM_EMS_PF && u < u1 && f0 + 16 < F;
            if (have) loadd(tf + 1);
        } else if (full && x16) {
            // the wave's 16 x D frame tile is contiguous in HBM and starts on a 16-byte
            // boundary (f0 is a multiple of 16)
            if (!have) loadx(tf);
            const int q128 = 128 / D, r128 = 128 - q128 * D; // element index step 128 in (row, column) form
            int e0 = 2 * l, r = e0 / D, d = e0 - r * D;
#pragma unroll
            for (int u2 = 0; u2 < NP; u2++) {
                if (u2 < NP - 1 || l + 64 * u2 < np) { // (only the last piece can be surplus: D >= DP - 4)
                    const int d1 = d + 1 < D ? d + 1 : 0, r1 = d + 1 < D ? r : r + 1;
                    xw[r * XS + d] = xn[u2][0] - ol[d];
                    xw[r1 * XS + d1] = xn[u2][1] - ol[d1];
                }
                r += q128;
                d += r128;
                if (d >= D) {
                    d -= D;
                    r++;
                }
            }
            // the wave's next frame tile (if it has one and it is a whole tile) is fetched now,
            // under this tile's matrix and vector work
            have = GHMM_EMS_PF && u < u1 && f0 + 32 <= F;
            if (have) loadx(tf + 1);
        } else {
            // ragged last tile of the corpus, or frames on an odd 8-byte boundary: 8 bytes per
            // lane, addresses clamped into the corpus (never predicated)
            constexpr int NL = (16 * (DP - 1) + 63) / 64;
            const long long base = f0 * D + l;
            double xs[NL];
            have = false;
#pragma unroll
            for (int u2 = 0; u2 < NL; u2++) {
                long long q = base + 64 * u2;
                q = q < FD ? q : FD - 1;
                xs[u2] = X[q];
            }
            const int q64 = 64 / D, r64 = 64 - q64 * D;
            int r = l / D, d = l - r * D;
#pragma unroll
            for (int u2 = 0; u2 < NL; u2++) {
                if (u2 < NL - 1 || l + 64 * u2 < 16 * D) xw[r * XS + d] = xs[u2] - ol[d];
                r += q64;
                d += r64;
                if (d >= D) {
                    d -= D;
                    r++;
                }
            }
        }
        // the x operands of this frame tile (x' and x'^2 of frame l & 15, k = 4s + (l >> 4))
        // stay in registers for all of its Gaussian tiles (168 VGPRs = 3 waves per SIMD; the
        // log-b variant used to re-read them from the slab per tile: 1.19 -> 1.13 ms at
        // configs[2] with them in registers)
        constexpr bool AREG = true;
        if (!DIRECT) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // LDS is in order per wave
#pragma unroll
            for (int s = 0; s < Q; s++) {
                a1[s] = xr[4 * s];
                a2[s] = a1[s] * a1[s];
            }
        }
        const long long fr = f0 + j;   // this lane's frame
        const bool fok = full || fr < F;
        if (GHMM_LAB & 64) { // (lab: frame tiles only)
            if (a1[0] + a2[AREG ? Q - 1 : 0] == 12345.678) b[fr] = a1[0];
            continue;
        }
        // a state's mixtures fill MPL adjacent slots of TPS consecutive tiles
        double zero = 0.0; // (opaque to the compiler below: orders the tiles of a multi-tile state)
        for (int ct = g0 * TPS; ct < g1 * TPS; ct += TPS) {
            double e[TPS][4];
#pragma unroll
            for (int tt = 0; tt < TPS; tt++) {
                v4d acc = {zero, zero, zero, zero};
                const double *Wt = Wl + (size_t)(ct + tt) * KS * 64 + l;
                if ((shm >> (ct + tt)) & 1u) {
                    // this tile's own offset: x'' = x' - (offset - oglob), from the slab
                    const double *dq = dl + (ct + tt) * DP + kq;
#pragma unroll
                    for (int s = 0; s < Q; s++) {
                        const double x1 = (AREG ? a1[AREG ? s : 0] : xr[4 * s]) - dq[4 * s];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wt[s * 64], x1, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wt[(Q + s) * 64], x1 * x1, acc, 0, 0, 0);
                    }
                } else if (AREG && (GHMM_LAB & 4)) {
                    const double w0 = Wt[0];
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[r] = -fabs(w0 * a1[AREG ? r : 0]);
                } else if (AREG) {
#pragma unroll
                    for (int s = 0; s < Q; s++) {
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wt[s * 64], a1[AREG ? s : 0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wt[(Q + s) * 64], a2[AREG ? s : 0], acc, 0, 0, 0);
                    }
                } else {
#pragma unroll 5
                    for (int s = 0; s < Q; s++) {
                        const double x1 = xr[4 * s];
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wt[s * 64], x1, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wt[(Q + s) * 64], x1 * x1, acc, 0, 0, 0);
                    }
                }
                // acc[r]: exponent of slot 4 kq + r at frame l & 15
                if ((bdm >> (ct + tt)) & 1u) {
                    // A slot whose Gaussian is ill-conditioned even around the tile's offset (a
                    // second variance-floored component in the tile): its expanded exponent is
                    // off by up to ~1e-7 absolutely.  Where that exponent is far below the
                    // underflow of exp() the density is 0 either way (TF:1834); only for the
                    // few frames next to such a component is it taken again in the reference's
                    // direct form (x - mu) inv (x - mu), TF:1829-1832, from the raw frame.
                    const unsigned bm = (unsigned)tfl[ct + tt] >> 16;
                    bool need = false;
#pragma unroll
                    for (int r = 0; r < 4; r++) need |= ((bm >> (4 * kq + r)) & 1u) && acc[r] > -760.0;
                    if (__any(need)) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (((bm >> (4 * kq + r)) & 1u) && acc[r] > -760.0 && fok) {
                                const int gq = gml[(ct + tt) * 16 + 4 * kq + r];
                                const double *mu = mean + (size_t)gq * D, *iv = inv_var + (size_t)gq * D;
                                const double *xf = X + fr * D;
                                double aux = 0.0;
                                for (int d = 0; d < D; d++) {
                                    const double dif = xf[d] - mu[d];
                                    aux += dif * iv[d] * dif;
                                }
                                acc[r] = -0.5 * aux;
                            }
                    }
                }
                if (OUT == 2) {
                    // wkl holds log(wk) here: keep the exponents, exponentiate after the max
                    const v4d wk4 = *(const v4d *)(wkl + (ct + tt) * 16 + 4 * kq);
#pragma unroll
                    for (int r = 0; r < 4; r++) e[tt][r] = acc[r] + wk4[r];
                } else {
                    if (GHMM_LAB & 1) {
#pragma unroll
                        for (int r = 0; r < 4; r++) e[tt][r] = acc[r];
                    } else {
                        exp_emis4(acc, etab, e[tt]);
                    }
                    const v4d wk4 = *(const v4d *)(wkl + (ct + tt) * 16 + 4 * kq);
#pragma unroll
                    for (int r = 0; r < 4; r++) e[tt][r] *= wk4[r];
                }
                // a state that spans several tiles: one tile at a time — left alone, the compiler runs
                // the MFMA chains of all of them first and then their exponentials side by side,
                // which takes more registers than there are.  The empty asm ties the next tile's
                // accumulator start to this tile's finished densities.
                if (TPS > 1)
                    asm volatile("" : "+v"(e[tt][0]), "+v"(e[tt][1]), "+v"(e[tt][2]), "+v"(e[tt][3]), "+v"(zero));
            }
            // first state of this lane and whether the lane stores it (one lane per state)
            const int st0 = ((c0 + ct) * 16 + 4 * kq) >> LOGMP;
            const bool holder = MPL <= 4 || (kq & (MPL / 4 - 1)) == 0;
            if (OUT == 2) {
                // log b_i = m + log(sum exp(e - m)), m = the state's largest exponent
                double m[NS], sm[NS], lb[NS];
#pragma unroll
                for (int v = 0; v < NS; v++) {
                    double t = e[0][v * GS];
#pragma unroll
                    for (int tt = 0; tt < TPS; tt++)
#pragma unroll
                        for (int r = 0; r < GS; r++) t = fmax(t, e[tt][v * GS + r]);
                    m[v] = kq_max<MPL>(t);
                    sm[v] = 0.0;
                }
#pragma unroll
                for (int tt = 0; tt < TPS; tt++) {
                    v4d dlt;
                    double ex[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) dlt[r] = e[tt][r] - m[r / GS];
                    exp_emis4(dlt, etab, ex);
#pragma unroll
                    for (int r = 0; r < 4; r++) sm[r / GS] += ex[r];
                }
#pragma unroll
                for (int v = 0; v < NS; v++) {
                    const double st = kq_sum<MPL>(sm[v]);
                    lb[v] = m[v] < -1.0e299 ? -INFINITY : m[v] + log(st);
                }
                if (fok && holder) {
#pragma unroll
                    for (int v = 0; v < NS; v++)
                        if (st0 + v < N) b[fr * N + st0 + v] = lb[v];
                }
                continue;
            }
            // state sums in the lane, then across the kq lanes of the state
            double sm[NS];
#pragma unroll
            for (int v = 0; v < NS; v++) {
                double t = 0.0;
#pragma unroll
                for (int tt = 0; tt < TPS; tt++) {
                    double ts;
                    if (GS == 4) ts = (e[tt][0] + e[tt][1]) + (e[tt][2] + e[tt][3]);
                    else if (GS == 2) ts = e[tt][2 * v] + e[tt][2 * v + 1];
                    else ts = e[tt][v];
                    t = tt == 0 ? ts : t + ts; // (no 0 + x: hipcc keeps that addition, it canonicalises)
                }
                sm[v] = (GHMM_LAB & 8) ? e[0][v] : kq_sum<MPL>(t);
            }
            if (fok && holder && !((GHMM_LAB & 2) && sm[0] != 12345.678)) {
#pragma unroll
                for (int v = 0; v < NS; v++)
                    if (st0 + v < N) b[fr * N + st0 + v] = sm[v];
            }
            if (OUT == 1) {
                // gauss[i][j] /= b_i, 0 when b_i == 0 (TF:1773-1778)
                double rr[NS], pre[NS];
                bool subn = false;
#pragma unroll
                for (int v = 0; v < NS; v++) {
                    pre[v] = 1.0;
                    rr[v] = (GHMM_LAB & 8) ? sm[v] : recip_post(sm[v], pre[v]);
                    subn |= pre[v] != 1.0;
                }
                const bool fix = __any(subn); // some state's sum is subnormal (rare, wave-uniform)
#pragma unroll
                for (int tt = 0; tt < TPS; tt++) {
                    double pv[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) pv[r] = e[tt][r] * rr[r / GS];
                    if (fix) {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int r = 0; r < 4; r++) pv[r] = (e[tt][r] * pre[r / GS]) * rr[r / GS];
                    }
                    if ((fum >> (ct + tt)) & 1u) {
                        // the lane's four posteriors are 32 contiguous, 16-byte aligned bytes
                        if (fok && !((GHMM_LAB & 2) && pv[0] + pv[1] + pv[2] + pv[3] != 12345.678)) {
                            if (GHMM_LAB & 16) {
                                // (lab: the same bytes as whole 128-byte lines, data misplaced)
                                double *pp = post + ((f0 + (l >> 3)) * G + gml[(ct + tt) * 16] + 2 * (l & 7));
                                *(v2d *)pp = (v2d){pv[0], pv[1]};
                                *(v2d *)(pp + 8 * G) = (v2d){pv[2], pv[3]};
                            } else if (GHMM_LAB & 512) {
                                // (lab: whole lines per instruction with the lanes of a line 8 apart: frames
                                // j & 7 by the first store, 8 + (j & 7) by the second; data misplaced)
                                double *pp = post + ((f0 + (j & 7)) * G + gml[(ct + tt) * 16] + 4 * kq + 2 * (j >> 3));
                                *(v2d *)pp = (v2d){pv[0], pv[1]};
                                *(v2d *)(pp + 8 * G) = (v2d){pv[2], pv[3]};
                            } else if (GHMM_LAB & 128) {
                                // (lab: every posterior store lands in one 4 MB window: L2 hits, no HBM writes)
                                double *pp = post + ((fr * G + gml[(ct + tt) * 16 + 4 * kq]) & 0x7FFFFll);
                                *(v2d *)pp = (v2d){pv[0], pv[1]};
                                *(v2d *)(pp + 2) = (v2d){pv[2], pv[3]};
                            } else {
                            double *pp = post + (fr * G + gml[(ct + tt) * 16 + 4 * kq]);
                            if (ntp) {
                                // (the empty asm keeps the two arms apart: merged into one, hipcc
                                // drops the hint and emits plain stores)
                                asm volatile("" ::: "memory");
                                __builtin_nontemporal_store((v2d){pv[0], pv[1]}, (v2d *)pp);
                                __builtin_nontemporal_store((v2d){pv[2], pv[3]}, (v2d *)(pp + 2));
                            } else {
                                *(v2d *)pp = (v2d){pv[0], pv[1]};
                                *(v2d *)(pp + 2) = (v2d){pv[2], pv[3]};
                            }
                            }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int gm = gml[(ct + tt) * 16 + 4 * kq + r];
                            if (fok && gm >= 0) post[fr * G + gm] = pv[r];
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

// e / d for small non-negative e (see divmod_small)
__device__ inline int div_small(int e, int d, float rd)
{
    int q, r;
    divmod_small(e, d, rd, q, r);
    return q;
}

constexpr int MSM_WAVES = 4;

// smask[stage] bit i = gamma_t(i) != 0 for some frame t of the 16-frame stage (N <= 32).  One
// wave per 4 stages = 64 frames = 64 N consecutive doubles, read coalesced (lane l takes elements
// l, l + 64, ...: N independent loads in flight; one frame per lane with a ballot per state ran N
// dependent round trips per wave: 76 us at 20 states x 300 000 frames, now a pass over gamma).
__global__ void __launch_bounds__(WAVE)
k_stage_masks(int N, long long F, const double *__restrict__ gamma, unsigned *__restrict__ smask)
{
    const int l = threadIdx.x;
    const long long f0 = (long long)blockIdx.x * WAVE;          // first frame of the wave
    const long long nel = (F - f0 < WAVE ? F - f0 : WAVE) * N;  // elements of its frames
    const double *g = gamma + f0 * N;
    // element e = l + 64 k: frame e / N, state e % N, advanced without divisions
    const int dq = WAVE / N, dr = WAVE % N;
    int fr = l / N, st = l % N;
    unsigned m[4] = {0u, 0u, 0u, 0u};
    for (long long e0 = 0; e0 < nel; e0 += 8 * WAVE) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const long long e = e0 + k * WAVE + l;
            v[k] = g[e < nel ? e : nel - 1]; // clamped, never predicated
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool in = e0 + k * WAVE + l < nel;
            const unsigned bit = (in && v[k] != 0.0) ? (1u << st) : 0u;
            const int q = fr >> 4;
            m[0] |= q == 0 ? bit : 0u;
            m[1] |= q == 1 ? bit : 0u;
            m[2] |= q == 2 ? bit : 0u;
            m[3] |= q == 3 ? bit : 0u;
            fr += dq;
            st += dr;
            if (st >= N) {
                st -= N;
                fr++;
            }
        }
    }
    // OR over the wave's lanes (as in k_mixstats_mfma's state_mask), the totals in lane 63
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int v = (int)m[q];
        v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
        v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
        v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true); // row_half_mirror
        v |= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true); // row_mirror
        v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast15 into rows 1 and 3
        v |= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast31 into rows 2 and 3
        m[q] = (unsigned)__builtin_amdgcn_readlane(v, 63);
    }
    const long long sg = (long long)blockIdx.x * 4 + l;
    if (l < 4 && sg * 16 < F) smask[sg] = l == 0 ? m[0] : (l == 1 ? m[1] : (l == 2 ? m[2] : m[3]));
}
constexpr int MSM_PD = 4; // k-steps of operands in flight per wave

// STAGED = true: frames go HBM -> registers -> LDS in 16-frame stages with fully
// coalesced 16-byte-per-lane loads (one stage ahead), MFMA operands come from LDS.
// Needs G and every chunk's gmin, GW even (16-byte alignment of the posterior rows' pieces; mixture
// padding and odd mixture counts are fine otherwise) and N <= 8 NGL (<= 32: the
// state masks are 32-bit words).
// STAGED = false: operands straight from HBM, 8 bytes per lane (any shape).
// MASKED (staged only): the states that carry weight in every 16-frame stage come from
// smask[stage] (k_stage_masks) instead of from gamma travelling a stage ahead, and the wave
// walks ONLY the stages in which a state of this chunk's tiles is occupied.  For models whose
// Gaussians take several chunks (64 mixtures: 8 chunks of little more than one state each) most
// stages of a chunk are empty, and an empty stage then costs a bit test instead of a trip to
// memory.
// NTL: the posteriors are read with non-temporal loads (compile time: behind a run-time flag hipcc
// merges the two arms into one plain load)
// NGL: 16-byte gamma pieces per lane and stage (2: N <= 16, 4: N <= 32)
template <int CT, int NE, bool STAGED, bool MASKED = false, bool NTL = false, int NGL = 2>
__global__ void __launch_bounds__(MSM_WAVES *WAVE, 1)
k_mixstats_mfma(int N, int M, int Mp, int D, int DP, int NT, long long F, int gmin, int GW,
                const double *__restrict__ X, const double *__restrict__ gamma,
                const double *__restrict__ post, const int *__restrict__ gmap,
                const double *__restrict__ oglob, double *__restrict__ part,
                const unsigned *__restrict__ smask = nullptr)
{
    extern __shared__ double lds[]; // fold: [CT*NE][4 rows][3 writers][64]; STAGED: per-wave frame stages
    const unsigned long long t_entry = (GHMM_LAB & 8192) ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long t_loop0 = 0, t_loop1 = 0;
    const int tid = threadIdx.x, l = tid & 63, j = l & 15, kq = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6); // wave index, in a scalar register
    const int G = N * M, ES = NE * 16;
    // staged: one launch per chunk; gmin = the chunk's first REAL Gaussian, which sits in the first
    // slot of the chunk's first tile (the host checks): its padded position gives the tile
    const int c0 = STAGED ? ((gmin / M) * Mp + gmin % M) / 16 : blockIdx.y * CT;
    int gmA[CT], stA[CT];
    const float rMp = 1.0f / (float)Mp, rM = 1.0f / (float)M, rN = 1.0f / (float)N;
#pragma unroll
    for (int c = 0; c < CT; c++) {
        const int gp = (c0 + c) * 16 + j;
        gmA[c] = (c0 + c < NT) ? gmap[gp] : -1;
        stA[c] = div_small(gp, Mp, rMp); // (float reciprocal + correction: the integer divisions of
                                         // this set-up were a third of the kernel's first 13 000 cycles)
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
    // floor(steps wi / nwaves) through one f64 division each: exact while steps wi < 2^52 (the two
    // 64-bit integer divisions were ~500 scalar instructions of a lone wave's set-up)
    const double rnw = (double)nwaves;
    const long long s0 = (long long)floor((double)(steps * wi) / rnw), s1 = (long long)floor((double)(steps * (wi + 1)) / rnw);

    int gmC[CT];
    double mk[CT]; // 1 for a real Gaussian, 0 for padding: masks by multiplication, so
                   // that hipcc cannot sink the loads under a branch
#pragma unroll
    for (int c = 0; c < CT; c++) {
        gmC[c] = gmA[c] >= 0 ? gmA[c] : (STAGED ? gmin : 0); // (padding: the chunk's first column, times 0)
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
        // pieces l + 64u.  Bounds: 8*D <= 64*NE, 8*GW <= 64*2*CT, 8*N <= 64*NGL.  Surplus lanes
        // repeat the last piece, load and store alike (same value to the same place): no
        // predicates, no branches in the stage writer.
        constexpr int NXL = NE, NPL = 2 * CT;
        const int nxp = 8 * D, npp = 8 * GW, ngp = 8 * N, ppr = GW / 2;
        const float rD = 1.0f / (float)D, rppr = 1.0f / (float)ppr;
        // gamma is EXACTLY 0 for most (frame, state) pairs (alpha^ beta^ underflows away from the
        // path: 87 % at BASELINE configs[1]), and a weight of 0 adds exactly nothing.  Per stage the
        // states with a non-zero gamma anywhere in its 16 frames form a bit mask (gamma travels one
        // stage ahead of the posteriors for that); posterior pieces of the other states are not
        // fetched (their loads are pointed at the stage's first piece: no branch, no HBM traffic,
        // whatever they bring is multiplied by 0) and a Gaussian tile none of whose states is in
        // the mask skips its LDS reads and MFMAs.
        v2d rx[NXL], rp[NPL], rg[NGL], rgn[NGL];
        unsigned offp[NPL]; // posterior piece -> element offset inside a stage (rows strided by G)
        unsigned pcp[NPL], pcx[NXL], pcg[NGL]; // clamped piece indices
        unsigned pst[NPL];                     // state of the piece's first Gaussian
        unsigned offpB[NPL], pbit[NPL], pcxB[NXL]; // byte offsets inside a stage, state bit, frame piece bytes
        unsigned gb0[NGL], gb1[NGL];           // state bits of the two gammas of a piece
        // the first two stages' gammas are asked for before anything else: the tables below are
        // built under their trip to memory (the kernel's first 13 000 cycles were 12 % of it)
#pragma unroll
        for (int u = 0; u < NGL; u++) {
            const int pc = l + 64 * u;
            pcg[u] = (unsigned)(pc < ngp ? pc : ngp - 1);
        }
        if (!MASKED && s0 < s1) {
            const v2d *g0 = (const v2d *)(gamma + uniform64(s0 * 16 * N));
            const v2d *g1 = (const v2d *)(gamma + uniform64((s0 + 1 < s1 ? s0 + 1 : s0) * 16 * N));
#pragma unroll
            for (int u = 0; u < NGL; u++) {
                rg[u] = g0[pcg[u]];
                rgn[u] = g1[pcg[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < NPL; u++) {
            int pc = l + 64 * u;
            pc = pc < npp ? pc : npp - 1;
            pcp[u] = (unsigned)pc;
            int row, c2;
            divmod_small(pc, ppr, rppr, row, c2);
            offp[u] = (unsigned)(row * G + gmin + 2 * c2);
            pst[u] = (unsigned)div_small(gmin + 2 * c2, M, rM);
            offpB[u] = (offp[u] - (unsigned)gmin) * 8u;
            // (an odd mixture count puts the piece's two Gaussians in two states now and then)
            pbit[u] = (1u << pst[u]) | (1u << (unsigned)div_small(gmin + 2 * c2 + 1, M, rM));
        }
#pragma unroll
        for (int u = 0; u < NGL; u++) {
            const int pc = l + 64 * u;
            pcg[u] = (unsigned)(pc < ngp ? pc : ngp - 1);
            int q0, r0, q1, r1;
            divmod_small((int)(2 * pcg[u]), N, rN, q0, r0);
            divmod_small((int)(2 * pcg[u] + 1), N, rN, q1, r1);
            gb0[u] = 1u << r0;
            gb1[u] = 1u << r1;
        }
        // states covered by Gaussian tile c of this chunk
        unsigned tm[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) {
            const int g_lo = (c0 + c) * 16, g_hi = g_lo + 15;
            const int s_lo = div_small(g_lo, Mp, rMp), s_hq = div_small(g_hi, Mp, rMp), s_hi = s_hq < N ? s_hq : N - 1;
            tm[c] = (c0 + c < NT && s_lo < N) ? ((2u << s_hi) - 1u) & ~((1u << s_lo) - 1u) : 0u;
        }
        // bit i = some frame of the stage whose gammas sit in `q` has gamma(state i) != 0
        auto state_mask = [&](const v2d (&q)[NGL]) {
            unsigned bits = 0;
#pragma unroll
            for (int u = 0; u < NGL; u++)
                bits |= (q[u][0] != 0.0 ? gb0[u] : 0u) | (q[u][1] != 0.0 ? gb1[u] : 0u);
            // OR over the wave's lanes: four levels inside the rows of 16, then row 0 -> 1, 2 -> 3 and
            // rows 0-1 -> 2-3 (row_bcast15 / row_bcast31), the total in lane 63 (a loop of one ballot
            // per state took 110 instructions of a lone wave per stage: 13 % of the kernel)
            int v = (int)bits;
            v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
            v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
            v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true); // row_half_mirror
            v |= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true); // row_mirror
            v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast15 into rows 1 and 3
            v |= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast31 into rows 2 and 3
            return (unsigned)__builtin_amdgcn_readlane(v, 63);
        };
        // where the two doubles of X piece u land: (slab offset << 8) | coefficient index
        unsigned xa[NXL], xb[NXL];
#pragma unroll
        for (int u = 0; u < NXL; u++) {
            int pc = l + 64 * u;
            pc = pc < nxp ? pc : nxp - 1;
            pcx[u] = (unsigned)pc;
            pcxB[u] = (unsigned)pc * 16u;
            const int e0 = 2 * pc;
            int r0, d0;
            divmod_small(e0, D, rD, r0, d0);
            const int r1 = d0 + 1 < D ? r0 : r0 + 1, d1 = d0 + 1 < D ? d0 + 1 : 0;
            xa[u] = ((unsigned)(r0 * XS + d0) << 8) | (unsigned)d0;
            xb[u] = ((unsigned)(r1 * XS + d1) << 8) | (unsigned)d1;
        }
        auto fetch_gamma = [&](long long stg, v2d (&q)[NGL]) {
            const v2d *gsrc = (const v2d *)(gamma + uniform64(stg * 16 * N));
#pragma unroll
            for (int u = 0; u < NGL; u++) q[u] = gsrc[pcg[u]];
        };
        // states of this chunk's tiles: a stage in which none of them carries weight is not
        // staged at all (with 64 mixtures a chunk of 5 tiles is little more than one state, and a
        // state is occupied in a small part of the frames)
        unsigned cmask = 0;
#pragma unroll
        for (int c = 0; c < CT; c++) cmask |= tm[c];
        // frames and the posteriors of the states in `smask` (the others: the stage's first piece;
        // no state of the chunk in the mask: the frames' first piece as well)
        auto fetch = [&](long long stg, unsigned smask) {
            const long long f = stg * 16;
            // wave-uniform bases in scalar registers + 32-bit per-lane BYTE offsets: the loads take the
            // saddr + voffset form, no 64-bit address arithmetic per load
            const char *xsrc = (const char *)(X + uniform64(f * D));
            const char *psrc = (const char *)(post + uniform64(f * G + gmin));
            const unsigned xm = (smask & cmask) != 0u ? ~0u : 0u; // wave-uniform
#pragma unroll
            for (int u = 0; u < NXL; u++) rx[u] = *(const v2d *)(xsrc + (size_t)(pcxB[u] & xm));
#pragma unroll
            for (int u = 0; u < NPL; u++) {
                const unsigned o = (smask & pbit[u]) != 0u ? offpB[u] : 0u;
                // NTL: the posteriors pass by once (the host picks this variant together with the
                // emission kernel's non-temporal stores: -1 % per iteration at 10x8)
                if (NTL) rp[u] = __builtin_nontemporal_load((const v2d *)(psrc + (size_t)o));
                else rp[u] = *(const v2d *)(psrc + (size_t)o);
            }
        };
        // MASKED: next stage >= from (< s1) with a state of this chunk in its mask, and that mask;
        // the masks of 64 consecutive stages sit one per lane, their "occupied" bits in a scalar pair
        long long blk_base = -64;
        unsigned long long blk_bits = 0;
        unsigned blk_val = 0;
        auto next_active = [&](long long from, unsigned &mval) -> long long {
            while (from < s1) {
                if (from >= blk_base + 64) {
                    blk_base = from;
                    const long long sl = from + l < s1 ? from + l : s1 - 1;
                    blk_val = smask[sl];
                    blk_bits = __ballot((blk_val & cmask) != 0u && from + l < s1);
                }
                const int rel = (int)(from - blk_base);
                const unsigned long long rem = blk_bits >> rel;
                if (rem) {
                    const int at = rel + __builtin_ctzll(rem);
                    mval = (unsigned)__builtin_amdgcn_readlane((int)blk_val, at);
                    return blk_base + at;
                }
                from = blk_base + 64;
            }
            mval = 0;
            return s1;
        };
        const unsigned long long t_tab = (GHMM_LAB & 8192) ? __builtin_amdgcn_s_memtime() : 0ull;
        unsigned smask_cur = 0;
        long long cur = s0; // MASKED: the stage being staged
        if (MASKED) {
            cur = next_active(s0, smask_cur);
            if (cur < s1) {
                fetch_gamma(cur, rg);
                fetch(cur, smask_cur);
            }
        } else if (s0 < s1) {
            smask_cur = state_mask(rg); // (rg, rgn: asked for at the top)
            fetch(s0, smask_cur);
        }
        const unsigned long long t_fet = (GHMM_LAB & 8192) ? __builtin_amdgcn_s_memtime() : 0ull;
        if (l < DP) ol[l] = og_l;
        if (l + WAVE < DP) ol[l + WAVE] = og_h;
        {   // constant columns only (the stage writer rewrites the others): the 1 at D, zeros in
            // D+1 .. DP-1 and DP+D .. XS-1 — two columns at D = 39, one LDS write per lane for all
            // 16 rows (filling the whole 16 x XS stage took 3 000 cycles of the kernel's set-up)
            const int nlo = DP - D, ncc = XS - 2 * D;
            for (int k = l; k < 16 * ncc; k += WAVE) {
                const int r = k / ncc, cc = k - r * ncc;
                const int col = cc < nlo ? D + cc : DP + D + (cc - nlo);
                fx[r * XS + col] = col == D ? 1.0 : 0.0;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the offsets of this lane's frame pieces, fixed for the whole kernel: in registers (the
        // stage writer of a lone wave pays ~10 cycles per instruction, and ten LDS reads per stage
        // for values that never change were 7 % of it)
        double oa[NXL], ob[NXL];
#pragma unroll
        for (int u = 0; u < NXL; u++) {
            oa[u] = ol[xa[u] & 255u];
            ob[u] = ol[xb[u] & 255u];
        }
        // (lab 8192: cycles of a wave spent waiting for its stage's loads / in the stage writer /
        // issuing the next fetch / in the k-steps, printed by a few waves)
        unsigned long long tw0 = 0, tw1 = 0, tw2 = 0, tw3 = 0, tst = 0;
        const unsigned long long tbeg = (GHMM_LAB & 8192) ? __builtin_amdgcn_s_memtime() : 0ull;
        for (long long stg = MASKED ? cur : s0; stg < s1; stg = MASKED ? cur : stg + 1) {
            unsigned long long ta = 0, tb = 0, tc = 0, td = 0;
            if (GHMM_LAB & 8192) {
                ta = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                tb = __builtin_amdgcn_s_memtime();
            }
            if (MASKED || (smask_cur & cmask) != 0u) {
#pragma unroll
                for (int u = 0; u < NXL; u++) {
                    const double x0 = rx[u][0] - oa[u], x1 = rx[u][1] - ob[u];
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
            }
            if (GHMM_LAB & 8192) tc = __builtin_amdgcn_s_memtime();
            unsigned smask_next;
            if (MASKED) {
                // the next occupied stage, in flight under this stage's MFMAs
                cur = next_active(stg + 1, smask_next);
                const long long nx = cur < s1 ? cur : stg;
                fetch_gamma(nx, rg);
                fetch(nx, cur < s1 ? smask_next : smask_cur);
            } else {
                // the next stage, in flight under this stage's MFMAs: its gammas arrived a stage ago
                smask_next = state_mask(rgn);
                fetch(stg + 1 < s1 ? stg + 1 : stg, smask_next);
#pragma unroll
                for (int u = 0; u < NGL; u++) rg[u] = rgn[u];
                fetch_gamma(stg + 2 < s1 ? stg + 2 : (stg + 1 < s1 ? stg + 1 : stg), rgn);
            }
            if (GHMM_LAB & 8192) {
                td = __builtin_amdgcn_s_memtime();
                tw0 += tb - ta; tw1 += tc - tb; tw2 += td - tc; tst++;
            }
            // Gaussian tiles with a state of this stage's mask
            unsigned tact = 0;
#pragma unroll
            for (int c = 0; c < CT; c++) tact |= ((smask_cur & tm[c]) != 0u ? 1u : 0u) << c;
            smask_cur = smask_next;
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
                for (int c = 0; c < CT; c++)
                    if ((tact >> c) & 1u) {
                        g[c] = gsl[c][og];
                        p[c] = psl[c][op];
                    }
#pragma unroll
                for (int n = 0; n < NE; n++) f[n] = fxl[of + 16 * n];
            };
            auto run = [&](const double (&g)[CT], const double (&p)[CT], const double (&f)[NE]) {
#pragma unroll
                for (int c = 0; c < CT; c++)
                    if ((tact >> c) & 1u) {
                        const double wv = g[c] * p[c] * mk[c];
#pragma unroll
                        for (int n = 0; n < NE; n++)
                            acc[c][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(wv, f[n], acc[c][n], 0, 0, 0);
                    }
            };
            if (tact == 0) continue; // nobody occupies any state of this chunk in these 16 frames
            if (GHMM_MIX_TILEMAJOR) {
                // Tile-major (round 3): the stage's 4 x NE frame operands are read once, then every
                // ACTIVE Gaussian tile runs its four k-steps in one straight block — one branch per
                // tile and stage.  The k-step-major order below tested every tile twice per k-step
                // (operand reads, MFMAs): 40 branches per stage of a lone wave, and with the 1.35
                // tiles that are active on average (gamma is 0 for 87 % of the pairs) more than half
                // of the k-step phase was not matrix-pipe time (in-kernel stamps, GHMM_LAB 8192).
                double fq[4][NE];
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int n = 0; n < NE; n++) fq[q][n] = fxl[4 * q * XS + 16 * n];
#pragma unroll
                for (int c = 0; c < CT; c++)
                    if ((tact >> c) & 1u) {
                        double wq[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) wq[q] = gsl[c][4 * q * N] * psl[c][4 * q * GW] * mk[c];
#pragma unroll
                        for (int q = 0; q < 4; q++)
#pragma unroll
                            for (int n = 0; n < NE; n++)
                                acc[c][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(wq[q], fq[q][n], acc[c][n], 0, 0, 0);
                    }
                continue;
            }
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
        if (GHMM_LAB & 8192) {
            const unsigned long long tend = __builtin_amdgcn_s_memtime();
            tw3 = (tend - tbeg) - tw0 - tw1 - tw2;
            if (l == 0 && (blockIdx.x % 64) == 3 && gmin == 0 && !(GHMM_LAB & 16384))
                printf("mixstats block %d wave %d: %llu stages, cycles total %llu: load wait %llu, stage writer %llu, fetch issue %llu, k-steps+rest %llu\n",
                       (int)blockIdx.x, w, tst, tend - tbeg, tw0, tw1, tw2, tw3);
            t_loop0 = tbeg; t_loop1 = tend;
            if (l == 0 && (blockIdx.x % 64) == 3 && gmin == 0 && w == 0)
                printf("mixstats block %d: entry -> tables done %llu, -> first fetch issued %llu, -> loop %llu\n", (int)blockIdx.x, t_tab - t_entry, t_fet - t_tab, tbeg - t_fet);
        }
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
    // Fold the block's waves in wave order and write the block's partial, all four waves at work,
    // in ONE round: wave W keeps register row W of every tile and parks its other three rows in LDS
    // (3/4 of 4 waves' tiles: exactly 150 KB at 25 tiles); behind the barrier it adds up rows W of
    // wave 0 + wave 1 + wave 2 + wave 3, in that order (its own from registers), and stores them.
    // (Two rounds of half the tiles with every row parked: 104 + 104 LDS accesses and 4 barriers per
    // wave instead of 75 + 75 and 2.)
    __syncthreads(); // the stages alias the fold buffer
    auto fold = [&](auto wc) {
        constexpr int W = decltype(wc)::value;
#pragma unroll
        for (int c = 0; c < CT; c++)
#pragma unroll
            for (int n = 0; n < NE; n++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (r != W) // slot (tile, row r, rank of the writer among the three waves != r)
                        lds[((size_t)((c * NE + n) * 4 + r) * 3 + (W < r ? W : W - 1)) * 64 + l] = acc[c][n][r];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CT; c++)
#pragma unroll
            for (int n = 0; n < NE; n++) {
                double v = 0.0;
#pragma unroll
                for (int ww = 0; ww < MSM_WAVES; ww++) {
                    const double x = ww == W ? acc[c][n][W]
                                             : lds[((size_t)((c * NE + n) * 4 + W) * 3 + (ww < W ? ww : ww - 1)) * 64 + l];
                    v = ww == 0 ? x : v + x;
                }
                const int gp = (c0 + c) * 16 + kq + 4 * W;
                if (c0 + c < NT) part[((size_t)blockIdx.x * NT * 16 + gp) * ES + 16 * n + j] = v;
            }
    };
    // (w is wave-uniform: one of four copies of the code runs, each with constant register indices)
    if (w == 0) fold(std::integral_constant<int, 0>{});
    else if (w == 1) fold(std::integral_constant<int, 1>{});
    else if (w == 2) fold(std::integral_constant<int, 2>{});
    else fold(std::integral_constant<int, 3>{});
    if (GHMM_LAB & 8192) {
        const unsigned long long t_exit = __builtin_amdgcn_s_memtime();
        if (l == 0 && (blockIdx.x % 64) == 3 && gmin == 0)
            printf("mixstats block %d wave %d: cycles before the stage loop %llu, behind it (fold + partial) %llu\n",
                   (int)blockIdx.x, w, t_loop0 - t_entry, t_exit - t_loop1);
    }
}

} // namespace ghmm
