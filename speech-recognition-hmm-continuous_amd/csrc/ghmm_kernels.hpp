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
           double *__restrict__ b, double *__restrict__ post, double *__restrict__ lognorm,
           const int *__restrict__ only_if, int epoch)
{
    extern __shared__ double lds[];
    // only_if[0] == epoch: the model's current preparation found an ill-conditioned Gaussian
    if (only_if && only_if[0] != epoch) return; // the matrix-core kernel has done the job
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

// several feature streams: b_i(t) <- b_i(t) * b^p_i(t), the running product of TF:1406-1409
__global__ void __launch_bounds__(256)
k_mul_streams(long long n, double *__restrict__ b, const double *__restrict__ bp)
{
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long long)gridDim.x * 256)
        b[k] *= bp[k];
}

// ------------------------------------------------------------ group helpers
// A "group" is L consecutive lanes (L = 16, 32 or 64) that own one utterance, lane i =
// state i.  Cross-lane traffic stays inside the group.
// For L = 16 a group is one DPP row; wider groups span rows.  Shifts and the butterfly sum
// are register moves (v_mov_b32_dpp, v_permlane*_swap), not LDS-crossbar shuffles
// (ds_bpermute) — they sit on the serial critical path of every time step.
template <int CTRL> __device__ inline double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true); // out-of-row source -> 0
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// value of lane i ^ 4: row_shl:4 into the lanes of banks 0 and 2, row_shr:4 into banks 1 and 3
__device__ inline double dpp_xor4_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    int tl = __builtin_amdgcn_update_dpp(0, lo, 0x104, 0xf, 0x5, false);
    int th = __builtin_amdgcn_update_dpp(0, hi, 0x104, 0xf, 0x5, false);
    tl = __builtin_amdgcn_update_dpp(tl, lo, 0x114, 0xf, 0xa, false);
    th = __builtin_amdgcn_update_dpp(th, hi, 0x114, 0xf, 0xa, false);
    return __hiloint2double(th, tl);
}
constexpr int DPP_ROW_SHL1 = 0x101, DPP_ROW_SHR1 = 0x111;
constexpr int DPP_ROW_ROR1 = 0x121, DPP_ROW_ROR2 = 0x122, DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;

// wave_shr:1 / wave_shl:1 (the GFX9 whole-wave DPP shifts): lane i-1 / i+1 across the rows, 0 at the
// wave's ends
constexpr int DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138;
// sum of v over the row pairs {0,1}, {2,3} (v_permlane16_swap with both operands the same value
// hands every lane its own row's and the partner row's value) / over the two halves of the wave
__device__ inline double rows_pair_sum(double v)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}
__device__ inline double halves_pair_sum(double v)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}

template <int L> __device__ inline double group_sum(double v)
{
    // rotate-and-add inside a 16-lane row: every lane of the row ends with the same bits (each
    // level adds the same unordered pair on both partners); wider groups then add the rows
    // pairwise with the lane-swap instructions, again the same unordered pair in every lane.
    // All of it is register moves (v_mov_b32_dpp / v_permlane*_swap), not LDS-crossbar
    // shuffles: the sum sits on the serial critical path of every time step.
    v += dpp_f64<DPP_ROW_ROR8>(v);
    v += dpp_f64<DPP_ROW_ROR4>(v);
    v += dpp_f64<DPP_ROW_ROR2>(v);
    v += dpp_f64<DPP_ROW_ROR1>(v);
    if (L >= 32) v = rows_pair_sum(v);
    if (L >= 64) v = halves_pair_sum(v);
    return v;
}
// value of lane i-1 (0 for the first lane of the group) / lane i+1 (0 for the last)
template <int L> __device__ inline double group_up1(double v)
{
    if (L == 16) return dpp_f64<DPP_ROW_SHR1>(v);
    const double r = dpp_f64<DPP_WAVE_SHR1>(v);
    return (L == WAVE || (threadIdx.x % L)) ? r : 0.0;
}
template <int L> __device__ inline double group_down1(double v)
{
    if (L == 16) return dpp_f64<DPP_ROW_SHL1>(v);
    const double r = dpp_f64<DPP_WAVE_SHL1>(v);
    return (L == WAVE || (threadIdx.x % L) != L - 1) ? r : 0.0;
}
// sum over the Mp (power of two <= 16) adjacent lanes that hold one state's mixtures
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141,
              DPP_ROW_MIRROR = 0x140;
__device__ inline double segment_sum(double v, int Mp)
{
    if (Mp >= 2) v += dpp_f64<DPP_QUAD_XOR1>(v);
    if (Mp >= 4) v += dpp_f64<DPP_QUAD_XOR2>(v);
    if (Mp >= 8) v += dpp_f64<DPP_ROW_HALF_MIRROR>(v);
    if (Mp >= 16) v += dpp_f64<DPP_ROW_MIRROR>(v);
    return v;
}

// exp(x) for the emission epilogue: x = k ln2 + r, |r| <= ln2/2, degree-13 Taylor
// polynomial (truncation 4e-18), v_ldexp_f64 for 2^k (gradual underflow to 0 like libm).
// Branch-free; NaN stays NaN; within ~1 ulp of glibc's exp on (-750, 1].
__device__ inline double exp_emis(double x)
{
    x = x < -750.0 ? -750.0 : x;
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;           // 1/13!
    p = fma(p, r, 2.08767569878680989792e-09);   // 1/12!
    p = fma(p, r, 2.50521083854417187751e-08);
    p = fma(p, r, 2.75573192239858906526e-07);
    p = fma(p, r, 2.75573192239858906526e-06);
    p = fma(p, r, 2.48015873015873015873e-05);
    p = fma(p, r, 1.98412698412698412698e-04);
    p = fma(p, r, 1.38888888888888888889e-03);
    p = fma(p, r, 8.33333333333333333333e-03);
    p = fma(p, r, 4.16666666666666666667e-02);
    p = fma(p, r, 1.66666666666666666667e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// 1/s: hardware reciprocal + two Newton steps (<= 1 ulp from the IEEE quotient the
// reference computes, TF:1414/1436) for normal s; the exact division otherwise so that
// s = 0 still yields the reference's inf -> NaN cascade
__device__ inline double recip_scale(double s)
{
    if (s >= 1.0e-290 && s <= 1.0e290) {
        double r = __builtin_amdgcn_rcp(s);
        r = fma(r, fma(-s, r, 1.0), r);
        r = fma(r, fma(-s, r, 1.0), r);
        return r;
    }
    return 1.0 / s;
}

// Stores of lanes without an output go to a sink, loads of lanes without an input read zeros:
// SINK_WAVES regions of [WAVE doubles written | WAVE zeros], one per wave (modulo), so that
// hundreds of waves do not hammer the same four cache lines of one L2 channel at every step.
// Partial sums of the utterance statistics, one slot per utterance or per (utterance, chunk),
// slot-minor so that k_reduce_all reads them coalesced: xi[(i, o)][slot], den[i][slot]
__device__ inline size_t pxi_at(int slot, int i, int o, int S) { return ((size_t)i * (MAX_DELTA + 1) + o) * S + slot; }
__device__ inline size_t pden_at(int slot, int i, int S) { return (size_t)i * S + slot; }

constexpr int SINK_WAVES = 4096;
__device__ inline double *wave_sink(double *sink)
{
    const unsigned wid = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    return sink + (size_t)(wid % SINK_WAVES) * 2 * WAVE + (threadIdx.x & (WAVE - 1));
}

constexpr int PF = 8;   // frames of b / alpha prefetched ahead of the serial recursion
#ifndef GHMM_PFF
#define GHMM_PFF 16 // (measurement builds override it: profiles/tools/lab.sh)
#endif
constexpr int PFF = GHMM_PFF; // the same for the forward pass (one operand stream, more room)
// frames of padding the host keeps in front of and behind the emission densities b[F][N]: the
// scans' operand cursors run up to 2 PFF frames past either end of an utterance
constexpr int B_PAD_FRAMES = 2 * PFF;

// ------------------------------------------------------------------ forward
// calc_alpha (TF:1380-1443 = RF:739-799) with pi = one-hot at state 0 (TF:232-234)
// and calc_probability (TF:1536-1553).  One group per utterance; the time loop is
// serial, utterances are the parallel axis (SURVEY.md §5 "long-context").
//   alpha_t(i) = (sum_j alpha^_{t-1}(j) a_ji) b_i(t);  c_t = 1/sum_i alpha_t(i);  alpha^ = alpha c_t
//   log P = -sum_t log c_t + log alpha^_{T-1}(N-1)   [+ sum_t lognorm_t in robust mode]
// The loop body is branch-free: lanes that own no output store into a sink, the
// reciprocal is a select between Newton's result and the raw v_rcp_f64 (0 / inf / tiny
// sums, where the reference's 1/0 -> NaN cascade must survive), and A's band structure
// is a template parameter chosen once per wave.
__device__ inline double recip_select(double s)
{
    const double r0 = __builtin_amdgcn_rcp(s); // inf for 0, 0 for inf, like 1.0/s
    double r = fma(r0, fma(-s, r0, 1.0), r0);
    r = fma(r, fma(-s, r, 1.0), r);
    // refined wherever 1/s is a normal number (the raw reciprocal is good to ~1e-8 only: a first
    // frame with b ~ 1e-294 used to carry that into alpha^_0 and c_0)
    return (s >= 0x1p-1020 && s <= 0x1p1020) ? r : r0;
}

// one Newton step on the hardware reciprocal: 2e-15 relative (measured), enough for a
// per-step scale that is renormalised every frame; the raw reciprocal keeps 0 -> inf
__device__ inline double recip_fast(double s)
{
    // s = 0, inf or subnormal gives NaN here where the reference's 1.0/s gives inf or 0;
    // either way every later alpha^ of the utterance, and its log P, is NaN (0 * inf)
    const double r0 = __builtin_amdgcn_rcp(s);
    return fma(r0, fma(-s, r0, 1.0), r0);
}

// -sum_t log c_t without a log per frame: the product of the mantissas and the sum of the exponents
// (v_frexp_mant/exp + one multiply and one add per value), one log at the end.  0, inf and NaN
// behave like log()'s sum: the mantissa of 0 / inf / NaN is itself.
struct log_product {
    double m = 1.0;
    long long e = 0;
    int n = 0;
    __device__ inline void mul(double c)
    {
        m *= __builtin_amdgcn_frexp_mant(c);
        e += __builtin_amdgcn_frexp_exp(c);
        if (++n == 512) { // keep the product of mantissas (each in [0.5, 1)) away from underflow
            e += __builtin_amdgcn_frexp_exp(m);
            m = __builtin_amdgcn_frexp_mant(m);
            n = 0;
        }
    }
    __device__ inline double log_value() const
    {
        return fma((double)e, 6.93147180369123816490e-01, fma((double)e, 1.90821492927058770002e-10, log(m)));
    }
};

template <int L, bool BANDED> struct fwd_state {
    double a, a_self, a_prev, a_next;
    double acol[BANDED ? 1 : L];
    int N;
    // Banded A (what the reference's trainer always produces): a 2-term update.  (Taking
    // the 16-lane sum off this chain via sum_j alpha^_{t-1}(j) g_j(t) was measured slower:
    // a lone wave per SIMD is bound by instruction count, ~13 cycles each, not by the chain.)
    // SINV: lane 1 keeps 1/c_t = sum_i alpha_t(i) for the reference-order backward pass
    // STORE_C = false (scoring only): c_t is not written; its logarithm's pieces are kept in `acc`
    template <bool SINV, bool STORE_C>
    __device__ inline void step_banded(double bt, double *__restrict__ pa, double *__restrict__ pcs, int i,
                                       log_product &acc)
    {
        const double v = fma(a, a_self, group_up1<L>(a) * a_prev) * bt;
        const double s = group_sum<L>(v);
        const double c = recip_fast(s);
        a = v * c;
        *pa = a;
        if (STORE_C) *pcs = (SINV && i != 0) ? s : c;
        else acc.mul(c);
    }
    template <bool SINV, bool STORE_C>
    __device__ inline void step_dense(double bt, double *__restrict__ pa, double *__restrict__ pcs, int i,
                                      log_product &acc)
    {
        double aux = 0.0;
#pragma unroll
        for (int j = 0; j < L; j++)
            if (j < N) aux += __shfl(a, j, L) * acol[BANDED ? 0 : j];
        const double v = aux * bt;
        const double s = group_sum<L>(v);
        const double c = recip_select(s);
        a = v * c;
        *pa = a;
        if (STORE_C) *pcs = (SINV && i != 0) ? s : c;
        else acc.mul(c);
    }
};

template <int L, bool BANDED, bool SINV = true, bool STORE_C = true>
__device__ __forceinline__ double forward_run(int N, int T, int i, bool act, const double *__restrict__ A,
                                     const double *__restrict__ bu, double *__restrict__ au,
                                     double *__restrict__ su, double *__restrict__ si,
                                     double *__restrict__ sink, int bstride, log_product *pacc = nullptr)
{
    log_product acc;
    fwd_state<L, BANDED> st;
    st.N = N;
    st.a_self = act ? A[i * N + i] : 0.0;
    st.a_prev = (act && i > 0) ? A[(i - 1) * N + i] : 0.0;
    st.a_next = (act && i + 1 < N) ? A[i * N + i + 1] : 0.0;
    if (!BANDED) {
#pragma unroll
        for (int j = 0; j < L; j++) st.acol[BANDED ? 0 : j] = (act && j < N) ? A[j * N + i] : 0.0;
    }
    // per-lane store cursors: alpha slot (stride N) or the sink (stride 0); lane 0 writes
    // c_t, lane 1 writes 1/c_t = sum_i alpha_t(i), the rest write the sink
    double *pa = (act && au) ? au + i : sink; // au == nullptr: scoring only, alpha^ not kept
    const int da = (act && au) ? N : 0;
    double *pcs = (i == 0) ? su : ((SINV && i == 1) ? si : sink);
    const int dc = (i == 0 || (SINV && i == 1)) ? 1 : 0;
    // b of frame 0, 1, 2, ... through a cursor that is never clamped and never predicated (a
    // load under a branch makes hipcc wait vmcnt(0) at every step; an index clamp costs six
    // vector instructions per load): the prefetch runs up to 2 PFF frames past the utterance,
    // into the next utterance's rows or the padding the host keeps behind b (B_PAD_FRAMES);
    // what it reads there is never used.  Idle lanes read the zero half of the sink buffer,
    // which nothing writes, with stride 0 (no `act ?` on the load, it would predicate it).
    const double *pl = act ? bu + i : sink + WAVE;
    const ptrdiff_t db = act ? bstride : 0;
    auto bnext = [&]() {
        const double v = *pl;
        pl += db;
        return v;
    };

    // t = 0
    {
        const double a0 = ((i == 0) ? 1.0 : 0.0) * bnext();
        const double s = group_sum<L>(a0);
        const double c = recip_select(s);
        st.a = a0 * c;
        *pa = st.a;
        if (STORE_C) *pcs = (SINV && i != 0) ? s : c;
        else acc.mul(c);
        pa += da; pcs += dc;
    }
    double bq[PFF];
    int t = 1;
#pragma unroll
    for (int k = 0; k < PFF; k++) bq[k] = bnext();
    for (; t + PFF <= T; t += PFF) {
        double bn[PFF];
#pragma unroll
        for (int k = 0; k < PFF; k++) bn[k] = bnext();
        if (BANDED) {
#pragma unroll
            for (int k = 0; k < PFF; k++) {
                st.template step_banded<SINV, STORE_C>(bq[k], pa, pcs, i, acc);
                pa += da; pcs += dc;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PFF; k++) {
                st.template step_dense<SINV, STORE_C>(bq[k], pa, pcs, i, acc);
                pa += da; pcs += dc;
            }
        }
#pragma unroll
        for (int k = 0; k < PFF; k++) bq[k] = bn[k];
    }
#pragma unroll
    for (int k = 0; k < PFF - 1; k++)
        if (t + k < T) {
            if (BANDED) {
                st.template step_banded<SINV, STORE_C>(bq[k], pa, pcs, i, acc);
            } else {
                st.template step_dense<SINV, STORE_C>(bq[k], pa, pcs, i, acc);
            }
            pa += da; pcs += dc;
        }
    if (!STORE_C && pacc) *pacc = acc;
    return st.a;
}

// calc_alpha + calc_probability for utterance u on the 16/64 lanes of one group
// SCORE: log P only (ghmm_score) — alpha^ and c_t are not written, the logarithm of their product
// is taken from its pieces in registers
template <int L, bool SINV = true, bool SCORE = false>
__device__ inline void forward_utt(int N, int u, int i, const double *__restrict__ A,
                                   const double *__restrict__ b, const long long *__restrict__ off,
                                   double *__restrict__ alpha, double *__restrict__ scale,
                                   double *__restrict__ sinv, const double *__restrict__ lognorm,
                                   double *__restrict__ loglik, double *__restrict__ sink,
                                   bool want_logp = true)
{
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    if (T <= 0) {
        if (i == 0 && want_logp) loglik[u] = 0.0;
        return;
    }
    const bool act = i < N;
    bool offband = false;
    for (int j = 0; j < N; j++)
        offband |= act && (A[j * N + i] != 0.0 && j != i && j != i - 1);
    const bool banded = !__any(offband);
    double *su = scale + f0, *si = sinv + f0;
    double *snk = wave_sink(sink);
    double a;
    if (SCORE) {
        log_product pc;
        if (banded)
            a = forward_run<L, true, false, false>(N, T, i, act, A, b + f0 * N, nullptr, su, si, snk, N, &pc);
        else
            a = forward_run<L, false, false, false>(N, T, i, act, A, b + f0 * N, nullptr, su, si, snk, N, &pc);
        double lp = 0.0;
        if (lognorm) {
            for (int t = i; t < T; t += L) lp += lognorm[f0 + t];
            lp = group_sum<L>(lp);
        }
        const double last = __shfl(a, N - 1, L);
        if (i == 0) loglik[u] = (lp - pc.log_value()) + log(last);
        return;
    }
    if (banded)
        a = forward_run<L, true, SINV>(N, T, i, act, A, b + f0 * N, alpha + f0 * N, su, si, snk, N);
    else
        a = forward_run<L, false, SINV>(N, T, i, act, A, b + f0 * N, alpha + f0 * N, su, si, snk, N);
    if (!want_logp) return; // k_combine takes the logs, spread over all of its waves
    // log P: the T logs are spread over the group's lanes instead of a serial loop
    __threadfence_block();
    double lp = 0.0;
    log_product pc;
    for (int t = i; t < T; t += L) {
        pc.mul(su[t]);
        if (lognorm) lp += lognorm[f0 + t];
    }
    lp = group_sum<L>(lp - pc.log_value());
    double last = __shfl(a, N - 1, L);
    if (i == 0) loglik[u] = lp + log(last);
}

template <int L>
__global__ void __launch_bounds__(WAVE)
k_forward(int N, int U, const double *__restrict__ A, const double *__restrict__ b,
          const long long *__restrict__ off, double *__restrict__ alpha,
          double *__restrict__ scale, double *__restrict__ sinv,
          const double *__restrict__ lognorm, double *__restrict__ loglik,
          double *__restrict__ sink, const int *__restrict__ order)
{
    const int slot = blockIdx.x * (WAVE / L) + threadIdx.x / L;
    const int i = threadIdx.x % L;
    if (slot >= U) return;
    const int u = order[slot]; // neighbours in length order share a wave
    forward_utt<L>(N, u, i, A, b, off, alpha, scale, sinv, lognorm, loglik, sink);
}

// The recogniser's vocabulary loop (RF:326-374) in one launch: blockIdx.y = word model.
// The emission densities of ALL models were computed by one launch over the concatenated
// Gaussians (b[F][NS], model k owns columns bo_k .. bo_k + N_k - 1); every (model,
// utterance) pair runs calc_alpha + calc_probability here.  alpha^ is not kept.
struct fwd_model {
    const double *A;
    int N, bo;
};

// ghmm_score_batch: word model k's Gaussians (ng of them from g0) copied into the concatenated model
struct gather_src {
    const double *c, *mean, *inv_var, *det;
    int g0, ng;
};
__global__ void __launch_bounds__(256)
k_gather_models(int D, const gather_src *__restrict__ src, double *__restrict__ c, double *__restrict__ mean,
                double *__restrict__ inv_var, double *__restrict__ det)
{
    const gather_src s = src[blockIdx.x];
    for (int k = threadIdx.x; k < s.ng; k += 256) {
        c[s.g0 + k] = s.c[k];
        det[s.g0 + k] = s.det[k];
    }
    const size_t n = (size_t)s.ng * D, o = (size_t)s.g0 * D;
    for (size_t k = threadIdx.x; k < n; k += 256) {
        mean[o + k] = s.mean[k];
        inv_var[o + k] = s.inv_var[k];
    }
}

template <int L>
__global__ void __launch_bounds__(WAVE)
k_forward_multi(int U, int NS, long long F, const fwd_model *__restrict__ tab,
                const double *__restrict__ b, const long long *__restrict__ off,
                double *__restrict__ scale, double *__restrict__ sinv,
                double *__restrict__ loglik, double *__restrict__ sink,
                const int *__restrict__ order)
{
    const int slot = blockIdx.x * (WAVE / L) + threadIdx.x / L;
    const int i = threadIdx.x % L;
    const int k = blockIdx.y;
    if (slot >= U) return;
    const int u = order[slot];
    const fwd_model mk = tab[k];
    const int N = mk.N;
    const double *A = mk.A;
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    if (T <= 0) {
        if (i == 0) loglik[(size_t)k * U + u] = 0.0;
        return;
    }
    const bool act = i < N;
    bool offband = false;
    for (int j = 0; j < N; j++)
        offband |= act && (A[j * N + i] != 0.0 && j != i && j != i - 1);
    const bool banded = !__any(offband);
    // (c_t is not written: 100 words x 150 000 frames of it were 240 MB per pass; the logarithm of
    // the product comes from its pieces in registers)
    double *snk = wave_sink(sink);
    const double *bu = b + f0 * NS + mk.bo;
    double a;
    log_product pc;
    if (banded)
        a = forward_run<L, true, false, false>(N, T, i, act, A, bu, nullptr, snk, snk, snk, NS, &pc);
    else
        a = forward_run<L, false, false, false>(N, T, i, act, A, bu, nullptr, snk, snk, snk, NS, &pc);
    double last = __shfl(a, N - 1, L);
    if (i == 0) loglik[(size_t)k * U + u] = log(last) - pc.log_value();
}

// ----------------------------------------------------------------- backward
// calc_beta (TF:1463-1516) fused with the per-utterance sums of
// calc_transition_probab (TF:1577-1620) and calc_den_mix_coef (TF:1642-1664):
//   beta^_{T-1}(i) = [i == N-1] c_{T-1};  beta^_t(i) = c_t sum_j beta^_{t+1}(j) a_ij b_j(t+1)
//   gamma_t(i) = alpha^_t(i) beta^_t(i) / c_t                                  -> gamma[F][N]
//   xi[u][i][o] = sum_{t<T-1} alpha^_t(i) a_{i,i+o} b_{i+o}(t+1) beta^_{t+1}(i+o),  o = 0..delta
//   dena[u][i] = sum_{t<T-1} gamma_t(i);   denc[u][i] = sum_{t<T} gamma_t(i)
// Per-utterance partial sums are written out and added in utterance order by
// k_reduce_utt (bitwise reproducible; no atomics).  gamma uses 1/c_t = sum_i alpha_t(i)
// kept by the forward pass instead of a division.
template <int L, bool BANDED> struct bwd_state {
    double be, a_self, a_next, dena, denc;
    bool wild = false; // a beta^ that is inf or NaN (see backward_run's return value)
    double arow[BANDED ? 1 : L];
    double aband[MAX_DELTA + 1], xi[MAX_DELTA + 1];
    int N, delta;
    // one step t (descending): bnext = b_i(t+1), al = alpha^_t(i), c = c_t, sv = 1/c_t
    __device__ inline void step(double bnext, double al, double c, double sv,
                                double *__restrict__ pbe, double *__restrict__ pg, int i)
    {
        const double w = be * bnext; // beta^_{t+1}(i) b_i(t+1) on lane i
        const double wd = group_down1<L>(w);
        double aux;
        if (BANDED) {
            aux = a_self * w + a_next * wd;
        } else {
            aux = 0.0;
#pragma unroll
            for (int j = 0; j < L; j++)
                if (j < N) aux += arow[j] * __shfl(w, j, L);
        }
        // xi band sums without the constant a_{i,i+o}: it multiplies the finished sum
        xi[0] = fma(al, w, xi[0]);
        xi[1] = fma(al, wd, xi[1]);
#pragma unroll
        for (int o = 2; o <= MAX_DELTA; o++)
            if (o <= delta) {
                double wj = __shfl_down(w, o, L);
                xi[o] += (i + o < N) ? al * wj : 0.0;
            }
        be = aux * c;
        if (BANDED) wild |= !(fabs(be) < INFINITY);
        const double g = (al * sv) * be;
        *pbe = be;
        *pg = g;
        dena += g;
    }
};

// Returns (BANDED only) whether a beta^ left the finite numbers: next to an overflowed beta^ the
// reference's dense inner loop (TF:1493-1510) multiplies inf by every a_ij = 0 of the row and
// turns the whole row NaN, which the band-only update does not reproduce; the caller then takes
// the utterance again with the dense form (k_backward_fix).
template <int L, bool BANDED>
__device__ __forceinline__ bool backward_run(int N, int T, int delta, int i, bool act, int u,
                                    const double *__restrict__ A, const double *__restrict__ bu,
                                    const double *__restrict__ au, const double *__restrict__ su,
                                    const double *__restrict__ si, double *__restrict__ beu,
                                    double *__restrict__ gu, double *__restrict__ part_xi,
                                    double *__restrict__ part_dena, double *__restrict__ part_denc,
                                    double *__restrict__ sink, int S)
{
    bwd_state<L, BANDED> st;
    st.N = N;
    st.delta = delta;
    st.dena = st.denc = 0.0;
    st.a_self = act ? A[i * N + i] : 0.0;
    st.a_next = (act && i + 1 < N) ? A[i * N + i + 1] : 0.0;
    if (!BANDED) {
#pragma unroll
        for (int j = 0; j < L; j++) st.arow[BANDED ? 0 : j] = (act && j < N) ? A[i * N + j] : 0.0;
    }
#pragma unroll
    for (int o = 0; o <= MAX_DELTA; o++) {
        st.aband[o] = (act && i + o < N && o <= delta) ? A[i * N + i + o] : 0.0;
        st.xi[o] = 0.0;
    }
    const int dn = act ? N : 0;
    // frame-indexed readers, clamped into the utterance (never predicated); lanes without
    // a state read / write the sink with stride 0
    const double *pa0 = act ? au + i : sink + WAVE, *pb0 = act ? bu + i : sink + WAVE;
    auto clampf = [&](int f) { return (size_t)(f < 0 ? 0 : f); };
    double *pbe = act ? beu + (size_t)(T - 1) * N + i : sink;
    double *pg = act ? gu + (size_t)(T - 1) * N + i : sink;
    {
        const double cT = su[T - 1];
        st.be = (i == N - 1) ? 1.0 * cT : 0.0;
        // (si == nullptr, k_backward_fix: the paired launch keeps no 1/c_t; it is divided out here)
        const double g = (act ? pa0[(size_t)(T - 1) * dn] : 0.0) * st.be * (si ? si[T - 1] : 1.0 / cT);
        *pbe = st.be;
        *pg = g;
        st.denc = g; // gamma_{T-1}: in den_c (t < T, TF:1660) but not in den_a (t < T-1, TF:1618)
    }
    // queues for step t (descending from T-2): b[t+1], alpha[t], c[t], 1/c[t]
    double qb[PF], qa[PF], qc[PF], qs[PF];
    int t = T - 2;
#pragma unroll
    for (int k = 0; k < PF; k++) {
        const size_t f = clampf(t - k);
        qb[k] = pb0[(f + 1 < (size_t)T ? f + 1 : (size_t)T - 1) * dn];
        qa[k] = pa0[f * dn];
        qc[k] = su[f];
        qs[k] = si ? si[f] : 1.0 / su[f];
    }
    pbe -= dn; pg -= dn;
    for (; t - PF + 1 >= 0; t -= PF) {
        double nb[PF], na[PF], nc[PF], ns[PF];
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const size_t f = clampf(t - PF - k);
            nb[k] = pb0[(f + 1 < (size_t)T ? f + 1 : (size_t)T - 1) * dn];
            na[k] = pa0[f * dn];
            nc[k] = su[f];
            ns[k] = si ? si[f] : 1.0 / su[f];
        }
#pragma unroll
        for (int k = 0; k < PF; k++) {
            st.step(qb[k], qa[k], qc[k], qs[k], pbe, pg, i);
            pbe -= dn; pg -= dn;
        }
#pragma unroll
        for (int k = 0; k < PF; k++) {
            qb[k] = nb[k]; qa[k] = na[k]; qc[k] = nc[k]; qs[k] = ns[k];
        }
    }
#pragma unroll
    for (int k = 0; k < PF - 1; k++)
        if (t - k >= 0) {
            st.step(qb[k], qa[k], qc[k], qs[k], pbe, pg, i);
            pbe -= dn; pg -= dn;
        }
    if (act) {
#pragma unroll
        for (int o = 0; o <= MAX_DELTA; o++) // compile-time indices: the arrays stay in registers
            if (o <= delta) part_xi[pxi_at(u, i, o, S)] = st.aband[o] * st.xi[o];
        part_dena[pden_at(u, i, S)] = st.dena;
        part_denc[pden_at(u, i, S)] = st.dena + st.denc;
    }
    return st.wild;
}

template <int L>
__global__ void __launch_bounds__(WAVE)
k_backward(int N, int U, int delta, const double *__restrict__ A, const double *__restrict__ b,
           const long long *__restrict__ off, const double *__restrict__ alpha,
           const double *__restrict__ scale, const double *__restrict__ sinv,
           double *__restrict__ beta, double *__restrict__ gamma, double *__restrict__ part_xi,
           double *__restrict__ part_dena, double *__restrict__ part_denc,
           double *__restrict__ sink, const int *__restrict__ order, int *__restrict__ fix_cnt,
           int *__restrict__ fix_list)
{
    const int slot = blockIdx.x * (WAVE / L) + threadIdx.x / L;
    const int i = threadIdx.x % L;
    if (slot >= U) return;
    const int u = order[slot];
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    const bool act = i < N;
    if (T <= 0) {
        if (act) {
            for (int o = 0; o <= delta; o++) part_xi[pxi_at(u, i, o, U)] = 0.0;
            part_dena[pden_at(u, i, U)] = 0.0;
            part_denc[pden_at(u, i, U)] = 0.0;
        }
        return;
    }
    bool offband = false;
    for (int j = 0; j < N; j++)
        offband |= act && (A[i * N + j] != 0.0 && j != i && j != i + 1);
    const bool banded = !__any(offband);
    double *snk = wave_sink(sink);
    if (banded) {
        const bool wild = backward_run<L, true>(N, T, delta, i, act, u, A, b + f0 * N, alpha + f0 * N, scale + f0,
                                                sinv + f0, beta + f0 * N, gamma + f0 * N, part_xi, part_dena,
                                                part_denc, snk, U);
        // (the ballot covers the wave's other utterances too: they are only taken twice)
        if (__any(wild) && i == 0) fix_list[atomicAdd(fix_cnt, 1)] = u;
    } else
        backward_run<L, false>(N, T, delta, i, act, u, A, b + f0 * N, alpha + f0 * N, scale + f0,
                               sinv + f0, beta + f0 * N, gamma + f0 * N, part_xi, part_dena,
                               part_denc, snk, U);
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
constexpr int MS_MAXG = 4096; // Gaussians the compact list of k_mixstats can hold
constexpr int MS_FS = 32; // frames staged per pass (fewer when a wide tile would not fit LDS)

__global__ void __launch_bounds__(MS_THREADS)
k_mixstats(int N, int M, int D, long long F, long long frames_per_block, int FS,
           const double *__restrict__ X, const double *__restrict__ gamma,
           const double *__restrict__ post, const double *__restrict__ mean,
           double *__restrict__ part_mu, double *__restrict__ part_var,
           const int *__restrict__ only_if, int epoch, const int *__restrict__ scls = nullptr,
           int Mp = 0)
{
    extern __shared__ double lds[];
    __shared__ int fl[MS_MAXG]; // scls != nullptr: the Gaussians of class 2 (ghmm_mfma.hpp), in index order
    __shared__ int nfl;
    if (only_if && only_if[0] != epoch) return; // matrix-core tier: nothing is ill-conditioned
    const int G = N * M, D1 = D + 1;
    const int tid = threadIdx.x;
    // Behind the matrix-core kernel only the ill-conditioned Gaussians' sums are used
    // (k_reduce_all): the element space shrinks to those Gaussians (a handful of variance-
    // floored components in a typical training run), every block builds the same list.
    int GE = G;
    if (scls) {
        if (tid == 0) {
            int n = 0;
            for (int g = 0; g < G; g++)
                if (scls[(g / M) * Mp + g % M] == 2) fl[n++] = g;
            nfl = n;
        }
        __syncthreads();
        GE = nfl;
    }
    const long long E = (long long)G * D1, EE = (long long)GE * D1; // real / compact element space
    const long long e0 = (long long)blockIdx.y * (MS_THREADS * MS_EPT);
    if (e0 >= EE) return;
    const long long e1 = (e0 + MS_THREADS * MS_EPT < EE) ? e0 + MS_THREADS * MS_EPT : EE;
    const int g0 = (int)(e0 / D1);
    const int g1 = (int)((e1 - 1) / D1); // inclusive (compact indices behind the matrix-core kernel)
    const int GW = g1 - g0 + 1;
    double *xs = lds;                // [FS][D1]
    double *ws = lds + FS * D1;      // [FS][GW]
    auto real = [&](int gc) { return scls ? fl[gc] : gc; };
    const int kmax = (int)((e1 - e0 + MS_THREADS - 1) / MS_THREADS); // element slots in use per thread

    int gx[MS_EPT], dx[MS_EPT];
    long long eo[MS_EPT]; // where the element's sums go (real layout)
    double mu[MS_EPT], acc_mu[MS_EPT], acc_var[MS_EPT];
#pragma unroll
    for (int k = 0; k < MS_EPT; k++) {
        long long e = e0 + tid + (long long)k * MS_THREADS;
        bool ok = e < e1;
        int gc = ok ? (int)(e / D1) : g0;
        int d = ok ? (int)(e - (long long)gc * D1) : D;
        const int g = real(gc);
        gx[k] = gc - g0;
        dx[k] = d;
        eo[k] = ok ? (long long)g * D1 + d : -1;
        mu[k] = (ok && d < D) ? mean[(size_t)g * D + d] : 0.0;
        acc_mu[k] = 0.0;
        acc_var[k] = 0.0;
    }
    const long long fb0 = (long long)blockIdx.x * frames_per_block;
    const long long fb1 = (fb0 + frames_per_block < F) ? fb0 + frames_per_block : F;
    // frames of the staged batch that carry weight for this block's Gaussians: gamma is exactly 0
    // for most (frame, state) pairs, and a weight of exactly 0 adds exactly nothing — such frames
    // are neither fetched nor visited
    __shared__ unsigned long long nzmask;
    for (long long fs = fb0; fs < fb1; fs += FS) {
        const int nf = (int)((fb1 - fs) < FS ? (fb1 - fs) : FS);
        __syncthreads();
        if (tid == 0) nzmask = 0ull;
        for (int k = tid; k < nf * GW; k += MS_THREADS) {
            int r = k / GW, gl = k - r * GW;
            int g = real(g0 + gl);
            ws[k] = gamma[(fs + r) * N + g / M] * post[(fs + r) * G + g];
        }
        __syncthreads();
        if (tid < nf) {
            bool nz = false;
            for (int gl = 0; gl < GW; gl++) nz |= ws[tid * GW + gl] != 0.0;
            if (nz) atomicOr(&nzmask, 1ull << tid);
        }
        __syncthreads();
        const unsigned long long nzm = nzmask; // FS <= 32 frames per batch
        if (nzm == 0ull) continue;
        for (int k = tid; k < nf * D1; k += MS_THREADS) {
            int r = k / D1, d = k - r * D1;
            if ((nzm >> r) & 1ull) xs[k] = d < D ? X[(fs + r) * D + d] : 1.0;
        }
        __syncthreads();
        for (int r = 0; r < nf; r++) {
            if (!((nzm >> r) & 1ull)) continue;
#pragma unroll
            for (int k = 0; k < MS_EPT; k++)
                if (k < kmax) { // block-uniform: a short element range leaves the later slots empty
                    double w = ws[r * GW + gx[k]];
                    double x = xs[r * D1 + dx[k]];
                    acc_mu[k] += w * x;
                    double dif = x - mu[k];
                    acc_var[k] += w * (dif * dif);
                }
        }
    }
#pragma unroll
    for (int k = 0; k < MS_EPT; k++)
        if (eo[k] >= 0) {
            part_mu[(size_t)blockIdx.x * E + eo[k]] = acc_mu[k];
            part_var[(size_t)blockIdx.x * E + eo[k]] = acc_var[k];
        }
}

// ------------------------------------------------------------------- reduce
// Ordered sums of all partials into the flat statistics vector (layout:
// include/ghmm.h) in ONE launch.  Every sum is taken in a fixed order (thread-strided
// ascending, then a fixed LDS tree), so the result does not depend on scheduling.
//   blocks [0, NG)        one per Gaussian: the frame-block partials of k_mixstats (vector
//                         ALU tier, [P1][G*(D+1)]) or of k_mixstats_mfma (matrix-core tier,
//                         [Pm][NT*16][ES] expanded sums S, converted here:
//                           num_c = S[D], num_mu_d = S_d + o_d num_c,
//                           num_var_d = S_{DP+d} - 2 mu'_d S_d + mu'_d^2 num_c, mu' = mu - o);
//                         an ill-conditioned Gaussian (cond > cond_max) takes the vector-ALU
//                         sums on the matrix-core tier too
//   blocks [NG, NG+NU)    per-utterance partials of k_backward: num_a, den_a, den_c, loglik
constexpr int RD_THREADS = 1024; // 128 feature lanes x 8 partial slices

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

struct reduce_args {
    int N, M, D, U, delta;
    // host-visible copy of (log P, utterances) + a sequence number behind them (pinned, fine-grained
    // memory: the trainer's stopping rule polls it instead of copying and waiting); nullptr: none
    long long *mbox;
    long long mbox_seq;
    int S; // slots of the utterance partials: one per utterance, or one per (utterance, chunk)
    // vector-ALU partials (vec == 0: k_mixstats was not launched; a class-2 Gaussian is then
    // recomputed here, exactly)
    int P1, vec;
    const double *part_mu, *part_var;
    // matrix-core partials (Pm == 0: tier not in use)
    int Pm, NT, DP, ES;
    const double *part_m, *condg, *oglob, *mean;
    const int *gmap;
    double cond_max;
    // utterance partials
    const double *part_xi, *part_dena, *part_denc, *loglik;
    // log P in pieces (ghmm_pair.hpp): -sum log c_t per slot and log alpha^_{T-1}(N-1) per utterance;
    // nullptr: loglik[u] is complete
    const double *lpart, *logk;
    // per-tile offsets of the expanded form for the NEXT model (ghmm_mfma.hpp): chosen here from
    // the model the statistics were taken with; nullptr: tier not in use
    double *otile;
    int *tnext;
    // statistics class per padded Gaussian (ghmm_mfma.hpp stats_class) and what the exact
    // recomputation of a class-1 Gaussian that fails its check reads
    const int *scls;
    const double *X, *gamma, *post;
    long long F;
    double kappa; // relative rounding bound of the expanded sums (accumulation length x eps)
    double *stats;
};

__global__ void __launch_bounds__(RD_THREADS) k_reduce_all(reduce_args a)
{
    __shared__ double sh[RD_THREADS];
    __shared__ double sh2[RD_THREADS];
    const int N = a.N, M = a.M, D = a.D, G = N * M, D1 = D + 1;
    const int NG = a.Pm > 0 ? a.NT * 16 : G;
    const int tid = threadIdx.x;
    double *num_a = a.stats, *den_a = num_a + (size_t)N * N, *den_c = den_a + N;
    double *num_c = den_c + N, *num_mu = num_c + G, *num_var = num_mu + (size_t)G * D;
    if ((int)blockIdx.x < NG) {
        const int gp = blockIdx.x;
        const int g = a.Pm > 0 ? a.gmap[gp] : gp;
        if (g < 0) return;
        constexpr int SL = RD_THREADS / 128;
        const int e = tid & 127, half = tid >> 7;
        const int cls = a.Pm > 0 ? a.scls[gp] : 2;
        // The reference's direct form over every frame that carries weight, by this block alone
        // (rare, ~0.1 ms).  Lane = frame for the weights (64 frames per wave step, waves
        // interleaved), then lane = coefficient for every such frame; fixed order throughout.
        auto recompute_exact = [&]() {
            const int st = g / M, w = tid >> 6, l = tid & 63, NW = RD_THREADS / 64;
            for (int d0 = 0; d0 < D; d0 += 64) { // (coefficient counts beyond 64: 64 at a time)
                const int d = d0 + l;
                const double mu = d < D ? a.mean[(size_t)g * D + d] : 0.0;
                double am = 0.0, av = 0.0, ac = 0.0;
                for (long long t0 = (long long)w * 64; t0 < a.F; t0 += (long long)NW * 64) {
                    const long long t = t0 + l;
                    const double wt = t < a.F ? a.gamma[t * N + st] * a.post[t * G + g] : 0.0;
                    unsigned long long m = __ballot(wt != 0.0);
                    while (m) {
                        const int k = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const double wk = __shfl(wt, k, 64);
                        const double x = d < D ? a.X[(t0 + k) * D + d] : 0.0;
                        const double dif = x - mu;
                        am += wk * x;
                        av += wk * (dif * dif);
                        ac += wk;
                    }
                }
                // waves in order: sh[w*64 + l] (mean sums), sh2 (variance sums); the count from lane 0
                __syncthreads();
                sh[tid] = am;
                sh2[tid] = av;
                __syncthreads();
                if (tid < 64) {
                    double sm = 0.0, sv = 0.0;
                    for (int q = 0; q < NW; q++) {
                        sm += sh[q * 64 + tid];
                        sv += sh2[q * 64 + tid];
                    }
                    if (d0 + tid < D) {
                        num_mu[(size_t)g * D + d0 + tid] = sm;
                        num_var[(size_t)g * D + d0 + tid] = sv;
                    }
                }
                __syncthreads();
                sh[tid] = ac;
                __syncthreads();
                if (tid == 0 && d0 == 0) {
                    double sc = 0.0;
                    for (int q = 0; q < NW; q++) sc += sh[q * 64];
                    num_c[g] = sc;
                }
            }
        };
        if (cls == 2 && !a.vec) {
            // k_mixstats was not launched (the host had not seen this preparation's class-2
            // Gaussian yet, ghmm_hip.hip run_accumulate): taken here, exactly
            recompute_exact();
        } else if (cls != 2) {
            // sixteen independent chains (fixed assignment of partials to chains: reproducible),
            // so that the loads of a thread overlap: 32 partials per thread are two rounds of
            // loads in flight, not eight
            constexpr int CH = 16;
            double vch[CH];
#pragma unroll
            for (int k = 0; k < CH; k++) vch[k] = 0.0;
            if (e < a.ES) {
                const size_t pst = (size_t)a.NT * 16 * a.ES;
                const double *pm = a.part_m + (size_t)gp * a.ES + e;
                int p = half;
                for (; p + (CH - 1) * SL < a.Pm; p += CH * SL) {
#pragma unroll
                    for (int k = 0; k < CH; k++) vch[k] += pm[(size_t)(p + k * SL) * pst];
                }
                for (; p < a.Pm; p += SL) vch[0] += pm[(size_t)p * pst];
            }
#pragma unroll
            for (int k = CH / 2; k > 0; k >>= 1)
#pragma unroll
                for (int q = 0; q < k; q++) vch[q] += vch[q + k];
            sh[tid] = vch[0];
            __syncthreads();
            if (tid < 128) {
                double t = sh[tid];
                for (int q = 1; q < SL; q++) t += sh[q * 128 + tid];
                sh[tid] = t;
            }
            __syncthreads();
            bool bad = false; // class 1: this coefficient's variance statistic is neither accurate
                              // nor certain to end under the floor
            if (tid < D1) {
                const double S0 = sh[D];
                if (tid == D) {
                    num_c[g] = S0;
                } else {
                    const double o = a.oglob[tid], mu = a.mean[(size_t)g * D + tid] - o;
                    const double S1 = sh[tid], S2 = sh[a.DP + tid];
                    const double V = (S2 - 2.0 * mu * S1) + mu * mu * S0;
                    num_mu[(size_t)g * D + tid] = S1 + o * S0;
                    num_var[(size_t)g * D + tid] = V;
                    if (cls == 1) {
                        // what cancelled, times the rounding bound of sums this long
                        const double E = a.kappa * (fabs(S2) + 2.0 * fabs(mu * S1) + mu * mu * fabs(S0));
                        const bool accurate = E <= 1.0e-9 * fabs(V);
                        const bool floored = V + E < FLOOR * S0 * (1.0 - 1.0e-9); // M-step: max(V / S0, 1e-5)
                        bad = !(accurate || floored);
                    }
                }
            }
            // A collapsed component whose variance statistic has come up to the floor
            if (cls == 1 && __syncthreads_or(bad ? 1 : 0)) recompute_exact();
        } else {
            const long long E = (long long)G * D1;
            for (int d0 = 0; d0 < D1; d0 += 128) {
                const int d = d0 + e;
                double vm = 0.0, vv = 0.0;
                if (d < D1) {
                    // four independent chains (fixed assignment of partials to chains: reproducible)
                    double m1 = 0.0, m2 = 0.0, m3 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
                    const double *pmu = a.part_mu + (size_t)g * D1 + d, *pva = a.part_var + (size_t)g * D1 + d;
                    int p = half;
                    for (; p + 3 * SL < a.P1; p += 4 * SL) {
                        vm += pmu[(size_t)p * E];
                        vv += pva[(size_t)p * E];
                        m1 += pmu[(size_t)(p + SL) * E];
                        v1 += pva[(size_t)(p + SL) * E];
                        m2 += pmu[(size_t)(p + 2 * SL) * E];
                        v2 += pva[(size_t)(p + 2 * SL) * E];
                        m3 += pmu[(size_t)(p + 3 * SL) * E];
                        v3 += pva[(size_t)(p + 3 * SL) * E];
                    }
                    for (; p < a.P1; p += SL) {
                        vm += pmu[(size_t)p * E];
                        vv += pva[(size_t)p * E];
                    }
                    vm = (vm + m1) + (m2 + m3);
                    vv = (vv + v1) + (v2 + v3);
                }
                sh[tid] = vm;
                sh2[tid] = vv;
                __syncthreads();
                if (tid < 128 && d < D1) {
                    double sm = sh[tid], sv = sh2[tid];
                    for (int q = 1; q < SL; q++) {
                        sm += sh[q * 128 + tid];
                        sv += sh2[q * 128 + tid];
                    }
                    if (d < D) {
                        num_mu[(size_t)g * D + d] = sm;
                        num_var[(size_t)g * D + d] = sv;
                    } else {
                        num_c[g] = sm;
                    }
                }
                __syncthreads();
            }
        }
        return;
    }
    const int q = blockIdx.x - NG, U = a.U;
    double v = 0.0;
    if (q < N * N) {
        int i = q / N, j = q % N, o = j - i;
        if (o >= 0 && o <= a.delta)
            for (int u = tid; u < a.S; u += RD_THREADS)
                v += a.part_xi[pxi_at(u, i, o, a.S)];
        v = block_sum_fixed(v, sh);
        if (tid == 0) num_a[q] = v;
    } else if (q < N * N + N) {
        int i = q - N * N;
        for (int u = tid; u < a.S; u += RD_THREADS) v += a.part_dena[pden_at(u, i, a.S)];
        v = block_sum_fixed(v, sh);
        if (tid == 0) den_a[i] = v;
    } else if (q < N * N + 2 * N) {
        int i = q - N * N - N;
        for (int u = tid; u < a.S; u += RD_THREADS) v += a.part_denc[pden_at(u, i, a.S)];
        v = block_sum_fixed(v, sh);
        if (tid == 0) den_c[i] = v;
    } else if (q == N * N + 2 * N) {
        if (a.lpart) {
            for (int u = tid; u < a.S; u += RD_THREADS) v += a.lpart[u];
            for (int u = tid; u < U; u += RD_THREADS) v += a.logk[u];
        } else {
            for (int u = tid; u < U; u += RD_THREADS) v += a.loglik[u];
        }
        v = block_sum_fixed(v, sh);
        if (tid == 0) {
            num_var[(size_t)G * D] = v;            // loglik
            num_var[(size_t)G * D + 1] = (double)U; // n_utt
            if (a.mbox) {
                __hip_atomic_store(a.mbox, __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(a.mbox + 1, __double_as_longlong((double)U), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(a.mbox + 2, a.mbox_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    } else if (a.otile) {
        // offset of the expanded form for tile t of the NEXT model: the mean of the tile's worst-
        // conditioned Gaussian if one exceeds the bound (first maximum), else the global offset
        const int t = q - (N * N + 2 * N + 1);
        if (t < a.NT) {
            int w = -1;
            double best = a.cond_max;
            for (int k = 0; k < 16; k++) {
                const double cv = a.condg[t * 16 + k];
                if (cv > best) {
                    best = cv;
                    w = k;
                }
            }
            if (tid == 0) a.tnext[t] = w >= 0 ? 1 : 0;
            if (w >= 0) {
                const int g = a.gmap[t * 16 + w];
                for (int d = tid; d < a.DP; d += RD_THREADS)
                    a.otile[(size_t)t * a.DP + d] = (d < D && g >= 0) ? a.mean[(size_t)g * D + d] : 0.0;
            }
        }
    }
}

// -------------------------------------------------------------------- mstep
// updating_transition_probab (TF:1862-1889), updating_mix_param (TF:1911-1955) with
// changing_zero_coef (TF:1338-1359), then calc_det (TF:1976) and inv_matrix
// (TF:2012) as main() chains them (TF:332-346) — including the reference's
// behaviour for a state whose den_c is 0 (its stored inverse variances go through
// det/inverse as if they were variances) — and the derived constants of k_prepare.
// One block per state.
constexpr int MS2_THREADS = 128;
__device__ inline void mstep_state(int N, int M, int D, const double *__restrict__ stats, double norm2pi,
                                   double *__restrict__ A, double *__restrict__ c, double *__restrict__ mean,
                                   double *__restrict__ inv_var, double *__restrict__ det,
                                   double *__restrict__ wk, double *__restrict__ logwk,
                                   double *__restrict__ logA, int lds_doubles, double *vs, int delta)
{
    const int G = N * M, i = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const double *num_a = stats, *den_a = num_a + (size_t)N * N, *den_c = den_a + N;
    const double *num_c = den_c + N, *num_mu = num_c + G, *num_var = num_mu + (size_t)G * D;
    for (int j = tid; j < N; j += nt) {
        double v = A[i * N + j];
        if (den_a[i] != 0.0) {
            // num_a is only ever accumulated for i <= j <= i + delta (TF:1601): outside that
            // band the reference's quotient is 0 / den_a = 0, whatever the vector holds there
            v = (j >= i && j <= i + delta) ? num_a[i * N + j] / den_a[i] : 0.0;
            A[i * N + j] = v;
        }
        logA[i * N + j] = v > 0.0 ? log(v) : -INFINITY;
    }
    __shared__ double csum;
    if (lds_doubles >= M * D + M) {
        // The state's variances and weights live in LDS from here to the end (no global
        // write-then-read between the steps).  den_c == 0: the state keeps what it had (TF:1935);
        // its cov_matrix slot then holds INVERSE variances, which calc_det / inv_matrix
        // (TF:343-346) treat like any other: reproduced.
        double *vw = vs + (size_t)M * D;
        const bool upd = den_c[i] != 0.0;
        for (int k = tid; k < M * D; k += nt) {
            const int g = i * M + k / D;
            const size_t q = (size_t)i * M * D + k;
            double v;
            if (upd) {
                mean[q] = num_mu[q] / num_c[g];
                v = num_var[q] / num_c[g];
                if (v < FLOOR) v = FLOOR;
            } else {
                v = inv_var[q];
            }
            vs[k] = v;
        }
        // changing_zero_coef (TF:1338-1359): floor the weights, renormalise by their sum taken
        // in the reference's order
        for (int m = tid; m < M; m += nt) {
            const double cv = upd ? num_c[i * M + m] / den_c[i] : c[i * M + m];
            vw[m] = cv < FLOOR ? FLOOR : cv;
        }
        __syncthreads();
        if (tid == 0) {
            double sum = 0.0;
            for (int k = 0; k < M; k++) sum += vw[k];
            csum = sum;
        }
        __syncthreads();
        // det = product of the (floored) variances in order, then the inverses (TF:343-346)
        for (int m = tid; m < M; m += nt) {
            const int g = i * M + m;
            const double cg = vw[m] / csum;
            c[g] = cg;
            const double *v = vs + (size_t)m * D;
            double d = 1.0;
            for (int k = 0; k < D; k++) d *= v[k];
            det[g] = d;
            const double den = norm2pi * sqrt(fabs(d));
            wk[g] = cg / den;
            logwk[g] = log(cg) - log(den);
        }
        for (int k = tid; k < M * D; k += nt) inv_var[(size_t)i * M * D + k] = 1.0 / vs[k];
        return;
    }
    // states too large for LDS: the same steps through global memory
    if (den_c[i] != 0.0) {
        for (int k = tid; k < M * D; k += nt) {
            const int g = i * M + k / D;
            const size_t q = (size_t)i * M * D + k;
            mean[q] = num_mu[q] / num_c[g];
            double v = num_var[q] / num_c[g];
            if (v < FLOOR) v = FLOOR;
            inv_var[q] = v;
        }
        for (int m = tid; m < M; m += nt) c[i * M + m] = num_c[i * M + m] / den_c[i];
    }
    __syncthreads();
    if (tid == 0) {
        double sum = 0.0;
        for (int k = 0; k < M; k++) {
            double v = c[i * M + k];
            if (v < FLOOR) v = FLOOR;
            c[i * M + k] = v;
            sum += v;
        }
        for (int k = 0; k < M; k++) c[i * M + k] /= sum;
    }
    __syncthreads();
    for (int m = tid; m < M; m += nt) {
        const int g = i * M + m;
        const double *v = inv_var + (size_t)g * D;
        double d = 1.0;
        for (int k = 0; k < D; k++) d *= v[k];
        det[g] = d;
        const double den = norm2pi * sqrt(fabs(d));
        wk[g] = c[g] / den;
        logwk[g] = log(c[g]) - log(den);
    }
    __syncthreads(); // every determinant is taken before any variance is inverted in place
    for (int k = tid; k < M * D; k += nt) {
        const size_t q = (size_t)i * M * D + k;
        inv_var[q] = 1.0 / inv_var[q];
    }
}

__global__ void __launch_bounds__(MS2_THREADS)
k_mstep(int N, int M, int D, const double *__restrict__ stats, double norm2pi,
        double *__restrict__ A, double *__restrict__ c, double *__restrict__ mean,
        double *__restrict__ inv_var, double *__restrict__ det, double *__restrict__ wk,
        double *__restrict__ logwk, double *__restrict__ logA, int lds_doubles, int delta)
{
    extern __shared__ double vs[];
    mstep_state(N, M, D, stats, norm2pi, A, c, mean, inv_var, det, wk, logwk, logA, lds_doubles, vs, delta);
}

// The cell bookkeeping between two k-means passes of the initial model (TF:1043-1270), on the device:
// new means = cell sums / counts, empty cells re-seeded from the cells of largest distortion, and
// (do_split) the split that opens the next round — the same operations in the same order as the
// host loop of ghmm_model_init_comm, block = state, thread = coefficient (every thread walks the
// same order of cells).  M <= INIT_MAXM.
constexpr int INIT_MAXM = 64;
__global__ void __launch_bounds__(64)
k_init_cells(int N, int M, int D, int n_cells, int do_split, int first, const double *__restrict__ stats,
             double *__restrict__ cells)
{
    __shared__ double dist[INIT_MAXM];
    __shared__ int idx[INIT_MAXM];
    const int k = blockIdx.x, tid = threadIdx.x, G = N * M;
    const double *num_c = stats + (size_t)N * N + 2 * (size_t)N, *num_mu = num_c + G, *num_var = num_mu + (size_t)G * D;
    double *ck = cells + (size_t)k * M * D;
    for (int j = tid; j < M; j += 64) {
        double d = 0.0;
        for (int l = 0; l < D; l++) d += num_var[((size_t)k * M + j) * D + l];
        dist[j] = d;
    }
    __syncthreads();
    // indices by decreasing key, adjacent-swap passes with strict '<' (TF:1289-1315)
    auto order_desc = [&](int n) {
        if (tid == 0) {
            for (int i = 0; i < n; i++) idx[i] = i;
            bool done = false;
            while (!done) {
                done = true;
                for (int i = 0; i < n - 1; i++)
                    if (dist[idx[i]] < dist[idx[i + 1]]) {
                        const int t = idx[i];
                        idx[i] = idx[i + 1];
                        idx[i + 1] = t;
                        done = false;
                    }
            }
        }
        __syncthreads();
    };
    auto split_cell = [&](int from, int to) { // TF:1138
        for (int l = tid; l < D; l += 64) {
            const double v = ck[(size_t)from * D + l];
            ck[(size_t)to * D + l] = v * 1.005;
            ck[(size_t)from * D + l] = v * 0.995;
        }
    };
    for (int j = 0; j < n_cells; j++)
        for (int l = tid; l < D; l += 64)
            ck[(size_t)j * D + l] = num_mu[((size_t)k * M + j) * D + l] / num_c[(size_t)k * M + j];
    if (!first) { // empty cells are re-seeded from the cells with the largest distortion (TF:1236-1270)
        order_desc(n_cells);
        int i = 0;
        for (int j = 0; j < n_cells; j++)
            if (num_c[(size_t)k * M + j] == 0.0) split_cell(idx[i++], j); // (a thread only ever touches its own coefficients)
    }
    if (do_split) {
        if (2 * n_cells < M) {
            for (int i = 0; i < n_cells; i++) split_cell(i, n_cells + i);
        } else {
            order_desc(n_cells);
            for (int i = 0; i < M - n_cells; i++) split_cell(idx[i], n_cells + i);
        }
    }
}


// --------------------------------------------------------------- init model
// creating_initial_model (TF:732-1317) on the device.  Its k-means passes are "hard"
// statistics: with gamma_t = one-hot(state that owns frame t under the uniform
// segmentation, TF:1005-1013) and post_t = one-hot(nearest cell of that state,
// classifying TF:1179-1215: squared Euclidean distance, strict '<', lowest cell wins
// ties), the mixture-statistics kernels give exactly what init_mix_mean / init_mix_param
// accumulate: num_c = vectors per cell, num_mu = their sum, sum_d num_var = the cell's
// distortion (and, in the last pass, the per-dimension squared deviations TF:883-891).
// A block takes FR (<= IC_FRAMES) consecutive frames: coalesced copy into LDS, one thread per
// frame for the classification, then the block writes the frames' one-hot rows as whole lines.
constexpr int IC_FRAMES = 64;
__global__ void __launch_bounds__(256)
k_init_classify(int N, int M, int D, int n_cells, int U, long long F, int FR,
                const double *__restrict__ X, const long long *__restrict__ off,
                const double *__restrict__ mean, double *__restrict__ gamma,
                double *__restrict__ post)
{
    extern __shared__ double xs[]; // [FR][D | 1]
    __shared__ int own_state[IC_FRAMES], own_gauss[IC_FRAMES];
    const int tid = threadIdx.x, DS = D | 1, G = N * M;
    const long long f0 = (long long)blockIdx.x * FR;
    const int nf = (int)((F - f0) < FR ? (F - f0) : FR);
    if (nf <= 0) return;
    {
        const double *src = X + f0 * D;
        int r = tid / D, d = tid - r * D; // element tid of the block's nf x D slice, then + 256 per step
        const int qs = 256 / D, rs = 256 - qs * D;
        for (int e = tid; e < nf * D; e += 256) {
            xs[r * DS + d] = src[e];
            r += qs;
            d += rs;
            if (d >= D) {
                d -= D;
                r++;
            }
        }
    }
    __syncthreads();
    if (tid < nf) {
        const long long f = f0 + tid;
        // utterance of this frame: last u with off[u] <= f
        int lo = 0, hi = U - 1;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (off[mid] <= f) lo = mid;
            else hi = mid - 1;
        }
        const int T = (int)(off[lo + 1] - off[lo]), j = (int)(f - off[lo]);
        // run k of an utterance of T frames: q = T/N frames, the first T%N runs one more
        const int q = T / N, r = T % N;
        int k;
        if (j < r * (q + 1)) k = j / (q + 1);
        else k = r + (q > 0 ? (j - r * (q + 1)) / q : 0);
        const double *x = xs + tid * DS;
        double best = 1.0e20;
        int cell = 0;
        for (int c = 0; c < n_cells; c++) {
            const double *mu = mean + ((size_t)k * M + c) * D;
            double dist = 0.0;
            for (int d = 0; d < D; d++) {
                double aux = mu[d] - x[d];
                dist += aux * aux;
            }
            if (dist < best) {
                best = dist;
                cell = c;
            }
        }
        own_state[tid] = k;
        own_gauss[tid] = k * M + cell;
    }
    __syncthreads();
    {
        double *g = gamma + f0 * N;
        int r = tid / N, i = tid - r * N;
        const int qs = 256 / N, rs = 256 - qs * N;
        for (int e = tid; e < nf * N; e += 256) {
            g[e] = i == own_state[r] ? 1.0 : 0.0;
            r += qs;
            i += rs;
            if (i >= N) {
                i -= N;
                r++;
            }
        }
    }
    {
        double *p = post + f0 * (size_t)G;
        int r = tid / G, e1 = tid - r * G;
        const int qs = 256 / G, rs = 256 - qs * G;
        for (int e = tid; e < nf * G; e += 256) {
            p[e] = e1 == own_gauss[r] ? 1.0 : 0.0;
            r += qs;
            e1 += rs;
            if (e1 >= G) {
                e1 -= G;
                r++;
            }
        }
    }
}

// ------------------------------------------------------------------ viterbi
// Max-plus lattice (absent from the reference; definition in oracle/ghmm_oracle.c):
//   delta_0(j) = (j == 0 ? 0 : -inf) + logb_j(0)
//   delta_t(j) = max_i (delta_{t-1}(i) + log a_ij) + logb_j(t), ties -> lowest i
//   score = delta_{T-1}(N-1); path by back-pointers from state N-1.
// One group of L lanes per utterance, lane = state.  Like the forward pass the time loop is
// branch-free (operand cursor without clamp, idle lanes store into the sink) and log A's band
// structure is chosen once per wave: for a left-to-right model only the candidates i = j-1 and
// i = j exist (every other log a_ij is -inf and can never win a strict `>`), taken with one DPP
// row shift instead of N lane reads.  Back-pointers are one byte, rows of L bytes per frame.
template <int L, bool BANDED>
__device__ __forceinline__ double viterbi_run(int N, int T, int j, bool act, const double *__restrict__ logA,
                                     const double *__restrict__ lb, unsigned char *__restrict__ ps,
                                     double *__restrict__ sink)
{
    double lacol[BANDED ? 1 : L];
    if (!BANDED) {
#pragma unroll
        for (int i = 0; i < L; i++) lacol[BANDED ? 0 : i] = (act && i < N) ? logA[i * N + j] : -INFINITY;
    }
    const double la_self = act ? logA[j * N + j] : -INFINITY;
    const double la_prev = (act && j > 0) ? logA[(j - 1) * N + j] : -INFINITY;
    // idle lanes: zeros from the sink's read-only half (stride 0), back-pointers into its other half
    const double *pl = act ? lb + j : sink + WAVE;
    const ptrdiff_t db = act ? N : 0;
    unsigned char *pp = act ? ps + j : (unsigned char *)sink;
    const ptrdiff_t dp = act ? L : 0;
    auto bnext = [&]() {
        const double v = *pl;
        pl += db;
        return v;
    };
    double d = ((j == 0) ? 0.0 : -INFINITY) + bnext();
    *pp = 0;
    pp += dp;
    auto step = [&](double q) {
        double best = -INFINITY;
        int arg = 0;
        if (BANDED) {
            const double c1 = group_up1<L>(d) + la_prev; // predecessor j-1 (the lower index first)
            const double c0 = d + la_self;
            if (c1 > best) {
                best = c1;
                arg = j - 1;
            }
            if (c0 > best) {
                best = c0;
                arg = j;
            }
        } else {
#pragma unroll
            for (int i = 0; i < L; i++)
                if (i < N) {
                    const double v = __shfl(d, i, L) + lacol[BANDED ? 0 : i];
                    if (v > best) {
                        best = v;
                        arg = i;
                    }
                }
        }
        d = best + q;
        *pp = (unsigned char)arg;
        pp += dp;
    };
    double q[PF];
    int t = 1;
#pragma unroll
    for (int k = 0; k < PF; k++) q[k] = bnext();
    for (; t + PF <= T; t += PF) {
        double qn[PF];
#pragma unroll
        for (int k = 0; k < PF; k++) qn[k] = bnext();
#pragma unroll
        for (int k = 0; k < PF; k++) step(q[k]);
#pragma unroll
        for (int k = 0; k < PF; k++) q[k] = qn[k];
    }
#pragma unroll
    for (int k = 0; k < PF - 1; k++)
        if (t + k < T) step(q[k]);
    return d;
}

// byte s (0..15) of a 16-byte back-pointer row
__device__ __forceinline__ int psi_byte(const uint4 r, int s)
{
    const unsigned lo = (s & 4) ? r.y : r.x, hi = (s & 4) ? r.w : r.z;
    const unsigned w = (s & 8) ? hi : lo;
    return (int)((w >> ((s & 3) * 8)) & 0xffu);
}

template <int L>
__global__ void __launch_bounds__(WAVE)
k_viterbi(int N, int U, const double *__restrict__ logA, const double *__restrict__ logb,
          const long long *__restrict__ off, unsigned char *__restrict__ psi,
          unsigned char *__restrict__ path, double *__restrict__ score, double *__restrict__ sink,
          const int *__restrict__ order)
{
    const int slot = blockIdx.x * (WAVE / L) + threadIdx.x / L;
    const int j = threadIdx.x % L;
    if (slot >= U) return;
    const int u = order[slot];
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    if (T <= 0) {
        if (j == 0) score[u] = 0.0;
        return;
    }
    const bool act = j < N;
    bool offband = false;
    for (int i = 0; i < N; i++)
        offband |= act && (logA[i * N + j] != -INFINITY && i != j && i != j - 1);
    const bool banded = !__any(offband);
    unsigned char *ps = psi + (size_t)f0 * L; // rows of L bytes
    double *snk = wave_sink(sink);
    double d;
    if (banded) d = viterbi_run<L, true>(N, T, j, act, logA, logb + f0 * N, ps, snk);
    else d = viterbi_run<L, false>(N, T, j, act, logA, logb + f0 * N, ps, snk);
    const double sc = __shfl(d, N - 1, L);
    __threadfence_block();
    if (j != 0) return;
    score[u] = sc;
    int s = N - 1;
    unsigned char *pu = path + f0;
    if (L == 16) {
        // whole 16-byte rows, read ahead of the chain (their addresses do not depend on it):
        // the chain itself is a byte select in registers
        constexpr int PB = 8;
        const uint4 *rows = (const uint4 *)ps;
        int t = T - 1;
        for (; t - PB + 1 >= 0; t -= PB) {
            uint4 r[PB];
#pragma unroll
            for (int k = 0; k < PB; k++) r[k] = rows[t - k];
#pragma unroll
            for (int k = 0; k < PB; k++) {
                pu[t - k] = (unsigned char)s;
                s = psi_byte(r[k], s);
            }
        }
        for (; t >= 0; t--) {
            pu[t] = (unsigned char)s;
            s = psi_byte(rows[t], s);
        }
    } else {
        for (int t = T - 1; t >= 0; t--) {
            pu[t] = (unsigned char)s;
            s = ps[(size_t)t * L + s];
        }
    }
}

} // namespace ghmm
