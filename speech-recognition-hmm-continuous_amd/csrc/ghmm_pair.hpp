// ghmm_pair.hpp — forward and backward recursions side by side.
//
// The reference's calc_beta (TF:1463-1516) multiplies every step by the forward pass's c_t, so
// it can only start when calc_alpha has finished; both are serial in t and bound by one
// utterance's dependent chain, with 1 000 utterances on a quarter of the chip's SIMDs.  Here
// the backward recursion carries its OWN normaliser, an exact power of two,
//     W_t(i)    = b_i(t+1) beta~_{t+1}(i)                       (kept, row t)
//     v_t       = A W_t,   beta~_t = v_t 2^(e_t),  e_t = BT_K - exponent(sum_i v_t(i))   (2^(e_t) kept)
//     beta~_{T-1} = e_{N-1}                                      (final-state constraint, TF:1484-1490)
// so that it needs nothing from the forward pass and runs beside it in the same launch
// (k_scan_pair, blockIdx.y = direction; k_scan_combine, waves 0 and 1 of a block, with the combine
// pass below behind a barrier in the same launch).  beta~_t is the reference's beta^_t up to a factor
// rho_t, and the reference's own scaling fixes that factor:
//     sum_i alpha^_t(i) beta^_t(i) = c_t kappa,  kappa = alpha^_{T-1}(N-1)      (induction on TF:1507)
//     =>  rho_t = c_t kappa / D_t,   D_t = sum_i alpha^_t(i) beta~_t(i)
// k_combine then forms, for every frame independently (chunks of an utterance in parallel),
//     gamma_t(i) = alpha^_t(i) beta^_t(i) / c_t = (alpha^_t(i) beta~_t(i) / D_t) kappa     (TF:1655-1660)
//     xi_t(i,j)  = alpha^_t(i) a_ij b_j(t+1) beta^_{t+1}(j) = (alpha^_t(i) a_ij W_t(j) c_{t+1} / D_{t+1}) kappa
// (TF:1601-1614) and, when asked for it, beta^_t = rho_t beta~_t.
//
// RANGE.  Rows of beta~ sum to about 2^BT_K (BT_K = 680, 5e204), not to 1: a component survives
// down to 2^-1022 absolutely = 1e-512 of its row's largest, and since the scale is a power of two
// the normalisation rounds nothing.  The reference keeps beta^_t(i) = rho_t beta~_t(i) down to
// 2^-1022 as well, i.e. it sees FURTHER below the row's largest only where rho_t > 2^BT_K.  Such
// an utterance (forward and backward mass more than 200 decades apart at some frame: start
// models far from their data), one whose D_t leaves the numbers altogether, and one with no path
// into the last state (kappa = 0: shorter than the model, or numerically dead; the reference's
// beta^ and NaN artefacts are what its own order of operations makes them) is put on a list by
// k_combine and taken again, whole, by k_backward_fix in the reference's order of operations
// (calc_beta scaled by c_t, dense inner loops as at TF:1493-1510).  Round 2 normalised the rows
// to a sum of 1 (v_t / s_t): 17 of 607 harsh shapes then lost components the reference keeps.
#pragma once
#include "ghmm_kernels.hpp"

namespace ghmm {

#ifndef GHMM_CB_CH
#define GHMM_CB_CH 8 // (measurement builds override it: profiles/tools/lab.sh)
#endif
constexpr int CB_CH = GHMM_CB_CH; // chunks of an utterance handled by different groups of k_combine
constexpr int SC_UPB = 4;     // utterances per block of k_scan_combine, at least
constexpr int SC_GROUPS = SC_UPB * GHMM_CB_CH; // its combine groups = utterances x chunks per utterance
constexpr int CB_PF = 4; // frames of operands read ahead in k_combine (x 2 register sets x 4 operands)
// rows of beta~ are scaled to a sum in [2^BT_K, 2^(BT_K+1)): W = b beta~ stays finite for densities
// up to 1e100 (a 39-d Gaussian at the 1e-5 variance floor peaks at 1e82) and keeps 512 decades
constexpr int BT_K = 680;

template <int L, bool BANDED>
__device__ __forceinline__ void backward_own_run(int N, int T, int i, bool act, const double *__restrict__ A,
                                        const double *__restrict__ bu, double *__restrict__ wu,
                                        double *__restrict__ sbu, double *__restrict__ sink)
{
    const double a_self = act ? A[i * N + i] : 0.0;
    const double a_next = (act && i + 1 < N) ? A[i * N + i + 1] : 0.0;
    double arow[BANDED ? 1 : L];
    if (!BANDED) {
#pragma unroll
        for (int j = 0; j < L; j++) arow[BANDED ? 0 : j] = (act && j < N) ? A[i * N + j] : 0.0;
    }
    const int dn = act ? N : 0;
    const double *pb0 = act ? bu + i : sink + WAVE;
    double be = (i == N - 1) ? ldexp(1.0, BT_K) : 0.0; // e_{N-1} at the rows' scale
    double *pw = act ? wu + (size_t)(T - 1) * N + i : sink;
    double *ps = (i == 0) ? sbu + (T - 1) : sink;
    const int ds = (i == 0) ? 1 : 0;
    *pw = be;  // row T-1 holds beta~_{T-1} itself (no W behind the last frame)
    *ps = 1.0;
    pw -= dn; ps -= ds;
    auto step = [&](double bnext) {
        const double w = be * bnext;
        *pw = w;
        double v;
        if (BANDED) {
            v = a_self * w + a_next * group_down1<L>(w);
        } else {
            v = 0.0;
#pragma unroll
            for (int j = 0; j < L; j++)
                if (j < N) v += arow[BANDED ? 0 : j] * __shfl(w, j, L);
        }
        // the row's scale: an exact power of two that puts its sum at 2^BT_K (three instructions
        // on the chain: exponent, subtract, ldexp; nothing is rounded).  s = 0: nothing can
        // follow, beta~ = 0 from here on like beta^; s = inf / NaN: the combine pass sees it
        const double s = group_sum<L>(v);
        const int e = BT_K - __builtin_amdgcn_frexp_exp(s);
        be = ldexp(v, e);
        *ps = ldexp(1.0, e);
        pw -= dn; ps -= ds;
    };
    // b of frame T-1, T-2, ... (frame t+1 for t = T-2 .. 0) through a descending cursor, PFF
    // steps ahead, neither clamped nor predicated: it runs up to 2 PFF frames in front of the
    // utterance (the previous utterance's rows or the padding in front of b, B_PAD_FRAMES)
    const double *pl = pb0 + (ptrdiff_t)(T - 1) * dn;
    const ptrdiff_t dnl = dn;
    auto bnext = [&]() {
        const double v = *pl;
        pl -= dnl;
        return v;
    };
    double bq[PFF];
    int t = T - 2;
#pragma unroll
    for (int k = 0; k < PFF; k++) bq[k] = bnext();
    for (; t - PFF + 1 >= 0; t -= PFF) {
        double bn[PFF];
#pragma unroll
        for (int k = 0; k < PFF; k++) bn[k] = bnext();
#pragma unroll
        for (int k = 0; k < PFF; k++) step(bq[k]);
#pragma unroll
        for (int k = 0; k < PFF; k++) bq[k] = bn[k];
    }
#pragma unroll
    for (int k = 0; k < PFF - 1; k++)
        if (t - k >= 0) step(bq[k]);
}

// blockIdx.y (or `only` when one direction is wanted): 0 = calc_alpha + calc_probability,
// 1 = the backward recursion with its own normaliser.  Groups of 16/64 lanes = utterances.
template <int L>
__global__ void __launch_bounds__(WAVE)
k_scan_pair(int N, int U, int only, const double *__restrict__ A, const double *__restrict__ b,
            const long long *__restrict__ off, double *__restrict__ alpha, double *__restrict__ scale,
            double *__restrict__ sinv, const double *__restrict__ lognorm, double *__restrict__ loglik,
            double *__restrict__ wrow, double *__restrict__ sb, double *__restrict__ sink,
            const int *__restrict__ order)
{
    // the wave's 4 (or 1) utterances are neighbours in the corpus' length order (longest
    // first): equal work inside a wave, the long chains start first
    const int slot = blockIdx.x * (WAVE / L) + threadIdx.x / L;
    const int i = threadIdx.x % L;
    if (slot >= U) return;
    const int u = order[slot];
    const int dir = only >= 0 ? only : (int)blockIdx.y;
    if (only == 2) { // log P only (ghmm_score): nothing but loglik[] is written
        forward_utt<L, false, true>(N, u, i, A, b, off, alpha, scale, sinv, lognorm, loglik, sink);
        return;
    }
    if (dir == 0) {
        // with the backward direction alongside, k_combine follows and takes the logs of log P
        forward_utt<L, false>(N, u, i, A, b, off, alpha, scale, sinv, lognorm, loglik, sink, only == 0);
        return;
    }
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    if (T <= 0) return;
    const bool act = i < N;
    bool offband = false;
    for (int j = 0; j < N; j++)
        offband |= act && (A[i * N + j] != 0.0 && j != i && j != i + 1);
    const bool banded = !__any(offband);
    double *snk = wave_sink(sink);
    if (banded)
        backward_own_run<L, true>(N, T, i, act, A, b + f0 * N, wrow + f0 * N, sb + f0, snk);
    else
        backward_own_run<L, false>(N, T, i, act, A, b + f0 * N, wrow + f0 * N, sb + f0, snk);
}

// a * b * c * d where the product is a normal number although a partial product may leave the
// range or turn subnormal (beta^ = beta~ * (1/D) * c * kappa, asked for by ghmm_fetch only)
__device__ inline double mul4_ranged(double a, double b, double c, double d)
{
    const double m = (__builtin_amdgcn_frexp_mant(a) * __builtin_amdgcn_frexp_mant(b)) *
                     (__builtin_amdgcn_frexp_mant(c) * __builtin_amdgcn_frexp_mant(d));
    const int e = (__builtin_amdgcn_frexp_exp(a) + __builtin_amdgcn_frexp_exp(b)) +
                  (__builtin_amdgcn_frexp_exp(c) + __builtin_amdgcn_frexp_exp(d));
    return ldexp(m, e); // (0, inf and NaN: the mantissa is the value itself, its exponent 0)
}

// gamma, the xi / den sums (one partial slot per (utterance, chunk)) and optionally beta^ from
// alpha^, c, W and the rows' scales.  Group = (utterance, chunk of its frames), frames descending.
// MD = widest band offset that can carry statistics (delta <= MD).  Returns whether the
// utterance has to be taken again in the reference's order (see RANGE above).
template <int L, bool BANDED, bool WANT_BETA, int MD = MAX_DELTA>
__device__ __forceinline__ bool combine_run(int N, int T, int delta, int i, bool act, int slot, int tlo, int thi,
                                   const double *__restrict__ A, const double *__restrict__ au,
                                   const double *__restrict__ su, const double *__restrict__ wu,
                                   const double *__restrict__ sbu, double *__restrict__ beu,
                                   double *__restrict__ gu, double *__restrict__ part_xi,
                                   double *__restrict__ part_dena, double *__restrict__ part_denc,
                                   double *__restrict__ sink, int S)
{
    const double a_self = act ? A[i * N + i] : 0.0;
    const double a_next = (act && i + 1 < N) ? A[i * N + i + 1] : 0.0;
    double arow[BANDED ? 1 : L];
    if (!BANDED) {
#pragma unroll
        for (int j = 0; j < L; j++) arow[BANDED ? 0 : j] = (act && j < N) ? A[i * N + j] : 0.0;
    }
    double aband[MD + 1], xi[MD + 1];
#pragma unroll
    for (int o = 0; o <= MD; o++) {
        aband[o] = (act && i + o < N && o <= delta) ? A[i * N + i + o] : 0.0;
        xi[o] = 0.0;
    }
    const int dn = act ? N : 0;
    const double *pa0 = act ? au + i : sink + WAVE, *pw0 = act ? wu + i : sink + WAVE;
    const double kappa = au[(size_t)(T - 1) * N + (N - 1)];
    // beta~_t(i) from row t of W (row T-1 holds beta~_{T-1} itself); wd = W_t(i+1)
    auto beta_own = [&](int t, double w, double &wd) {
        wd = group_down1<L>(w);
        double v;
        if (BANDED) {
            v = a_self * w + a_next * wd;
        } else {
            v = 0.0;
#pragma unroll
            for (int j = 0; j < L; j++)
                if (j < N) v += arow[BANDED ? 0 : j] * __shfl(w, j, L);
        }
        return t == T - 1 ? w : v * sbu[t];
    };
    // 1 / D (0 where D has left the numbers: the frame is then reported, see `dead`)
    auto recipD = [&](double D) {
        const double r0 = __builtin_amdgcn_rcp(D);
        double r = fma(r0, fma(-D, r0, 1.0), r0);
        r = fma(r, fma(-D, r, 1.0), r);
        return (D > 0.0 && D < INFINITY) ? r : 0.0;
    };
    // 1/D of the frame behind the chunk, for the chunk's first xi
    double rDn = 0.0, cn = 0.0;
    if (thi < T) {
        double wd;
        const double bt = beta_own(thi, pw0[(size_t)thi * dn], wd);
        rDn = recipD(group_sum<L>(pa0[(size_t)thi * dn] * bt));
        cn = su[thi];
    }
    double dena = 0.0, denc = 0.0;
    double rhomax = 0.0; // max over the chunk's frames of c_t / D_t = rho_t / kappa
    bool dead = false;   // a frame whose D_t is 0, inf or NaN
    double *pg = act ? gu + (size_t)(thi - 1) * N + i : sink;
    double *pbe = act ? beu + (size_t)(thi - 1) * N + i : sink;
    auto frame = [&](int t, double w, double al, double ct, double sbt) {
        double wd = group_down1<L>(w);
        double v;
        if (BANDED) {
            v = a_self * w + a_next * wd;
        } else {
            v = 0.0;
#pragma unroll
            for (int j = 0; j < L; j++)
                if (j < N) v += arow[BANDED ? 0 : j] * __shfl(w, j, L);
        }
        const double bt = t == T - 1 ? w : v * sbt; // (sbt is a power of two: the scan's beta~ bit for bit)
        const double p = al * bt;
        const double D = group_sum<L>(p);
        const double rD = recipD(D);
        // the frame's share first (<= 1), kappa last: neither product can leave the range
        // unless gamma itself does
        const double g = (p * rD) * kappa;
        *pg = g;
        denc += g;
        const double inner = t < T - 1 ? 1.0 : 0.0; // the last frame has no transition behind it
        dena = fma(g, inner, dena);
        const double rc1 = (cn * rDn) * inner; // c_{t+1} / D_{t+1} = rho_{t+1} / kappa
        xi[0] = fma(al * w, rc1, xi[0]);
        xi[1] = fma(al * wd, rc1, xi[1]);
#pragma unroll
        for (int o = 2; o <= MD; o++)
            if (o <= delta) {
                const double wj = __shfl_down(w, o, L);
                xi[o] += (i + o < N) ? al * wj * rc1 : 0.0;
            }
        const double crd = ct * rD;
        rhomax = fmax(rhomax, crd);
        dead |= rD == 0.0;
        if (WANT_BETA) {
            *pbe = mul4_ranged(bt, rD, ct, kappa);
            pbe -= dn;
        }
        rDn = rD;
        cn = ct;
        pg -= dn;
    };
    // operands of frame t: W_t(i), alpha^_t(i), c_t, 2^(e_t), read CB_PF frames ahead with
    // addresses clamped into the chunk (never predicated)
    auto cl = [&](int t) { return (size_t)(t < tlo ? tlo : t); };
    double qw[CB_PF], qa[CB_PF], qc[CB_PF], qs[CB_PF];
    int t = thi - 1;
#pragma unroll
    for (int k = 0; k < CB_PF; k++) {
        const size_t f = cl(t - k);
        qw[k] = pw0[f * dn]; qa[k] = pa0[f * dn]; qc[k] = su[f]; qs[k] = sbu[f];
    }
    for (; t - CB_PF + 1 >= tlo; t -= CB_PF) {
        double nw[CB_PF], na[CB_PF], nc[CB_PF], ns[CB_PF];
#pragma unroll
        for (int k = 0; k < CB_PF; k++) {
            const size_t f = cl(t - CB_PF - k);
            nw[k] = pw0[f * dn]; na[k] = pa0[f * dn]; nc[k] = su[f]; ns[k] = sbu[f];
        }
#pragma unroll
        for (int k = 0; k < CB_PF; k++) frame(t - k, qw[k], qa[k], qc[k], qs[k]);
#pragma unroll
        for (int k = 0; k < CB_PF; k++) {
            qw[k] = nw[k]; qa[k] = na[k]; qc[k] = nc[k]; qs[k] = ns[k];
        }
    }
#pragma unroll
    for (int k = 0; k < CB_PF - 1; k++)
        if (t - k >= tlo) frame(t - k, qw[k], qa[k], qc[k], qs[k]);
    if (act) {
#pragma unroll
        for (int o = 0; o <= MD; o++) // compile-time indices: the arrays stay in registers
            if (o <= delta) part_xi[pxi_at(slot, i, o, S)] = (aband[o] * xi[o]) * kappa;
        part_dena[pden_at(slot, i, S)] = dena;
        part_denc[pden_at(slot, i, S)] = denc;
    }
    // the reference's order of operations decides when: there is no path into the last state
    // (kappa is 0 or NaN), a frame's D left the numbers, or rho_t = kappa c_t / D_t reaches the
    // rows' scale 2^BT_K (rho'_t = rho_t 2^-BT_K >= 1: the reference's beta^ sees further down)
    // (every term is uniform over the group's lanes: D_t comes from group_sum)
    return !(kappa > 0.0) || dead || !(kappa * rhomax < 1.0);
}

// DENSE = false: A is known (on the host, ghmm_model_set) to be band-diagonal with a_ij = 0
// unless j = i or i + 1, only that form is compiled (half the registers, twice the waves in
// flight); DENSE = true decides per wave on the device like the one-pass kernels.
// group qs = (utterance in length order, chunk), lane i of the group
// Returns whether the utterance has to be taken again in the reference's order; with fix_mark the
// utterance is put on k_backward_fix's list here (once, whichever of its chunks asks first).
template <int L, bool WANT_BETA, bool DENSE>
__device__ __forceinline__ bool combine_group(int qs, int i, int N, int U, int delta, const double *__restrict__ A,
          const long long *__restrict__ off,
          const double *__restrict__ alpha, const double *__restrict__ scale,
          const double *__restrict__ wrow, const double *__restrict__ sb, double *__restrict__ beta,
          double *__restrict__ gamma, double *__restrict__ part_xi, double *__restrict__ part_dena,
          double *__restrict__ part_denc, double *__restrict__ sink,
          const double *__restrict__ lognorm, double *__restrict__ lpart, double *__restrict__ logk,
          const int *__restrict__ order, int *__restrict__ fix_mark, int stamp, int *__restrict__ fix_cnt,
          int *__restrict__ fix_list, int nch = CB_CH)
{   // nch = chunks per utterance: CB_CH in k_combine's launch, 2 .. CB_CH in k_scan_combine's
    const int k = qs % nch;
    if (qs / nch >= U) return false;
    const int u = order[qs / nch]; // longest utterances first
    const int q = u * nch + k;     // partial-sum slot of (utterance, chunk)
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    const bool act = i < N;
    const int tlo = (int)((long long)T * k / nch), thi = (int)((long long)T * (k + 1) / nch);
    if (lpart) {
        // calc_probability (TF:1536-1553) in pieces: this chunk's -sum log c_t (+ the robust
        // mode's normalisers), and log alpha^_{T-1}(N-1) from the chunk that ends the utterance
        double lp = 0.0;
        log_product pc;
        for (int t = tlo + i; t < thi; t += L) {
            pc.mul(scale[f0 + t]);
            if (lognorm) lp += lognorm[f0 + t];
        }
        lp = group_sum<L>(lp - pc.log_value());
        if (i == 0) {
            lpart[q] = lp;
            if (k == nch - 1) logk[u] = T > 0 ? log(alpha[(f0 + T - 1) * N + (N - 1)]) : 0.0;
        }
    }
    if (T <= 0 || thi <= tlo) {
        if (act) {
            for (int o = 0; o <= MAX_DELTA; o++) part_xi[pxi_at(q, i, o, U * nch)] = 0.0;
            part_dena[pden_at(q, i, U * nch)] = 0.0;
            part_denc[pden_at(q, i, U * nch)] = 0.0;
        }
        return false;
    }
    double *snk = wave_sink(sink);
    bool banded = true;
    if (DENSE) {
        bool offband = false;
        for (int j = 0; j < N; j++)
            offband |= act && (A[i * N + j] != 0.0 && j != i && j != i + 1);
        banded = !__any(offband);
    }
    bool again = false;
    if (banded)
        again = combine_run<L, true, WANT_BETA, DENSE ? MAX_DELTA : 1>(N, T, delta, i, act, q, tlo, thi, A, alpha + f0 * N, scale + f0,
                                        wrow + f0 * N, sb + f0, beta + f0 * N, gamma + f0 * N, part_xi,
                                        part_dena, part_denc, snk, U * nch);
    else if (DENSE)
        again = combine_run<L, false, WANT_BETA>(N, T, delta, i, act, q, tlo, thi, A, alpha + f0 * N, scale + f0,
                                         wrow + f0 * N, sb + f0, beta + f0 * N, gamma + f0 * N, part_xi,
                                         part_dena, part_denc, snk, U * nch);
    // the utterance goes on k_backward_fix's list once, whichever of its chunks asks first
    // (marks carry the pass's stamp: nothing is ever cleared)
    if (fix_mark && again && i == 0 && atomicExch(&fix_mark[u], stamp) != stamp) fix_list[atomicAdd(fix_cnt, 1)] = u;
    return again;
}

template <int L, bool WANT_BETA, bool DENSE>
__global__ void __launch_bounds__(WAVE, DENSE ? 2 : 4)
k_combine(int N, int U, int delta, const double *__restrict__ A, const long long *__restrict__ off,
          const double *__restrict__ alpha, const double *__restrict__ scale,
          const double *__restrict__ wrow, const double *__restrict__ sb, double *__restrict__ beta,
          double *__restrict__ gamma, double *__restrict__ part_xi, double *__restrict__ part_dena,
          double *__restrict__ part_denc, double *__restrict__ sink,
          const double *__restrict__ lognorm, double *__restrict__ lpart, double *__restrict__ logk,
          const int *__restrict__ order, int *__restrict__ fix_mark, int stamp, int *__restrict__ fix_cnt,
          int *__restrict__ fix_list)
{
    combine_group<L, WANT_BETA, DENSE>(blockIdx.x * (WAVE / L) + threadIdx.x / L, threadIdx.x % L, N, U, delta, A, off,
                                       alpha, scale, wrow, sb, beta, gamma, part_xi, part_dena, part_denc, sink,
                                       lognorm, lpart, logk, order, fix_mark, stamp, fix_cnt, fix_list);
}

// A band-diagonal A (known on the host): both scans and the combine pass in ONE launch.  A block owns
// four utterances; at 16-lane groups wave 0 runs their forward recursion, wave 1 their backward
// recursion (wider groups: more scan waves), the other waves wait at the barrier, then the block's
// CB_CH waves take one chunk of every utterance each.  The combine pass of a block starts the moment
// ITS scans end: no launch boundary (the separate combine launch drains, starts and ramps up: 27 -> 18 us at 1 000 utterances), its
// operands were written by this compute unit a moment ago, and on large corpora the chip always
// holds blocks in both phases (12 500 utterances: 0.85 -> 0.73 ms).
// k_backward_fix's loop body for one utterance of k_scan_combine's block (a function of its own: the
// dense recursion's registers stay out of the kernel's main path)
template <int L>
__device__ __noinline__ void fix_in_block(int N, int U, int delta, int u, int i, const double *__restrict__ A,
                                          const double *__restrict__ b, const long long *__restrict__ off,
                                          const double *__restrict__ alpha, const double *__restrict__ scale,
                                          double *__restrict__ beta, double *__restrict__ gamma,
                                          double *__restrict__ part_xi, double *__restrict__ part_dena,
                                          double *__restrict__ part_denc, double *__restrict__ snk,
                                          int *__restrict__ fix_cnt, int nch)
{
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    if (T <= 0) return;
    const bool act = i < N;
    const int S = U * nch;
    backward_run<L, false>(N, T, delta, i, act, u * nch, A, b + f0 * N, alpha + f0 * N, scale + f0,
                           (const double *)nullptr, beta + f0 * N, gamma + f0 * N, part_xi, part_dena, part_denc,
                           snk, S);
    if (act)
        for (int k = 1; k < nch; k++) {
            for (int o = 0; o <= MAX_DELTA; o++) part_xi[pxi_at(u * nch + k, i, o, S)] = 0.0;
            part_dena[pden_at(u * nch + k, i, S)] = 0.0;
            part_denc[pden_at(u * nch + k, i, S)] = 0.0;
        }
    if (i == 0) atomicAdd(fix_cnt, 1);
}

// An utterance the combine pass wants taken again in the reference's order (see RANGE; none on data a
// model fits) is taken again right here, by wave 0 of its block with k_backward_fix's loop, behind a
// second barrier: the block owns every chunk of it.  fix_cnt counts them, fix_cnt_next is zeroed for
// the next pass (as k_backward_fix does behind k_combine).
template <int L, bool WANT_BETA>
__global__ void __launch_bounds__(CB_CH *WAVE)
k_scan_combine(int N, int U, int delta, int upb, const double *__restrict__ A, const double *__restrict__ b,
               const long long *__restrict__ off, double *__restrict__ alpha, double *__restrict__ scale,
               double *__restrict__ wrow, double *__restrict__ sb, double *__restrict__ beta,
               double *__restrict__ gamma, double *__restrict__ part_xi, double *__restrict__ part_dena,
               double *__restrict__ part_denc, double *__restrict__ sink,
               const double *__restrict__ lognorm, double *__restrict__ lpart, double *__restrict__ logk,
               const int *__restrict__ order, int *__restrict__ fix_cnt, int *__restrict__ fix_cnt_next)
{
    // upb = SC_UPB (4), 8 or 16 utterances per block (the host: more of them, in fewer chunks each,
    // the shorter they are): 2 upb / gpw scan waves — forward and backward of WAVE / L utterances
    // each, at most the block's eight — and upb x nch = SC_GROUPS combine groups in 1, 2 or 4 passes
    // of the eight waves.  (With WAVE / L utterances per block a 20-state model made 500 blocks of
    // 1 000 utterances, two rounds over the compute units; with eight chunks for a 30-frame word
    // the combine pass was all set-up and the reduction read 80 000 slots.)
    constexpr int gpw = WAVE / L, PASSES = SC_GROUPS / (CB_CH * gpw);
    __shared__ int fix_flag[SC_GROUPS / 2];
    const int nch = SC_GROUPS / upb, nsw = 2 * (upb / gpw);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), l = threadIdx.x % WAVE;
    if (threadIdx.x < SC_GROUPS / 2) fix_flag[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) *fix_cnt_next = 0; // the counter of the next pass
    if (w < nsw) {
        const int slot = blockIdx.x * upb + (w >> 1) * gpw + l / L;
        const int i = l % L;
        if (slot < U) {
            const int u = order[slot];
            const long long f0 = off[u];
            const int T = (int)(off[u + 1] - f0);
            if (T > 0) {
                // (the scan waves' sink regions side by side: through wave_sink they would sit 8 KB
                // apart, one per block, on a fraction of the L2 channels)
                double *snk = sink + (size_t)((blockIdx.x * nsw + w) % SINK_WAVES) * 2 * WAVE + l;
                if ((w & 1) == 0)
                    forward_run<L, true, false>(N, T, i, i < N, A, b + f0 * N, alpha + f0 * N, scale + f0,
                                                (double *)nullptr, snk, N);
                else
                    backward_own_run<L, true>(N, T, i, i < N, A, b + f0 * N, wrow + f0 * N, sb + f0, snk);
            }
        }
    }
    __syncthreads(); // (orders the scans' global stores before the block's reads of them)
#pragma unroll 1
    for (int pass = 0; pass < PASSES; pass++) {
        const int gid = (pass * CB_CH + w) * gpw + l / L; // (utterance of the block, chunk) = (gid / nch, gid % nch)
        const bool again = combine_group<L, WANT_BETA, false>(blockIdx.x * SC_GROUPS + gid, l % L, N, U, delta, A,
                                           off, alpha, scale, wrow, sb, beta, gamma, part_xi, part_dena, part_denc,
                                           sink, lognorm, lpart, logk, order, (int *)nullptr, 0, (int *)nullptr,
                                           (int *)nullptr, nch);
        if (again && (l % L) == 0) fix_flag[gid / nch] = 1;
    }
    __syncthreads();
    // wave w takes the listed ones among utterances w gpw .. w gpw + gpw - 1 (+ 8 gpw, ... for upb > 8 gpw: never)
    if (w >= upb / gpw) return;
    const int g = w * gpw + l / L, slot = blockIdx.x * upb + g;
    if (slot >= U || !fix_flag[g]) return;
    fix_in_block<L>(N, U, delta, order[slot], l % L, A, b, off, alpha, scale, beta, gamma, part_xi, part_dena, part_denc,
                    wave_sink(sink), fix_cnt, nch);
}

// The utterances k_combine listed, whole, in the reference's own order of operations:
// calc_beta scaled by the forward pass's c_t with the dense inner loop of TF:1493-1510 (an
// a_ij = 0 next to an overflowed beta^ makes the reference's NaN; the band-only update would
// not), gamma = alpha^ beta^ / c_t (TF:1655-1660), xi and den sums on the chain (TF:1601-1618).
// Their gamma (and beta^) rows and their partial-sum slots are rewritten: chunk slot 0 takes the
// utterance's sums, the other CB_CH - 1 slots 0.  Launched behind every k_combine (and behind
// k_backward, for utterances whose band-only update met an overflowed beta^); leaves at once when
// the list is empty.
template <int L>
__global__ void __launch_bounds__(WAVE)
k_backward_fix(int N, int U, int delta, const double *__restrict__ A, const double *__restrict__ b,
               const long long *__restrict__ off, const double *__restrict__ alpha,
               const double *__restrict__ scale, double *__restrict__ beta, double *__restrict__ gamma,
               double *__restrict__ part_xi, double *__restrict__ part_dena, double *__restrict__ part_denc,
               double *__restrict__ sink, const int *__restrict__ fix_cnt, const int *__restrict__ fix_list,
               int *__restrict__ fix_cnt_next, int spu, const double *__restrict__ sinv)
{   // spu = partial-sum slots per utterance: CB_CH behind k_combine, 1 behind k_backward
    const int n = *fix_cnt;
    if (blockIdx.x == 0 && threadIdx.x == 0) *fix_cnt_next = 0; // the counter of the next pass
    if (n == 0) return;
    const int i = threadIdx.x % L;
    const bool act = i < N;
    const int S = U * spu;
    double *snk = wave_sink(sink);
    for (int idx = blockIdx.x * (WAVE / L) + threadIdx.x / L; idx < n; idx += gridDim.x * (WAVE / L)) {
        const int u = fix_list[idx];
        const long long f0 = off[u];
        const int T = (int)(off[u + 1] - f0);
        if (T <= 0) continue; // (never listed)
        backward_run<L, false>(N, T, delta, i, act, u * spu, A, b + f0 * N, alpha + f0 * N, scale + f0,
                               sinv ? sinv + f0 : (const double *)nullptr, beta + f0 * N, gamma + f0 * N,
                               part_xi, part_dena, part_denc, snk, S);
        if (act)
            for (int k = 1; k < spu; k++) {
                for (int o = 0; o <= MAX_DELTA; o++) part_xi[pxi_at(u * spu + k, i, o, S)] = 0.0;
                part_dena[pden_at(u * spu + k, i, S)] = 0.0;
                part_denc[pden_at(u * spu + k, i, S)] = 0.0;
            }
    }
}

} // namespace ghmm

namespace ghmm {
// log P per utterance from the pieces k_combine left (only when somebody asks for the vector)
__global__ void __launch_bounds__(256)
k_loglik_assemble(int U, int nch, const double *__restrict__ lpart, const double *__restrict__ logk,
                  double *__restrict__ loglik)
{
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= U) return;
    double lp = 0.0;
    for (int k = 0; k < nch; k++) lp += lpart[(size_t)u * nch + k];
    loglik[u] = lp + logk[u];
}
} // namespace ghmm
