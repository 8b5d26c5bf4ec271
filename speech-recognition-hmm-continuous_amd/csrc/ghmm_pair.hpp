// ghmm_pair.hpp — forward and backward recursions side by side.
//
// The reference's calc_beta (TF:1463-1516) multiplies every step by the forward pass's c_t, so
// it can only start when calc_alpha has finished; both are serial in t and bound by one
// utterance's dependent chain, with 1 000 utterances on a quarter of the chip's SIMDs.  Here
// the backward recursion carries its OWN normaliser,
//     W_t(i)    = b_i(t+1) beta~_{t+1}(i)                       (kept, row t)
//     v_t       = A W_t,   s_t = sum_i v_t(i),   beta~_t = v_t / s_t     (1/s_t kept)
//     beta~_{T-1} = e_{N-1}                                      (final-state constraint, TF:1484-1490)
// so that it needs nothing from the forward pass and runs beside it in the same launch
// (k_scan_pair, blockIdx.y = direction).  beta~_t is the reference's beta^_t up to a factor
// rho_t, and the reference's own scaling fixes that factor:
//     sum_i alpha^_t(i) beta^_t(i) = c_t kappa,  kappa = alpha^_{T-1}(N-1)      (induction on TF:1507)
//     =>  rho_t = c_t kappa / D_t,   D_t = sum_i alpha^_t(i) beta~_t(i)
// k_combine then forms, for every frame independently (chunks of an utterance in parallel),
//     gamma_t(i) = alpha^_t(i) beta^_t(i) / c_t = kappa alpha^_t(i) beta~_t(i) / D_t     (TF:1655-1660)
//     xi_t(i,j)  = alpha^_t(i) a_ij b_j(t+1) beta^_{t+1}(j) = alpha^_t(i) a_ij W_t(j) rho_{t+1}   (TF:1601-1614)
// and, when asked for it, beta^_t = rho_t beta~_t.  An utterance that cannot end in the last
// state (kappa = 0: shorter than the model, or numerically dead) has gamma = xi = 0 here as in
// the reference, whose alpha^ beta^ products are all 0 then; its beta^ is rebuilt from
// rho_t = c_t s_t rho_{t+1} instead.
#pragma once
#include "ghmm_kernels.hpp"

namespace ghmm {

#ifndef GHMM_CB_CH
#define GHMM_CB_CH 8 // (measurement builds override it: profiles/tools/lab.sh)
#endif
constexpr int CB_CH = GHMM_CB_CH; // chunks of an utterance handled by different groups of k_combine
constexpr int CB_PF = 4; // frames of operands read ahead in k_combine (x 2 register sets x 4 operands)

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
    double be = (i == N - 1) ? 1.0 : 0.0;
    double *pw = act ? wu + (size_t)(T - 1) * N + i : sink;
    double *ps = (i == 0) ? sbu + (T - 1) : sink;
    const int ds = (i == 0) ? 1 : 0;
    *pw = be;  // row T-1 holds beta~_{T-1} itself
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
        const double s = group_sum<L>(v);
        const double r0 = __builtin_amdgcn_rcp(s);
        double r = fma(r0, fma(-s, r0, 1.0), r0);
        r = s > 0.0 ? r : 0.0; // nothing can follow (or NaN): beta~ = 0 from here on, like beta^
        be = v * r;
        *ps = r;
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

// gamma, the xi / den sums (one partial slot per (utterance, chunk)) and optionally beta^ from
// alpha^, c, W and 1/s.  Group = (utterance, chunk of its frames), frames descending.
// MD = widest band offset that can carry statistics (delta <= MD)
template <int L, bool BANDED, bool WANT_BETA, int MD = MAX_DELTA>
__device__ __forceinline__ void combine_run(int N, int T, int delta, int i, bool act, int slot, int tlo, int thi,
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
    // kappa / D, 0 when the utterance has no path into the last state (everything is 0 then)
    auto factor = [&](double D) {
        const double r0 = __builtin_amdgcn_rcp(D);
        double r = fma(r0, fma(-D, r0, 1.0), r0);
        r = fma(r, fma(-D, r, 1.0), r);
        return (D > 0.0 && D < INFINITY) ? kappa * r : 0.0;
    };
    // rho of the frame behind the chunk, for the chunk's first xi
    double facn = 0.0, cn = 0.0, rhon = 0.0;
    if (thi < T) {
        double wd;
        const double bt = beta_own(thi, pw0[(size_t)thi * dn], wd);
        const double D = group_sum<L>(pa0[(size_t)thi * dn] * bt);
        facn = factor(D);
        cn = su[thi];
    }
    if (WANT_BETA && !(kappa > 0.0)) {
        // no path into the last state: rho from its own recursion, rho_t = c_t s_t rho_{t+1}
        rhon = su[T - 1];
        // (c_t s_t first: c_t rho_{t+1} alone can be beyond the largest double where rho_t is not)
        for (int t = T - 2; t >= thi; t--) rhon = sbu[t] > 0.0 ? rhon * (su[t] / sbu[t]) : 0.0;
    }
    double dena = 0.0, denc = 0.0;
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
        const double bt = t == T - 1 ? w : v * sbt;
        const double p = al * bt;
        const double D = group_sum<L>(p);
        const double fac = factor(D);
        const double g = p * fac;
        *pg = g;
        denc += g;
        const double inner = t < T - 1 ? 1.0 : 0.0; // the last frame has no transition behind it
        dena = fma(g, inner, dena);
        const double rho1 = cn * facn * inner; // rho_{t+1}
        xi[0] = fma(al * w, rho1, xi[0]);
        xi[1] = fma(al * wd, rho1, xi[1]);
#pragma unroll
        for (int o = 2; o <= MD; o++)
            if (o <= delta) {
                const double wj = __shfl_down(w, o, L);
                xi[o] += (i + o < N) ? al * wj * rho1 : 0.0;
            }
        if (WANT_BETA) {
            if (kappa > 0.0) {
                *pbe = bt * (ct * fac);
            } else {
                if (t < T - 1) rhon = sbt > 0.0 ? rhon * (ct / sbt) : 0.0;
                *pbe = bt > 0.0 ? bt * rhon : 0.0; // (a zero stays zero when rho has overflowed)
            }
            pbe -= dn;
        }
        facn = fac;
        cn = ct;
        pg -= dn;
    };
    // operands of frame t: W_t(i), alpha^_t(i), c_t, 1/s_t, read CB_PF frames ahead with
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
            if (o <= delta) part_xi[pxi_at(slot, i, o, S)] = aband[o] * xi[o];
        part_dena[pden_at(slot, i, S)] = dena;
        part_denc[pden_at(slot, i, S)] = denc;
    }
}

// DENSE = false: A is known (on the host, ghmm_model_set) to be band-diagonal with a_ij = 0
// unless j = i or i + 1, only that form is compiled (half the registers, twice the waves in
// flight); DENSE = true decides per wave on the device like the one-pass kernels.
template <int L, bool WANT_BETA, bool DENSE>
__global__ void __launch_bounds__(WAVE, DENSE ? 2 : 4)
k_combine(int N, int U, int delta, const double *__restrict__ A, const long long *__restrict__ off,
          const double *__restrict__ alpha, const double *__restrict__ scale,
          const double *__restrict__ wrow, const double *__restrict__ sb, double *__restrict__ beta,
          double *__restrict__ gamma, double *__restrict__ part_xi, double *__restrict__ part_dena,
          double *__restrict__ part_denc, double *__restrict__ sink,
          const double *__restrict__ lognorm, double *__restrict__ lpart, double *__restrict__ logk,
          const int *__restrict__ order)
{
    const int qs = blockIdx.x * (WAVE / L) + threadIdx.x / L;
    const int i = threadIdx.x % L;
    const int k = qs % CB_CH;
    if (qs / CB_CH >= U) return;
    const int u = order[qs / CB_CH]; // longest utterances first
    const int q = u * CB_CH + k;     // partial-sum slot of (utterance, chunk)
    const long long f0 = off[u];
    const int T = (int)(off[u + 1] - f0);
    const bool act = i < N;
    const int tlo = (int)((long long)T * k / CB_CH), thi = (int)((long long)T * (k + 1) / CB_CH);
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
            if (k == CB_CH - 1) logk[u] = T > 0 ? log(alpha[(f0 + T - 1) * N + (N - 1)]) : 0.0;
        }
    }
    if (T <= 0 || thi <= tlo) {
        if (act) {
            for (int o = 0; o <= MAX_DELTA; o++) part_xi[pxi_at(q, i, o, U * CB_CH)] = 0.0;
            part_dena[pden_at(q, i, U * CB_CH)] = 0.0;
            part_denc[pden_at(q, i, U * CB_CH)] = 0.0;
        }
        return;
    }
    double *snk = wave_sink(sink);
    bool banded = true;
    if (DENSE) {
        bool offband = false;
        for (int j = 0; j < N; j++)
            offband |= act && (A[i * N + j] != 0.0 && j != i && j != i + 1);
        banded = !__any(offband);
    }
    if (banded)
        combine_run<L, true, WANT_BETA, DENSE ? MAX_DELTA : 1>(N, T, delta, i, act, q, tlo, thi, A, alpha + f0 * N, scale + f0,
                                        wrow + f0 * N, sb + f0, beta + f0 * N, gamma + f0 * N, part_xi,
                                        part_dena, part_denc, snk, U * CB_CH);
    else if (DENSE)
        combine_run<L, false, WANT_BETA>(N, T, delta, i, act, q, tlo, thi, A, alpha + f0 * N, scale + f0,
                                         wrow + f0 * N, sb + f0, beta + f0 * N, gamma + f0 * N, part_xi,
                                         part_dena, part_denc, snk, U * CB_CH);
}

} // namespace ghmm

namespace ghmm {
// log P per utterance from the pieces k_combine left (only when somebody asks for the vector)
__global__ void __launch_bounds__(256)
k_loglik_assemble(int U, const double *__restrict__ lpart, const double *__restrict__ logk,
                  double *__restrict__ loglik)
{
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= U) return;
    double lp = 0.0;
    for (int k = 0; k < CB_CH; k++) lp += lpart[(size_t)u * CB_CH + k];
    loglik[u] = lp + logk[u];
}
} // namespace ghmm
