// ghmm_hip.hip — the C ABI of include/ghmm.h on top of the gfx950 kernels.
//
// Host logic only: contexts, device buffers, launch geometry, HIP-event timing.
// All arithmetic of the path happens in the kernels (ghmm_kernels.hpp,
// ghmm_mfma.hpp, ghmm_pair.hpp).  No CPU fallback: without a gfx950 device every entry point that
// needs one fails with GHMM_ERR_NODEVICE.
#include "ghmm.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include <chrono>
#include <cstring>
#include "ghmm_kernels.hpp"
#include "ghmm_mfma.hpp"
#include "ghmm_pair.hpp"
#include "ghmm_wide.hpp"

extern "C" void ghmm_set_error(const char *fmt, ...); // ghmm_io.c

using namespace ghmm;

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            ghmm_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                     \
            return GHMM_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

#define ARG_CHECK(cond, msg)                                                              \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            ghmm_set_error("%s: %s", __func__, msg);                                      \
            return GHMM_ERR_ARG;                                                          \
        }                                                                                 \
    } while (0)

struct ktimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    double ms = 0.0;
    int64_t launches = 0;
};

struct ghmm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cus = 256, dev_cus = 256; // grid sizing (GHMM_OPT_CUS) / the device's count
    int64_t delta = 1, robust = 0, kernels = 0, timing = 0, partials = 0, vec_stats = 0, nt_post = 0, fused_scan = 0;
    // workspace (grown on demand, never shrunk)
    size_t cap_b = 0, cap_post = 0, cap_alpha = 0, cap_beta = 0, cap_gamma = 0, cap_scale = 0,
           cap_lognorm = 0, cap_loglik = 0, cap_pxi = 0, cap_pdena = 0, cap_pdenc = 0, cap_pmu = 0,
           cap_pvar = 0, cap_psi = 0, cap_path = 0, cap_pm = 0, cap_sinv = 0, cap_sink = 0;
    double *b = nullptr, *post = nullptr, *alpha = nullptr, *beta = nullptr, *gamma = nullptr;
    double *scale = nullptr, *sinv = nullptr, *lognorm = nullptr, *loglik = nullptr;
    double *sink = nullptr; // [0,64): idle lanes' stores land here; [64,128): zeros they read
    double *part_xi = nullptr, *part_dena = nullptr, *part_denc = nullptr;
    // paired scans (ghmm_pair.hpp): W rows and 1/s_t of the backward pass with its own normaliser
    // pinned bounce buffers for large device-to-host copies into pageable memory
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    double *wrow = nullptr, *sb = nullptr, *lpart = nullptr, *logk = nullptr;
    size_t cap_wrow = 0, cap_sb = 0, cap_lpart = 0, cap_logk = 0;
    // utterances k_combine hands to k_backward_fix (ghmm_pair.hpp, RANGE): per-utterance marks
    // stamped with the pass number, the list, and two counters used alternately (the fix-up
    // kernel of pass n zeroes the counter of pass n + 1)
    int *fix_mark = nullptr, *fix_list = nullptr, *fix_cnt = nullptr;
    int *wide_flag = nullptr; // models of more than 64 states: A (or log A) has entries off the band
    // ghmm_score_batch: the concatenated vocabulary model and the pass's tables, kept between calls
    ghmm_model *bt_cat = nullptr;
    char *bt_tab = nullptr;
    double *bt_scale = nullptr, *bt_sinv = nullptr, *bt_ll = nullptr;
    size_t cap_bt_tab = 0, cap_bt_scale = 0, cap_bt_sinv = 0, cap_bt_ll = 0;
    size_t cap_fix_mark = 0, cap_fix_list = 0;
    int fix_stamp = 0;
    long long launch_mark = 0, sync_mark = 0; // preparations enqueued / covered by a completed wait (stream_sync)
    // one page of fine-grained pinned host memory, 64 slots of 64 bytes: the models' host-visible
    // statistics flags (ghmm_model::hflag_host).  One allocation per context: hipHostMalloc takes
    // milliseconds, a training job creates its model inside the time a user waits for.
    int *hflag_page = nullptr, *hflag_page_dev = nullptr;
    unsigned long long hflag_used = 0;
    // a second such page: 32-byte mailboxes (log P, utterances, sequence) of the context's statistics
    // vectors, written by k_reduce_all, polled by ghmm_stats_loglik
    long long *mbox_page = nullptr, *mbox_page_dev = nullptr;
    unsigned long long mbox_used = 0;
    long long mbox_seq = 0;
    bool loglik_pieces = false; // log P of the last E-step is in lpart / logk, loglik[] not assembled
    int lp_nch = 0;             // chunks per utterance of those pieces
    bool own_bwd_done = false; // k_scan_pair ran the backward direction for the current alpha
    bool beta_valid = false;   // ctx->beta holds the reference's beta^
    // what the last gamma / xi pass ran on, so that ghmm_fetch(GHMM_BUF_BETA) can form beta^ when
    // ghmm_estep skipped it; cleared when that model changes or either object goes away
    ghmm_model *last_m = nullptr;
    ghmm_corpus *last_c = nullptr;
    int slots = 0;             // partial-sum slots filled by the last backward / combine pass
    double *part_mu = nullptr, *part_var = nullptr, *part_m = nullptr;
    // several feature streams: the stream being evaluated (b^p) and every stream's posteriors
    double *b_stream = nullptr;
    double *b_alloc = nullptr; // b sits B_PAD_FRAMES rows inside this allocation (grow_b)
    unsigned *smask = nullptr; // states occupied in each 16-frame stage (k_stage_masks)
    size_t cap_smask = 0;
    size_t b_pad = 0;
    size_t cap_b_stream = 0;
    std::vector<double *> post_s;
    std::vector<size_t> cap_post_s;
    unsigned char *psi = nullptr;
    unsigned char *path = nullptr; // Viterbi state per frame (N <= 255): one byte on the device
    // shape of what the workspace currently holds (for ghmm_fetch)
    long long F = 0;
    int U = 0, N = 0, G = 0;
    bool b_is_log = false;
    // what the emission densities in the workspace belong to (ghmm_forward / ghmm_backward /
    // ghmm_accumulate check it): model, its preparation count, corpus
    const ghmm_model *em_m = nullptr;
    const ghmm_corpus *em_c = nullptr;
    int em_epoch = -1;
    // kernels whose dynamic-LDS limit has been raised on THIS context's device (the attribute
    // is per device and per function; nothing process-wide, nothing shared between threads)
    std::vector<const void *> lds_fns;
    ktimer kt[GHMM_K_COUNT];
};

// every wait for the context's stream goes through here: what was enqueued before it has
// completed, which run_accumulate uses to know whether the host's view of a model's statistics
// flag (hflag_host) is up to date
static hipError_t stream_sync(ghmm_ctx *ctx)
{
    const long long mark = ctx->launch_mark;
    const hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) ctx->sync_mark = mark;
    return e;
}

struct ghmm_model {
    int N = 0, M = 0, D = 0;
    double *A = nullptr, *c = nullptr, *mean = nullptr, *inv_var = nullptr, *det = nullptr;
    double *wk = nullptr, *logwk = nullptr, *logA = nullptr;
    // matrix-core tier (ghmm_mfma.hpp): padded geometry and B fragments
    int Mp = 0, NT = 0, DP = 0, TC = 0, tps = 1;
    int TCs[3] = {0, 0, 0}; // tiles per chunk of k_emission_sched, per output mode
    size_t em_lds = 0;
    bool mfma_ok = false;
    double *Wm = nullptr, *offs = nullptr, *wkp = nullptr, *logwkp = nullptr, *condp = nullptr;
    double *oglob = nullptr, *condg = nullptr;
    int *gmap = nullptr, *anyflag = nullptr;
    // per-tile offsets of the expanded form (ghmm_mfma.hpp): otile[NT][DP] the offset itself,
    // dtile = otile - oglob, tshift[NT] = the tile has its own offset; condt = conditioning
    // relative to the tile's offset; sflag: like anyflag, for the statistics (global offset)
    double *otile = nullptr, *dtile = nullptr, *condt = nullptr;
    int *tshift = nullptr, *sflag = nullptr, *tnext = nullptr; // tnext: the choice for the next preparation
    int *scls = nullptr;  // [NT*16] how each padded Gaussian's statistics are taken (stats_class)
    int *tfull = nullptr; // [NT] the tile's slots are 16 consecutive real Gaussians, even start, G even
    // sflag's value as the HOST sees it, some launches late (pinned, mapped memory that the
    // preparing kernels write directly): the last preparation that found a class-2 Gaussian.
    // run_accumulate launches the vector-ALU k_mixstats only while that is recent (or the model
    // has just been set from the host); whenever the launch is left out and a class-2 Gaussian
    // is there after all, k_reduce_all recomputes it exactly itself — the flag only ever chooses
    // between two exact paths, so a stale value costs time, never correctness.
    int *hflag_host = nullptr, *hflag_dev = nullptr;
    int hflag_slot = -1;
    int vec_until = 0; // epoch up to which k_mixstats is launched unconditionally (after ghmm_model_set)
    long long prep_mark = 0; // ctx->launch_mark when this model's current preparation was enqueued
    bool banded = false; // A as last set from the host has a_ij = 0 unless j = i or i + 1
    int epoch = 0; // preparation count; anyflag[0] == epoch: this model holds an ill-conditioned Gaussian
    int NE = 0, CT = 0; // statistics kernel: feature tiles, Gaussian tiles per wave
};

struct ghmm_corpus {
    const double *X = nullptr;
    bool own = false;
    long long *off = nullptr; // device, U+1
    int *order = nullptr;     // device, U: utterance indices by decreasing length (stable)
    std::vector<int32_t> len;
    long long F = 0;
    int U = 0, D = 0, Tmax = 0;
};

struct ghmm_stats {
    int N = 0, M = 0, D = 0;
    double *v = nullptr;
    bool own = false;
    size_t n = 0;
    // mailbox of (log P, utterances) in the context's pinned page (own vectors only: a wrapped one
    // can change behind the library's back); valid while nothing has rewritten v since k_reduce_all
    int mbox_slot = -1;
    long long mbox_expect = 0;
    long long mbox_mark = 0; // ctx->launch_mark when the k_reduce_all that fills it was enqueued
    bool mbox_valid = false;
};

static const char *k_names[GHMM_K_COUNT] = {"emission", "forward", "backward", "mixstats",
                                            "reduce",   "mstep",   "viterbi",  "prepare"};

extern "C" const char *ghmm_kernel_name(int k)
{
    return (k >= 0 && k < GHMM_K_COUNT) ? k_names[k] : "?";
}

// ----------------------------------------------------------------- helpers

template <class T> static int dev_grow(T **p, size_t *cap, size_t need)
{
    if (need <= *cap && *p) return GHMM_OK;
    if (*p) HIP_TRY(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    size_t n = need ? need : 1;
    HIP_TRY(hipMalloc((void **)p, n * sizeof(T)));
    *cap = n;
    return GHMM_OK;
}

template <class T> static int dev_alloc(T **p, size_t n)
{
    HIP_TRY(hipMalloc((void **)p, (n ? n : 1) * sizeof(T)));
    return GHMM_OK;
}

struct kscope { // brackets one kernel launch with HIP events when timing is on
    ghmm_ctx *ctx;
    int k;
    hipEvent_t a = nullptr, b = nullptr;
    kscope(ghmm_ctx *c, int kid) : ctx(c), k(kid)
    {
        if (!ctx->timing) return;
        ktimer &t = ctx->kt[k];
        if (!t.pool.empty()) {
            a = t.pool.back().first;
            b = t.pool.back().second;
            t.pool.pop_back();
        } else {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
                a = b = nullptr;
                return;
            }
        }
        (void)hipEventRecord(a, ctx->stream);
    }
    ~kscope()
    {
        if (!a) return;
        (void)hipEventRecord(b, ctx->stream);
        ctx->kt[k].pending.push_back({a, b});
    }
};

// Device -> pageable host memory.  A plain hipMemcpy stages through small internal buffers
// (measured 1.4 GB/s on 12 MB of Viterbi paths); here the DMA goes into two pinned 16 MB
// buffers in turn while the host copies the previous one out.  Synchronises the stream.
constexpr size_t PIN_CHUNK = 16u << 20;
// copy-out of one landed chunk: plain bytes, or bytes widened to int32 (Viterbi paths travel
// as one byte per frame and are widened into the caller's int32 array)
static inline void chunk_out(void *dst, size_t dst_off, const void *pin, size_t n, bool widen)
{
    if (!widen) {
        memcpy((char *)dst + dst_off, pin, n);
        return;
    }
    int32_t *o = (int32_t *)dst + dst_off;
    const unsigned char *in = (const unsigned char *)pin;
    for (size_t k = 0; k < n; k++) o[k] = in[k];
}

static int d2h_pageable(ghmm_ctx *ctx, void *dst, const void *src, size_t bytes, bool widen = false)
{
    if (bytes < (1u << 20) && !widen) {
        if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(stream_sync(ctx));
        return GHMM_OK;
    }
    if (bytes == 0) {
        HIP_TRY(stream_sync(ctx));
        return GHMM_OK;
    }
    for (int k = 0; k < 2; k++)
        if (!ctx->pin[k]) {
            HIP_TRY(hipHostMalloc(&ctx->pin[k], PIN_CHUNK, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&ctx->pin_ev[k], hipEventDisableTiming));
        }
    size_t off = 0, prev_off = 0, prev_n = 0;
    int k = 0;
    // (widening: 4 MB pieces, so that the host loop over one piece runs under the next DMA)
    const size_t piece = widen ? (PIN_CHUNK / 4 < bytes ? PIN_CHUNK / 4 : bytes) : PIN_CHUNK;
    while (off < bytes) {
        const size_t n = bytes - off < piece ? bytes - off : piece;
        const int cur = k & 1;
        HIP_TRY(hipMemcpyAsync(ctx->pin[cur], (const char *)src + off, n, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipEventRecord(ctx->pin_ev[cur], ctx->stream));
        if (prev_n) {
            HIP_TRY(hipEventSynchronize(ctx->pin_ev[cur ^ 1]));
            chunk_out(dst, prev_off, ctx->pin[cur ^ 1], prev_n, widen);
        }
        prev_off = off;
        prev_n = n;
        off += n;
        k++;
    }
    HIP_TRY(hipEventSynchronize(ctx->pin_ev[(k - 1) & 1]));
    chunk_out(dst, prev_off, ctx->pin[(k - 1) & 1], prev_n, widen);
    HIP_TRY(stream_sync(ctx));
    return GHMM_OK;
}

// Raise a kernel's dynamic shared-memory limit to 150 KB on the context's device, once per
// (context, kernel).  Per-context state: two contexts on two devices each set it for their
// own device, two host threads never share a flag.
static int lds_attr(ghmm_ctx *ctx, const void *fn)
{
    for (const void *p : ctx->lds_fns)
        if (p == fn) return GHMM_OK;
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    ctx->lds_fns.push_back(fn);
    return GHMM_OK;
}

static int launch_ok(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ghmm_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return GHMM_ERR_HIP;
    }
    return GHMM_OK;
}

static int use(ghmm_ctx *ctx)
{
    if (!ctx) {
        ghmm_set_error("null context");
        return GHMM_ERR_ARG;
    }
    HIP_TRY(hipSetDevice(ctx->device));
    return GHMM_OK;
}

// ----------------------------------------------------------------- context

extern "C" int ghmm_ctx_create(int device, void *hip_stream, ghmm_ctx **out)
{
    ARG_CHECK(out, "null output");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        ghmm_set_error("no HIP device visible; the GMM-HMM path has no CPU fallback");
        return GHMM_ERR_NODEVICE;
    }
    ARG_CHECK(device >= 0 && device < n, "device index out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        ghmm_set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device,
                       prop.gcnArchName);
        return GHMM_ERR_NODEVICE;
    }
    HIP_TRY(hipSetDevice(device));
    ghmm_ctx *ctx = new (std::nothrow) ghmm_ctx();
    if (!ctx) return GHMM_ERR_ALLOC;
    ctx->device = device;
    ctx->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->dev_cus = ctx->cus;
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            ghmm_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            return GHMM_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    *out = ctx;
    return GHMM_OK;
}

extern "C" void ghmm_ctx_destroy(ghmm_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->bt_cat) {
        ghmm_model_destroy(ctx, ctx->bt_cat);
        ctx->bt_cat = nullptr;
    }
    void *bufs[] = {ctx->bt_tab, ctx->bt_scale, ctx->bt_sinv, ctx->bt_ll, ctx->b_alloc, ctx->post,      ctx->alpha,     ctx->beta,    ctx->gamma,
                    ctx->scale,   ctx->lognorm,   ctx->loglik,    ctx->part_xi, ctx->part_dena,
                    ctx->part_denc, ctx->part_mu, ctx->part_var,  ctx->psi,     ctx->path,
                    ctx->part_m,  ctx->sinv,      ctx->sink,      ctx->wrow,    ctx->sb,
                    ctx->lpart,   ctx->logk,      ctx->b_stream,  ctx->smask,
                    ctx->fix_mark, ctx->fix_list, ctx->fix_cnt, ctx->wide_flag};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    for (double *p : ctx->post_s)
        if (p) (void)hipFree(p);
    if (ctx->hflag_page) (void)hipHostFree(ctx->hflag_page);
    if (ctx->mbox_page) (void)hipHostFree(ctx->mbox_page);
    for (int k = 0; k < 2; k++) {
        if (ctx->pin[k]) (void)hipHostFree(ctx->pin[k]);
        if (ctx->pin_ev[k]) (void)hipEventDestroy(ctx->pin_ev[k]);
    }
    for (auto &t : ctx->kt) {
        for (auto &e : t.pending) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        for (auto &e : t.pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int ghmm_ctx_sync(ghmm_ctx *ctx)
{
    int rc = use(ctx);
    if (rc) return rc;
    HIP_TRY(stream_sync(ctx));
    return GHMM_OK;
}

extern "C" int ghmm_ctx_set_option(ghmm_ctx *ctx, int option, int64_t value)
{
    ARG_CHECK(ctx, "null context");
    switch (option) {
    case GHMM_OPT_DELTA:
        ARG_CHECK(value >= 0 && value <= MAX_DELTA, "delta out of range (0..7)");
        ctx->delta = value;
        break;
    case GHMM_OPT_ROBUST:
        ARG_CHECK(value == 0 || value == 1, "robust must be 0 or 1");
        ctx->robust = value;
        break;
    case GHMM_OPT_KERNELS:
        ARG_CHECK(value >= 0 && value <= 3, "kernels must be 0..3");
        ctx->kernels = value;
        break;
    case GHMM_OPT_TIMING:
        ctx->timing = value ? 1 : 0;
        break;
    case GHMM_OPT_PARTIALS:
        ARG_CHECK(value >= 0 && value <= 65535, "partials out of range");
        ctx->partials = value;
        break;
    case GHMM_OPT_VEC_STATS:
        ARG_CHECK(value >= 0 && value <= 2, "vec_stats must be 0, 1 or 2");
        ctx->vec_stats = value;
        break;
    case GHMM_OPT_FUSED_SCAN:
        ARG_CHECK(value >= 0 && value <= 2, "fused_scan must be 0, 1 or 2");
        ctx->fused_scan = value;
        break;
    case GHMM_OPT_NT_POST:
        ARG_CHECK(value >= 0 && value <= 2, "nt_post must be 0, 1 or 2");
        ctx->nt_post = value;
        break;
    case GHMM_OPT_CUS:
        ARG_CHECK(value >= 0 && value <= ctx->dev_cus, "compute units out of range");
        ctx->cus = value ? (int)value : ctx->dev_cus;
        break;
    default:
        ghmm_set_error("unknown option %d", option);
        return GHMM_ERR_ARG;
    }
    return GHMM_OK;
}

extern "C" int ghmm_ctx_get_option(ghmm_ctx *ctx, int option, int64_t *value)
{
    ARG_CHECK(ctx && value, "null argument");
    switch (option) {
    case GHMM_OPT_DELTA: *value = ctx->delta; break;
    case GHMM_OPT_ROBUST: *value = ctx->robust; break;
    case GHMM_OPT_KERNELS: *value = ctx->kernels; break;
    case GHMM_OPT_TIMING: *value = ctx->timing; break;
    case GHMM_OPT_PARTIALS: *value = ctx->partials; break;
    case GHMM_OPT_CUS: *value = ctx->cus; break;
    case GHMM_OPT_VEC_STATS: *value = ctx->vec_stats; break;
    case GHMM_OPT_NT_POST: *value = ctx->nt_post; break;
    case GHMM_OPT_FUSED_SCAN: *value = ctx->fused_scan; break;
    case GHMM_OPT_REFORDER_COUNT: {
        int n = 0, rc = use(ctx);
        if (rc) return rc;
        if (ctx->fix_cnt && ctx->fix_stamp > 0) {
            HIP_TRY(hipMemcpyAsync(&n, ctx->fix_cnt + (ctx->fix_stamp & 1), sizeof n, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(stream_sync(ctx));
        }
        *value = n;
        break;
    }
    default:
        ghmm_set_error("unknown option %d", option);
        return GHMM_ERR_ARG;
    }
    return GHMM_OK;
}

extern "C" int ghmm_ctx_kernel_time(ghmm_ctx *ctx, int kernel, double *total_ms, int64_t *launches)
{
    int rc = use(ctx);
    if (rc) return rc;
    ARG_CHECK(kernel >= 0 && kernel < GHMM_K_COUNT, "kernel id out of range");
    HIP_TRY(stream_sync(ctx));
    ktimer &t = ctx->kt[kernel];
    for (auto &e : t.pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
        t.ms += ms;
        t.launches++;
        t.pool.push_back(e);
    }
    t.pending.clear();
    if (total_ms) *total_ms = t.ms;
    if (launches) *launches = t.launches;
    return GHMM_OK;
}

extern "C" int ghmm_ctx_kernel_time_reset(ghmm_ctx *ctx)
{
    int rc = use(ctx);
    if (rc) return rc;
    HIP_TRY(stream_sync(ctx));
    for (auto &t : ctx->kt) {
        for (auto &e : t.pending) t.pool.push_back(e);
        t.pending.clear();
        t.ms = 0.0;
        t.launches = 0;
    }
    return GHMM_OK;
}

// ------------------------------------------------------------------- model

// preparations for which the vector-ALU statistics kernel stays launched after the host has
// last seen (or could not yet have seen) a class-2 Gaussian
constexpr int VEC_WINDOW = 64;

static int model_prepare(ghmm_ctx *ctx, ghmm_model *m, bool base)
{
    // pow(2*pi, D/2): the reference's aux1 (TF:1821-1823), evaluated by the host libm
    const double norm2pi = pow(2.0 * M_PI, m->D / 2.0);
    const int G = m->N * m->M;
    int rc;
    if (base) { // after ghmm_model_set; k_mstep derives the same constants itself
        int work = G > m->N * m->N ? G : m->N * m->N;
        int blocks = (work + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        kscope ks(ctx, GHMM_K_PREPARE);
        hipLaunchKernelGGL(k_prepare, dim3(blocks), dim3(256), 0, ctx->stream, m->N, m->M, m->A,
                           m->c, m->det, norm2pi, m->wk, m->logwk, m->logA);
    }
    if ((rc = launch_ok("k_prepare")) || !m->mfma_ok) return rc;
    {
        kscope ks(ctx, GHMM_K_PREPARE);
        hipLaunchKernelGGL(k_prepare_offsets, dim3((unsigned)(m->NT + m->D)), dim3(64), 0, ctx->stream,
                           m->N, m->M, m->D, m->Mp, m->NT, m->DP, m->mean, m->offs, m->oglob);
        hipLaunchKernelGGL(k_prepare_tiles, dim3((unsigned)m->NT), dim3(64), 0, ctx->stream, m->N, m->M,
                           m->D, m->Mp, m->NT, m->DP, m->mean, m->inv_var, m->oglob, m->otile, m->tnext);
        hipLaunchKernelGGL(k_prepare_mfma, dim3((unsigned)(m->NT * 16)), dim3(64), 0,
                           ctx->stream, m->N, m->M, m->D, m->Mp, m->NT, m->DP, m->mean, m->inv_var,
                           m->wk, m->logwk, m->otile, m->tnext, m->oglob, m->Wm, m->wkp, m->logwkp,
                           m->gmap, m->condt, m->condg, m->anyflag, m->sflag, m->epoch, m->dtile,
                           m->tshift, m->scls, m->hflag_dev);
    }
    return launch_ok("k_prepare_mfma");
}

extern "C" int ghmm_model_create(ghmm_ctx *ctx, int N, int M, int D, ghmm_model **out)
{
    int rc = use(ctx);
    if (rc) return rc;
    ARG_CHECK(out && N > 0 && M > 0 && D > 0, "bad dimensions");
    ARG_CHECK((long long)N * M < (1ll << 30), "too many Gaussians");
    ghmm_model *m = new (std::nothrow) ghmm_model();
    if (!m) return GHMM_ERR_ALLOC;
    m->N = N; m->M = M; m->D = D;
    size_t G = (size_t)N * M;
    if ((rc = dev_alloc(&m->A, (size_t)N * N)) || (rc = dev_alloc(&m->c, G)) ||
        (rc = dev_alloc(&m->mean, G * D)) || (rc = dev_alloc(&m->inv_var, G * D)) ||
        (rc = dev_alloc(&m->det, G)) || (rc = dev_alloc(&m->wk, G)) ||
        (rc = dev_alloc(&m->logwk, G)) || (rc = dev_alloc(&m->logA, (size_t)N * N))) {
        ghmm_model_destroy(ctx, m);
        return rc;
    }
    // matrix-core geometry: mixtures padded to a power of two (<= 16) or a multiple of 16
    {
        int Mp = 1;
        if (M <= 16) while (Mp < M) Mp <<= 1;
        else Mp = (M + 15) / 16 * 16;
        m->Mp = Mp;
        m->tps = Mp > 16 ? Mp / 16 : 1;
        m->NT = (int)(((size_t)N * Mp + 15) / 16);
        m->DP = (D + 1 + 3) / 4 * 4;
        const size_t per_tile = (size_t)(m->DP / 2) * 64 * 8;
        const size_t slabs = (size_t)EM_WAVES * 16 * (2 * m->DP + 1) * 8;
        int tcmax = (int)((150 * 1024 - slabs) / per_tile);
        if (tcmax > 6) tcmax = 6;
        int TC = tcmax / m->tps * m->tps;
        if (TC > m->NT) TC = m->NT;
        m->TC = TC;
        // the scheduled kernel: as many whole states per chunk as fit beside its slabs (none in
        // the b-only variant) — every chunk reads the frames again
        for (int out = 0; out < 3; out++) {
            int tcs = ems_tc_cap(Mp, out) / m->tps * m->tps;
            while (tcs > m->tps && ems_lds_bytes(tcs, m->DP, ems_waves(Mp, out), Mp, out) > 159 * 1024) tcs -= m->tps;
            if (tcs < m->tps) tcs = m->tps;
            if (tcs > m->NT) tcs = m->NT;
            m->TCs[out] = tcs;
        }
        m->em_lds = (size_t)TC * per_tile + slabs;
        m->mfma_ok = TC >= m->tps && TC > 0 && m->em_lds <= 150 * 1024 && 16 * m->D <= 64 * EM_XR;
        if (m->mfma_ok) {
            size_t nw = (size_t)m->NT * (m->DP / 2) * 64;
            if ((rc = dev_alloc(&m->Wm, nw)) || (rc = dev_alloc(&m->offs, (size_t)m->NT * m->DP)) ||
                (rc = dev_alloc(&m->wkp, (size_t)m->NT * 16)) ||
                (rc = dev_alloc(&m->logwkp, (size_t)m->NT * 16)) ||
                (rc = dev_alloc(&m->condp, (size_t)m->NT * 16)) ||
                (rc = dev_alloc(&m->gmap, (size_t)m->NT * 16)) ||
                (rc = dev_alloc(&m->condg, (size_t)m->NT * 16)) ||
                (rc = dev_alloc(&m->oglob, (size_t)m->DP)) || (rc = dev_alloc(&m->anyflag, 1)) ||
                (rc = dev_alloc(&m->otile, (size_t)m->NT * m->DP)) || (rc = dev_alloc(&m->dtile, (size_t)m->NT * m->DP)) ||
                (rc = dev_alloc(&m->condt, (size_t)m->NT * 16)) || (rc = dev_alloc(&m->tshift, (size_t)m->NT)) ||
                (rc = dev_alloc(&m->sflag, 1)) || (rc = dev_alloc(&m->tnext, (size_t)m->NT)) ||
                (rc = dev_alloc(&m->tfull, (size_t)m->NT)) || (rc = dev_alloc(&m->scls, (size_t)m->NT * 16))) {
                ghmm_model_destroy(ctx, m);
                return rc;
            }
            // the host's late view of sflag (see ghmm_model): a 64-byte slot of the context's page of
            // pinned, mapped, fine-grained memory (a kernel's store is seen without a synchronisation)
            if (!ctx->hflag_page) {
                void *hp = nullptr, *dp = nullptr;
                if (hipHostMalloc(&hp, 4096, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
                    hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
                    if (hp) (void)hipHostFree(hp);
                    ghmm_set_error("ghmm_model_create: no pinned host memory for the statistics flags");
                    ghmm_model_destroy(ctx, m);
                    return GHMM_ERR_ALLOC;
                }
                ctx->hflag_page = (int *)hp;
                ctx->hflag_page_dev = (int *)dp;
            }
            {
                int slot = 0;
                while (slot < 64 && ((ctx->hflag_used >> slot) & 1ull)) slot++;
                if (slot < 64) {
                    ctx->hflag_used |= 1ull << slot;
                    m->hflag_slot = slot;
                    m->hflag_host = ctx->hflag_page + slot * 16;
                    m->hflag_dev = ctx->hflag_page_dev + slot * 16;
                    *(volatile int *)m->hflag_host = -(1 << 30); // no preparation has found a class-2 Gaussian
                } else {
                    // a vocabulary of more than 64 models on one context: this one goes without a
                    // host-visible flag (the vector-ALU statistics kernel is then always launched
                    // for it, as in round 2); the kernels' store lands on the device flag itself
                    m->hflag_host = nullptr;
                    m->hflag_dev = m->sflag;
                }
            }
            // on the context's stream, like every consumer of these buffers
            hipError_t e = hipMemsetAsync(m->dtile, 0, (size_t)m->NT * m->DP * 8, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(m->otile, 0, (size_t)m->NT * m->DP * 8, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(m->tshift, 0, (size_t)m->NT * sizeof(int), ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(m->condt, 0, (size_t)m->NT * 16 * 8, ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(m->sflag, 0, sizeof(int), ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(m->anyflag, 0, sizeof(int), ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(m->tnext, 0, (size_t)m->NT * sizeof(int), ctx->stream);
            if (e == hipSuccess) e = hipMemsetAsync(m->scls, 0, (size_t)m->NT * 16 * sizeof(int), ctx->stream);
            // which tiles hold 16 consecutive real Gaussians (a property of N, M and the padding)
            std::vector<int> tf((size_t)m->NT, 0);
            for (int t = 0; t < m->NT; t++) {
                const long long gp0 = 16ll * t, gp1 = gp0 + 15;
                tf[t] = (Mp == M && gp1 / Mp < N && (G % 2) == 0) ? 1 : 0; // no padding: g = gp, even start
            }
            if (e == hipSuccess)
                e = hipMemcpyAsync(m->tfull, tf.data(), tf.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream); // tf is a local
            if (e != hipSuccess) {
                ghmm_set_error("ghmm_model_create: hipMemsetAsync failed: %s", hipGetErrorString(e));
                ghmm_model_destroy(ctx, m);
                return GHMM_ERR_HIP;
            }
            // statistics kernel: NE feature tiles of 16 over [x', 1, x'^2]; CT Gaussian tiles
            // per wave so that CT*NE accumulator tiles (8 VGPRs each) stay near 200 VGPRs
            m->NE = (2 * m->DP + 15) / 16;
            static const int ct_of_ne[9] = {0, 8, 8, 8, 6, 5, 4, 3, 3};
            m->CT = m->NE <= 8 ? ct_of_ne[m->NE] : 0;
        }
    }
    *out = m;
    return GHMM_OK;
}

extern "C" void ghmm_model_destroy(ghmm_ctx *ctx, ghmm_model *m)
{
    if (!m) return;
    if (ctx && ctx->last_m == m) ctx->last_m = nullptr;
    if (ctx && ctx->em_m == m) ctx->em_m = nullptr;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    void *bufs[] = {m->A,  m->c,    m->mean, m->inv_var, m->det,   m->wk,    m->logwk, m->logA,
                    m->Wm, m->offs, m->wkp,  m->condp,   m->gmap,  m->oglob, m->condg, m->anyflag,
                    m->logwkp, m->otile, m->dtile, m->condt, m->tshift, m->sflag, m->tnext, m->tfull, m->scls};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (m->hflag_slot >= 0 && ctx) ctx->hflag_used &= ~(1ull << m->hflag_slot);
    delete m;
}

extern "C" int ghmm_model_set(ghmm_ctx *ctx, ghmm_model *m, const double *A, const double *c,
                              const double *mean, const double *inv_var, const double *det)
{
    int rc = use(ctx);
    if (ctx && ctx->last_m == m) ctx->last_m = nullptr; // alpha^ / W on the device belong to the old parameters
    if (rc) return rc;
    ARG_CHECK(m && A && c && mean && inv_var && det, "null argument");
    size_t G = (size_t)m->N * m->M, NN = (size_t)m->N * m->N;
    m->banded = true;
    for (int i = 0; i < m->N; i++)
        for (int j = 0; j < m->N; j++)
            if (A[(size_t)i * m->N + j] != 0.0 && j != i && j != i + 1) m->banded = false;
    HIP_TRY(hipMemcpyAsync(m->A, A, NN * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(m->c, c, G * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(m->mean, mean, G * m->D * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(m->inv_var, inv_var, G * m->D * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(m->det, det, G * 8, hipMemcpyHostToDevice, ctx->stream));
    // pageable host memory: the copies above have consumed the buffers on return
    HIP_TRY(stream_sync(ctx));
    m->epoch++; // a new set of parameters
    // the host cannot know yet what this model's statistics classes are: k_mixstats is launched
    // (and leaves at once where it has nothing to do) for the first preparations behind this one
    m->vec_until = m->epoch + VEC_WINDOW;
    m->prep_mark = ++ctx->launch_mark;
    return model_prepare(ctx, m, true);
}

extern "C" int ghmm_model_get(ghmm_ctx *ctx, ghmm_model *m, double *A, double *c, double *mean,
                              double *inv_var, double *det)
{
    int rc = use(ctx);
    if (rc) return rc;
    ARG_CHECK(m, "null model");
    size_t G = (size_t)m->N * m->M, NN = (size_t)m->N * m->N;
    if (A) HIP_TRY(hipMemcpyAsync(A, m->A, NN * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (c) HIP_TRY(hipMemcpyAsync(c, m->c, G * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (mean) HIP_TRY(hipMemcpyAsync(mean, m->mean, G * m->D * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (inv_var)
        HIP_TRY(hipMemcpyAsync(inv_var, m->inv_var, G * m->D * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (det) HIP_TRY(hipMemcpyAsync(det, m->det, G * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(stream_sync(ctx));
    return GHMM_OK;
}

extern "C" int ghmm_model_dims(const ghmm_model *m, int *N, int *M, int *D)
{
    ARG_CHECK(m, "null model");
    if (N) *N = m->N;
    if (M) *M = m->M;
    if (D) *D = m->D;
    return GHMM_OK;
}

// ------------------------------------------------------------------ corpus

static int corpus_make(ghmm_ctx *ctx, const double *X_host, const double *X_dev, const int32_t *len,
                       int n_utt, int D, ghmm_corpus **out)
{
    int rc = use(ctx);
    if (rc) return rc;
    ARG_CHECK(out && len && n_utt >= 0 && D > 0, "bad arguments");
    ghmm_corpus *c = new (std::nothrow) ghmm_corpus();
    if (!c) return GHMM_ERR_ALLOC;
    c->U = n_utt;
    c->D = D;
    c->len.assign(len, len + n_utt);
    std::vector<long long> off((size_t)n_utt + 1, 0);
    for (int u = 0; u < n_utt; u++) {
        if (len[u] < 0) {
            delete c;
            ghmm_set_error("negative utterance length");
            return GHMM_ERR_ARG;
        }
        off[u + 1] = off[u] + len[u];
        if (len[u] > c->Tmax) c->Tmax = len[u];
    }
    c->F = off[n_utt];
    // length order for the scans: a wave holds 4 utterances and runs as long as its longest,
    // and the longest chains should start first (stable: equal lengths keep corpus order)
    std::vector<int> ord((size_t)n_utt);
    for (int u = 0; u < n_utt; u++) ord[u] = u;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return len[a] > len[b]; });
    if ((rc = dev_alloc(&c->off, off.size())) || (rc = dev_alloc(&c->order, ord.size()))) {
        if (c->off) (void)hipFree(c->off);
        delete c;
        return rc;
    }
    hipError_t e = hipMemcpyAsync(c->off, off.data(), off.size() * sizeof(long long), hipMemcpyHostToDevice,
                                  ctx->stream);
    if (e == hipSuccess && n_utt)
        e = hipMemcpyAsync(c->order, ord.data(), ord.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && X_host) {
        double *xd = nullptr;
        if ((rc = dev_alloc(&xd, (size_t)c->F * D))) {
            (void)hipFree(c->off);
            (void)hipFree(c->order);
            delete c;
            return rc;
        }
        c->X = xd;
        c->own = true;
        if (c->F) e = hipMemcpyAsync(xd, X_host, (size_t)c->F * D * 8, hipMemcpyHostToDevice, ctx->stream);
    } else {
        c->X = X_dev;
    }
    // pageable host memory (and the local `off`): consumed when the stream has drained
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        ghmm_set_error("corpus upload failed: %s", hipGetErrorString(e));
        if (c->own) (void)hipFree((void *)c->X);
        (void)hipFree(c->off);
        (void)hipFree(c->order);
        delete c;
        return GHMM_ERR_HIP;
    }
    *out = c;
    return GHMM_OK;
}

extern "C" int ghmm_corpus_create(ghmm_ctx *ctx, const double *X_host, const int32_t *len, int n_utt,
                                  int D, ghmm_corpus **out)
{
    ARG_CHECK(X_host || n_utt == 0, "null frames");
    static const double dummy = 0.0;
    return corpus_make(ctx, X_host ? X_host : &dummy, nullptr, len, n_utt, D, out);
}

extern "C" int ghmm_corpus_wrap(ghmm_ctx *ctx, const double *X_dev, const int32_t *len, int n_utt,
                                int D, ghmm_corpus **out)
{
    ARG_CHECK(X_dev, "null device pointer");
    return corpus_make(ctx, nullptr, X_dev, len, n_utt, D, out);
}

extern "C" void ghmm_corpus_destroy(ghmm_ctx *ctx, ghmm_corpus *c)
{
    if (!c) return;
    if (ctx && ctx->last_c == c) ctx->last_c = nullptr;
    if (ctx && ctx->em_c == c) ctx->em_c = nullptr;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (c->own && c->X) (void)hipFree((void *)c->X);
    if (c->off) (void)hipFree(c->off);
    if (c->order) (void)hipFree(c->order);
    delete c;
}

extern "C" int64_t ghmm_corpus_frames(const ghmm_corpus *c) { return c ? c->F : -1; }
extern "C" int ghmm_corpus_utterances(const ghmm_corpus *c) { return c ? c->U : -1; }

// ------------------------------------------------------------------- stats

extern "C" size_t ghmm_stats_len(int N, int M, int D)
{
    return (size_t)N * N + 2 * (size_t)N + (size_t)N * M * (2 * (size_t)D + 1) + 2;
}

extern "C" int ghmm_stats_create(ghmm_ctx *ctx, int N, int M, int D, ghmm_stats **out)
{
    int rc = use(ctx);
    if (rc) return rc;
    ARG_CHECK(out && N > 0 && M > 0 && D > 0, "bad dimensions");
    ghmm_stats *s = new (std::nothrow) ghmm_stats();
    if (!s) return GHMM_ERR_ALLOC;
    s->N = N; s->M = M; s->D = D;
    s->n = ghmm_stats_len(N, M, D);
    s->own = true;
    if ((rc = dev_alloc(&s->v, s->n))) {
        delete s;
        return rc;
    }
    hipError_t e = hipMemsetAsync(s->v, 0, s->n * 8, ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(s->v);
        delete s;
        ghmm_set_error("hipMemsetAsync failed: %s", hipGetErrorString(e));
        return GHMM_ERR_HIP;
    }
    // a mailbox slot (64 per context; without one ghmm_stats_loglik copies and waits as before)
    if (!ctx->mbox_page) {
        void *hp = nullptr, *dp = nullptr;
        if (hipHostMalloc(&hp, 4096, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
            hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            memset(hp, 0, 4096);
            ctx->mbox_page = (long long *)hp;
            ctx->mbox_page_dev = (long long *)dp;
        } else if (hp) {
            (void)hipHostFree(hp);
        }
    }
    if (ctx->mbox_page) {
        int slot = 0;
        while (slot < 64 && ((ctx->mbox_used >> slot) & 1ull)) slot++;
        if (slot < 64) {
            ctx->mbox_used |= 1ull << slot;
            s->mbox_slot = slot;
        }
    }
    *out = s;
    return GHMM_OK;
}

extern "C" int ghmm_stats_wrap(ghmm_ctx *ctx, int N, int M, int D, double *dev_ptr, ghmm_stats **out)
{
    int rc = use(ctx);
    if (rc) return rc;
    ARG_CHECK(out && dev_ptr && N > 0 && M > 0 && D > 0, "bad arguments");
    ghmm_stats *s = new (std::nothrow) ghmm_stats();
    if (!s) return GHMM_ERR_ALLOC;
    s->N = N; s->M = M; s->D = D;
    s->n = ghmm_stats_len(N, M, D);
    s->v = dev_ptr;
    s->own = false;
    *out = s;
    return GHMM_OK;
}

extern "C" void ghmm_stats_destroy(ghmm_ctx *ctx, ghmm_stats *s)
{
    if (!s) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (s->own && s->v) (void)hipFree(s->v);
    if (ctx && s->mbox_slot >= 0) ctx->mbox_used &= ~(1ull << s->mbox_slot);
    delete s;
}

extern "C" double *ghmm_stats_device_ptr(ghmm_stats *s) { return s ? s->v : nullptr; }

extern "C" int ghmm_stats_download(ghmm_ctx *ctx, ghmm_stats *s, double *host)
{
    int rc = use(ctx);
    if (rc) return rc;
    ARG_CHECK(s && host, "null argument");
    HIP_TRY(hipMemcpyAsync(host, s->v, s->n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(stream_sync(ctx));
    return GHMM_OK;
}

extern "C" int ghmm_stats_loglik(ghmm_ctx *ctx, ghmm_stats *s, double out[2])
{
    int rc = use(ctx);
    if (rc) return rc;
    ARG_CHECK(s && out, "null argument");
    // loglik and n_utt are the last two doubles of the vector (layout in ghmm.h).  Straight behind an
    // E-step they are also in the vector's mailbox in pinned memory, a sequence number behind them:
    // polling that costs the kernel's own latency, a 16-byte copy and a stream wait ~0.1 ms more.
    if (s->mbox_valid && s->mbox_slot >= 0 && ctx->mbox_page) {
        volatile long long *mb = ctx->mbox_page + 4 * s->mbox_slot;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; spins++) {
            if (__atomic_load_n(mb + 2, __ATOMIC_ACQUIRE) == s->mbox_expect) {
                long long a = mb[0], b = mb[1];
                memcpy(&out[0], &a, 8);
                memcpy(&out[1], &b, 8);
                // everything enqueued before that k_reduce_all has completed: as good as a wait for
                // the host's view of the models' flags (run_accumulate's exact decision)
                if (s->mbox_mark > ctx->sync_mark) ctx->sync_mark = s->mbox_mark;
                return GHMM_OK;
            }
            if ((spins & 1023u) == 1023u &&
                std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200))
                break; // (a launch that failed, a device far behind: the copy below waits for it)
        }
    }
    HIP_TRY(hipMemcpyAsync(out, s->v + (s->n - 2), 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(stream_sync(ctx));
    return GHMM_OK;
}

extern "C" int ghmm_stats_upload(ghmm_ctx *ctx, ghmm_stats *s, const double *host)
{
    int rc = use(ctx);
    if (rc) return rc;
    ARG_CHECK(s && host, "null argument");
    s->mbox_valid = false;
    HIP_TRY(hipMemcpyAsync(s->v, host, s->n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(stream_sync(ctx));
    return GHMM_OK;
}

// --------------------------------------------------------------- workspace

static int check_pair(const ghmm_model *m, const ghmm_corpus *c)
{
    if (!m || !c) {
        ghmm_set_error("null model or corpus");
        return GHMM_ERR_ARG;
    }
    if (m->D != c->D) {
        ghmm_set_error("model has %d coefficients per frame, corpus has %d", m->D, c->D);
        return GHMM_ERR_ARG;
    }
    return GHMM_OK;
}

// b[F][N] with B_PAD_FRAMES rows of padding on either side: the scans' operand cursors run past
// the ends of an utterance without a clamp (ghmm_kernels.hpp forward_run); nothing read there is used
static int grow_b(ghmm_ctx *ctx, size_t F, size_t N)
{
    const size_t pad = (size_t)B_PAD_FRAMES * N, need = F * N;
    if (ctx->b_alloc && ctx->b_pad >= pad && ctx->b_pad + need + pad <= ctx->cap_b) return GHMM_OK;
    if (ctx->b_alloc) HIP_TRY(hipFree(ctx->b_alloc));
    ctx->b_alloc = ctx->b = nullptr;
    ctx->cap_b = ctx->b_pad = 0;
    HIP_TRY(hipMalloc((void **)&ctx->b_alloc, (need + 2 * pad + 1) * sizeof(double)));
    // (defined contents for the rows no frame owns)
    HIP_TRY(hipMemsetAsync(ctx->b_alloc, 0, pad * sizeof(double), ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->b_alloc + pad + need, 0, (pad + 1) * sizeof(double), ctx->stream));
    ctx->cap_b = need + 2 * pad + 1;
    ctx->b_pad = pad;
    ctx->b = ctx->b_alloc + pad;
    return GHMM_OK;
}

static int ws_frames(ghmm_ctx *ctx, const ghmm_model *m, const ghmm_corpus *c, bool want_post)
{
    int rc;
    size_t F = (size_t)c->F, N = (size_t)m->N, G = (size_t)m->N * m->M;
    if ((rc = grow_b(ctx, F, N))) return rc;
    if (want_post && (rc = dev_grow(&ctx->post, &ctx->cap_post, F * G))) return rc;
    if ((rc = dev_grow(&ctx->scale, &ctx->cap_scale, F))) return rc;
    if ((rc = dev_grow(&ctx->sinv, &ctx->cap_sinv, F))) return rc;
    if (!ctx->sink) {
        if ((rc = dev_grow(&ctx->sink, &ctx->cap_sink, (size_t)2 * WAVE * SINK_WAVES))) return rc;
        HIP_TRY(hipMemsetAsync(ctx->sink, 0, (size_t)2 * WAVE * SINK_WAVES * sizeof(double), ctx->stream));
    }
    if ((rc = dev_grow(&ctx->lognorm, &ctx->cap_lognorm, F))) return rc;
    if ((rc = dev_grow(&ctx->loglik, &ctx->cap_loglik, (size_t)c->U))) return rc;
    ctx->F = c->F;
    ctx->U = c->U;
    ctx->N = m->N;
    ctx->G = (int)G;
    return GHMM_OK;
}

static int ws_fb(ghmm_ctx *ctx, const ghmm_model *m, const ghmm_corpus *c)
{
    int rc;
    size_t FN = (size_t)c->F * m->N, UN = (size_t)c->U * m->N * CB_CH; // one slot per (utterance, chunk)
    if ((rc = dev_grow(&ctx->wrow, &ctx->cap_wrow, FN))) return rc;
    if ((rc = dev_grow(&ctx->sb, &ctx->cap_sb, (size_t)c->F))) return rc;
    if ((rc = dev_grow(&ctx->lpart, &ctx->cap_lpart, (size_t)c->U * CB_CH))) return rc;
    if ((rc = dev_grow(&ctx->logk, &ctx->cap_logk, (size_t)c->U))) return rc;
    if ((rc = dev_grow(&ctx->alpha, &ctx->cap_alpha, FN))) return rc;
    if ((rc = dev_grow(&ctx->beta, &ctx->cap_beta, FN))) return rc;
    if ((rc = dev_grow(&ctx->gamma, &ctx->cap_gamma, FN))) return rc;
    if ((rc = dev_grow(&ctx->part_xi, &ctx->cap_pxi, UN * (MAX_DELTA + 1)))) return rc;
    if ((rc = dev_grow(&ctx->part_dena, &ctx->cap_pdena, UN))) return rc;
    if ((rc = dev_grow(&ctx->part_denc, &ctx->cap_pdenc, UN))) return rc;
    if ((size_t)c->U > ctx->cap_fix_mark) {
        if ((rc = dev_grow(&ctx->fix_mark, &ctx->cap_fix_mark, (size_t)c->U))) return rc;
        HIP_TRY(hipMemsetAsync(ctx->fix_mark, 0, ctx->cap_fix_mark * sizeof(int), ctx->stream));
        ctx->fix_stamp = 0; // (stamps start at 1: a zeroed mark never equals one)
    }
    if ((rc = dev_grow(&ctx->fix_list, &ctx->cap_fix_list, (size_t)c->U))) return rc;
    if (!ctx->fix_cnt) {
        if ((rc = dev_alloc(&ctx->fix_cnt, 2))) return rc;
        HIP_TRY(hipMemsetAsync(ctx->fix_cnt, 0, 2 * sizeof(int), ctx->stream));
    }
    return GHMM_OK;
}

// --------------------------------------------------------------- emission

static bool nt_posteriors(const ghmm_ctx *ctx, const ghmm_model *m, const ghmm_corpus *c)
{
    // posteriors are written once and mostly never read (gamma is 0 for most states of a frame):
    // non-temporal stores (emission) and loads (statistics) while an iteration's OTHER buffers can
    // stay in the 256 MB Infinity Cache thanks to it (10x8, 192 MB of posteriors: 0.245 -> 0.230 ms
    // per iteration); with 19 GB of them (64 mixtures) nothing stays anyway and the hint costs
    // 0.4 %.  GHMM_OPT_NT_POST 1 / 2 force it on / off (profiles/tools/nt_ab.py).
    return ctx->nt_post == 1 ||
           (ctx->nt_post == 0 && (double)c->F * m->N * m->M * 8.0 <= 1024.0 * 1048576.0);
}

static int run_emission(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, int mode, bool want_post)
{
    // the workspace is rewritten: alpha^ / W / 1/s of an earlier E-step no longer go with its b
    // (ghmm_fetch(GHMM_BUF_BETA) refuses instead of rebuilding beta^ from mixed buffers)
    ctx->last_m = nullptr;
    ctx->last_c = nullptr;
    ctx->em_m = m;
    ctx->em_c = c;
    ctx->em_epoch = m->epoch;
    if (c->F == 0) return GHMM_OK;
    const long long blocks = (c->F + WAVE - 1) / WAVE;
    const size_t lds = (size_t)WAVE * (m->D | 1) * sizeof(double);
    if (lds > 160 * 1024) {
        ghmm_set_error("coefficient count %d too large for the emission tile", m->D);
        return GHMM_ERR_UNSUPPORTED;
    }
    double *post = want_post ? ctx->post : nullptr;
    const bool sched_ok = m->mfma_ok && ctx->kernels != 1 &&
                          (m->Mp <= 16 || m->Mp == 32 || m->Mp == 64) &&
                          m->DP >= 8 && m->DP <= 48;
    if ((mode == 2 || mode == 0) && sched_ok) {
        // The scheduled matrix-core kernel: compile-time K steps, KS = DP / 2 for D = 4 .. 47 (DP = D + 1
        // rounded up to a multiple of four, so D always reaches into the last group of four columns,
        // which its direct-operand variant reads unclamped up to there) — and mixture padding.  It needs no fallback launch: a Gaussian that is ill-conditioned even
        // around its tile's offset is re-evaluated in direct form inside the kernel, for the few
        // frames where its density is not 0.
        int rc;
        const long long ntf = (c->F + 15) / 16;
        const int po = mode == 2 ? 2 : (post ? 1 : 0);
        const int tcs = m->TCs[po];
        const int chunks = (m->NT + tcs - 1) / tcs;
        const int wv = ems_waves(m->Mp, po);
        const size_t lds_s = ems_lds_bytes(tcs, m->DP, wv, m->Mp, po);
        long long gxs = (ntf + wv - 1) / wv;
        if (gxs > ctx->cus) gxs = ctx->cus;
        const double *wk = mode == 2 ? m->logwkp : m->wkp; // OUT = 2 adds log wk to the exponents
        const int ntp = nt_posteriors(ctx, m, c) ? 1 : 0; // non-temporal posterior stores
        kscope ks(ctx, GHMM_K_EMISSION);
#define GHMM_EMSK(KSV, MP, PO)                                                                    \
    do {                                                                                          \
        if ((rc = lds_attr(ctx, (const void *)k_emission_sched<KSV, MP, PO>))) return rc;        \
        hipLaunchKernelGGL((k_emission_sched<KSV, MP, PO>), dim3((unsigned)gxs, (unsigned)chunks), \
                           dim3((unsigned)(wv * WAVE)), lds_s, ctx->stream, m->N, m->M, m->D, m->NT, \
                           tcs, c->F, c->X, m->Wm, m->oglob, wk, m->gmap, ctx->b, post, m->dtile, \
                           m->tshift, m->tfull, m->condt, m->mean, m->inv_var, ntp);              \
    } while (0)
#define GHMM_EMS(MP, PO)                                                                          \
    do {                                                                                          \
        switch (m->DP) {                                                                          \
        case 8: GHMM_EMSK(4, MP, PO); break;                                                      \
        case 12: GHMM_EMSK(6, MP, PO); break;                                                     \
        case 16: GHMM_EMSK(8, MP, PO); break;                                                     \
        case 20: GHMM_EMSK(10, MP, PO); break;                                                    \
        case 24: GHMM_EMSK(12, MP, PO); break;                                                    \
        case 28: GHMM_EMSK(14, MP, PO); break;                                                    \
        case 32: GHMM_EMSK(16, MP, PO); break;                                                    \
        case 36: GHMM_EMSK(18, MP, PO); break;                                                    \
        case 40: GHMM_EMSK(20, MP, PO); break;                                                    \
        case 44: GHMM_EMSK(22, MP, PO); break;                                                    \
        default: GHMM_EMSK(24, MP, PO); break;                                                    \
        }                                                                                         \
    } while (0)
#define GHMM_EMS3(MP)                                                                             \
    do {                                                                                          \
        if (po == 2) GHMM_EMS(MP, 2);                                                             \
        else if (po == 1) GHMM_EMS(MP, 1);                                                        \
        else GHMM_EMS(MP, 0);                                                                     \
    } while (0)
        switch (m->Mp) {
        case 1: GHMM_EMS3(1); break;
        case 2: GHMM_EMS3(2); break;
        case 4: GHMM_EMS3(4); break;
        case 8: GHMM_EMS3(8); break;
        case 16: GHMM_EMS3(16); break;
        case 32: GHMM_EMS3(32); break;
        default: GHMM_EMS3(64); break;
        }
        ctx->b_is_log = (mode == 2);
        return launch_ok("k_emission_sched");
    }
    if (mode == 0 && m->mfma_ok && ctx->kernels != 1) {
        // other coefficient counts: the generic matrix-core kernel (direct-form tiles inside)
        int rc;
        if ((rc = lds_attr(ctx, (const void *)k_emission_mfma))) return rc;
        const long long ntf = (c->F + 15) / 16;
        const int chunks = (m->NT + m->TC - 1) / m->TC;
        long long gx = (ntf + EM_WAVES - 1) / EM_WAVES;
        if (gx > ctx->cus) gx = ctx->cus; // one 8-wave block per CU, chunks in grid.y
        {
            kscope ks(ctx, GHMM_K_EMISSION);
            hipLaunchKernelGGL(k_emission_mfma, dim3((unsigned)gx, (unsigned)chunks),
                               dim3(EM_WAVES * WAVE), m->em_lds, ctx->stream, m->N, m->M, m->Mp, m->D,
                               m->DP, m->NT, m->TC, c->F, c->X, m->Wm, m->oglob, m->wkp, m->gmap,
                               m->condt, m->mean, m->inv_var, ctx->b, post, (const int *)nullptr,
                               m->epoch, m->tshift, m->dtile);
        }
        ctx->b_is_log = false;
        return launch_ok("k_emission_mfma");
    }
    const int *only_if = nullptr;
    {
        kscope ks(ctx, GHMM_K_EMISSION);
        if (mode == 0)
            hipLaunchKernelGGL(k_emission<0>, dim3((unsigned)blocks), dim3(WAVE), lds, ctx->stream,
                               m->N, m->M, m->D, c->F, c->X, m->mean, m->inv_var, m->wk, m->logwk,
                               ctx->b, post, ctx->lognorm, only_if, m->epoch);
        else if (mode == 1)
            hipLaunchKernelGGL(k_emission<1>, dim3((unsigned)blocks), dim3(WAVE), lds, ctx->stream,
                               m->N, m->M, m->D, c->F, c->X, m->mean, m->inv_var, m->wk, m->logwk,
                               ctx->b, post, ctx->lognorm, only_if, m->epoch);
        else
            hipLaunchKernelGGL(k_emission<2>, dim3((unsigned)blocks), dim3(WAVE), lds, ctx->stream,
                               m->N, m->M, m->D, c->F, c->X, m->mean, m->inv_var, m->wk, m->logwk,
                               ctx->b, post, ctx->lognorm, only_if, m->epoch);
    }
    ctx->b_is_log = (mode == 2);
    return launch_ok("k_emission");
}

extern "C" int ghmm_emission(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, int want_post)
{
    int rc = use(ctx);
    if (rc || (rc = check_pair(m, c))) return rc;
    if ((rc = ws_frames(ctx, m, c, want_post != 0))) return rc;
    return run_emission(ctx, m, c, ctx->robust ? 1 : 0, want_post != 0);
}

// ------------------------------------------------------- forward / backward

// A statement once per group width, LL the width as a constant (template argument)
#define GHMM_BY_LANES(L, ...)                                                                      \
    do {                                                                                           \
        if ((L) == 16) { constexpr int LL = 16; __VA_ARGS__; }                                     \
        else if ((L) == 32) { constexpr int LL = 32; __VA_ARGS__; }                                \
        else { constexpr int LL = 64; __VA_ARGS__; }                                               \
    } while (0)

static int fb_lanes(const ghmm_model *m, int *L)
{   // L = 0: more states than lanes, the one-wave-per-utterance kernels of ghmm_wide.hpp
    if (m->N <= 16) *L = 16;
    else if (m->N <= 32) *L = 32;
    else if (m->N <= 64) *L = 64;
    else if (m->N <= WIDE_MAX) *L = 0;
    else {
        ghmm_set_error("%d states: the forward / backward / Viterbi kernels take models of up to %d",
                       m->N, WIDE_MAX);
        return GHMM_ERR_UNSUPPORTED;
    }
    return GHMM_OK;
}

// models of more than 64 states: is A (zero = 0) or log A (zero = -inf) band-diagonal?  Decided on
// the device at every pass (the M-step may have changed it), read by the wide kernels behind it.
static int wide_band_flag(ghmm_ctx *ctx, const ghmm_model *m, const double *A, double zero)
{
    int rc;
    if (!ctx->wide_flag && (rc = dev_alloc(&ctx->wide_flag, 1))) return rc;
    HIP_TRY(hipMemsetAsync(ctx->wide_flag, 0, sizeof(int), ctx->stream));
    const long long n = (long long)m->N * m->N;
    hipLaunchKernelGGL(k_wide_band, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, m->N, A, zero,
                       ctx->wide_flag);
    return launch_ok("k_wide_band");
}

// The paired scans of ghmm_pair.hpp unless GHMM_OPT_KERNELS = 1 asks for the reference's order
// (calc_alpha, then calc_beta scaled by its c_t) in the one-pass kernels.
static bool use_pair(const ghmm_ctx *ctx, const ghmm_model *m) { return ctx->kernels != 1 && m->N <= 64; }

static int run_forward(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, bool with_backward = false,
                       bool score_only = false)
{   // score_only (ghmm_score): log P alone, alpha^ and c_t are not written

    if (c->U == 0) return GHMM_OK;
    int L, rc;
    if ((rc = fb_lanes(m, &L))) return rc;
    const double *ln = ctx->robust ? ctx->lognorm : nullptr;
    ctx->last_m = nullptr; // alpha^ is rewritten: run_backward re-arms these
    ctx->last_c = nullptr;
    ctx->own_bwd_done = false;
    ctx->beta_valid = false;
    if (L == 0) { // more than 64 states: one wave per utterance, the reference's order (ghmm_wide.hpp)
        ctx->loglik_pieces = false;
        if ((rc = wide_band_flag(ctx, m, m->A, 0.0))) return rc;
        kscope ks(ctx, GHMM_K_FORWARD);
        hipLaunchKernelGGL(k_forward_wide, dim3((unsigned)c->U), dim3(WAVE), (size_t)4 * m->N * sizeof(double),
                           ctx->stream, m->N, c->U, m->A, ctx->b, c->off, ctx->alpha, ctx->scale, ctx->sinv, ln,
                           ctx->loglik, c->order, ctx->wide_flag, m->N, (const double *)nullptr);
        return launch_ok("k_forward_wide");
    }
    const int gpw = WAVE / L;
    const unsigned blocks = (unsigned)((c->U + gpw - 1) / gpw);
    ctx->loglik_pieces = use_pair(ctx, m) && with_backward; // k_combine will take the logs
    {
        kscope ks(ctx, GHMM_K_FORWARD);
        if (use_pair(ctx, m)) {
            const unsigned ny = with_backward ? 2u : 1u;
            const int only = with_backward ? -1 : (score_only ? 2 : 0);
            GHMM_BY_LANES(L, hipLaunchKernelGGL(k_scan_pair<LL>, dim3(blocks, ny), dim3(WAVE), 0, ctx->stream, m->N, c->U,
                                                only, m->A, ctx->b, c->off, ctx->alpha, ctx->scale, ctx->sinv, ln,
                                                ctx->loglik, ctx->wrow, ctx->sb, ctx->sink, c->order));
            ctx->own_bwd_done = with_backward;
        } else
            GHMM_BY_LANES(L, hipLaunchKernelGGL(k_forward<LL>, dim3(blocks), dim3(WAVE), 0, ctx->stream, m->N, c->U,
                                                m->A, ctx->b, c->off, ctx->alpha, ctx->scale, ctx->sinv, ln,
                                                ctx->loglik, ctx->sink, c->order));
    }
    return launch_ok("k_forward");
}

#ifndef GHMM_FIX_BLOCKS
#define GHMM_FIX_BLOCKS 256u
#endif
// The utterances the pass before it listed (none on data a model fits), again, whole, in the
// reference's own order of operations with its dense inner loops: one small launch that leaves
// at once on an empty list.  spu = partial-sum slots per utterance of that pass.
static int run_backward_fix(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, int L, unsigned blocks, int spu,
                            const double *sinv)
{
    const unsigned fb = blocks < GHMM_FIX_BLOCKS ? blocks : GHMM_FIX_BLOCKS;
    int *cnt = ctx->fix_cnt + (ctx->fix_stamp & 1), *nxt = ctx->fix_cnt + ((ctx->fix_stamp + 1) & 1);
    GHMM_BY_LANES(L, hipLaunchKernelGGL(k_backward_fix<LL>, dim3(fb), dim3(WAVE), 0, ctx->stream, m->N, c->U,
                                        (int)ctx->delta, m->A, ctx->b, c->off, ctx->alpha, ctx->scale, ctx->beta,
                                        ctx->gamma, ctx->part_xi, ctx->part_dena, ctx->part_denc, ctx->sink, cnt,
                                        ctx->fix_list, nxt, spu, sinv));
    return launch_ok("k_backward_fix");
}

// gamma, the xi / den partial sums and (want_beta) the reference's beta^
static int run_backward(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, bool want_beta = true)
{
    if (c->U == 0) {
        ctx->slots = 0; // nothing was accumulated: every utterance sum is empty
        return GHMM_OK;
    }
    int L, rc;
    if ((rc = fb_lanes(m, &L))) return rc;
    if (L == 0) {
        if ((rc = wide_band_flag(ctx, m, m->A, 0.0))) return rc;
        if ((rc = lds_attr(ctx, (const void *)k_backward_wide))) return rc; // (64 KB at 512 states)
        kscope ks(ctx, GHMM_K_BACKWARD);
        hipLaunchKernelGGL(k_backward_wide, dim3((unsigned)c->U), dim3(WAVE),
                           (size_t)(8 + MAX_DELTA + 1) * m->N * sizeof(double), ctx->stream, m->N, c->U,
                           (int)ctx->delta, m->A, ctx->b, c->off, ctx->alpha, ctx->scale, ctx->beta, ctx->gamma,
                           ctx->part_xi, ctx->part_dena, ctx->part_denc, c->order, ctx->wide_flag);
        ctx->beta_valid = true;
        ctx->slots = c->U;
        return launch_ok("k_backward_wide");
    }
    const int gpw = WAVE / L;
    const unsigned blocks = (unsigned)((c->U + gpw - 1) / gpw);
    {
        kscope ks(ctx, GHMM_K_BACKWARD);
        if (use_pair(ctx, m)) {
            if (!ctx->own_bwd_done) { // the forward pass ran alone (row API): the other direction now
                GHMM_BY_LANES(L, hipLaunchKernelGGL(k_scan_pair<LL>, dim3(blocks, 1u), dim3(WAVE), 0, ctx->stream,
                                                    m->N, c->U, 1, m->A, ctx->b, c->off, ctx->alpha, ctx->scale,
                                                    ctx->sinv, (const double *)nullptr, ctx->loglik, ctx->wrow,
                                                    ctx->sb, ctx->sink, c->order));
                ctx->own_bwd_done = true;
            }
            const unsigned cb = (unsigned)(((long long)c->U * CB_CH + gpw - 1) / gpw);
#define GHMM_COMBINE(LL, WB, DN)                                                                   \
    hipLaunchKernelGGL((k_combine<LL, WB, DN>), dim3(cb), dim3(WAVE), 0, ctx->stream, m->N, c->U,  \
                       (int)ctx->delta, m->A, c->off, ctx->alpha, ctx->scale, ctx->wrow, ctx->sb, \
                       ctx->beta, ctx->gamma, ctx->part_xi, ctx->part_dena, ctx->part_denc, ctx->sink, \
                       ctx->robust ? ctx->lognorm : (const double *)nullptr,                     \
                       ctx->loglik_pieces ? ctx->lpart : (double *)nullptr, ctx->logk, c->order,  \
                       ctx->fix_mark, ctx->fix_stamp, ctx->fix_cnt + (ctx->fix_stamp & 1), ctx->fix_list)
            // the M-step keeps a band-diagonal A band-diagonal as long as it re-estimates
            // no transition beyond i -> i + 1
            const bool band2 = m->banded && ctx->delta <= 1;
            if (++ctx->fix_stamp == 0x7fffffff) { // (2^31 passes: start the marks over)
                HIP_TRY(hipMemsetAsync(ctx->fix_mark, 0, ctx->cap_fix_mark * sizeof(int), ctx->stream));
                HIP_TRY(hipMemsetAsync(ctx->fix_cnt, 0, 2 * sizeof(int), ctx->stream));
                ctx->fix_stamp = 1;
            }
            if (band2) {
                if (want_beta) GHMM_BY_LANES(L, GHMM_COMBINE(LL, true, false));
                else GHMM_BY_LANES(L, GHMM_COMBINE(LL, false, false));
            } else {
                if (want_beta) GHMM_BY_LANES(L, GHMM_COMBINE(LL, true, true));
                else GHMM_BY_LANES(L, GHMM_COMBINE(LL, false, true));
            }
            if ((rc = run_backward_fix(ctx, m, c, L, blocks, CB_CH, nullptr))) return rc;
            ctx->lp_nch = CB_CH;
            ctx->beta_valid = want_beta;
            ctx->last_m = m;
            ctx->last_c = c;
            ctx->slots = c->U * CB_CH;
        } else {
            if (++ctx->fix_stamp == 0x7fffffff) ctx->fix_stamp = 1; // (no marks on this tier: a wave lists each of its utterances once)
            int *cnt = ctx->fix_cnt + (ctx->fix_stamp & 1);
            GHMM_BY_LANES(L, hipLaunchKernelGGL(k_backward<LL>, dim3(blocks), dim3(WAVE), 0, ctx->stream, m->N, c->U,
                                                (int)ctx->delta, m->A, ctx->b, c->off, ctx->alpha, ctx->scale,
                                                ctx->sinv, ctx->beta, ctx->gamma, ctx->part_xi, ctx->part_dena,
                                                ctx->part_denc, ctx->sink, c->order, cnt, ctx->fix_list));
            // utterances whose band-only update met an overflowed beta^: again, dense (TF:1493-1510)
            if ((rc = run_backward_fix(ctx, m, c, L, blocks, 1, ctx->sinv))) return rc;
            ctx->beta_valid = true;
            ctx->slots = c->U;
        }
    }
    return launch_ok("k_backward");
}

// ghmm_estep on a band-diagonal A: forward, backward and the gamma / xi pass in one launch
// (k_scan_combine; never slower than the separate launches, 1 000 .. 12 500 utterances measured:
// profiles/tools/fused_ab.py).  Returns GHMM_OK with *done = false when the separate launches apply.
static int run_scan_combine(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, bool *done)
{
    *done = false;
    int L, rc;
    if (c->U == 0 || !use_pair(ctx, m) || ctx->fused_scan == 2) return GHMM_OK;
    if ((rc = fb_lanes(m, &L))) return rc;
    const bool band2 = m->banded && ctx->delta <= 1;
    if (!band2) return GHMM_OK;
    // utterances per block: more of them, in fewer chunks each, the shorter they are (a 30-frame word
    // in eight chunks is all set-up, and the reduction then reads 8 slots per word) — as far as
    // the block's eight waves can scan them (2 upb / (WAVE / L) <= 8)
    int upb = SC_UPB;
    {
        const long long tmean = c->F / c->U;
        const int want = tmean >= 160 ? SC_UPB : (tmean >= 80 ? 2 * SC_UPB : 4 * SC_UPB);
        while (upb < want && 2 * (2 * upb) / (WAVE / L) <= CB_CH) upb *= 2;
    }
    const int nch = SC_GROUPS / upb;
    const unsigned blocks = (unsigned)((c->U + upb - 1) / upb);
    ctx->own_bwd_done = true;
    ctx->loglik_pieces = true; // the combine phase takes the logs of log P
    if (++ctx->fix_stamp == 0x7fffffff) { // (2^31 passes: start the marks over)
        HIP_TRY(hipMemsetAsync(ctx->fix_mark, 0, ctx->cap_fix_mark * sizeof(int), ctx->stream));
        HIP_TRY(hipMemsetAsync(ctx->fix_cnt, 0, 2 * sizeof(int), ctx->stream));
        ctx->fix_stamp = 1;
    }
    {
        kscope ks(ctx, GHMM_K_FORWARD);
        GHMM_BY_LANES(L, hipLaunchKernelGGL((k_scan_combine<LL, false>), dim3(blocks), dim3(CB_CH * WAVE), 0, ctx->stream,
                                            m->N, c->U, (int)ctx->delta, upb, m->A, ctx->b, c->off, ctx->alpha, ctx->scale,
                                            ctx->wrow, ctx->sb, ctx->beta, ctx->gamma, ctx->part_xi, ctx->part_dena,
                                            ctx->part_denc, ctx->sink,
                                            ctx->robust ? ctx->lognorm : (const double *)nullptr, ctx->lpart, ctx->logk,
                                            c->order, ctx->fix_cnt + (ctx->fix_stamp & 1),
                                            ctx->fix_cnt + ((ctx->fix_stamp + 1) & 1)));
    }
    if ((rc = launch_ok("k_scan_combine"))) return rc;
    ctx->lp_nch = nch;
    ctx->beta_valid = false; // beta^ on demand (ghmm_fetch)
    ctx->last_m = m;
    ctx->last_c = c;
    ctx->slots = c->U * nch;
    *done = true;
    return GHMM_OK;
}

static int need_emission(ghmm_ctx *ctx, const ghmm_model *m, const ghmm_corpus *c)
{
    if (!ctx->b || ctx->F != c->F || ctx->U != c->U || ctx->N != m->N || ctx->b_is_log ||
        ctx->em_m != m || ctx->em_c != c || ctx->em_epoch != m->epoch) {
        ghmm_set_error("call ghmm_emission on this model and corpus first");
        return GHMM_ERR_ARG;
    }
    return GHMM_OK;
}

extern "C" int ghmm_forward(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c)
{
    int rc = use(ctx);
    if (rc || (rc = check_pair(m, c)) || (rc = need_emission(ctx, m, c))) return rc;
    if ((rc = ws_fb(ctx, m, c))) return rc;
    return run_forward(ctx, m, c);
}

extern "C" int ghmm_backward(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c)
{
    int rc = use(ctx);
    if (rc || (rc = check_pair(m, c)) || (rc = need_emission(ctx, m, c))) return rc;
    if ((rc = ws_fb(ctx, m, c))) return rc;
    return run_backward(ctx, m, c);
}

// -------------------------------------------------------------- statistics

template <int CT, int NE>
static int launch_mixstats_mfma(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, int P, int chunks)
{
    const bool ntl = nt_posteriors(ctx, m, c); // (single-chunk launches only: small models)
    // fold buffer: three of the four register rows of every tile, four waves (ghmm_mfma.hpp, end of k_mixstats_mfma)
    const size_t fold = (size_t)CT * NE * 4 * 3 * 64 * sizeof(double);
    // staged variant: every chunk of CT tiles must map to an even-aligned, even-length run of
    // real Gaussians that starts in the first slot of its first tile (mixture padding and odd
    // mixture counts included: 10 x 3, 10 x 5 ... qualify, 5 x 3 with its 15 Gaussians does not) and
    // N <= 32 (two 16-byte gamma pieces per lane and stage up to 16 states, four beyond)
    const int G = m->N * m->M;
    bool staged = (((unsigned long long)c->X) & 15ull) == 0 && // (its frame pieces are 16-byte loads)
                  m->N <= 32 && (G % 2) == 0 && chunks <= 64 && ctx->kernels != 3;
    int cg0[64], cgw[64], GWmax = 0; // per chunk: first real Gaussian, number of real Gaussians
    for (int ch = 0; staged && ch < chunks; ch++) {
        const int p0 = ch * CT * 16, p1 = (ch + 1) * CT * 16 < m->NT * 16 ? (ch + 1) * CT * 16 : m->NT * 16;
        int first = -1, n = 0;
        for (int gp = p0; gp < p1; gp++)
            if (gp / m->Mp < m->N && gp % m->Mp < m->M) {
                if (first < 0) first = (gp / m->Mp) * m->M + gp % m->Mp;
                n++;
            }
        const bool head_real = p0 / m->Mp < m->N && p0 % m->Mp < m->M;
        if (n == 0 || !head_real || (first % 2) != 0 || (n % 2) != 0) staged = false;
        cg0[ch] = first;
        cgw[ch] = n;
        if (n > GWmax) GWmax = n;
    }
    const bool wide = m->N > 16;
    int rc;
    if (staged) {
        size_t stage = (size_t)MSM_WAVES * (16 * (NE * 16 + GWmax + m->N) + m->DP) * sizeof(double);
        size_t lds = stage > fold ? stage : fold;
        if (lds <= 150 * 1024) {
            // Several chunks (e.g. 64 mixtures: 8 chunks of little more than one state): each
            // chunk's launch walks only the stages in which one of its states is occupied, from
            // a per-stage state mask taken once from gamma.
            bool masked = false;
            if constexpr (NE == 5) {
                if (chunks > 1 && c->F >= 16) {
                    const size_t nst = (size_t)((c->F + 15) / 16);
                    if ((rc = dev_grow(&ctx->smask, &ctx->cap_smask, nst))) return rc;
                    hipLaunchKernelGGL(k_stage_masks, dim3((unsigned)((c->F + WAVE - 1) / WAVE)), dim3(WAVE), 0,
                                       ctx->stream, m->N, c->F, ctx->gamma, ctx->smask);
                    masked = true;
                }
            }
#define GHMM_MIXSTATS(...)                                                                         \
    do {                                                                                           \
        if ((rc = lds_attr(ctx, (const void *)k_mixstats_mfma<__VA_ARGS__>))) return rc;          \
        hipLaunchKernelGGL((k_mixstats_mfma<__VA_ARGS__>), dim3((unsigned)P, 1u), dim3(MSM_WAVES * WAVE), lds,  \
                           ctx->stream, m->N, m->M, m->Mp, m->D, m->DP, m->NT, c->F, gmin, GW, c->X, ctx->gamma, \
                           ctx->post, m->gmap + 0, m->oglob, ctx->part_m,                          \
                           masked ? ctx->smask : (const unsigned *)nullptr);                       \
    } while (0)
            // one launch per chunk so that gmin / GW are plain arguments (chunks == 1 at 10x8)
            for (int ch = 0; ch < chunks; ch++) {
                const int gmin = cg0[ch], GW = cgw[ch];
                if constexpr (NE == 5) {
                    if (masked) {
                        if (wide) GHMM_MIXSTATS(CT, NE, true, true, false, 4);
                        else GHMM_MIXSTATS(CT, NE, true, true);
                        continue;
                    }
                }
                if (wide) GHMM_MIXSTATS(CT, NE, true, false, false, 4);
                else if (ntl) GHMM_MIXSTATS(CT, NE, true, false, true);
                else GHMM_MIXSTATS(CT, NE, true);
            }
#undef GHMM_MIXSTATS
            return GHMM_OK;
        }
    }
    if ((rc = lds_attr(ctx, (const void *)k_mixstats_mfma<CT, NE, false>))) return rc;
    hipLaunchKernelGGL((k_mixstats_mfma<CT, NE, false>), dim3((unsigned)P, (unsigned)chunks),
                       dim3(MSM_WAVES * WAVE), fold, ctx->stream, m->N, m->M, m->Mp, m->D, m->DP, m->NT,
                       c->F, 0, 0, c->X, ctx->gamma, ctx->post, m->gmap, m->oglob, ctx->part_m,
                       (const unsigned *)nullptr);
    return GHMM_OK;
}

static int run_accumulate(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, ghmm_stats *s)
{
    const int N = m->N, M = m->M, D = m->D, G = N * M, D1 = D + 1;
    const long long E = (long long)G * D1;
    const int NB = (int)((E + MS_THREADS * MS_EPT - 1) / (MS_THREADS * MS_EPT));
    const bool mfma = m->mfma_ok && m->CT > 0 && ctx->kernels != 1;
    int rc;
    // frame-block partials: enough blocks to fill the chip a few times over, few
    // enough that the partial sums stay small next to the frame data
    long long P = ctx->partials > 0 ? ctx->partials : (4LL * ctx->cus + NB - 1) / NB; // (4 blocks per CU: 0.70 -> 0.41 ms at configs[1] against 2)
    if (mfma && ctx->partials <= 0 && G <= MS_MAXG) {
        // behind the matrix-core kernel this one only sees the few ill-conditioned Gaussians
        // (one block of elements): parallelism has to come from the frame axis
        P = 4LL * ctx->cus;
        const long long cap = (256LL << 20) / (E * 16);
        if (P > cap) P = cap;
    }
    if (P < 1) P = 1;
    // frames staged through LDS per pass: x[FS][D+1] + w[FS][GW], GW = Gaussians that a
    // batch of 256*EPT elements can touch; keep the tile under 48 KB
    int GWmax = (MS_THREADS * MS_EPT) / D1 + 2;
    if (GWmax > G) GWmax = G;
    int FS = MS_FS;
    while (FS > 1 && (size_t)FS * (D1 + GWmax) * sizeof(double) > 48 * 1024) FS /= 2;
    const size_t lds = (size_t)FS * (D1 + GWmax) * sizeof(double);
    if (lds > 64 * 1024) {
        ghmm_set_error("coefficient count %d too large for the statistics tile", D);
        return GHMM_ERR_UNSUPPORTED;
    }
    long long fpb = (c->F + P - 1) / P;
    fpb = ((fpb + FS - 1) / FS) * FS;
    if (fpb < FS) fpb = FS;
    P = c->F > 0 ? (c->F + fpb - 1) / fpb : 0;
    const int *only_if = mfma ? m->sflag : nullptr; // statistics: conditioning around the global offset
    if (P > 0) {
        size_t need = (size_t)P * (size_t)E;
        if ((rc = dev_grow(&ctx->part_mu, &ctx->cap_pmu, need))) return rc;
        if ((rc = dev_grow(&ctx->part_var, &ctx->cap_pvar, need))) return rc;
    }
    int Pm = 0;
    size_t nsum = 0;
    if (mfma) {
        // matrix-core statistics: one partial per block, one block per CU
        const int chunks = (m->NT + m->CT - 1) / m->CT;
        Pm = ctx->cus;
        const long long steps = c->F / 16;
        if (steps < (long long)Pm * MSM_WAVES) Pm = (int)((steps + MSM_WAVES - 1) / MSM_WAVES);
        if (Pm < 1) Pm = 1;
        nsum = (size_t)m->NT * 16 * m->NE * 16;
        if ((rc = dev_grow(&ctx->part_m, &ctx->cap_pm, (size_t)Pm * nsum))) return rc;
        {
            kscope ks(ctx, GHMM_K_MIXSTATS);
            switch (m->NE) {
            case 1: rc = launch_mixstats_mfma<8, 1>(ctx, m, c, Pm, chunks); break;
            case 2: rc = launch_mixstats_mfma<8, 2>(ctx, m, c, Pm, chunks); break;
            case 3: rc = launch_mixstats_mfma<8, 3>(ctx, m, c, Pm, chunks); break;
            case 4: rc = launch_mixstats_mfma<6, 4>(ctx, m, c, Pm, chunks); break;
            case 5: rc = launch_mixstats_mfma<5, 5>(ctx, m, c, Pm, chunks); break;
            case 6: rc = launch_mixstats_mfma<4, 6>(ctx, m, c, Pm, chunks); break;
            case 7: rc = launch_mixstats_mfma<3, 7>(ctx, m, c, Pm, chunks); break;
            default: rc = launch_mixstats_mfma<3, 8>(ctx, m, c, Pm, chunks); break;
            }
        }
        if (rc || (rc = launch_ok("k_mixstats_mfma"))) return rc;
    }
    // matrix-core tier: the vector-ALU kernel has work only when the model holds class-2
    // Gaussians (ghmm_mfma.hpp stats_class), which the host learns some launches late from
    // pinned memory (hflag_host).  Launched while that was recently the case or the model has
    // just come from the host; otherwise left out (4.5 us of an idle 1 000-block launch per
    // iteration), and k_reduce_all recomputes a class-2 Gaussian exactly should one be there.
    bool vec = P > 0;
    if (vec && mfma && G <= MS_MAXG && m->hflag_host) {
        const int seen = *(volatile int *)m->hflag_host;
        if (ctx->sync_mark >= m->prep_mark)
            vec = seen == m->epoch; // the preparation has completed (a wait covered it): the flag is current
        else
            // the host runs ahead of the device by an unknown number of iterations (a bench loop
            // never waits): launch while a class-2 Gaussian was seen within the last VEC_WINDOW
            // preparations or the model has come from the host that recently
            vec = m->epoch <= m->vec_until || (m->epoch - seen) <= VEC_WINDOW;
        if (ctx->vec_stats) vec = ctx->vec_stats == 1;
    }
    if (vec) {
        // vector-ALU statistics: the whole job on that tier, or (matrix-core tier) only
        // when the model holds ill-conditioned Gaussians, whose sums it then supplies
        kscope ks(ctx, GHMM_K_MIXSTATS);
        hipLaunchKernelGGL(k_mixstats, dim3((unsigned)P, (unsigned)NB), dim3(MS_THREADS), lds,
                           ctx->stream, N, M, D, c->F, fpb, FS, c->X, ctx->gamma, ctx->post, m->mean,
                           ctx->part_mu, ctx->part_var, only_if, m->epoch,
                           (mfma && G <= MS_MAXG) ? m->scls : (const int *)nullptr, m->Mp);
        if ((rc = launch_ok("k_mixstats"))) return rc;
    }
    {
        reduce_args ra;
        ra.N = N; ra.M = M; ra.D = D; ra.U = c->U; ra.delta = (int)ctx->delta;
        ra.S = ctx->slots;
        ra.lpart = ctx->loglik_pieces ? ctx->lpart : nullptr;
        ra.logk = ctx->logk;
        ra.P1 = vec ? (int)P : 0; ra.part_mu = ctx->part_mu; ra.part_var = ctx->part_var;
        ra.vec = vec ? 1 : 0;
        ra.Pm = (mfma && c->F > 0) ? Pm : 0;
        ra.NT = m->NT; ra.DP = m->DP; ra.ES = m->NE * 16;
        ra.part_m = ctx->part_m; ra.condg = m->condg; ra.oglob = m->oglob; ra.mean = m->mean;
        ra.gmap = m->gmap; ra.cond_max = COND_MAX;
        ra.otile = ra.Pm > 0 ? m->otile : nullptr;
        ra.tnext = m->tnext;
        ra.scls = m->scls;
        ra.X = c->X; ra.gamma = ctx->gamma; ra.post = ctx->post; ra.F = c->F;
        // rounding bound of a sum of `len` terms accumulated in sequence (a wave's share of the
        // frames) and then over the partials: (len + partials) * eps, doubled
        ra.kappa = 2.0 * 1.1102230246251565e-16 * ((double)c->F / (double)(ctx->cus * MSM_WAVES) + 2.0 * ctx->cus + 64.0);
        ra.part_xi = ctx->part_xi; ra.part_dena = ctx->part_dena; ra.part_denc = ctx->part_denc;
        ra.loglik = ctx->loglik; ra.stats = s->v;
        ra.mbox = (s->mbox_slot >= 0 && ctx->mbox_page_dev) ? ctx->mbox_page_dev + 4 * s->mbox_slot : nullptr;
        ra.mbox_seq = ++ctx->mbox_seq;
        s->mbox_expect = ra.mbox_seq;
        s->mbox_mark = ctx->launch_mark;
        s->mbox_valid = ra.mbox != nullptr;
        if (mfma && c->F == 0) ra.P1 = 0; // nothing accumulated: every sum is empty
        const int NGb = ra.Pm > 0 ? m->NT * 16 : G;
        kscope ks(ctx, GHMM_K_REDUCE);
        const int tile_blocks = ra.otile ? m->NT : 0; // the next model's tile offsets
        hipLaunchKernelGGL(k_reduce_all, dim3((unsigned)(NGb + N * N + 2 * N + 1 + tile_blocks)), dim3(RD_THREADS), 0,
                           ctx->stream, ra);
    }
    return launch_ok("k_reduce_all");
}

static int check_stats(const ghmm_model *m, const ghmm_stats *s)
{
    if (!s || s->N != m->N || s->M != m->M || s->D != m->D) {
        ghmm_set_error("statistics vector does not match the model's shape");
        return GHMM_ERR_ARG;
    }
    return GHMM_OK;
}

extern "C" int ghmm_accumulate(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, ghmm_stats *s)
{
    int rc = use(ctx);
    if (rc || (rc = check_pair(m, c)) || (rc = need_emission(ctx, m, c)) || (rc = check_stats(m, s)))
        return rc;
    if (!ctx->gamma || !ctx->post) {
        ghmm_set_error("call ghmm_emission(want_post=1), ghmm_forward and ghmm_backward first");
        return GHMM_ERR_ARG;
    }
    return run_accumulate(ctx, m, c, s);
}

static int fetch_impl(ghmm_ctx *ctx, int which, bool whole, size_t first, double *host, size_t n)
{
    int rc = use(ctx);
    if (rc) return rc;
    ARG_CHECK(host || n == 0, "null destination");
    const double *src = nullptr;
    size_t have = 0;
    size_t F = (size_t)ctx->F, N = (size_t)ctx->N, G = (size_t)ctx->G, U = (size_t)ctx->U;
    switch (which) {
    case GHMM_BUF_B: src = ctx->b; have = F * N; break;
    case GHMM_BUF_POST: src = ctx->post; have = F * G; break;
    case GHMM_BUF_ALPHA: src = ctx->alpha; have = F * N; break;
    case GHMM_BUF_BETA:
        if (!ctx->beta_valid && ctx->beta) {
            // ghmm_estep leaves beta^ out; form it now from the same alpha^, W and 1/s
            if (!ctx->last_m || !ctx->last_c) {
                ghmm_set_error("beta^ of the last E-step is gone (the model changed): call ghmm_backward");
                return GHMM_ERR_ARG;
            }
            if ((rc = run_backward(ctx, ctx->last_m, ctx->last_c, true))) return rc;
        }
        src = ctx->beta;
        have = F * N;
        break;
    case GHMM_BUF_SCALE: src = ctx->scale; have = F; break;
    case GHMM_BUF_GAMMA: src = ctx->gamma; have = F * N; break;
    case GHMM_BUF_LOGLIK:
        if (ctx->loglik_pieces && ctx->loglik && U) {
            hipLaunchKernelGGL(k_loglik_assemble, dim3((unsigned)((U + 255) / 256)), dim3(256), 0, ctx->stream,
                               (int)U, ctx->lp_nch, ctx->lpart, ctx->logk, ctx->loglik);
            ctx->loglik_pieces = false; // loglik[] is complete now (the pieces stay valid too)
        }
        src = ctx->loglik;
        have = U;
        break;
    case GHMM_BUF_LOGNORM: src = ctx->lognorm; have = F; break;
    default:
        ghmm_set_error("unknown buffer %d", which);
        return GHMM_ERR_ARG;
    }
    if (!src || (whole ? n != have : (first > have || n > have - first))) {
        ghmm_set_error("buffer %d holds %zu doubles, [%zu, %zu) requested", which, src ? have : 0, first,
                       first + n);
        return GHMM_ERR_ARG;
    }
    return d2h_pageable(ctx, host, src + first, n * 8);
}

extern "C" int ghmm_fetch(ghmm_ctx *ctx, int which, double *host, size_t n)
{
    return fetch_impl(ctx, which, true, 0, host, n);
}

extern "C" int ghmm_fetch_range(ghmm_ctx *ctx, int which, size_t first, double *host, size_t n)
{
    return fetch_impl(ctx, which, false, first, host, n);
}

// -------------------------------------------------------------------- fused

extern "C" int ghmm_estep(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, ghmm_stats *s)
{
    int rc = use(ctx);
    if (rc || (rc = check_pair(m, c)) || (rc = check_stats(m, s))) return rc;
    if ((rc = ws_frames(ctx, m, c, true)) || (rc = ws_fb(ctx, m, c))) return rc;
    if ((rc = run_emission(ctx, m, c, ctx->robust ? 1 : 0, true))) return rc;
    bool fused = false;
    if ((rc = run_scan_combine(ctx, m, c, &fused))) return rc;
    if (!fused) {
        if ((rc = run_forward(ctx, m, c, true))) return rc;
        if ((rc = run_backward(ctx, m, c, false))) return rc; // beta^ on demand (ghmm_fetch)
    }
    return run_accumulate(ctx, m, c, s);
}

extern "C" int ghmm_mstep(ghmm_ctx *ctx, ghmm_model *m, ghmm_stats *s)
{
    int rc = use(ctx);
    if (ctx && ctx->last_m == m) ctx->last_m = nullptr; // alpha^ / W on the device belong to the old parameters
    if (rc) return rc;
    ARG_CHECK(m, "null model");
    if ((rc = check_stats(m, s))) return rc;
    m->epoch++; // a new set of parameters (and a new preparation of the matrix-core form)
    m->prep_mark = ++ctx->launch_mark;
    // transitions are re-estimated inside the band i <= j <= i + delta only (TF:1601): a
    // band-diagonal A stays band-diagonal exactly when that band is i, i + 1
    m->banded = m->banded && ctx->delta <= 1;
    {
        kscope ks(ctx, GHMM_K_MSTEP);
        size_t md = (size_t)m->M * m->D;
        // the state's variances + weights in LDS (+ offsets and a reduction buffer: under 64 KB)
        int lds_doubles = (md + m->M) * 8 <= 56 * 1024 ? (int)(md + m->M) : 0;
        if (m->mfma_ok) {
            // M-step and matrix-core preparation of the new model in one launch
            hipLaunchKernelGGL(k_mstep_mfma, dim3((unsigned)m->N), dim3(MSF_THREADS),
                               ((size_t)lds_doubles + m->DP + MSF_THREADS) * 8, ctx->stream, m->N, m->M, m->D, s->v,
                               pow(2.0 * M_PI, m->D / 2.0), m->A, m->c, m->mean, m->inv_var, m->det, m->wk,
                               m->logwk, m->logA, lds_doubles, m->Mp, m->NT, m->DP, m->oglob, m->Wm, m->wkp,
                               m->logwkp, m->gmap, m->condg, m->anyflag, m->epoch, m->otile, m->tnext,
                               m->condt, m->dtile, m->tshift, m->sflag, (int)ctx->delta, m->scls, m->hflag_dev);
            return launch_ok("k_mstep_mfma");
        }
        hipLaunchKernelGGL(k_mstep, dim3((unsigned)m->N), dim3(MS2_THREADS), (size_t)lds_doubles * 8,
                           ctx->stream, m->N, m->M, m->D, s->v, pow(2.0 * M_PI, m->D / 2.0), m->A, m->c,
                           m->mean, m->inv_var, m->det, m->wk, m->logwk, m->logA, lds_doubles,
                           (int)ctx->delta);
    }
    if ((rc = launch_ok("k_mstep"))) return rc;
    return model_prepare(ctx, m, false);
}

// creating_initial_model (TF:732-1317): distance / accumulation passes on the device (hard
// statistics through the vector-ALU statistics kernel: direct (x - mean)^2, no expanded
// form), cell bookkeeping on the host exactly as the reference orders it.
// new means, everything else as last set (the k-means passes of ghmm_model_init: one copy instead of five)
static int model_set_means(ghmm_ctx *ctx, ghmm_model *m, const double *mean)
{
    if (ctx->last_m == m) ctx->last_m = nullptr;
    HIP_TRY(hipMemcpyAsync(m->mean, mean, (size_t)m->N * m->M * m->D * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(stream_sync(ctx)); // (pageable host memory, see ghmm_model_set)
    m->epoch++;
    m->vec_until = m->epoch + VEC_WINDOW;
    m->prep_mark = ++ctx->launch_mark;
    return model_prepare(ctx, m, true);
}

extern "C" int ghmm_model_init(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c)
{
    return ghmm_model_init_comm(ctx, m, c, nullptr);
}

extern "C" int ghmm_model_init_comm(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, ghmm_comm *comm)
{
    int rc = use(ctx);
    if (rc || (rc = check_pair(m, c))) return rc;
    ARG_CHECK(c->U > 0 && c->F > 0, "empty corpus");
    const int N = m->N, M = m->M, D = m->D, G = N * M;
    if ((rc = ws_frames(ctx, m, c, true)) || (rc = ws_fb(ctx, m, c))) return rc;
    // the utterance-level partials feed k_reduce_all too; they carry nothing here
    HIP_TRY(hipMemsetAsync(ctx->part_xi, 0, (size_t)c->U * N * (MAX_DELTA + 1) * 8, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->part_dena, 0, (size_t)c->U * N * 8, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->part_denc, 0, (size_t)c->U * N * 8, ctx->stream));
    ctx->slots = c->U;
    ctx->loglik_pieces = false; // log P plays no part here: the zeroed loglik[] is what gets summed
    HIP_TRY(hipMemsetAsync(ctx->loglik, 0, (size_t)c->U * 8, ctx->stream));
    ghmm_stats *st = nullptr;
    if ((rc = ghmm_stats_create(ctx, N, M, D, &st))) return rc;
    const size_t ns = st->n;
    std::vector<double> sv(ns), A((size_t)N * N), cw((size_t)G, 1.0 / M), cells((size_t)G * D, 0.0),
        ones((size_t)G * D, 1.0), det1((size_t)G, 1.0), dist((size_t)G, 0.0);
    std::vector<int> idx((size_t)M);
    const double *num_c = sv.data() + (size_t)N * N + 2 * (size_t)N;
    const double *num_mu = num_c + G, *num_var = num_mu + (size_t)G * D;
    // one-step left-to-right transitions, uniform over the allowed band (TF:774-806)
    const int delta = 1;
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            double a;
            if (j > delta + i || j < i) a = 0.0;
            else if (delta + 1 > N - i) a = 1.0 / (double)(N - i);
            else a = 1.0 / (double)(delta + 1);
            A[(size_t)i * N + j] = a;
        }
    const int64_t saved_kernels = ctx->kernels;
    // The k-means passes only need the cells' means (a plain quotient of sums) and the ORDER of
    // their distortions: they run on the context's tier (matrix-core sums, 0.08 ms a pass).  The
    // last pass gives the model's variances: direct-form statistics (x - mean)^2 on the vector
    // ALU, like the reference takes them (0.7 ms).
    bool first_pass = true;
    auto pass = [&](int n_cells, bool exact) -> int {
        ctx->kernels = exact ? 1 : saved_kernels;
        // (only the cells' means change from pass to pass)
        int r = first_pass ? ghmm_model_set(ctx, m, A.data(), cw.data(), cells.data(), ones.data(), det1.data())
                           : model_set_means(ctx, m, cells.data());
        first_pass = false;
        if (r) return r;
        {
            kscope ks(ctx, GHMM_K_PREPARE);
            int FR = IC_FRAMES; // frames per block: their copy in LDS stays under 48 KB
            while (FR > 1 && (size_t)FR * (D | 1) * sizeof(double) > 48 * 1024) FR /= 2;
            hipLaunchKernelGGL(k_init_classify, dim3((unsigned)((c->F + FR - 1) / FR)), dim3(256),
                               (size_t)FR * (D | 1) * sizeof(double), ctx->stream, N, M, D, n_cells,
                               c->U, c->F, FR, c->X, c->off, m->mean, ctx->gamma, ctx->post);
        }
        if ((r = launch_ok("k_init_classify")) || (r = run_accumulate(ctx, m, c, st))) return r;
        // a corpus sharded over ranks: the cell sums of all shards (every rank then does the
        // same bookkeeping on the same numbers)
        if (comm && (r = ghmm_stats_allreduce(ctx, st, comm))) return r;
        if ((r = ghmm_stats_download(ctx, st, sv.data()))) return r;
        for (int g = 0; g < G; g++) {
            double d = 0.0;
            for (int l = 0; l < D; l++) d += num_var[(size_t)g * D + l];
            dist[g] = d;
        }
        return GHMM_OK;
    };
    // indices by decreasing key, adjacent-swap passes with strict '<' (TF:1289-1315)
    auto order_desc = [&](const double *key, int n) {
        for (int i = 0; i < n; i++) idx[i] = i;
        bool done = false;
        while (!done) {
            done = true;
            for (int i = 0; i < n - 1; i++)
                if (key[idx[i]] < key[idx[i + 1]]) {
                    std::swap(idx[i], idx[i + 1]);
                    done = false;
                }
        }
    };
    auto split_cell = [&](double *ck, int from, int to) { // TF:1138
        for (int l = 0; l < D; l++) ck[(size_t)to * D + l] = ck[(size_t)from * D + l] * 1.005;
        for (int l = 0; l < D; l++) ck[(size_t)from * D + l] = ck[(size_t)from * D + l] * 0.995;
    };
    auto finish = [&](int r) {
        ctx->kernels = saved_kernels;
        ghmm_stats_destroy(ctx, st);
        return r;
    };
    bool dev_done = false;
    if (M <= INIT_MAXM) {
        // The k-means rounds without a trip to the host: a pass leaves the cell sums in `st`,
        // k_init_cells turns them into the next pass's means on the device (same operations, same
        // order as the host loop below, which stays for wider mixtures), the matrix-core form of the
        // new means is prepared, the next pass is launched.  (Downloading the sums and uploading
        // the means was ~80 us of round trips per pass, eleven passes at eight mixtures.)
        auto dev_pass = [&](int n_cells, int do_split, bool first) -> int {
            int r;
            if (first) {
                if ((r = ghmm_model_set(ctx, m, A.data(), cw.data(), cells.data(), ones.data(), det1.data()))) return r;
            } else {
                if (ctx->last_m == m) ctx->last_m = nullptr;
                m->epoch++;
                m->vec_until = m->epoch + VEC_WINDOW;
                m->prep_mark = ++ctx->launch_mark;
                if ((r = model_prepare(ctx, m, true))) return r;
            }
            {
                kscope ks(ctx, GHMM_K_PREPARE);
                int FR = IC_FRAMES;
                while (FR > 1 && (size_t)FR * (D | 1) * sizeof(double) > 48 * 1024) FR /= 2;
                hipLaunchKernelGGL(k_init_classify, dim3((unsigned)((c->F + FR - 1) / FR)), dim3(256),
                                   (size_t)FR * (D | 1) * sizeof(double), ctx->stream, N, M, D, n_cells,
                                   c->U, c->F, FR, c->X, c->off, m->mean, ctx->gamma, ctx->post);
            }
            if ((r = launch_ok("k_init_classify")) || (r = run_accumulate(ctx, m, c, st))) return r;
            if (comm && (r = ghmm_stats_allreduce(ctx, st, comm))) return r;
            hipLaunchKernelGGL(k_init_cells, dim3((unsigned)N), dim3(64), 0, ctx->stream, N, M, D, n_cells,
                               do_split, first ? 1 : 0, st->v, m->mean);
            return launch_ok("k_init_cells");
        };
        ctx->kernels = saved_kernels;
        if ((rc = dev_pass(1, 1 < M ? 1 : 0, true))) return finish(rc);
        int nc = 1;
        while (nc < M) {
            nc = (2 * nc < M) ? 2 * nc : M;
            for (int it = 0; it < 3; it++)
                if ((rc = dev_pass(nc, (it == 2 && nc < M) ? 1 : 0, false))) return finish(rc);
        }
        // the final means, for the last (direct-form) pass's bookkeeping below
        HIP_TRY(hipMemcpyAsync(cells.data(), m->mean, (size_t)G * D * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(stream_sync(ctx));
        first_pass = false;
        dev_done = true;
    }
    if (!dev_done) {
    // one cell per state: the mean of the state's frames
    if ((rc = pass(1, false))) return finish(rc);
    for (int k = 0; k < N; k++)
        for (int l = 0; l < D; l++)
            cells[((size_t)k * M) * D + l] = num_mu[((size_t)k * M) * D + l] / num_c[(size_t)k * M];
    int n_cells = 1;
    while (n_cells < M) {
        for (int k = 0; k < N; k++) {
            double *ck = cells.data() + (size_t)k * M * D;
            if (2 * n_cells < M) {
                for (int i = 0; i < n_cells; i++) split_cell(ck, i, n_cells + i);
            } else {
                order_desc(dist.data() + (size_t)k * M, n_cells);
                for (int i = 0; i < M - n_cells; i++) split_cell(ck, idx[i], n_cells + i);
            }
        }
        n_cells = (2 * n_cells < M) ? 2 * n_cells : M;
        for (int it = 0; it < 3; it++) { // TF:1043
            if ((rc = pass(n_cells, false))) return finish(rc);
            for (int k = 0; k < N; k++) {
                double *ck = cells.data() + (size_t)k * M * D;
                for (int j = 0; j < n_cells; j++)
                    for (int l = 0; l < D; l++)
                        ck[(size_t)j * D + l] =
                            num_mu[((size_t)k * M + j) * D + l] / num_c[(size_t)k * M + j];
                // empty cells are re-seeded from the cells with the largest distortion (TF:1236-1270)
                order_desc(dist.data() + (size_t)k * M, n_cells);
                int i = 0;
                for (int j = 0; j < n_cells; j++)
                    if (num_c[(size_t)k * M + j] == 0.0) split_cell(ck, idx[i++], j);
            }
        }
    }
    } // (!dev_done)
    // per-cell variance around the final means and cell weights (TF:883-933)
    if ((rc = pass(M, true))) return finish(rc);
    std::vector<double> iv((size_t)G * D), dt((size_t)G);
    for (int k = 0; k < N; k++) {
        double dur = 0.0, sum = 0.0;
        for (int j = 0; j < M; j++) dur += num_c[(size_t)k * M + j];
        for (int j = 0; j < M; j++) {
            const size_t g = (size_t)k * M + j;
            double d = 1.0;
            for (int l = 0; l < D; l++) {
                double v = num_var[g * D + l] / num_c[g];
                if (v < FLOOR) v = FLOOR;
                iv[g * D + l] = v;
            }
            for (int l = 0; l < D; l++) d *= iv[g * D + l];
            dt[g] = d;
            for (int l = 0; l < D; l++) iv[g * D + l] = 1.0 / iv[g * D + l];
            double w = num_c[g] / dur;
            if (w < FLOOR) w = FLOOR;
            cw[g] = w;
            sum += w;
        }
        for (int j = 0; j < M; j++) cw[(size_t)k * M + j] /= sum;
    }
    ctx->kernels = saved_kernels;
    rc = ghmm_model_set(ctx, m, A.data(), cw.data(), cells.data(), iv.data(), dt.data());
    ghmm_stats_destroy(ctx, st);
    return rc;
}

extern "C" int ghmm_score(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, double *loglik_host)
{
    int rc = use(ctx);
    if (rc || (rc = check_pair(m, c))) return rc;
    ARG_CHECK(loglik_host || c->U == 0, "null destination");
    if ((rc = ws_frames(ctx, m, c, false)) || (rc = ws_fb(ctx, m, c))) return rc;
    if ((rc = run_emission(ctx, m, c, ctx->robust ? 1 : 0, false))) return rc;
    if ((rc = run_forward(ctx, m, c, false, true))) return rc;
    if (c->U)
        HIP_TRY(hipMemcpyAsync(loglik_host, ctx->loglik, (size_t)c->U * 8, hipMemcpyDeviceToHost,
                               ctx->stream));
    HIP_TRY(stream_sync(ctx));
    return GHMM_OK;
}

// ------------------------------------------------------ several feature streams

static int check_streams(ghmm_ctx *ctx, ghmm_model *const *models, ghmm_corpus *const *corpora, int P)
{
    ARG_CHECK(models && corpora && P >= 1 && P <= GHMM_MAX_STREAMS, "bad stream arguments");
    for (int p = 0; p < P; p++) {
        int rc = check_pair(models[p], corpora[p]);
        if (rc) return rc;
        if (models[p]->N != models[0]->N) {
            ghmm_set_error("stream %d has %d states, stream 0 has %d", p, models[p]->N, models[0]->N);
            return GHMM_ERR_ARG;
        }
        if (corpora[p]->U != corpora[0]->U || corpora[p]->len != corpora[0]->len) {
            ghmm_set_error("stream %d: utterance count or lengths differ from stream 0", p);
            return GHMM_ERR_ARG;
        }
    }
    if (P > 1 && ctx->robust) {
        ghmm_set_error("GHMM_OPT_ROBUST is not available with several feature streams");
        return GHMM_ERR_UNSUPPORTED;
    }
    return GHMM_OK;
}

// emission of every stream; ctx->b ends up holding the product, ctx->post_s[p] stream p's posteriors
static int streams_emission(ghmm_ctx *ctx, ghmm_model *const *models, ghmm_corpus *const *corpora, int P,
                            bool want_post)
{
    int rc;
    const size_t FN = (size_t)corpora[0]->F * models[0]->N;
    if ((int)ctx->post_s.size() < P) {
        ctx->post_s.resize((size_t)P, nullptr);
        ctx->cap_post_s.resize((size_t)P, 0);
    }
    for (int p = 0; p < P; p++) {
        if ((rc = ws_frames(ctx, models[p], corpora[p], false))) return rc;
        if (want_post &&
            (rc = dev_grow(&ctx->post_s[p], &ctx->cap_post_s[p], (size_t)corpora[p]->F * models[p]->N * models[p]->M)))
            return rc;
    }
    if ((rc = dev_grow(&ctx->b_stream, &ctx->cap_b_stream, FN))) return rc;
    double *const b_all = ctx->b, *const post_own = ctx->post;
    for (int p = 0; p < P && !rc; p++) {
        ctx->b = p == 0 ? b_all : ctx->b_stream;
        ctx->post = want_post ? ctx->post_s[p] : nullptr;
        rc = run_emission(ctx, models[p], corpora[p], 0, want_post);
        if (!rc && p > 0 && FN) {
            long long blocks = ((long long)FN + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            kscope ks(ctx, GHMM_K_EMISSION);
            hipLaunchKernelGGL(k_mul_streams, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (long long)FN,
                               b_all, ctx->b_stream);
            rc = launch_ok("k_mul_streams");
        }
    }
    ctx->b = b_all;
    ctx->post = post_own;
    // the product belongs to stream 0's (model, corpus) pair as far as the row API is concerned
    ctx->em_m = models[0];
    ctx->em_c = corpora[0];
    ctx->em_epoch = models[0]->epoch;
    ctx->N = models[0]->N;
    ctx->G = models[0]->N * models[0]->M;
    return rc;
}

extern "C" int ghmm_estep_streams(ghmm_ctx *ctx, ghmm_model *const *models, ghmm_corpus *const *corpora,
                                  int P, ghmm_stats *const *stats)
{
    int rc = use(ctx);
    if (rc || (rc = check_streams(ctx, models, corpora, P))) return rc;
    ARG_CHECK(stats, "null statistics");
    for (int p = 0; p < P; p++)
        if ((rc = check_stats(models[p], stats[p]))) return rc;
    if (P == 1) return ghmm_estep(ctx, models[0], corpora[0], stats[0]);
    if ((rc = streams_emission(ctx, models, corpora, P, true))) return rc;
    if ((rc = ws_fb(ctx, models[0], corpora[0]))) return rc;
    if ((rc = run_forward(ctx, models[0], corpora[0], true))) return rc;
    if ((rc = run_backward(ctx, models[0], corpora[0], false))) return rc;
    // calc_mix_param per stream (TF:306-315) with the common gamma; the transition sums, den_c,
    // log P and the exemplar count go into every stream's vector
    double *const post_own = ctx->post;
    for (int p = 0; p < P && !rc; p++) {
        ctx->post = ctx->post_s[p];
        rc = run_accumulate(ctx, models[p], corpora[p], stats[p]);
    }
    ctx->post = post_own;
    return rc;
}

extern "C" int ghmm_score_streams(ghmm_ctx *ctx, ghmm_model *const *models, ghmm_corpus *const *corpora,
                                  int P, double *loglik_host)
{
    int rc = use(ctx);
    if (rc || (rc = check_streams(ctx, models, corpora, P))) return rc;
    if (P == 1) return ghmm_score(ctx, models[0], corpora[0], loglik_host);
    ghmm_corpus *c = corpora[0];
    ARG_CHECK(loglik_host || c->U == 0, "null destination");
    if ((rc = streams_emission(ctx, models, corpora, P, false))) return rc;
    if ((rc = ws_fb(ctx, models[0], c)) || (rc = run_forward(ctx, models[0], c))) return rc;
    if (c->U)
        HIP_TRY(hipMemcpyAsync(loglik_host, ctx->loglik, (size_t)c->U * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(stream_sync(ctx));
    return GHMM_OK;
}

extern "C" int ghmm_score_batch(ghmm_ctx *ctx, ghmm_model *const *models, int n_models,
                                ghmm_corpus *c, double *loglik_host)
{
    int rc = use(ctx);
    if (rc) return rc;
    ARG_CHECK(models && n_models > 0 && c, "null argument");
    ARG_CHECK(loglik_host || c->U == 0, "null destination");
    for (int k = 0; k < n_models; k++) ARG_CHECK(models[k], "null model");
    const int M = models[0]->M, D = models[0]->D;
    int NS = 0, Nmax = 0;
    for (int k = 0; k < n_models; k++) {
        if (models[k]->M != M || models[k]->D != D) {
            ghmm_set_error("ghmm_score_batch: every model must have the same M and D");
            return GHMM_ERR_UNSUPPORTED;
        }
        NS += models[k]->N;
        if (models[k]->N > Nmax) Nmax = models[k]->N;
    }
    if (D != c->D) {
        ghmm_set_error("models have %d coefficients per frame, corpus has %d", D, c->D);
        return GHMM_ERR_ARG;
    }
    if (c->U == 0) return GHMM_OK;
    if (Nmax > 64) { // the one-launch vocabulary loop holds one state per lane: word by word instead
        for (int k = 0; k < n_models; k++)
            if ((rc = ghmm_score(ctx, models[k], c, loglik_host + (size_t)k * c->U))) return rc;
        return GHMM_OK;
    }
    if (ctx->robust) {
        // per-frame normalisation couples the models' densities; score them one by one
        for (int k = 0; k < n_models; k++)
            if ((rc = ghmm_score(ctx, models[k], c, loglik_host + (size_t)k * c->U))) return rc;
        return GHMM_OK;
    }
    // The concatenated model: NS states x M mixtures (transition matrix unused).  It and the
    // pass's tables live in the context from one call to the next (a recogniser scores batch after
    // batch against one vocabulary: creating and freeing twenty device buffers per call was 0.75 ms
    // of a 0.9 ms call); the words' parameters are gathered into it by ONE launch.
    if (ctx->bt_cat && (ctx->bt_cat->N != NS || ctx->bt_cat->M != M || ctx->bt_cat->D != D)) {
        ghmm_model_destroy(ctx, ctx->bt_cat);
        ctx->bt_cat = nullptr;
    }
    if (!ctx->bt_cat) {
        if ((rc = ghmm_model_create(ctx, NS, M, D, &ctx->bt_cat))) return rc;
        HIP_TRY(hipMemsetAsync(ctx->bt_cat->A, 0, (size_t)NS * NS * 8, ctx->stream));
    }
    ghmm_model *cat = ctx->bt_cat;
    std::vector<fwd_model> tab((size_t)n_models);
    std::vector<gather_src> src((size_t)n_models);
    {
        int go = 0, so = 0;
        for (int k = 0; k < n_models; k++) {
            const ghmm_model *m = models[k];
            tab[k].A = m->A;
            tab[k].N = m->N;
            tab[k].bo = so;
            src[k].c = m->c; src[k].mean = m->mean; src[k].inv_var = m->inv_var; src[k].det = m->det;
            src[k].g0 = go; src[k].ng = m->N * M;
            go += m->N * M;
            so += m->N;
        }
    }
    const size_t tab_bytes = tab.size() * sizeof(fwd_model), src_bytes = src.size() * sizeof(gather_src);
    if ((rc = dev_grow(&ctx->bt_tab, &ctx->cap_bt_tab, tab_bytes + src_bytes + 16))) return rc;
    fwd_model *dtab = (fwd_model *)ctx->bt_tab;
    gather_src *dsrc = (gather_src *)(ctx->bt_tab + ((tab_bytes + 15) / 16) * 16);
    if ((rc = dev_grow(&ctx->bt_scale, &ctx->cap_bt_scale, (size_t)n_models * c->F + 1)) ||
        (rc = dev_grow(&ctx->bt_sinv, &ctx->cap_bt_sinv, (size_t)n_models * c->F + 1)) ||
        (rc = dev_grow(&ctx->bt_ll, &ctx->cap_bt_ll, (size_t)n_models * c->U)))
        return rc;
    // (the tables leave pageable host vectors: the copies complete before the call returns them)
    HIP_TRY(hipMemcpyAsync(dtab, tab.data(), tab_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dsrc, src.data(), src_bytes, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_gather_models, dim3((unsigned)n_models), dim3(256), 0, ctx->stream, D, dsrc, cat->c,
                       cat->mean, cat->inv_var, cat->det);
    if ((rc = launch_ok("k_gather_models"))) return rc;
    cat->epoch++;
    cat->prep_mark = ++ctx->launch_mark;
    if ((rc = model_prepare(ctx, cat, true)) || (rc = ws_frames(ctx, cat, c, false)) ||
        (rc = run_emission(ctx, cat, c, 0, false)))
        return rc;
    {
        const int L = Nmax <= 16 ? 16 : Nmax <= 32 ? 32 : 64, gpw = WAVE / L;
        const unsigned blocks = (unsigned)((c->U + gpw - 1) / gpw);
        kscope ks(ctx, GHMM_K_FORWARD);
        GHMM_BY_LANES(L, hipLaunchKernelGGL(k_forward_multi<LL>, dim3(blocks, (unsigned)n_models), dim3(WAVE), 0,
                                            ctx->stream, c->U, NS, c->F, dtab, ctx->b, c->off, ctx->bt_scale,
                                            ctx->bt_sinv, ctx->bt_ll, ctx->sink, c->order));
    }
    if ((rc = launch_ok("k_forward_multi"))) return rc;
    HIP_TRY(hipMemcpyAsync(loglik_host, ctx->bt_ll, (size_t)n_models * c->U * 8, hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(stream_sync(ctx));
    ctx->b_is_log = true; // the workspace b belongs to the concatenated model: not reusable
    return GHMM_OK;
}

extern "C" int ghmm_viterbi(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, int32_t *path_host,
                            double *score_host)
{
    int rc = use(ctx);
    if (rc || (rc = check_pair(m, c))) return rc;
    ARG_CHECK((path_host && score_host) || c->U == 0, "null destination");
    int L;
    if ((rc = fb_lanes(m, &L))) return rc;
    ARG_CHECK(m->N <= 255, "too many states for byte back-pointers");
    if ((rc = ws_frames(ctx, m, c, false))) return rc;
    if ((rc = dev_grow(&ctx->psi, &ctx->cap_psi, (size_t)c->F * (L ? L : m->N) + 16))) return rc; // rows of L (N) bytes
    if ((rc = dev_grow(&ctx->path, &ctx->cap_path, (size_t)c->F))) return rc;
    if ((rc = run_emission(ctx, m, c, 2, false))) return rc;
    if (c->U) {
        const int gpw = L ? WAVE / L : 1;
        const unsigned blocks = (unsigned)((c->U + gpw - 1) / gpw);
        if (L == 0) {
            if ((rc = wide_band_flag(ctx, m, m->logA, -INFINITY))) return rc;
            kscope ks(ctx, GHMM_K_VITERBI);
            hipLaunchKernelGGL(k_viterbi_wide, dim3(blocks), dim3(WAVE), (size_t)2 * m->N * sizeof(double),
                               ctx->stream, m->N, c->U, m->logA, ctx->b, c->off, ctx->psi, ctx->path, ctx->loglik,
                               c->order, ctx->wide_flag);
        } else {
            kscope ks(ctx, GHMM_K_VITERBI);
            GHMM_BY_LANES(L, hipLaunchKernelGGL(k_viterbi<LL>, dim3(blocks), dim3(WAVE), 0, ctx->stream, m->N, c->U,
                                                m->logA, ctx->b, c->off, ctx->psi, ctx->path, ctx->loglik,
                                                ctx->sink, c->order));
        }
        if ((rc = launch_ok("k_viterbi"))) return rc;
        HIP_TRY(hipMemcpyAsync(score_host, ctx->loglik, (size_t)c->U * 8, hipMemcpyDeviceToHost,
                               ctx->stream));
        if (c->F && (rc = d2h_pageable(ctx, path_host, ctx->path, (size_t)c->F, true))) return rc;
    }
    HIP_TRY(stream_sync(ctx));
    return GHMM_OK;
}

// ------------------------------------------------------------ collective (RCCL)
// One sum of the statistics vector over ranks per EM iteration.  RCCL is
// resolved at run time from librccl.so.1: a single-GPU process never loads it, and a process
// that already holds an RCCL (e.g. torch's) shares that copy instead of mapping a second one.
#include <dlfcn.h>
#include <rccl/rccl.h> // types and enums; the five entry points come from dlsym
#include <sys/stat.h>
#include <unistd.h>

#include <mutex>

namespace {
struct rccl_api {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = {0};
};
rccl_api g_rccl;          // written once under g_rccl_once, read-only afterwards
std::once_flag g_rccl_once;

const rccl_api *rccl()
{
    std::call_once(g_rccl_once, [] {
        const char *env = getenv("GHMM_RCCL_LIB");
        const char *names[] = {env, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.handle) break;
            snprintf(g_rccl.why, sizeof g_rccl.why, "%s", dlerror());
        }
        if (!g_rccl.handle) return;
        g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.handle, "ncclGetUniqueId");
        g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.handle, "ncclCommInitRank");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.handle, "ncclCommDestroy");
        g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(g_rccl.handle, "ncclAllReduce");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.handle, "ncclGetErrorString");
        if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce ||
            !g_rccl.GetErrorString) {
            snprintf(g_rccl.why, sizeof g_rccl.why, "librccl lacks an entry point");
            g_rccl.handle = nullptr;
        }
    });
    return g_rccl.handle ? &g_rccl : nullptr;
}
} // namespace

struct ghmm_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

#define RCCL_TRY(api, expr)                                                                 \
    do {                                                                                    \
        ncclResult_t r_ = (expr);                                                           \
        if (r_ != ncclSuccess) {                                                            \
            ghmm_set_error("%s failed: %s", #expr, (api)->GetErrorString(r_));              \
            return GHMM_ERR_HIP;                                                            \
        }                                                                                   \
    } while (0)

static const rccl_api *rccl_or_error()
{
    const rccl_api *api = rccl();
    if (!api) ghmm_set_error("RCCL is not available: %s", g_rccl.why);
    return api;
}

extern "C" int ghmm_comm_unique_id(void *id_bytes)
{
    ARG_CHECK(id_bytes, "null id");
    static_assert(sizeof(ncclUniqueId) == GHMM_COMM_ID_BYTES, "id size");
    const rccl_api *api = rccl_or_error();
    if (!api) return GHMM_ERR_UNSUPPORTED;
    ncclUniqueId id;
    RCCL_TRY(api, api->GetUniqueId(&id));
    memcpy(id_bytes, &id, sizeof id);
    return GHMM_OK;
}

extern "C" int ghmm_comm_create(ghmm_ctx *ctx, const void *id_bytes, int rank, int world, ghmm_comm **out)
{
    int rc = use(ctx);
    if (rc) return rc;
    ARG_CHECK(out && id_bytes && world > 0 && rank >= 0 && rank < world, "bad arguments");
    *out = nullptr;
    const rccl_api *api = rccl_or_error();
    if (!api) return GHMM_ERR_UNSUPPORTED;
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    ghmm_comm *cm = new (std::nothrow) ghmm_comm();
    if (!cm) return GHMM_ERR_ALLOC;
    cm->rank = rank;
    cm->world = world;
    cm->device = ctx->device;
    ncclResult_t r = api->CommInitRank(&cm->comm, world, id, rank); // on the context's device (use())
    if (r != ncclSuccess) {
        ghmm_set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, ctx->device,
                       api->GetErrorString(r));
        delete cm;
        return GHMM_ERR_HIP;
    }
    *out = cm;
    return GHMM_OK;
}

extern "C" int ghmm_comm_create_file(ghmm_ctx *ctx, const char *path, int rank, int world, double timeout_s,
                                     ghmm_comm **out)
{
    ARG_CHECK(path && *path && out, "bad arguments");
    unsigned char id[GHMM_COMM_ID_BYTES];
    int rc;
    if (rank == 0 && (rc = ghmm_comm_unique_id(id))) return rc;
    // the id travels by the host-only file protocol of ghmm_rendezvous.c (nonce handshake,
    // bounded waits, rank 0 removes the file once every rank holds the id)
    if ((rc = ghmm_rendezvous_file(path, rank, world, timeout_s, id))) return rc;
    return ghmm_comm_create(ctx, id, rank, world, out);
}

extern "C" void ghmm_comm_destroy(ghmm_comm *cm)
{
    if (!cm) return;
    const rccl_api *api = rccl();
    if (api && cm->comm) {
        (void)hipSetDevice(cm->device);
        (void)api->CommDestroy(cm->comm);
    }
    delete cm;
}

extern "C" int ghmm_comm_rank(const ghmm_comm *cm) { return cm ? cm->rank : -1; }
extern "C" int ghmm_comm_size(const ghmm_comm *cm) { return cm ? cm->world : -1; }

extern "C" int ghmm_stats_allreduce(ghmm_ctx *ctx, ghmm_stats *s, ghmm_comm *cm)
{
    int rc = use(ctx);
    if (rc) return rc;
    ARG_CHECK(s && cm && cm->comm, "null argument");
    ARG_CHECK(cm->device == ctx->device, "communicator and context are on different devices");
    const rccl_api *api = rccl_or_error();
    if (!api) return GHMM_ERR_UNSUPPORTED;
    s->mbox_valid = false; // (the sum over ranks replaces what k_reduce_all left in the mailbox)
    // in place, on the stream that carries the E-step before it and the M-step after it
    RCCL_TRY(api, api->AllReduce(s->v, s->v, s->n, ncclDouble, ncclSum, cm->comm, ctx->stream));
    return GHMM_OK;
}
