/*
 * ghmm.h — C ABI of the MI355X-native continuous-density GMM-HMM core.
 *
 * This is the drop-in boundary for the diagonal-covariance hot path of
 * edielsonpf/speech-recognition-hmm-continuous.  The reference has no library
 * or FFI layer: its numerical core is a set of file-local C functions called
 * from main().  Every entry point below names the reference
 * function(s) it replaces.  Path aliases:
 *
 *   TF = train/source/hmm-fs/hmm_continuous_fs.c            (trainer, diagonal)
 *   RF = test/source/recognition-fs/recognition_continuous_fs.c (recogniser, diagonal)
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns GHMM_OK (0) or an error
 *     code and never calls exit(); ghmm_last_error() holds the detail text.
 *   - all arithmetic on the path is IEEE double ("f64"), like the reference.
 *   - a model is held as flat struct-of-arrays (the reference's `struct state`
 *     TF:53-64 is array-of-structs with fixed capacity):
 *         A[N*N] row-major, c[N*M], mean[N*M*D], inv_var[N*M*D], det[N*M]
 *     with the reference's meaning: inv_var = 1/sigma^2 (TF:2012), det = prod
 *     sigma^2 of the NON-inverted variances (TF:1976), exactly what a .hmm file holds.
 *   - frames are row-major X[F][D] (F = all frames of all utterances, back to
 *     back), i.e. the payload order of the reference's .perfil files (TF:527-544).
 *   - per-frame outputs are frame-major: b[F][N], post[F][N*M], alpha[F][N] ...
 *     (the reference keeps them state-major with a 500-frame cap, TF:107-114).
 *   - one ghmm_ctx per GPU per host thread; no global mutable state.
 *   - the library needs a gfx950 device: ghmm_ctx_create fails with
 *     GHMM_ERR_NODEVICE otherwise.  There is no CPU fallback.
 */
#ifndef GHMM_H
#define GHMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GHMM_VERSION 200

enum {
    GHMM_OK = 0,
    GHMM_ERR_ARG = 1,         /* bad argument / shape mismatch */
    GHMM_ERR_ALLOC = 2,       /* host or device allocation failed */
    GHMM_ERR_HIP = 3,         /* a HIP runtime call failed */
    GHMM_ERR_NODEVICE = 4,    /* no usable gfx950 device */
    GHMM_ERR_UNSUPPORTED = 5, /* valid request outside what is built (e.g. more than 512 states in the recursions) */
    GHMM_ERR_IO = 6,          /* file could not be opened / read / written */
    GHMM_ERR_FORMAT = 7       /* file content is not a .perfil / .hmm */
};

const char *ghmm_strerror(int code);
const char *ghmm_last_error(void); /* thread-local detail of the last failure */
int ghmm_version(void);

/* ------------------------------------------------------------------ context */

typedef struct ghmm_ctx ghmm_ctx;

/* `hip_stream` may be NULL (the library creates its own stream) or a
 * hipStream_t owned by the caller (e.g. torch's current stream); every kernel
 * and copy of this context is issued on it. */
int ghmm_ctx_create(int device, void *hip_stream, ghmm_ctx **out);
void ghmm_ctx_destroy(ghmm_ctx *ctx);
int ghmm_ctx_sync(ghmm_ctx *ctx);

enum {
    /* band of transitions that receive statistics: i <= j <= i+delta.
     * The reference hard-codes DELTA 1 (TF:38, TF:1601). Default 1. */
    GHMM_OPT_DELTA = 1,
    /* 0 (default): emission densities in the reference's linear domain
     *    (exp() underflows exactly where the reference's does, TF:1821-1836);
     * 1: per-frame max-normalised densities (the log of the normaliser is added back
     *    into the log-likelihood): finite where the reference's densities underflow as a
     *    whole frame.  It is still a linear-domain recursion: a frame whose REACHABLE states
     *    lie more than ~308 decades below its best state ends the utterance as in the
     *    reference (profiles/fuzz_robust.py). */
    GHMM_OPT_ROBUST = 2,
    /* 0 auto, 1 vector-ALU kernels and the reference's order of the recursions (calc_alpha,
     * then calc_beta scaled by its c_t, one pass each), 2 MFMA (f64 16x16x4) kernels and the
     * forward / backward recursions side by side (what auto picks); 3 is a measurement variant
     * of 2 (statistics kernel with its operands straight from HBM instead of staged through LDS) */
    GHMM_OPT_KERNELS = 3,
    /* 1: bracket every kernel with HIP events on the context's stream */
    GHMM_OPT_TIMING = 4,
    /* number of frame-block partial sums kept by the statistics kernel (0 auto) */
    GHMM_OPT_PARTIALS = 5,
    /* compute units the one-block-per-CU kernels size their grids for (0 = all of the
     * device's, the default): for a caller whose stream is restricted to part of the device
     * (hipExtStreamCreateWithCUMask) */
    GHMM_OPT_CUS = 6,
    /* read-only (ghmm_ctx_get_option; synchronises the stream): utterances the last gamma / xi
     * pass of the default tier took again in the reference's own order of operations — no path
     * into the last state, or forward and backward mass more than 200 decades apart at some
     * frame (ghmm_pair.hpp, RANGE).  0 on data the model fits. */
    GHMM_OPT_REFORDER_COUNT = 7,
    /* matrix-core tier, Gaussians too ill-conditioned for the expanded sums although their
     * variances are not at the floor ("class 2"): 0 (default) their direct-form sums come from
     * the vector-ALU statistics kernel, launched while the host has recently seen such a
     * Gaussian, and from an exact recomputation inside the reduction otherwise; 1 always
     * launch that kernel; 2 never (always the recomputation).  All three are exact; a
     * measurement / test switch. */
    GHMM_OPT_VEC_STATS = 8,
    /* mixture posteriors (gaus_probab_dens, TF:110) written with non-temporal stores: 0 (default)
     * when they are at most 1 GiB, 1 always, 2 never.  Same bytes either way; a measurement
     * switch (profiles/tools/nt_ab.py). */
    GHMM_OPT_NT_POST = 9,
    /* ghmm_estep's recursions: both scans and the gamma / xi pass in ONE launch (a block scans its
     * utterances with two waves, then all of its waves take the chunks) whenever A is band-diagonal
     * — 0 (default) and 1; 2 = the separate launches.  Same operations either way; a measurement
     * switch (profiles/tools/fused_ab.py). */
    GHMM_OPT_FUSED_SCAN = 10
};
int ghmm_ctx_set_option(ghmm_ctx *ctx, int option, int64_t value);
int ghmm_ctx_get_option(ghmm_ctx *ctx, int option, int64_t *value);

/* kernel ids for ghmm_ctx_kernel_time() */
enum {
    GHMM_K_EMISSION = 0,
    GHMM_K_FORWARD = 1,  /* forward recursion; inside ghmm_estep on a band-diagonal A the one launch that
                          * holds both recursions and the gamma / xi pass (then GHMM_K_BACKWARD counts nothing) */
    GHMM_K_BACKWARD = 2, /* backward recursion's share: gamma / xi pass and the fix-up launch */
    GHMM_K_MIXSTATS = 3,
    GHMM_K_REDUCE = 4,
    GHMM_K_MSTEP = 5,
    GHMM_K_VITERBI = 6,
    GHMM_K_PREPARE = 7,
    GHMM_K_COUNT = 8
};
/* Sum of HIP-event durations and launch count since the last reset (needs
 * GHMM_OPT_TIMING = 1).  Synchronises the context's stream. */
int ghmm_ctx_kernel_time(ghmm_ctx *ctx, int kernel, double *total_ms, int64_t *launches);
int ghmm_ctx_kernel_time_reset(ghmm_ctx *ctx);
const char *ghmm_kernel_name(int kernel);

/* -------------------------------------------------------------------- model */

typedef struct ghmm_model ghmm_model;

/* Device-resident model, one feature stream (the reference's param_number P;
 * every BASELINE configuration uses P = 1). Replaces `struct state
 * state_mix[P][N]` + `transition_probab[N][N]` (TF:104, TF:146). */
int ghmm_model_create(ghmm_ctx *ctx, int N, int M, int D, ghmm_model **out);
void ghmm_model_destroy(ghmm_ctx *ctx, ghmm_model *m);
/* host -> device; also rebuilds the derived per-Gaussian constants
 * (pow(2*pi, D/2) * sqrt(|det|), TF:1821-1827). */
int ghmm_model_set(ghmm_ctx *ctx, ghmm_model *m, const double *A, const double *c,
                   const double *mean, const double *inv_var, const double *det);
/* device -> host (synchronises); any pointer may be NULL */
int ghmm_model_get(ghmm_ctx *ctx, ghmm_model *m, double *A, double *c, double *mean,
                   double *inv_var, double *det);
int ghmm_model_dims(const ghmm_model *m, int *N, int *M, int *D);

/* ------------------------------------------------------------------- corpus */

typedef struct ghmm_corpus ghmm_corpus;

/* A batch of utterances resident in HBM.  `len[u]` = frames of utterance u.
 * _create copies host frames to the device; _wrap adopts a device pointer the
 * caller keeps alive (no copy). Replaces the per-frame fread loop TF:282-288. */
int ghmm_corpus_create(ghmm_ctx *ctx, const double *X_host, const int32_t *len, int n_utt, int D,
                       ghmm_corpus **out);
int ghmm_corpus_wrap(ghmm_ctx *ctx, const double *X_dev, const int32_t *len, int n_utt, int D,
                     ghmm_corpus **out);
void ghmm_corpus_destroy(ghmm_ctx *ctx, ghmm_corpus *c);
int64_t ghmm_corpus_frames(const ghmm_corpus *c);
int ghmm_corpus_utterances(const ghmm_corpus *c);

/* creating_initial_model (TF:732-1317) on the device, from a corpus resident in HBM:
 * uniform segmentation, LBG splitting (x1.005 / x0.995), three nearest-mean passes per
 * split, per-cell variance floored at 1e-5, weights floored and renormalised, one-step
 * left-to-right transitions.  The distance / accumulation passes run on the GPU (they
 * are the statistics kernels fed with one-hot weights), the cell bookkeeping (splitting
 * order, empty cells) on the host.  Fills model `m` (its N, M, D).  Synchronises.
 */
int ghmm_model_init(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c);

/* ---------------------------------------------------- sufficient statistics */

/* Flat Baum-Welch accumulator vector — the ONLY thing that crosses GPUs
 * (one all-reduce(SUM) per EM iteration).  Layout, in doubles:
 *   num_a[N*N]   TF:1614  (only the band i <= j <= i+delta is ever non-zero)
 *   den_a[N]     TF:1618
 *   den_c[N]     TF:1660
 *   num_c[N*M]   TF:1716
 *   num_mu[N*M*D]  TF:1718
 *   num_var[N*M*D] TF:1720-1722 (around the OLD mean)
 *   loglik       TF:318  (sum of per-utterance log P)
 *   n_utt        TF:320
 */
typedef struct ghmm_stats ghmm_stats;
size_t ghmm_stats_len(int N, int M, int D);
int ghmm_stats_create(ghmm_ctx *ctx, int N, int M, int D, ghmm_stats **out);
/* adopt caller-owned device memory of ghmm_stats_len() doubles (e.g. a torch
 * tensor that torch.distributed all-reduces in place) */
int ghmm_stats_wrap(ghmm_ctx *ctx, int N, int M, int D, double *dev_ptr, ghmm_stats **out);
void ghmm_stats_destroy(ghmm_ctx *ctx, ghmm_stats *s);
double *ghmm_stats_device_ptr(ghmm_stats *s);
int ghmm_stats_download(ghmm_ctx *ctx, ghmm_stats *s, double *host);
/* the two numbers the EM driver's stopping rule reads every iteration (TF:318-325):
 * out[0] = sum of log P over the utterances (`probab`), out[1] = utterance count
 * (`exemplar_number`) — a 16-byte download instead of the whole vector; straight behind an
 * E-step into a vector the library owns, a poll of pinned host memory the reduction kernel wrote
 * (returns as soon as the two numbers exist: the stream is NOT drained) */
int ghmm_stats_loglik(ghmm_ctx *ctx, ghmm_stats *s, double out[2]);
int ghmm_stats_upload(ghmm_ctx *ctx, ghmm_stats *s, const double *host);

/* ---------------------------------------------- the path, one row at a time */

/* calc_symbol_probab + calc_gaus, TF:1749-1841 (want_post = 1: also the
 * within-state mixture posteriors `gauss[i][j]`, TF:1773-1778) and RF:860-947
 * (want_post = 0).  Results stay in the context workspace. */
int ghmm_emission(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, int want_post);
/* calc_alpha TF:1380-1443 / RF:739-799 + calc_probability TF:1536-1553.
 * Models of up to 512 states (the reference's cap is 20, TF:41): one state per lane up to 64,
 * one wave per utterance with the states strided over its lanes beyond (GHMM_ERR_UNSUPPORTED
 * above 512; ghmm_viterbi: 255, its back-pointers are bytes). */
int ghmm_forward(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c);
/* calc_beta TF:1463-1516, fused with the per-utterance part of
 * calc_transition_probab TF:1577-1620 and calc_den_mix_coef TF:1642-1664 */
int ghmm_backward(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c);
/* calc_mix_param TF:1691-1727 over every frame, then the ordered reduction of
 * all partial sums into `stats` */
int ghmm_accumulate(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, ghmm_stats *stats);

/* workspace buffers readable with ghmm_fetch (all double, frame-major) */
enum {
    GHMM_BUF_B = 0,      /* b[F][N]        symbol_probab        TF:107 */
    GHMM_BUF_POST = 1,   /* post[F][N*M]   gaus_probab_dens     TF:110 */
    GHMM_BUF_ALPHA = 2,  /* alpha^[F][N]                        TF:112 */
    GHMM_BUF_BETA = 3,   /* beta^[F][N]                         TF:114; after ghmm_estep it is formed on
                          * this call (the E-step itself only needs gamma and xi) */
    GHMM_BUF_SCALE = 4,  /* c_t[F]         scaling_factor       TF:116 */
    GHMM_BUF_GAMMA = 5,  /* gamma[F][N] = alpha^*beta^/c_t      TF:1657 */
    GHMM_BUF_LOGLIK = 6, /* log P per utterance [U]             TF:1536 */
    GHMM_BUF_LOGNORM = 7 /* log of the per-frame normaliser [F] (GHMM_OPT_ROBUST) */
};
int ghmm_fetch(ghmm_ctx *ctx, int which, double *host, size_t n_doubles);
/* the same for doubles [first, first + n_doubles) of the buffer (e.g. a block of frames of a
 * b[F][N] too large for host memory: BASELINE's 2 000-state x 1 M-frame emission is 16 GB) */
int ghmm_fetch_range(ghmm_ctx *ctx, int which, size_t first, double *host, size_t n_doubles);

/* -------------------------------------------------- the path, batched/fused */

/* One E-step over the whole corpus: emission -> forward and backward recursions ->
 * gamma / xi -> statistics -> ordered reduction (TF:244-321).  `stats` is overwritten (the
 * zeroing of TF:244-270 is implied).  Asynchronous on the context's stream. */
int ghmm_estep(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, ghmm_stats *stats);
/* M-step from (possibly all-reduced) statistics, on the device, in place:
 * updating_transition_probab TF:1862-1889, updating_mix_param TF:1911-1955 with
 * changing_zero_coef TF:1338-1359, calc_det TF:1976 + inv_matrix TF:2012 as
 * called at TF:343-346.  num_a is read inside the band i <= j <= i + GHMM_OPT_DELTA only —
 * the only entries calc_transition_probab ever accumulates (TF:1601); a_ij outside it
 * becomes 0 / den_a = 0 as in the reference.  Asynchronous. */
int ghmm_mstep(ghmm_ctx *ctx, ghmm_model *m, ghmm_stats *stats);
/* Forward-algorithm score per utterance (RF:354-366): emission without
 * posteriors + the forward recursion + log P.  Synchronises, writes loglik[U] on the host.
 * (Only log P is kept: alpha^ and c_t stay in registers; ghmm_forward leaves them in the workspace.) */
int ghmm_score(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, double *loglik_host);
/* The recogniser's whole vocabulary loop (RF:326-374) in two launches: ONE emission launch
 * over the concatenated Gaussians of all `n_models` word models and ONE forward launch
 * over every (model, utterance) pair.  All models must share M and D (states may
 * differ).  loglik_host[k*U + u] = log P(utterance u | model k).  Synchronises. */
int ghmm_score_batch(ghmm_ctx *ctx, ghmm_model *const *models, int n_models, ghmm_corpus *c,
                     double *loglik_host);
/* Several feature streams (param_number P > 1): every recursion runs on the product over streams
 * of the emission densities, b_i(t) = prod_p b^p_i(t) in stream order (calc_alpha TF:1406-1409 /
 * 1429-1432, calc_beta TF:1501-1504, calc_transition_probab TF:1607-1610); calc_symbol_probab
 * (TF:278-288) and calc_mix_param (TF:306-315) run once per stream with that stream's own
 * mixtures and posteriors.  models[p] / corpora[p] / stats[p] = stream p: same states, same
 * utterances and lengths, own M_p and D_p; the transitions are models[0]'s.  stats[p] has the
 * single-stream layout (the common sums are written into every one), so that ghmm_mstep(models[p],
 * stats[p]) for every p is the M-step (TF:332-346: all of them write the same A).  The product
 * b is what GHMM_BUF_B then holds.  GHMM_OPT_ROBUST is not available with several streams. */
int ghmm_estep_streams(ghmm_ctx *ctx, ghmm_model *const *models, ghmm_corpus *const *corpora,
                       int n_streams, ghmm_stats *const *stats);
/* forward score per utterance of a P-stream model (RF:349-366) */
int ghmm_score_streams(ghmm_ctx *ctx, ghmm_model *const *models, ghmm_corpus *const *corpora,
                       int n_streams, double *loglik_host);

/* Max-plus lattice with the reference's one-hot start (RF:249-251) and
 * final-state termination (TF:1487, TF:1549); ties take the lowest predecessor.
 * ABSENT from the reference: defined by oracle/.
 * path_host[F] = state per frame, score_host[U] = best log score.  (The device keeps
 * one byte per frame; it is widened to int32 on the way into path_host.) */
int ghmm_viterbi(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, int32_t *path_host,
                 double *score_host);

/* -------------------------------------- several GPUs: the one collective */

/* Utterances shard data-parallel over ranks (one rank = one process or host thread with
 * its own ghmm_ctx on its own GPU); the accumulators are plain sums over utterances
 * (TF:1614, 1618, 1660, 1716-1722, 318-320), so the ONLY exchange per EM iteration is one
 * sum of the flat statistics vector over ranks: ncclAllReduce(ncclDouble, ncclSum) of RCCL,
 * in place, on the context's stream.  Every rank then applies the same
 * ghmm_mstep redundantly (identical inputs, identical models, no broadcast).  RCCL
 * (librccl.so.1) is loaded on the first ghmm_comm_* call; a process that never makes one
 * does not touch it. */
typedef struct ghmm_comm ghmm_comm;
#define GHMM_COMM_ID_BYTES 128
/* rank 0: a fresh rendezvous id (ncclGetUniqueId) to hand to every rank out of band */
int ghmm_comm_unique_id(void *id_bytes);
/* collective over all `world` ranks (ncclCommInitRank) */
int ghmm_comm_create(ghmm_ctx *ctx, const void *id_bytes, int rank, int world, ghmm_comm **out);
/* the same with the id passed through a file: rank 0 creates the id and writes `path`
 * (atomically, via rename), the other ranks wait up to timeout_s for it to appear; rank 0
 * removes the file once every rank has joined (ghmm_rendezvous_file below).  A `path` per job is
 * good practice, no longer a requirement. */
int ghmm_comm_create_file(ghmm_ctx *ctx, const char *path, int rank, int world, double timeout_s,
                          ghmm_comm **out);
/* the id exchange of ghmm_comm_create_file on its own — host code, no GPU, no RCCL (also in
 * libghmm_host.so): rank 0 passes its id IN and returns once every other rank has taken it;
 * the other ranks receive it in id_bytes.  Ranks announce themselves in `path`.join.<rank>
 * with a fresh nonce that the published `path` echoes, so a file left by an earlier job is
 * never taken for this job's id; a rank 0 that starts late is waited for; every wait ends
 * after timeout_s with GHMM_ERR_IO.  world = 1: returns at once, nothing is written. */
int ghmm_rendezvous_file(const char *path, int rank, int world, double timeout_s, void *id_bytes);
void ghmm_comm_destroy(ghmm_comm *comm);
int ghmm_comm_rank(const ghmm_comm *comm);
int ghmm_comm_size(const ghmm_comm *comm);
/* stats <- sum over ranks of stats, in place, asynchronous on the context's stream */
int ghmm_stats_allreduce(ghmm_ctx *ctx, ghmm_stats *stats, ghmm_comm *comm);
/* ghmm_model_init over a corpus sharded across ranks: the k-means sums of every pass are
 * all-reduced, every rank does the same cell bookkeeping and ends with the same model.
 * comm == NULL: this rank's corpus alone (= ghmm_model_init). */
int ghmm_model_init_comm(ghmm_ctx *ctx, ghmm_model *m, ghmm_corpus *c, ghmm_comm *comm);

/* ------------------------------------------------- host side: file formats */

#define GHMM_MAX_WORD 256

/* .perfil: int32 D, then T*D doubles, T implied by EOF (TF:527-581).
 * *X is malloc'ed; the caller frees it with ghmm_free(). */
int ghmm_perfil_read(const char *path, int *D, int *T, double **X);
int ghmm_perfil_write(const char *path, int D, int T, const double *X);
/* coefficient count and frame count from the header and the file size, nothing else read */
int ghmm_perfil_stat(const char *path, int *D, int *T);
void ghmm_free(void *p);

/* Length-balanced shard of `rank` among `world` (sort by length, deal in
 * turn): index[0 .. *n_out) = this rank's utterances in ascending index order (index has
 * room for (n_utt + world - 1) / world entries).  Frames per rank differ by at most the
 * longest utterance. */
int ghmm_shard_balanced(const int32_t *len, int n_utt, int rank, int world, int32_t *index,
                        int *n_out);

/* .hmm model file (writer TF:2043-2146, readers TF:604-711, RF:595-715), one
 * stream.  The reader accepts both a 4-byte and an 8-byte length prefix (the
 * shipped models come from a 32-bit build); the writer emits `len_bytes`
 * (8 = what a 64-bit build of the reference writes, or 4). Arrays are malloc'ed
 * by the reader. */
typedef struct ghmm_host_model {
    char word[GHMM_MAX_WORD];
    int N, M, D;
    double *A, *c, *mean, *inv_var, *det;
} ghmm_host_model;
int ghmm_host_model_alloc(ghmm_host_model *hm, int N, int M, int D);
void ghmm_host_model_free(ghmm_host_model *hm);
int ghmm_hmm_read(const char *path, ghmm_host_model *hm);
int ghmm_hmm_write(const char *path, const ghmm_host_model *hm, int len_bytes);
/* The same for models of several feature streams (the reference's param_number P, TF:2084-2099:
 * int M[P], int D[P], then per stream the states' mixtures): hm[p] = stream p with its own M and
 * D; word, N and A are common (the reader fills them into every hm[p]).  The reader takes up
 * to max_streams (<= GHMM_MAX_STREAMS) and reports the file's count. */
#define GHMM_MAX_STREAMS 8
int ghmm_hmm_read_streams(const char *path, ghmm_host_model *hm, int max_streams, int *n_streams);
int ghmm_hmm_write_streams(const char *path, const ghmm_host_model *hm, int n_streams, int len_bytes);

/* creating_initial_model TF:732-1317 (uniform segmentation, LBG splitting with
 * factors 1.005/0.995, three k-means passes, per-cell variance floored at 1e-5)
 * on utterances already in host memory.  Host code. */
int ghmm_init_model(const double *X, const int32_t *len, int n_utt, int N, int M, int D,
                    ghmm_host_model *hm);

/* -------------------------------------------- host side: synthetic corpora */

#define GHMM_SYNTH_SEED 20260104ull

/* ground truth: mean[N*M*D] ~ N(0, 2^2), stddev[N*M*D] ~ U[0.5, 1.5] */
int ghmm_synth_truth(uint64_t seed, int N, int M, int D, double *mean, double *stddev);
/* utterances first_utt .. first_utt+n_utt-1 of the corpus (seed): left-to-right
 * walk over the N states, one mixture per frame.  X holds sum(len)*D doubles. */
int ghmm_synth_utterances(uint64_t seed, int N, int M, int D, const double *mean,
                          const double *stddev, int64_t first_utt, int n_utt,
                          const int32_t *len, double *X);
/* starting model = truth perturbed by +-perturb (relative on stddev, in units
 * of stddev on the mean), uniform mixture weights, one-step left-to-right A */
int ghmm_synth_start_model(uint64_t seed, int N, int M, int D, const double *mean,
                           const double *stddev, double perturb, double *A, double *c,
                           double *mu0, double *inv_var0, double *det0);

#ifdef __cplusplus
}
#endif
#endif /* GHMM_H */
