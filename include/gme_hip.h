/*
 * gme_hip.h -- C ABI of libgme_hip.so, the MI355X (gfx950) implementation of the
 * per-frame-pair hot path of Samaretas/global-motion-estimation.
 *
 * The reference is pure Python and has no FFI; its boundary is the module surface
 * of global_motion_estimation/{bbme,motion,utils}.py.  Each entry point below names
 * the reference function(s) it replaces (paths relative to
 * /root/reference/global_motion_estimation/).  The Python mirror of that surface
 * (global-motion-estimation_amd/{bbme,motion,utils}.py) binds these symbols with
 * ctypes; INTEGRATION.md shows the stub a maintainer would add upstream.
 *
 * Conventions
 *   - plain C types only; every function returns 0 (GME_OK) or a negative GME_ERR_*;
 *     gme_last_error() gives the text for the calling thread's last failure.
 *   - host images are uint8, row-major, `stride` bytes between rows; motion fields
 *     are int32[h][w][2] with [..][0] = column (x) and [..][1] = row (y)
 *     displacement, position in `cur` minus position in `prev` (bbme.py:176-177).
 *   - the caller allocates every output; the library owns device memory inside the
 *     opaque gme_ctx / gme_seq objects; one HIP stream per context.  Every entry point locks
 *     its context for the whole call, so threads may share a context (their calls serialise;
 *     the reference is single-threaded); use one context per thread for
 *     concurrency.
 *   - integer results (motion vectors, masks, model fields, compensated frames,
 *     squared-error sums) are bit-exact with the reference; the normal-equation
 *     sums are bit-exact float64; the 3x3 solve stays on the host (NumPy) because
 *     LAPACK builds differ in the last bits.
 */
#ifndef GME_HIP_H
#define GME_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define GME_API __attribute__((visibility("default")))
#else
#define GME_API
#endif

typedef struct gme_ctx gme_ctx;   /* one device + one stream + scratch */
typedef struct gme_seq gme_seq;   /* a frame sequence resident in HBM + per-pair results */

enum {
    GME_OK = 0,
    GME_ERR_ARG = -1,        /* bad pointer / size / index (reference: IndexError, bbme.py:27,60) */
    /* -2 is unused (rounds 1-3 reserved it for "inexact": outside float32's exact-integer range, bbme.py:61-64, the kernels
     * sum in NumPy's float32 order instead of refusing) */
    GME_ERR_GEOMETRY = -3,   /* frame smaller than the search needs (reference: AssertionError, bbme.py:59) */
    GME_ERR_HIP = -4,        /* HIP runtime failure */
    GME_ERR_STATE = -5,      /* call order violated (e.g. fit before begin) */
    GME_ERR_NOMEM = -6
};

/* bbme.py:609-614 searching_procedures, bbme.py:608 pnorm_distances */
enum { GME_SEARCH_EXHAUSTIVE = 0, GME_SEARCH_THREESTEP = 1, GME_SEARCH_TWODLOG = 2, GME_SEARCH_DIAMOND = 3 };
enum { GME_NORM_MAE = 0, GME_NORM_MSE = 1 };

GME_API const char *gme_last_error(void);
GME_API int gme_device_count(void);
GME_API gme_ctx *gme_create(int device_id);
GME_API void gme_destroy(gme_ctx *ctx);
GME_API int gme_sync(gme_ctx *ctx);
GME_API int gme_device_info(gme_ctx *ctx, char *name, int name_len, int *cu_count, int *clock_khz);
/* PCI bus id of the context's device ("0000:05:00.0"); makes that device current and checks that it answers.  The
 * all-ranks-agree bring-up of the RCCL communicator (sequence.comm_init) publishes it in its first phase, so that a
 * lost device or two ranks on one device end in a clean refusal on every rank instead of inside ncclCommInitRank.
 * No reference counterpart (the reference is single-process, results.py:41-50). */
GME_API int gme_device_bus_id(gme_ctx *ctx, char *out, int out_len);
/* opaque handle of the context's HIP stream (hipStream_t), for callers that
 * want to order their own work or events after the library's */
GME_API void *gme_stream(gme_ctx *ctx);

/* Diagnostics of the context's last block-matching call (gme_bbme_u8, gme_seq_bbme, or the last
 * level search of a staged GME run): the kernel / tile shape / schedule the launch plan chose, e.g.
 * "k_exh_sea16p<3,6> tiles 2x4 persistent-dynamic grid 2048 lds 34864", and for the successive-
 * elimination kernels how many candidate patches the bound was applied to and how many it left
 * for exact evaluation (both 0 for kernels that evaluate every candidate, bbme.py:146-174), and how many
 * tiles it handed to the brute-force redo kernel because the bound pruned too little there.
 * No reference counterpart; tests use it to assert which kernel instance they exercised.
 * Any pointer may be NULL.  Synchronises. */
GME_API int gme_last_bbme_info(gme_ctx *ctx, char *plan, int plan_len, int64_t *patches, int64_t *surviving,
                       int64_t *redo_tiles);
/* Same call, one more figure: the patches the FIRST upper bound of each block left (what a single evaluation round
 * would have scored); gme_last_bbme_info's `surviving` is what the ordered two-round evaluation really scored
 * (the exactness argument of bbme.py:171 -- only a smaller key replaces the best one -- holds for any upper bound that
 * is a real candidate's cost, so tightening it between the rounds changes the work, never the result).  Synchronises. */
GME_API int gme_last_bbme_listed(gme_ctx *ctx, int64_t *listed);

/* HIP-event stopwatch on the context's stream (bench.py: kernel time of the timed region) */
GME_API int gme_timer_start(gme_ctx *ctx);
GME_API int gme_timer_stop(gme_ctx *ctx, float *elapsed_ms);   /* synchronises */

/* ---------------------------------------------------------------------------
 * Single-pair calls on host buffers (H2D, kernel, D2H inside the call).
 * ------------------------------------------------------------------------- */

/* bbme.get_motion_field (bbme.py:12-38) with the four searches (bbme.py:105-534).
 * mf_out: int32[H/bs][W/bs][2]. */
GME_API int gme_bbme_u8(gme_ctx *ctx, const uint8_t *prev, const uint8_t *cur, int H, int W, int stride,
                int block_size, int search_window, int procedure, int pnorm, int32_t *mf_out);

/* cv2.pyrDown as called by utils.get_pyramids (utils.py:34-51); dst is ((H+1)/2) x ((W+1)/2), tight. */
GME_API int gme_pyrdown_u8(gme_ctx *ctx, const uint8_t *src, int H, int W, int stride, uint8_t *dst);

/* motion.get_motion_field_affine (motion.py:139-157): int16[h][w][2], round-half-even. */
GME_API int gme_affine_field(gme_ctx *ctx, const double params[6], int h, int w, int16_t *mf_out);

/* motion.compensate_frame (motion.py:289-321); mf is int32[h][w][2]; out is H x W, tight. */
GME_API int gme_compensate_u8(gme_ctx *ctx, const uint8_t *frame, int H, int W, int stride,
                      const int32_t *mf, int h, int w, uint8_t *out);

/* integer core of utils.PSNR (utils.py:100-116): sum over pixels of (a-b)^2. */
GME_API int gme_sse_u8(gme_ctx *ctx, const uint8_t *a, const uint8_t *b, int H, int W, int stride_a,
               int stride_b, int64_t *sse_out);

/* ---------------------------------------------------------------------------
 * Sequence API: N frames resident in HBM; pair p = (frame p, frame p + fd) as in
 * results.py:41-48.  This is the batch form of the hot path and what bench.py times.
 * ------------------------------------------------------------------------- */
GME_API gme_seq *gme_seq_create(gme_ctx *ctx, int n_frames, int H, int W);
GME_API void gme_seq_destroy(gme_seq *seq);
GME_API int gme_seq_upload(gme_seq *seq, int first, int count, const uint8_t *frames, int row_stride,
                   int64_t frame_stride);
/* Use only the first n_frames frames (1 <= n_frames <= the count given to gme_seq_create) from now on: every stage call covers
 * the pairs of frames [0, n_frames) -- results.py:41-48 over a shorter list -- while the buffers stay sized for the full count.
 * How one sequence serves chunks of different lengths of a longer video (sequence.StreamEstimator).  Ends a staged GME run. */
GME_API int gme_seq_set_frames(gme_seq *seq, int n_frames);
/* deterministic synthetic frames t0 .. t0+N-1 generated on the device */
GME_API int gme_seq_synth(gme_seq *seq, uint64_t seed, int t0);
/* mark the pyramid levels stale (upload and synth do so themselves) */
GME_API int gme_seq_invalidate(gme_seq *seq);
/* level 2 = full resolution, 1 and 0 = pyramid levels (valid after gme_seq_gme_begin) */
GME_API int gme_seq_read_frame(gme_seq *seq, int level, int index, uint8_t *out);

/* bbme.get_motion_field over every pair of the sequence; results stay on the device */
GME_API int gme_seq_bbme(gme_seq *seq, int frame_distance, int block_size, int search_window,
                 int procedure, int pnorm);
GME_API int gme_seq_read_mv(gme_seq *seq, int first_pair, int count, int32_t *mf_out);

/* The same search for frames that still live in host memory (results.py:41-50 hands the path a list of
 * host arrays, utils.py:9-31): frames [0, count) are uploaded in chunks of `chunk_frames` on a copy
 * stream while the search of the previous chunk runs, and each chunk's fields are read back into
 * mf_out[count - fd][H/bs][W/bs][2] behind its kernel.  The frames stay resident in `seq` afterwards.
 * Page-locked `frames` (gme_host_alloc) cross at link speed; pageable memory works but is staged by
 * the HIP runtime.  Returns when mf_out is complete. */
GME_API int gme_seq_bbme_streamed(gme_seq *seq, const uint8_t *frames, int row_stride, int64_t frame_stride,
                          int count, int frame_distance, int block_size, int search_window, int procedure,
                          int pnorm, int chunk_frames, int32_t *mf_out);
/* page-locked host memory for frame loaders (hipHostMalloc); NULL on failure */
GME_API void *gme_host_alloc(size_t bytes);
GME_API void gme_host_free(void *p);

/* motion.global_motion_estimation (motion.py:109-136), staged so that the host does
 * the 3x3 solves (motion.py:262-264,280-282) between levels:
 *   begin : pyramids (utils.py:34-51) of all frames, diamond BBME at the three levels
 *           (motion.py:27-29,224-229), first parameters (motion.py:176-188) -> float32[P][6]
 *   fit   : for level 1 or 2 and parameters already projected by the caller
 *           (motion.py:191-207): model field, L1 difference, threshold, outlier mask,
 *           sequential float64 normal-equation sums (motion.py:232-279)
 *           -> sums_out[P][15] = F (9, row-major) | Sx (3) | Sy (3)
 * `procedure` / `search_window` select the BBME used at levels 1 and 2 (the reference
 * hard-codes diamond, GME_SEARCH_DIAMOND; exhaustive serves BASELINE config 4).
 * The level searches are launched so that the caller's work between the stages overlaps them:
 * begin returns once the first parameters are on the host with the level-1 search already
 * running; fit(level 1) returns its sums with the level-2 search running.  New frame data
 * (upload / synth / invalidate) ends a staged run: fit then reports GME_ERR_STATE until begin. */
GME_API int gme_seq_gme_begin(gme_seq *seq, int frame_distance, int bbme_block_size, int procedure,
                      int search_window, float *params0_out);
GME_API int gme_seq_gme_fit(gme_seq *seq, int level, const double *params_in, double outlier_fraction,
                    double *sums_out);
/* begin + projection of the first parameters (motion.py:191-207: two exact doublings of the float32 vector) + fit(level 1) in
 * one call: the first parameters go from the dense field to the level-1 fit on the device, so a staged run needs three
 * dependent host round trips instead of four (level-1 sums, level-2 sums, squared errors).  Returns the level-1 sums
 * (split-phase: once gme_seq_wait has returned) with the level-2 search already queued.  params0_out may be NULL. */
GME_API int gme_seq_gme_begin_fit(gme_seq *seq, int frame_distance, int bbme_block_size, int procedure, int search_window,
                          double outlier_fraction, float *params0_out, double *sums1_out);
/* stage read-back for parity tests; any pointer may be NULL.
 * level 0: gt = dense field (bs 2); levels 1, 2: gt, model (int16), mask, threshold */
GME_API int gme_seq_gme_read_stage(gme_seq *seq, int level, int pair, int32_t *gt, int16_t *model,
                           uint8_t *mask, int64_t *threshold);

/* motion.get_motion_field_affine((H/bs, W/bs), params) + motion.compensate_frame(prev, field)
 * + sum of squared error against `cur` for every pair (results.py:52-59,109).
 * Compensated frames stay on the device; sse_out[P] may be NULL. */
GME_API int gme_seq_compensate(gme_seq *seq, int frame_distance, int block_size, const double *params,
                       int64_t *sse_out);
GME_API int gme_seq_read_compensated(gme_seq *seq, int pair, uint8_t *out);
/* the compensated frames of pairs first .. first+count-1 into out[count][H][W] (tight), one wait for all of them: a finished
 * chunk of a streamed video for results.py's writers (results.py:59-76) */
GME_API int gme_seq_read_compensated_range(gme_seq *seq, int first, int count, uint8_t *out);
/* Opt-in one-call form of begin_fit -> solve -> fit(2) -> solve -> compensate with the two 3x3 solves of
 * motion.py:262-264,280-282 on the device: one host round trip per estimate instead of three.  LAPACK's last bits are not
 * reproduced: params_out[P][6] is within rtol 1e-10 of the staged path's (motion.py:109-136); model fields, masks,
 * compensated frames (results.py:52-59) and sse_out[P] (may be NULL) are bit-equal to it for every pair whose flags_out[p]
 * is 0.  A non-zero flag -- bit 1 / 2: a model displacement within 1e-9 of a rounding tie at level 2 / in the final field;
 * bit 4: a singular system (numpy.linalg.LinAlgError upstream, motion.py:262) -- tells the caller to redo that pair through
 * the staged calls.  Split-phase like them (gme_seq_set_split_phase). */
/* The device solve by itself: sums[P][15] = F (9, row-major) | Sx | Sy -> params_out[P][6] (motion.py:262-286; first
 * components doubled if `project`, motion.py:191-207) and flags_out[P] (bit 1: a displacement of the h x w model field of
 * those parameters within 1e-9 of a rounding tie, motion.py:139-157; bit 4: singular).  Host pointers. */
GME_API int gme_solve_fit_sums(gme_ctx *ctx, const double *sums, int pairs, int project, int h, int w, double *params_out,
                       int32_t *flags_out);
GME_API int gme_seq_gme_device_solve(gme_seq *seq, int frame_distance, int bbme_block_size, int procedure, int search_window,
                             double outlier_fraction, double *params_out, int64_t *sse_out, int32_t *flags_out);


/* Split-phase form of the three calls above (no counterpart in the reference, whose stages are plain function calls,
 * motion.py:109-136; this is how ONE host thread keeps several streams busy).  With the switch on,
 * gme_seq_gme_begin / gme_seq_gme_fit / gme_seq_compensate return as soon as their work is queued; their output
 * buffer (page-locked: gme_host_alloc) is valid after gme_seq_wait, which waits for the result of the last such call
 * only -- the level search queued behind it keeps running.  gme_sync still drains the stream and reports walk
 * overruns.  gme_seq_upload is split-phase too: it returns with its copies queued on the context's stream, and the
 * host frames must stay untouched until a later gme_seq_wait / gme_sync returns -- how sequence.estimate_stream keeps the
 * link busy with chunk k + 1 of a video in host memory (results.py:41-50, utils.py:9-31) while chunk k is estimated. */
GME_API int gme_seq_set_split_phase(gme_seq *seq, int on);
GME_API int gme_seq_wait(gme_seq *seq);
/* 1: the result of the last split-phase call has arrived (gme_seq_wait would not block), 0: not yet, < 0: error */
GME_API int gme_seq_poll(gme_seq *seq);

/* ---------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, contiguous pair ranges per rank (results.py:41-50 carries no state
 * between pairs), and ONE exchange: the all-gather of the per-pair parameter rows over RCCL / xGMI on
 * the context's stream.  The reference has no distributed code; SURVEY.md §8(e) defines this split.
 *   gme_comm_unique_id   rank 0 makes the 128-byte RCCL id; the launcher hands it to the other ranks
 *                        (file / env / socket -- before or after their contexts exist)
 *   gme_comm_init        collective over all ranks (ncclCommInitRank)
 *   gme_shard_gather     rows[n_local][k] of every rank -> out[world][n_max][k], blocks zero-padded to
 *                        n_max rows (n_max = the largest shard, the same on every rank); synchronises
 *   gme_comm_allreduce_max   element-wise max of v[n] over the ranks, in place (barrier; max-over-ranks time)
 *   gme_comm_probe       loads librccl.so and nothing else: a launcher lets every rank probe and agree BEFORE any rank
 *                        enters the collective gme_comm_init (which cannot time out), sequence.comm_init does
 *   gme_comm_info        rank and rank count as RCCL reports them (ncclCommUserRank / ncclCommCount)
 *   gme_seq_mv_summary   per-pair summary rows of the last gme_seq_bbme field, float64[P][6] = modal vector x, y (over the
 *                        vectors inside [-64, 64)^2, ties to the smaller (x+64)*128 + (y+64)), its block count, sum of
 *                        x, sum of y, checksum sum_i ((i mod 251) + 1)(3 x_i + 5 y_i) over the blocks in row-major
 *                        order -- what a sharded bbme.get_motion_field run (bbme.py:12-38 per pair, results.py:41-50 over
 *                        pairs) exchanges instead of its 10.8 kB fields; synchronises
 *   gme_seq_mv_summary_gather   the same rows all-gathered device to device: out[world][n_max][6], blocks zero-padded
 *                        to n_max >= this rank's pairs (the same on every rank); split-phase aware (gme_seq_wait)
 * ------------------------------------------------------------------------- */
GME_API int gme_comm_probe(void);
GME_API int gme_comm_info(gme_ctx *ctx, int *rank_out, int *world_out);
GME_API int gme_seq_mv_summary(gme_seq *seq, double *rows_out);
GME_API int gme_seq_mv_summary_gather(gme_seq *seq, int n_max, double *out);
GME_API int gme_comm_unique_id(char id_out[128]);
GME_API int gme_comm_init(gme_ctx *ctx, const char id[128], int rank, int world);
GME_API int gme_comm_destroy(gme_ctx *ctx);
GME_API int gme_shard_gather(gme_ctx *ctx, const double *rows, int n_local, int k, int n_max, double *out);
GME_API int gme_comm_allreduce_max(gme_ctx *ctx, double *v, int n);

#ifdef __cplusplus
}
#endif
#endif /* GME_HIP_H */
