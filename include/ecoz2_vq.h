/*
 * ecoz2_vq.h -- C-ABI of libecoz2vq.so: the MI355X-native drop-in for the VQ hot path of
 * mbari-org/ecoz2rs (`ecoz2 vq learn` / `ecoz2 vq quantize`).
 *
 * Part 1 declares exactly the symbols the reference's Rust FFI binds for this path
 * (extern "C" block, /root/reference/src/ecoz2_lib/mod.rs:72-178), with the same argument
 * order and meaning, so the reference's front-end can link this library in place of the
 * static `ecoz2_lib` that build.rs:59-77 compiles.  Part 2 is the resident-data session
 * API the same entry points are built on; bench.py, the tests and multi-GPU runs use it.
 *
 * Plain pointers and sizes only; no HIP or torch types.  All compute runs in hand-written
 * HIP kernels on the GPU; there is no CPU fallback: every entry point fails (non-zero
 * return + message on stderr, e2vq_last_error()) when no HIP device is usable.
 */
#ifndef ECOZ2_VQ_H
#define ECOZ2_VQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ========================================================================================
 * Part 1 -- the reference's FFI surface for this path
 * ====================================================================================== */

/* replaces `fn ecoz2_version() -> *const c_char`            src/ecoz2_lib/mod.rs:73 */
const char *ecoz2_version(void);

/* Per-codebook-size callback: (target, M, avg_distortion, sigma, inertia).
 * `callback: extern "C" fn(*mut Ecoz2ObserverRef, c_int, c_double, c_double, c_double)`
 *                                                            src/ecoz2_lib/mod.rs:104, 241-250
 * `target` is opaque here and handed back unchanged as argument 0. */
typedef void (*ecoz2_vq_learn_callback_t)(void *target, int M, double avg_distortion,
                                          double sigma, double inertia);

/* replaces `fn ecoz2_vq_learn(prediction_order, epsilon, codebook_class_name,
 *           predictor_filenames, num_predictors, target, callback)`
 *                                                            src/ecoz2_lib/mod.rs:96-105
 * LBG training over all frames of the given .prd files, M = 2, 4, ... ; writes
 * data/codebooks/<class>/eps_<eps>_M_<%04d>.cbook (+ .rpt) under the working directory
 * and invokes the callback once per codebook size, on the calling thread.
 * The Rust declaration has no return value; the int returned here is ignored by that
 * caller (ABI-safe) and is 0 on success. All strings are borrowed for the call. */
int ecoz2_vq_learn(int prediction_order, double epsilon, const char *codebook_class_name,
                   const char *const *predictor_filenames, int num_predictors, void *target,
                   ecoz2_vq_learn_callback_t callback);

/* replaces `fn ecoz2_vq_learn_using_base_codebook(base_codebook, epsilon,
 *           predictor_filenames, num_predictors, target, callback)`
 *                                                            src/ecoz2_lib/mod.rs:107-115
 * Same, starting from a saved codebook of size M0: the first trained size is 2*M0
 * (CHANGELOG.md:366-368). P and the class name come from the base codebook. */
int ecoz2_vq_learn_using_base_codebook(const char *base_codebook, double epsilon,
                                       const char *const *predictor_filenames,
                                       int num_predictors, void *target,
                                       ecoz2_vq_learn_callback_t callback);

/* replaces `fn ecoz2_vq_quantize(nom_raas, predictor_filenames, num_predictors,
 *           show_filenames)`                                  src/ecoz2_lib/mod.rs:117-122
 * Per .prd file: nearest-codeword symbol per frame -> data/sequences/M<M>/<class>/<name>.seq
 * (layout pinned by src/sequence/mod.rs:49-75). */
int ecoz2_vq_quantize(const char *nom_raas, const char *const *predictor_filenames,
                      int num_predictors, int show_filenames);

/* replaces `fn ecoz2_vq_show(codebook_filename, from, to)`   src/ecoz2_lib/mod.rs:132
 * Prints the reflection coefficients [from, to] of every codeword (-1 = whole range). */
int ecoz2_vq_show(const char *codebook_filename, int from, int to);

/* replaces `fn ecoz2_vq_classify(cb_filenames, num_codebooks, prd_filenames, num_predictors, show_ranked)`
 *                                                            src/ecoz2_lib/mod.rs:124-130
 * VQ-based classification: every .prd is quantised against every class codebook (same sweep kernel as
 * quantize); the predicted class is the codebook with the smallest average distortion (ties: first codebook).
 * Prints per-class and overall accuracy; with show_ranked, the ranked codebooks of each misclassified file. */
int ecoz2_vq_classify(const char *const *cb_filenames, int num_codebooks,
                      const char *const *prd_filenames, int num_predictors, int show_ranked);

/* replaces `fn ecoz2_prd_show_file(prd_filename, show_reflections, from, to)`   src/ecoz2_lib/mod.rs:89-94
 * (caller src/prd/mod.rs:99).  Prints the header and the coefficient range [from, to] of every predictor vector of a
 * .prd file (to <= 0: up to P), as in notes.md:77-85; with show_reflections the reflection coefficients k<n> of each
 * vector (Levinson recursion of src/lpc/lpca_r_rs.rs:8-43 on its autocorrelation) instead of the r<n>. */
int ecoz2_prd_show_file(const char *prd_filename, int show_reflections, int from, int to);

/* Knobs the reference has no argument for: environment variables, none of which changes a result.  The one table of them
 * (19) is INTEGRATION.md section 2; the ones a caller of these entry points is likely to set:
 *   ECOZ2_VQ_MAX_CODEBOOK_SIZE  last codebook size trained (default 2048, notes.md:147)
 *   ECOZ2_VQ_OUT_ROOT           prefix for the data/... outputs (default ".")
 *   ECOZ2_VQ_DEVICE             HIP device ordinal (default 0)
 *   ECOZ2_VQ_GPUS               vq learn: shard the frames over this many in-process ranks / GPUs (default 1);
 *                               vq quantize: deal the files to this many workers (no collective; same .seq files)
 *   ECOZ2_VQ_COLLECTIVE         rccl | p2p: the in-process exchange of ECOZ2_VQ_GPUS > 1 (default: RCCL -- librccl.so is
 *                               loaded with dlopen -- when every rank has a device of its own, else the peer-to-peer kernel)
 *   ECOZ2_VQ_QUIET              no progress lines on stderr                                   */

/* ========================================================================================
 * Part 2 -- session API (resident training set, one session per GPU / per rank)
 * ====================================================================================== */

typedef struct e2vq_session e2vq_session;

typedef struct {
    int M;                /* codebook size of this level                              */
    int passes;           /* assignment passes run at this level                      */
    double DD;            /* sum over frames of (min distortion - 1), last pass       */
    double avg_distortion;/* DD / T                                                   */
    double sigma;         /* std deviation of (min distortion - 1)                    */
    double inertia;       /* sum ||r - cell mean||^2 in autocorrelation space         */
    int64_t empty_cells;
    int64_t failed_cells; /* cells whose Levinson recursion failed in the last update */
} e2vq_level_stats;

/* Collective hook for N > 1 ranks: reduce `count` 64-bit integers in place on the device,
 * enqueued on `stream` (a hipStream_t). op 0 = sum (int64), 1 = max (uint64).
 * bench.py passes a torch.distributed (RCCL) all_reduce here. */
typedef int (*e2vq_allreduce_fn)(void *user, void *device_buf, int64_t count, int op, void *stream);

const char *e2vq_last_error(void);
int e2vq_device_count(void);

int e2vq_session_create(int device, int prediction_order, e2vq_session **out);
void e2vq_session_destroy(e2vq_session *s);
/* use an existing hipStream_t (e.g. torch's current stream); NULL = the session's own */
int e2vq_set_stream(e2vq_session *s, void *hip_stream);
int e2vq_set_allreduce(e2vq_session *s, e2vq_allreduce_fn fn, void *user, int rank, int world);
/* Device time of the exchange: HIP events on the session's stream around every call of the all-reduce hook.
 * e2vq_collective_timing synchronises the stream and returns the totals since the timing was switched on. */
int e2vq_enable_collective_timing(e2vq_session *s, int on);
int e2vq_collective_timing(e2vq_session *s, double *total_ms, int64_t *calls, int64_t *bytes);
/* 0 = every pass of this session on the plain FP64 sweep, 1 = prefiltered sweep again (same results either way: bench.py
 * re-runs its timed level both ways in one process and compares the codebooks bit for bit).  Switching it on needs the
 * images a session makes when it is created and given its frames with the prefilter enabled. */
int e2vq_set_prefilter(e2vq_session *s, int on);

/* In-process group: the ranks ecoz2_vq_learn runs for ECOZ2_VQ_GPUS > 1, as an object a host can drive from its own
 * threads -- one session per rank, rank r on devices[r].  collective: "rccl" (ncclCommInitAll + ncclAllReduce(int64) inside
 * the library, librccl.so loaded with dlopen: one device per rank), "p2p" (the library's reduce-scatter + all-gather kernel
 * over peer-to-peer memory: ranks may share devices), or NULL / "" (RCCL when every rank has a device of its own and
 * librccl.so loads, else p2p).  e2vq_group_bind makes the group's exchange the all-reduce of a session (which must live on
 * the rank's device); the exchange is a rendezvous -- every rank's thread must make the same calls.  A rank that gives up
 * calls e2vq_group_fail so that the others return with an error instead of waiting.  Destroy the sessions first. */
typedef struct e2vq_group e2vq_group;
int e2vq_group_create(int num_ranks, const int *devices, const char *collective, e2vq_group **out);
int e2vq_group_bind(e2vq_group *g, int rank, e2vq_session *s);
const char *e2vq_group_collective(e2vq_group *g); /* one line: which exchange the group uses */
int e2vq_group_uses_rccl(e2vq_group *g);
void e2vq_group_fail(e2vq_group *g);
void e2vq_group_destroy(e2vq_group *g);

/* training set: T x (P+1) doubles, row-major (the .prd payload), T <= 2^31 - 65. Copies / re-lays it out in HBM.
 * e2vq_set_frames_device reads `device_frames` on the session's stream: the caller must have finished writing the
 * buffer (or have written it on that stream); the read is complete when the call returns. */
int e2vq_set_frames_host(e2vq_session *s, const double *frames, int64_t T);
int e2vq_set_frames_device(e2vq_session *s, const void *device_frames, int64_t T);
/* data statistics (max |x|, global sums, sum of squares) incl. the cross-rank reduction */
int e2vq_prepare(e2vq_session *s);

/* codebook state (reflection coefficients, M x (P+1), row-major) */
int e2vq_set_codebook(e2vq_session *s, const double *reflections, int M);
int e2vq_get_codebook(e2vq_session *s, double *reflections, int *M);
int e2vq_init_codebook(e2vq_session *s); /* M = 1 centroid of the whole set */
int e2vq_grow(e2vq_session *s);          /* M -> 2M split                   */

/* one LBG iteration, in pieces: assignment+accumulation (+ all-reduce), statistics, update.
 * device_sym / device_dmin: optional device buffers of T uint16 / T doubles (NULL to skip).
 * e2vq_pass_stats and e2vq_update work from the rows of the last e2vq_pass over the CURRENT codebook: once an update (or
 * e2vq_set_codebook / e2vq_grow) has changed the codebook they fail until the next pass (the distortion sums in the rows
 * are fixed-point numbers scaled for the codebook the pass ran on). */
int e2vq_pass(e2vq_session *s, void *device_sym, void *device_dmin);
int e2vq_pass_stats(e2vq_session *s, e2vq_level_stats *out);
/* HIP events around the sweep kernel of e2vq_pass, on the session's stream */
int e2vq_enable_timing(e2vq_session *s, int on);
int e2vq_last_pass_kernel_ms(e2vq_session *s, float *ms);
/* sum of those event-measured kernel times over all passes since e2vq_enable_timing(1), and their number: lets a
 * caller time K iterations without a host synchronisation per iteration */
int e2vq_timing_total(e2vq_session *s, double *total_ms, int64_t *passes);
/* the same for the sweep kernels alone: a pass whose accumulate is a kernel of its own (recorded contributions folded by
 * k_reduce_records) counts with both kernels in e2vq_timing_total, with the sweep only here */
int e2vq_timing_sweep_total(e2vq_session *s, double *total_ms, int64_t *passes);
/* which sweep served the last e2vq_pass: *prefiltered = 1 when the f16-prefiltered sweep ran (P = 36, large M),
 * *fallback_frames = frames it handed to the full FP64 sweep (synchronises the stream) */
int e2vq_last_pass_info(e2vq_session *s, int *prefiltered, int64_t *fallback_frames);
/* *recorded = 1 when the last e2vq_pass recorded its contributions to the cell sums for k_reduce_records (0: it added them
 * itself); *records = how many the last recorded pass of this level wrote (-1: none yet; valid once e2vq_pass_stats has
 * returned for that pass).  Few records switch the rest of a level to the burst of atomics (ECOZ2_VQ_RECORDS_FEW_DIV). */
int e2vq_last_pass_records(e2vq_session *s, int *recorded, int64_t *records);
/* how the last e2vq_pass swept (round 5): *kind = 0 plain FP64 sweep, 1 round 4's fused prefiltered kernel, 2 candidate sweep +
 * finishing kernel + reduce (frames in their natural order), 3 the fused pass over frames grouped by cell; *two_stage = 1
 * when the sweep ran its coarse stage first; *flagged_fraction = share of the (tile, column block) jobs the level's first
 * two-stage pass had to finish with all weight levels (-1: not measured at this level; valid once e2vq_pass_stats has
 * returned for that pass) */
int e2vq_last_pass_sweep(e2vq_session *s, int *kind, int *two_stage, double *flagged_fraction);
/* what the fused sorted passes (kind 3) EXECUTED since the last reset, counted by the kernels themselves (one atomic per
 * wave, every pass): *jobs = (tile, 32-frame column block) jobs of the two-stage passes -- each ran the coarse k-steps --,
 * *flagged = those of them that also ran stage 2 with all k-steps, *one_stage_jobs = jobs of the passes that ran without a
 * coarse stage (all k-steps each).  Synchronises the stream; reset != 0 zeroes the counts. */
int e2vq_sweep_executed(e2vq_session *s, int64_t *flagged, int64_t *jobs, int64_t *one_stage_jobs, int reset);
/* The two switches the host makes from what a pass measured; both only choose kernels -- codebooks, symbols and statistics are
 * the same bits either way.  two_stage_max_fraction: flagged share of a level's first sorted pass above which the rest of the
 * level runs without the coarse stage.  max_uncertified_fraction: share of a pass's frames the prefiltered sweep may leave to
 * the FP64 fallback sweep before the plain FP64 sweep takes over from that codebook size on (data whose distortions are small
 * differences of large terms certify poorly).  Negative: unchanged.  Clears earlier decisions. */
int e2vq_set_sweep_policy(e2vq_session *s, double two_stage_max_fraction, double max_uncertified_fraction);
/* what the passes so far decided: sorted passes run one stage up to *one_stage_until_M (0: none); training passes run the plain
 * sweep from *plain_from_M on (0: none); *uncertified = frames the last prefiltered pass left to the fallback sweep (-1: the
 * last pass was not prefiltered).  Valid once e2vq_pass_stats has returned for the pass. */
int e2vq_sweep_policy_state(e2vq_session *s, int *one_stage_until_M, int *plain_from_M, int64_t *uncertified);
/* number of training-pass sweep launches so far, by kernel family (k_pass_pre / k_pass_mfma+generic): lets a
 * kernel trace of a whole run be cut to the dispatches of a timed region */
int e2vq_sweep_launch_counts(e2vq_session *s, int64_t *prefiltered, int64_t *plain);
/* the same by kernel: *pass_pre_lds = launches of round 4's fused kernel (k_pass_pre_lds), *sweep_cand = launches of round 5's
 * k_sweep_cand (the fused pass over grouped frames, or the candidate sweep in front of k_finish), *plain = FP64 sweeps */
int e2vq_launch_counts_by_kernel(e2vq_session *s, int64_t *pass_pre_lds, int64_t *sweep_cand, int64_t *plain);
int e2vq_update(e2vq_session *s);
/* ECOZ2_VQ_VERIFY_PUBLISH=1 (read at session creation): after every pass the statistics the update kernel published
 * through host-mapped memory (level sums, within-cell terms, L1 maximum, failed recursions) are recomputed on the host
 * from a copy of the accumulator rows and compared bit for bit; a mismatch fails the call.  Number of passes checked: */
int e2vq_verified_passes(e2vq_session *s, int64_t *passes);
/* one whole LBG iteration in a single call: e2vq_pass + e2vq_pass_stats + e2vq_update */
int e2vq_iterate(e2vq_session *s, void *device_sym, void *device_dmin, e2vq_level_stats *out);
/* the reduced accumulator rows of the last pass (M x row_stride int64) copied to the host.  Row of a cell: [0, 2 NC) the
 * limb pairs of its coefficient sums, [2 NC] its frame count, [2 NC + 1, 2 NC + 5) distortion sums (e and e^2 as limb
 * pairs).  Only the COLUMN TOTALS of the four distortion elements are defined (the level statistics read nothing else):
 * the prefiltered sweep adds a wave's distortion sums to one row, not frame by frame to the frame's cell. */
int e2vq_row_stride(int prediction_order);
int e2vq_get_rows(e2vq_session *s, int64_t *rows);

/* whole LBG ladder from the current codebook up to max_M. out_root NULL = no files. */
int e2vq_learn(e2vq_session *s, double epsilon, int max_M, const char *class_name,
               const char *out_root, void *target, ecoz2_vq_learn_callback_t callback,
               e2vq_level_stats *levels, int max_levels, int *num_levels);

/* DDprv of the stopping rule (DDprv - DD)/DD < eps; it carries over between codebook sizes (notes.md:128-153).
 * After restoring an earlier codebook with e2vq_set_codebook, restore its level's DD here and e2vq_learn repeats the
 * next level exactly as the uninterrupted ladder ran it. */
int e2vq_set_prev_distortion(e2vq_session *s, double DDprv);
/* One saved point of the ladder (device copies): codebook, DDprv, and the accumulator rows and cells of the last pass --
 * what the seeded first pass of the next codebook size starts from.  Save right after the pass + statistics that ended
 * a level; restoring puts the session back there, so that e2vq_learn repeats the next level exactly as the
 * uninterrupted ladder runs it (seeded first pass included).  bench.py times the M = 1024 level this way. */
int e2vq_save_state(e2vq_session *s);
int e2vq_restore_state(e2vq_session *s);
int e2vq_get_prev_distortion(e2vq_session *s, double *DDprv);

/* nearest-codeword assignment of arbitrary frames against the session's codebook.  Frames must be finite
 * (the file entry points check this and fail); T <= 2^31 - 65 per call. */
int e2vq_quantize_host(e2vq_session *s, const double *frames, int64_t T, uint16_t *sym,
                       double *dmin);
int e2vq_quantize_device(e2vq_session *s, const void *device_frames, int64_t T, void *device_sym,
                         void *device_dmin);
int e2vq_synchronize(e2vq_session *s);
/* average distortion sum_t (dmin_t - 1) / T of frames against the session's codebook (frame order, plain f64 sum) */
int e2vq_avg_distortion_host(e2vq_session *s, const double *frames, int64_t T, double *avg);

/* ---- files (.prd / .cbook / .seq) and synthetic data ------------------------------------- */
int e2vq_prd_info(const char *path, char class_name[96], int *P, int64_t *T);
int e2vq_prd_read(const char *path, double *frames, int64_t capacity_frames);
int e2vq_prd_write(const char *path, const char *class_name, int P, const double *frames, int64_t T);
int e2vq_cbook_info(const char *path, char class_name[96], int *P, int *M);
int e2vq_cbook_read(const char *path, double *reflections, int capacity_codewords);
int e2vq_cbook_write(const char *path, const char *class_name, int P, int M, const double *reflections);
int e2vq_seq_write(const char *path, const char *class_name, int M, const uint16_t *sym, int64_t T);

/* Synthetic gain-normalised autocorrelation frames [first, first+count) of the stream
 * (seed, n_classes): counter-based, so any shard regenerates identical frames. */
int e2vq_synth_frames(uint64_t seed, int n_classes, int P, int64_t first, int64_t count, double *frames);
/* The same stream with its shape exposed (bench.py's config.robustness, tests):
 *   kind 0  class prototypes + noise: reflection k_i = prototype(class, i) + noise * g, g ~ N(0, 1) (12 uniforms);
 *           e2vq_synth_frames is (kind 0, noise 0.05).
 *   kind 1  a continuum, no classes: the reflections of frame t follow a smooth trajectory in t (a few slow sinusoids per
 *           coefficient around a low-gain mean: r[0] = 1 / E about 2-3 as in the reference's whale-song predictor file,
 *           notes.md:80-85; neighbouring frames overlap as 45 ms windows at 15 ms offsets do) + noise * g.  n_classes is
 *           the number of sinusoids per coefficient.
 * Frames are gain-normalised autocorrelation sequences r / E built by the inverse Levinson recursion, as lpc_rs.rs:116-131
 * hands them to vq learn. */
int e2vq_synth_frames_kind(uint64_t seed, int kind, int n_classes, double noise, int P, int64_t first, int64_t count,
                           double *frames);

#ifdef __cplusplus
}
#endif
#endif
