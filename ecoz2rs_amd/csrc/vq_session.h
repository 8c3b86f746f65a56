// vq_session.h -- what the host translation units of libecoz2vq.so share (internal; the C-ABI is include/ecoz2_vq.h):
// the session object, error plumbing, and the few internal functions that cross files.
//   vq_host.cpp   session life cycle, training set, codebook, save / restore, the LBG ladder, quantize on resident data
//   vq_pass.cpp   one LBG iteration: the pass (kernel choice and launches), statistics, the speculative update
//   vq_group.cpp  the in-process group: peer-to-peer exchange, RCCL loaded with dlopen
//   vq_entry.cpp  the reference's entry points: file readers, upload, ecoz2_vq_learn / quantize / classify / show
#pragma once
#include "../../include/ecoz2_vq.h"
#include "vq_device.h"
#include "vq_fixed.h"
#include "vq_io.h"
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <errno.h>
#include <float.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

using e2vq::DevScalars;
#define E2VQ_MAX_PASSES 1000  // safety cap per codebook size (same in the oracle)
typedef long long i64;
typedef unsigned long long u64;

// errors: the message of the calling thread (e2vq_last_error); e2vq_set_error returns 1
char* e2vq_err_buf();                 // 1024 bytes, thread-local
int e2vq_set_error(const char* fmt, ...);

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return e2vq_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct e2vq_session {
    int device = 0, P = 0, NC = 0, FB = 64, RS = 0, NPAD = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // training set (blocked layout)
    double* d_blk = nullptr;
    double* d_aos = nullptr;  // row-major copy padded with zero rows to whole blocks (k_pass_pre_lds stages it in LDS)
    i64 T = 0, nblocks = 0, T_total = 0;
    bool prepared = false;
    bool maxabs_scanned = false;  // d_maxabs / d_flags hold this rank's scan from the re-layout kernel
    // codebook
    int M = 0, M_cap = 0;
    double* d_refl = nullptr;      // current reflections [M][NC]
    double* d_refl_next = nullptr; // grow target
    double* d_cbq = nullptr;       // [M][NPAD] pre-doubled raas rows
    double* d_cbm = nullptr;       // MFMA operand layout of the same codewords (P = 36)
    double* d_cbT = nullptr;       // P > 40: scratch for the transposed codebook of the LDS-staged generic sweep
    u64* d_l1max = nullptr;
    // shadow codebook: the centroid update of a pass is launched speculatively into these right after the
    // statistics kernel, while the host reads DD and decides; e2vq_update commits by swapping pointers
    double* d_refl_spec = nullptr;
    double* d_cbq_spec = nullptr;
    double* d_cbm_spec = nullptr;
    u64* d_l1max_spec = nullptr;
    bool spec_valid = false;
    bool spec_zeroed = false;  // the pass prologue zeroed d_l1max_spec and the shadow image's scalars
    hipEvent_t ev_stats = nullptr;
    struct HostStats { i64 l[64 * 8]; u64 l1bits; volatile u64 seq; i64 failed; volatile u64 seq2; volatile u64 err; volatile i64 rec_total; volatile u64 sw_flagged, sw_jobs; volatile i64 fb; }* h_stats = nullptr;  // pinned, host-mapped
    long verified_passes = 0;
    bool verify_publish = false;  // ECOZ2_VQ_VERIFY_PUBLISH: recompute every published statistic on the host from the rows
    bool failed_pending = false;              // the failed-recursion count of stats_seq has not been read yet (seq2)
    e2vq_level_stats* failed_patch = nullptr;  // e2vq_learn: the level record that still waits for that count
    u64 stats_seq = 0;
    double stats_wait_s = 0.0;                // how long the host waited for the last pass's statistics (spin_for_sequence)
    double* h_within = nullptr;                                      // pinned, M_cap doubles
    // statistics
    DevScalars* d_sc = nullptr;
    DevScalars h_sc{};
    u64* d_maxabs = nullptr;
    int* d_flags = nullptr;   // [0] bad data, [1] init status
    i64* d_stats = nullptr;   // [2NC+3]: global sums, sum sq limbs, T
    i64* d_rows = nullptr;    // [M][RS]
    double* d_S = nullptr;    // [M][NC]
    double* d_within = nullptr;
    i64* d_lstats = nullptr;  // [64 slots][8]: dist, dist2 limbs, empty, failed (slots are summed on the host)
    bool lstats_dirty = false;  // a centroid kernel added to the slots after they were published
    bool stats_valid = false;
    bool rows_fresh = false;  // d_rows hold the sums of a pass over the codebook that is still the current one
    e2vq_level_stats last{};
    double DDprv = DBL_MAX / 1e5;  // "e+303" in notes.md:128
    // quantize scratch
    double* d_qaos = nullptr;
    double* d_qblk = nullptr;
    unsigned short* d_qsym = nullptr;
    double* d_qdmin = nullptr;
    i64 q_cap = 0, qblk_cap = 0;
    // HIP events around the sweep kernel (bench.py's roofline figures)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr;  // (ev_mid: between the sweep and its accumulate kernel)
    bool timing_mid = false;        // the pending pass has an ev_mid
    double timing_sweep_ms = 0.0;   // the sweep kernels alone (e2vq_timing_sweep_total)
    bool timing = false, timed = false;
    double timing_sum_ms = 0.0;  // kernel time of the timed passes already folded in (e2vq_timing_total)
    long timing_count = 0;
    bool timing_pending = false;  // ev0/ev1 hold a pass that is not in the sum yet
    // prefiltered sweep (P = 36, M >= pre_min_M): f16 limb images of the frames / the codebook, fallback list
    bool pre_enabled = false;
    int pre_min_M = 256;    // (training passes with the recorded accumulate: 128, see rec_enabled)
    int pre_min_M_quant = 256;
    unsigned long long* d_colmax = nullptr;
    int* d_ea = nullptr;
    void* d_fimg = nullptr;
    float* d_fg = nullptr;
    // two codebook limb images + per-pass scalars: slot img_cur serves the current codebook, the other one is built
    // for the speculative (shadow) codebook right after the statistics are published, while the host reads them
    void* d_cimg2[2] = {nullptr, nullptr};
    void* d_ps2[2] = {nullptr, nullptr};
    bool img_valid[2] = {false, false};
    int img_cur = 0, img_last = 0;  // img_last: the slot whose scalars hold the last pass's fallback count
    int cimg_cap = 0;
    void* d_ps = nullptr;  // quantize
    int* d_fblist = nullptr;
    bool last_prefiltered = false;
    i64 n_pre_launches = 0, n_plain_launches = 0, n_sweep_launches = 0;  // (n_sweep_launches: those of n_pre_launches that ran k_sweep_cand)  // sweep launches of this session's training passes, by kernel family
    // quantize through the prefiltered sweep: scratch images of the frames handed in and of the codebook
    int* d_ea_q = nullptr;
    void* d_qfimg = nullptr;
    float* d_qfg = nullptr;
    int* d_qfblist = nullptr;
    void* d_qcimg = nullptr;
    i64 qpre_cap = 0;
    i64 qfb_cap = 0;
    u64 cb_version = 1;    // bumped whenever the codebook in d_cbq changes
    u64 scale_version = 0; // codebook version whose limb-image scale e2vq_grow's update kernel has already found ...
    int scale_img = -1;    // ... in the scalars of this image (zeroed there too)
    u64 qimg_version = 0;  // codebook version d_qcimg / d_ea_q / d_ps were built for
    int qcimg_cap = 0;
    // incremental accumulation (prefiltered passes): the rank's own rows and every frame's cell persist between
    // passes of one codebook size; a pass then moves only the frames whose cell changed (vq_accum.h)
    bool incr_enabled = true, incr_valid = false;
    int incr_M = 0;
    unsigned short* d_prev_sym = nullptr;
    // round 4: the RECORDED accumulate -- the accumulating sweep writes an 8-byte record per contribution into the region
    // of (sweeping workgroup, bin of cells), k_reduce_records folds the records into the rows through LDS tables
    // (vq_prefilter.hip).  ECOZ2_VQ_ACCUMULATE=burst: the fused burst of atomics; rec_max_bytes bounds the record buffer (it is sized for
    // the worst case, every frame of a workgroup in one bin, i.e. 16 bytes x frames x bins)
    bool rec_enabled = true;
    int rec_min_M = 64;
    size_t rec_max_bytes = (size_t)8192 << 20;
    // Few contributions (the later passes of a level run to a small epsilon): the burst of atomics inside the sweep hides
    // under the sweep and is cheaper than a second kernel.  k_reduce_records publishes the pass's record count; once it falls
    // below frames / rec_few_div the rest of the level runs the burst (ECOZ2_VQ_RECORDS_FEW_DIV, 0 = never switch).
    // Measured on levels of 11-12 passes (profiles/r04_records.txt): 1/3 is best or within noise of it at M = 256 / 512 /
    // 1024; the usual three-pass level (65 % / 46 % / 31 % of the frames recorded) stays on records throughout
    int rec_few_div = 3;
    bool last_recorded = false;    // the last pass recorded its contributions
    bool rec_pending = false;      // the pass in flight publishes its record count
    bool rec_level_burst = false;  // this level has switched to the burst
    i64 rec_last_total = -1;       // records of the last recorded pass (-1: none yet at this level)
    void* d_recs = nullptr;
    size_t recs_cap = 0;
    int* d_rec_counts = nullptr;
    // round 5 (vq_sweep.hip): recorded passes run as sort (once per level) + candidate sweep + finishing kernel + reduce.
    // ECOZ2_VQ_ACCUMULATE=records keeps round 4's fused kernel (A/B).  The sweep's two-stage keys need the frames grouped by cell
    // and data whose near codewords share tiles: the finishing kernel publishes the flagged fraction of every two-stage
    // sweep, and above two_stage_max_frac the rest of the level (and the next one) runs the one-stage sweep.
    bool sweep2_enabled = true;
    bool two_blocks_always = false;  // ECOZ2_VQ_ACCUMULATE=sorted (the tests' small shards reach the two-block turn)
    bool fused_enabled = true;       // ECOZ2_VQ_ACCUMULATE=sweep: grouped passes as sweep + finishing kernel + reduce too (A/B)
    int sweep_min_M = 256;           // smallest codebook of the round-5 kernels (ECOZ2_VQ_ACCUMULATE=sorted / sweep: 64)
    double two_stage_max_frac = 0.45;
    // round 6: the share of a pass's frames the prefiltered sweep could not certify is published with the statistics; above
    // pre_max_uncertified the FP64 fallback sweep of those frames costs more than the prefilter saves (data whose distortions
    // are a small difference of large terms: DESIGN 4.2) -- the plain sweep takes over from this codebook size on, until a
    // codebook is defined from outside
    double pre_max_uncertified = 0.40;  // (round 4's kernel + the list-driven FP64 sweep of a share u of 2^21 frames at M = 1024:
                                        // 1.0 + 4.8 u ms against the plain sweep's 2.9: bench.py config.robustness)
    bool last_first_of_level = false;
    int pre_off_from_M = 0;          // plain FP64 sweep for training passes while M >= this (0: never)
    i64 last_fb = -1;                // uncertified frames of the last prefiltered pass (-1: not a prefiltered pass)
    int two_stage_off_until_M = 0;   // one-stage sweeps while M <= this
    bool sw_pending = false;         // a two-stage sweep's counters have not been read yet
    u64 sw_host_flagged = 0, sw_host_jobs = 0;  // counters the host has fetched since the last e2vq_sweep_executed(reset)
    u64 sw_one_stage_jobs = 0;                  // jobs of the fused passes that ran one-stage since then
    void* d_fimgF = nullptr;         // frame-major limb image (gathered through d_perm)
    unsigned* d_perm = nullptr;      // slot -> frame, grouped by the cell at the level's start
    unsigned* d_cand = nullptr;      // per frame: the two candidates + flags
    void* d_sort = nullptr;
    int perm_M = 0;                  // codebook size d_perm was sorted for (0: none)
    double last_flagged_frac = -1.0;
    int last_kind = 0;               // e2vq_last_pass_sweep
    bool last_two_stage = false;
    i64* d_rows_local = nullptr;  // world > 1: the un-reduced rows (d_rows holds the all-reduced copy)
    int rows_local_cap = 0;
    // the seeded first pass of a level (vq_update.hip: k_seed_family): the rank's own rows of the last pass at the previous
    // size, stashed by e2vq_grow, and the side table of the in-family arrivals
    bool fam_enabled = true, fam_pending = false;
    int fam_M = 0, fam_cap = 0;
    int cells_M = 0;             // codebook size d_prev_sym's cells belong to (0: not valid)
    bool rows_local_is_current = false;  // d_rows_local (not d_rows) holds this rank's rows of the last pass
    bool rows_are_local = false; // the rows of the last pass are this rank's own sums (no collective, or d_rows_local)
    i64* d_rows_parent = nullptr;
    i64* d_fam = nullptr;
    // e2vq_save_state / e2vq_restore_state: one saved point of the ladder (codebook, DDprv, rows, cells)
    struct Saved {
        bool valid = false;
        int M = 0, cells_M = 0, incr_M = 0;
        double DDprv = 0.0;
        bool rows_fresh = false, rows_are_local = false, rows_local_is_current = false, incr_valid = false;
        double* refl = nullptr;
        i64* rows = nullptr;
        i64* rows_local = nullptr;
        unsigned short* cells = nullptr;
        int cap_M = 0;
        i64 cap_T = 0;
        i64 nblocks = 0;  // of the training set the rows and cells were saved for
    } sv;
    // collective hook
    e2vq_allreduce_fn allreduce = nullptr;
    void* ar_user = nullptr;
    int rank = 0, world = 1;
    bool ar_force = false;  // call the hook even for one rank (a 1-rank RCCL group: exercises the plumbing on one GPU)
    const volatile bool* group_failed = nullptr;  // in-process group: its failed flag (the statistics spin looks at it)
    // e2vq_enable_collective_timing: HIP events on the session's stream around every call of the hook
    bool ar_timing = false;
    struct ArTimed { hipEvent_t a, b; };
    std::vector<ArTimed> ar_pending, ar_free;
    double ar_ms = 0.0;
    long ar_calls = 0, ar_bytes = 0;
};

// ---- internal functions that cross translation units ----
int e2vq_reduce(e2vq_session* s, void* buf, i64 count, int op);                       // vq_host.cpp: the all-reduce hook, timed
int e2vq_ensure_codebook_capacity(e2vq_session* s, int M);                           // vq_host.cpp
// (callers that redefine the codebook from outside: set / init / grow / restore; the update of a pass calls it with redefined = false)
int e2vq_codebook_prepare(e2vq_session* s, bool redefined = true, bool grown = false, int zeroed_with_scale = -1);  // vq_host.cpp
int e2vq_set_frames_device_impl(e2vq_session* s, const void* device_frames, int64_t T, bool* adopt);  // vq_host.cpp
int e2vq_pass_mode(const e2vq_session* s);                                           // vq_pass.cpp
bool e2vq_use_prefilter(const e2vq_session* s, int mode);                            // vq_pass.cpp
int e2vq_fold_pending_timing(e2vq_session* s);                                       // vq_pass.cpp
int e2vq_pass_stats_impl(e2vq_session* s, e2vq_level_stats* out, bool wait_failed);  // vq_pass.cpp
int e2vq_resolve_failed_cells(e2vq_session* s);                                      // vq_pass.cpp
int e2vq_env_int(const char* name, int dflt);                                        // vq_entry.cpp
const char* e2vq_env_str(const char* name, const char* dflt);

