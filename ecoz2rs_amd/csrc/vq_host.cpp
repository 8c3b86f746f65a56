// vq_host.cpp -- host side of libecoz2vq.so: resident-data session, LBG driver and the
// reference's entry points (include/ecoz2_vq.h).  All arithmetic of the hot path runs in
// the HIP kernels of vq_device.hip; the host only sequences launches, reads a few scalars
// per pass to take the convergence decision the reference takes on the CPU
// (loop shape: /root/reference/notes.md:122-153), and does file I/O.
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

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

int e2vq_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    fprintf(stderr, "ecoz2vq: ERROR: %s\n", g_err);
    return 1;
}

extern "C" const char* e2vq_last_error(void) { return g_err; }

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return e2vq_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char* ecoz2_version(void) { return "ecoz2vq-mi355x 0.1.0 (HIP gfx950)"; }

extern "C" int e2vq_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------
// session
// ------------------------------------------------------------------------------------------
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
    struct HostStats { i64 l[64 * 8]; u64 l1bits; volatile u64 seq; i64 failed; volatile u64 seq2; volatile u64 err; volatile i64 rec_total; volatile u64 sw_flagged, sw_jobs; }* h_stats = nullptr;  // pinned, host-mapped
    long verified_passes = 0;
    bool verify_publish = false;  // ECOZ2_VQ_VERIFY_PUBLISH: recompute every published statistic on the host from the rows
    bool failed_pending = false;              // the failed-recursion count of stats_seq has not been read yet (seq2)
    e2vq_level_stats* failed_patch = nullptr;  // e2vq_learn: the level record that still waits for that count
    u64 stats_seq = 0;
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
    i64 n_pre_launches = 0, n_plain_launches = 0;  // sweep launches of this session's training passes, by kernel family
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
    bool plain_first = true;  // first (full) pass of the smallest prefiltered levels on the plain hybrid kernel
    int incr_M = 0;
    unsigned short* d_prev_sym = nullptr;
    // round 4: the RECORDED accumulate -- the accumulating sweep writes an 8-byte record per contribution into the region
    // of (sweeping workgroup, bin of cells), k_reduce_records folds the records into the rows through LDS tables
    // (vq_prefilter.hip).  ECOZ2_VQ_RECORDS=0: the fused burst of atomics; ECOZ2_VQ_RECORDS_MAX_MB bounds the record buffer (default 8192: it is sized for the worst case, every
    // frame of a workgroup in one bin, i.e. 16 bytes x frames x bins)
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
    // ECOZ2_VQ_SPLIT_SWEEP=0 keeps round 4's fused kernel (A/B).  The sweep's two-stage keys need the frames grouped by cell
    // and data whose near codewords share tiles: the finishing kernel publishes the flagged fraction of every two-stage
    // sweep, and above two_stage_max_frac the rest of the level (and the next one) runs the one-stage sweep.
    bool sweep2_enabled = true;
    bool fused_enabled = true;       // ECOZ2_VQ_FUSED_SORTED=0: grouped passes as sweep + finishing kernel + reduce too (A/B)
    int fused_min_M = 256;           // ECOZ2_VQ_FUSED_MIN_M
    bool two_stage_enabled = true;   // ECOZ2_VQ_TWO_STAGE=0: one-stage sweep always
    double two_stage_max_frac = 0.45;
    int two_stage_off_until_M = 0;   // one-stage sweeps while M <= this
    bool sw_pending = false;         // a two-stage sweep's counters have not been read yet
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
    // the seeded first pass of a level (vq_device.hip: k_seed_family): the rank's own rows of the last pass at the previous
    // size, stashed by e2vq_grow, and the side table of the in-family arrivals
    bool fam_enabled = true, fam_pending = false;
    int fam_M = 0, fam_cap = 0;
    int fam_min_M = 512;  // smallest size whose first pass is seeded: at M = 256 the atomics of 2^21 frames crowd onto 384
                          // rows and the plain first pass with its workgroup LDS table is faster (0.96 vs 1.08 ms)
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

static int pass_mode(const e2vq_session* s);
static bool use_prefilter(const e2vq_session* s, int mode);

static int ensure_codebook_capacity(e2vq_session* s, int M)
{
    if (M <= s->M_cap) return 0;
    int cap = std::max(M, std::max(2 * s->M_cap, 64));
    HIPCHK(hipSetDevice(s->device));
    double *refl, *refl_next, *cbq, *S, *within, *cbm = nullptr, *refl_spec, *cbq_spec, *cbm_spec = nullptr, *hw;
    i64* rows;
    HIPCHK(hipMalloc(&refl, (size_t)cap * s->NC * 8));
    HIPCHK(hipMalloc(&refl_next, (size_t)cap * s->NC * 8));
    HIPCHK(hipMalloc(&cbq, (size_t)cap * s->NPAD * 8 + 512));
    HIPCHK(hipMalloc(&refl_spec, (size_t)cap * s->NC * 8));
    HIPCHK(hipMalloc(&cbq_spec, (size_t)cap * s->NPAD * 8 + 512));
    HIPCHK(hipHostMalloc(&hw, (size_t)cap * 8, hipHostMallocMapped | hipHostMallocCoherent));
    if (e2vq::uses_mfma(s->NC)) {
        HIPCHK(hipMalloc(&cbm, (size_t)e2vq::cbm_doubles(s->NC, cap) * 8));
        HIPCHK(hipMalloc(&cbm_spec, (size_t)e2vq::cbm_doubles(s->NC, cap) * 8));
    } else {
        if (s->d_cbT) (void)hipFree(s->d_cbT);
        s->d_cbT = nullptr;
        HIPCHK(hipMalloc(&s->d_cbT, (size_t)e2vq::generic_scratch_doubles(s->NC, cap) * 8));
    }
    HIPCHK(hipMalloc(&S, (size_t)cap * s->NC * 8));
    // (+ the flags k_cell_update's publishing workgroup polls: two per cell)
    HIPCHK(hipMalloc(&within, (size_t)cap * 8 + (size_t)cap * 2 * sizeof(unsigned)));
    HIPCHK(hipMemset(within + cap, 0, (size_t)cap * 2 * sizeof(unsigned)));
    HIPCHK(hipMalloc(&rows, (size_t)cap * s->RS * 8));
    if (s->M > 0) {
        HIPCHK(hipMemcpyAsync(refl, s->d_refl, (size_t)s->M * s->NC * 8, hipMemcpyDeviceToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(cbq, s->d_cbq, (size_t)s->M * s->NPAD * 8, hipMemcpyDeviceToDevice, s->stream));
        if (cbm)
            HIPCHK(hipMemcpyAsync(cbm, s->d_cbm, (size_t)e2vq::cbm_doubles(s->NC, s->M) * 8, hipMemcpyDeviceToDevice,
                                  s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    (void)hipFree(s->d_refl);
    (void)hipFree(s->d_refl_next);
    (void)hipFree(s->d_cbq);
    (void)hipFree(s->d_cbm);
    (void)hipFree(s->d_refl_spec);
    (void)hipFree(s->d_cbq_spec);
    (void)hipFree(s->d_cbm_spec);
    if (s->h_within) (void)hipHostFree(s->h_within);
    (void)hipFree(s->d_S);
    (void)hipFree(s->d_within);
    (void)hipFree(s->d_rows);
    s->d_refl = refl;
    s->d_refl_next = refl_next;
    s->d_cbq = cbq;
    s->d_cbm = cbm;
    s->d_refl_spec = refl_spec;
    s->d_cbq_spec = cbq_spec;
    s->d_cbm_spec = cbm_spec;
    s->h_within = hw;
    s->spec_valid = false;
    s->d_S = S;
    s->d_within = within;
    s->d_rows = rows;
    s->incr_valid = false;
    s->M_cap = cap;
    return 0;
}

static int session_init(e2vq_session* s)
{
    HIPCHK(hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking));
    s->stream = s->own_stream;
    HIPCHK(hipMalloc(&s->d_sc, sizeof(DevScalars)));
    HIPCHK(hipMalloc(&s->d_maxabs, 8));
    HIPCHK(hipMalloc(&s->d_l1max, 8));
    HIPCHK(hipMalloc(&s->d_flags, 2 * sizeof(int)));
    HIPCHK(hipMalloc(&s->d_stats, (size_t)(2 * s->NC + 3) * 8));
    HIPCHK(hipMalloc(&s->d_lstats, 64 * 8 * 8));
    HIPCHK(hipMemset(s->d_lstats, 0, 64 * 8 * 8));
    HIPCHK(hipMemset(s->d_sc, 0, sizeof(DevScalars)));
    HIPCHK(hipEventCreate(&s->ev0));
    HIPCHK(hipEventCreate(&s->ev1));
    HIPCHK(hipEventCreate(&s->ev_mid));
    HIPCHK(hipEventCreateWithFlags(&s->ev_stats, hipEventDisableTiming));
    HIPCHK(hipMalloc(&s->d_l1max_spec, 8));
    HIPCHK(hipHostMalloc(&s->h_stats, sizeof(*s->h_stats), hipHostMallocMapped | hipHostMallocCoherent));
    s->h_stats->seq = 0;
    s->h_stats->seq2 = 0;
    s->h_stats->err = 0;
    if (const char* vp = getenv("ECOZ2_VQ_VERIFY_PUBLISH")) s->verify_publish = atoi(vp) != 0;
    // ECOZ2_VQ_PREFILTER=0 keeps every pass on the FP64 sweep; ECOZ2_VQ_PREFILTER_MIN_M moves the switch-over size
    const char* pf = getenv("ECOZ2_VQ_PREFILTER");
    s->pre_enabled = e2vq::prefilter_supports(s->NC, 64) && !(pf && atoi(pf) == 0);
    if (const char* mm = getenv("ECOZ2_VQ_PREFILTER_MIN_M")) s->pre_min_M = s->pre_min_M_quant = std::max(64, atoi(mm));
    if (const char* inc = getenv("ECOZ2_VQ_INCREMENTAL")) s->incr_enabled = atoi(inc) != 0;
    if (const char* pf1 = getenv("ECOZ2_VQ_PLAIN_FIRST")) s->plain_first = atoi(pf1) != 0;
    if (const char* fm = getenv("ECOZ2_VQ_FAMILY")) s->fam_enabled = atoi(fm) != 0;
    if (const char* fm = getenv("ECOZ2_VQ_FAMILY_MIN_M")) s->fam_min_M = std::max(64, atoi(fm));
    if (const char* rc = getenv("ECOZ2_VQ_RECORDS")) s->rec_enabled = atoi(rc) != 0;
    if (const char* rc = getenv("ECOZ2_VQ_RECORDS_MIN_M")) s->rec_min_M = std::max(64, atoi(rc));
    if (const char* rc = getenv("ECOZ2_VQ_RECORDS_MAX_MB")) s->rec_max_bytes = (size_t)std::max(0, atoi(rc)) << 20;
    if (const char* rc = getenv("ECOZ2_VQ_RECORDS_FEW_DIV")) s->rec_few_div = std::max(0, atoi(rc));
    if (const char* sw = getenv("ECOZ2_VQ_SPLIT_SWEEP")) s->sweep2_enabled = atoi(sw) != 0;
    if (const char* sw = getenv("ECOZ2_VQ_TWO_STAGE")) s->two_stage_enabled = atoi(sw) != 0;
    if (const char* sw = getenv("ECOZ2_VQ_FUSED_SORTED")) s->fused_enabled = atoi(sw) != 0;
    if (const char* sw = getenv("ECOZ2_VQ_FUSED_MIN_M")) s->fused_min_M = std::max(64, atoi(sw));
    // With the recorded accumulate the prefiltered pass also wins at M = 128 (0.36 vs 0.43 ms per pass on 2^21 frames; not
    // at 64: 0.30 vs 0.28), and a seeded first pass halves the records of every prefiltered level's first pass
    if (s->rec_enabled && e2vq::prefilter_lds_stage(s->NC)) {
        if (!getenv("ECOZ2_VQ_PREFILTER_MIN_M")) s->pre_min_M = 128;
        if (!getenv("ECOZ2_VQ_FAMILY_MIN_M")) s->fam_min_M = 128;
    }
    if (s->pre_enabled) {
        HIPCHK(hipMalloc(&s->d_colmax, (size_t)s->NC * 8));
        HIPCHK(hipMalloc(&s->d_ea, (size_t)s->NC * sizeof(int)));
        HIPCHK(hipMalloc(&s->d_ps, e2vq::prefilter_scalars_bytes()));
        HIPCHK(hipMalloc(&s->d_ps2[0], e2vq::prefilter_scalars_bytes()));
        HIPCHK(hipMalloc(&s->d_ps2[1], e2vq::prefilter_scalars_bytes()));
    }
    return 0;
}

extern "C" void e2vq_session_destroy(e2vq_session* s);

extern "C" int e2vq_session_create(int device, int prediction_order, e2vq_session** out)
{
    *out = nullptr;
    if (prediction_order < 1 || prediction_order > E2VQ_MAX_P)
        return e2vq_set_error("prediction order %d out of range [1, %d]", prediction_order, E2VQ_MAX_P);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return e2vq_set_error("no HIP device available (%s); this library has no CPU path",
                              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return e2vq_set_error("device %d not in [0, %d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    e2vq_session* s = new e2vq_session();
    s->device = device;
    s->P = prediction_order;
    s->NC = prediction_order + 1;
    s->FB = 64;  // every kernel works on blocks of 64 frames
    s->RS = e2vq::row_stride(s->NC);
    s->NPAD = e2vq::cb_pad(s->NC);
    if (session_init(s)) {  // message already set; release whatever was created
        e2vq_session_destroy(s);
        return 1;
    }
    *out = s;
    return 0;
}

extern "C" void e2vq_session_destroy(e2vq_session* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    void* ptrs[] = {s->d_cbT, s->sv.refl, s->sv.rows, s->sv.rows_local, s->sv.cells, s->d_rows_parent, s->d_fam, s->d_aos, s->d_refl_spec, s->d_cbq_spec, s->d_cbm_spec, s->d_l1max_spec, s->d_cbm, s->d_blk,   s->d_refl,  s->d_refl_next, s->d_cbq,  s->d_l1max, s->d_sc,   s->d_maxabs, s->d_flags,
                    s->d_stats, s->d_rows,  s->d_S,         s->d_within, s->d_lstats, s->d_qaos, s->d_qblk,   s->d_qsym,
                    s->d_qdmin, s->d_colmax, s->d_ea, s->d_fimg, s->d_fg, s->d_cimg2[0], s->d_cimg2[1], s->d_ps2[0], s->d_ps2[1], s->d_ps, s->d_fblist, s->d_prev_sym, s->d_rows_local, s->d_recs, s->d_rec_counts, s->d_fimgF, s->d_perm, s->d_cand, s->d_sort,
                    s->d_ea_q, s->d_qfimg, s->d_qfg, s->d_qfblist, s->d_qcimg};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (auto* list : {&s->ar_pending, &s->ar_free})
        for (auto& ev : *list) {
            (void)hipEventDestroy(ev.a);
            (void)hipEventDestroy(ev.b);
        }
    if (s->ev_stats) (void)hipEventDestroy(s->ev_stats);
    if (s->h_stats) (void)hipHostFree(s->h_stats);
    if (s->h_within) (void)hipHostFree(s->h_within);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->ev_mid) (void)hipEventDestroy(s->ev_mid);
    if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
    delete s;
}

extern "C" int e2vq_set_stream(e2vq_session* s, void* hip_stream)
{
    HIPCHK(hipStreamSynchronize(s->stream));
    s->stream = hip_stream ? (hipStream_t)hip_stream : s->own_stream;
    return 0;
}

extern "C" int e2vq_set_allreduce(e2vq_session* s, e2vq_allreduce_fn fn, void* user, int rank, int world)
{
    s->allreduce = fn;
    s->ar_user = user;
    s->rank = rank;
    s->world = fn ? world : 1;
    return 0;
}

extern "C" int e2vq_synchronize(e2vq_session* s)
{
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

static int reduce(e2vq_session* s, void* buf, i64 count, int op)
{
    // ECOZ2_VQ_FORCE_ALLREDUCE: call the hook even for a single rank (tests exercise the RCCL plumbing on one GPU)
    if (!s->allreduce || (s->world <= 1 && !s->ar_force && !getenv("ECOZ2_VQ_FORCE_ALLREDUCE"))) return 0;
    e2vq_session::ArTimed ev{nullptr, nullptr};
    if (s->ar_timing) {
        if (!s->ar_free.empty()) {
            ev = s->ar_free.back();
            s->ar_free.pop_back();
        } else {
            HIPCHK(hipEventCreate(&ev.a));
            HIPCHK(hipEventCreate(&ev.b));
        }
        HIPCHK(hipEventRecord(ev.a, s->stream));
    }
    const int rc = s->allreduce(s->ar_user, buf, count, op, (void*)s->stream);
    if (rc != 0) return e2vq_set_error("all-reduce hook failed (%d)", rc);
    if (s->ar_timing) {
        HIPCHK(hipEventRecord(ev.b, s->stream));
        s->ar_pending.push_back(ev);
        s->ar_calls += 1;
        s->ar_bytes += (long)count * 8;
    }
    return 0;
}

// Device time of the exchange: events on the session's stream around every call of the all-reduce hook (what lies
// between them is the collective's kernels and their wait for the other ranks).  e2vq_collective_timing synchronises the
// stream and returns the totals since the timing was switched on.
extern "C" int e2vq_enable_collective_timing(e2vq_session* s, int on)
{
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (auto& ev : s->ar_pending) s->ar_free.push_back(ev);
    s->ar_pending.clear();
    s->ar_timing = on != 0;
    s->ar_ms = 0.0;
    s->ar_calls = s->ar_bytes = 0;
    return 0;
}

extern "C" int e2vq_collective_timing(e2vq_session* s, double* total_ms, int64_t* calls, int64_t* bytes)
{
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (auto& ev : s->ar_pending) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
        s->ar_ms += ms;
        s->ar_free.push_back(ev);
    }
    s->ar_pending.clear();
    if (total_ms) *total_ms = s->ar_ms;
    if (calls) *calls = s->ar_calls;
    if (bytes) *bytes = s->ar_bytes;
    return 0;
}

// Switches the prefiltered sweep off (every pass on the plain FP64 sweep) or back on for this session: same results
// either way -- bench.py re-runs its timed level both ways in one process and compares the codebooks bit for bit.
// (Switching it ON needs the images a session makes when it is created and given its frames with the prefilter enabled.)
extern "C" int e2vq_set_prefilter(e2vq_session* s, int on)
{
    const bool want = on != 0;
    if (want && !(s->d_ps2[0] && e2vq::prefilter_supports(s->NC, 64)))
        return e2vq_set_error("this session has no prefilter images (created with ECOZ2_VQ_PREFILTER=0, or an unsupported order)");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->pre_enabled = want;
    s->incr_valid = false;
    s->fam_pending = false;
    s->img_valid[0] = s->img_valid[1] = false;
    s->spec_valid = false;
    return 0;
}

// ---- training set ------------------------------------------------------------------------

// adopt (optional): the caller's buffer holds at least ((T + 63) / 64) * 64 rows and may be kept -- when the session wants a
// row-major copy of its own (the LDS-staged prefiltered pass), it takes the buffer as that copy instead of making one (3 GB less
// to allocate, touch and copy for 10 M frames) and sets *adopt; the caller then must not free it.
static int set_frames_device_impl(e2vq_session* s, const void* device_frames, int64_t T, bool* adopt)
{
    if (adopt) *adopt = false;
    if (T < 1) return e2vq_set_error("empty training set");
    // frame indices travel as 32-bit ints in the fallback lists / block tables of the kernels
    if (T > (int64_t)INT32_MAX - 64) return e2vq_set_error("%lld frames exceed the per-session limit of 2^31 - 65", (long long)T);
    HIPCHK(hipSetDevice(s->device));
    // whatever happens below, the previous training set is gone: nothing may sweep a stale or null buffer
    s->prepared = false;
    s->stats_valid = false;
    s->rows_fresh = false;
    s->spec_valid = false;
    s->incr_valid = false;
    s->fam_pending = false;
    s->cells_M = 0;
    s->maxabs_scanned = false;
    s->T = 0;
    s->nblocks = 0;
    s->sv.valid = false;  // (rows and cells of a saved point describe the frames that are going away)
    HIPCHK(hipStreamSynchronize(s->stream));  // no kernel of this session still reads the old blocks
    if (s->d_blk) HIPCHK(hipFree(s->d_blk));
    s->d_blk = nullptr;
    const i64 nblocks = (T + s->FB - 1) / s->FB;
    HIPCHK(hipMalloc(&s->d_blk, (size_t)nblocks * s->NC * s->FB * 8));
    s->T = T;
    s->nblocks = nblocks;
    HIPCHK(hipMemsetAsync(s->d_maxabs, 0, 8, s->stream));
    HIPCHK(hipMemsetAsync(s->d_flags, 0, 2 * sizeof(int), s->stream));
    s->maxabs_scanned = e2vq::launch_blockify((const double*)device_frames, T, s->NC, s->FB, s->d_blk, s->nblocks,
                                              s->d_maxabs, s->d_flags, s->stream);
    HIPCHK(hipGetLastError());
    if (s->pre_enabled) {
        if (s->d_fimg) HIPCHK(hipFree(s->d_fimg));
        if (s->d_fg) HIPCHK(hipFree(s->d_fg));
        if (s->d_fblist) HIPCHK(hipFree(s->d_fblist));
        if (s->d_prev_sym) HIPCHK(hipFree(s->d_prev_sym));
        if (s->d_aos) HIPCHK(hipFree(s->d_aos));
        s->d_aos = nullptr;
        s->d_prev_sym = nullptr;
        s->d_fimg = nullptr;
        s->d_fg = nullptr;
        s->d_fblist = nullptr;
        // 234 B per frame beside the 296 B of the blocked frames.  If the device cannot hold them, the session
        // simply keeps to the plain FP64 sweep (same results): use_prefilter() looks at d_fimg.
        // (prev_sym: + 256 B, k_pass_pre_lds fetches the 64 cells of a block as 64 dwords)
        const bool fits = hipMalloc(&s->d_fimg, e2vq::prefilter_frame_image_bytes(s->NC, s->nblocks)) == hipSuccess &&
                          hipMalloc(&s->d_fg, (size_t)s->nblocks * 64 * sizeof(float)) == hipSuccess &&
                          hipMalloc(&s->d_fblist, (size_t)s->nblocks * 64 * sizeof(int)) == hipSuccess &&
                          hipMalloc(&s->d_prev_sym, (size_t)s->nblocks * 64 * sizeof(unsigned short) + 256) == hipSuccess;
        // the accumulating prefiltered pass stages the FP64 frames of a block in LDS from a row-major copy (another 296 B
        // per frame; without it the pass keeps to the round-2 kernel, which reads the blocked layout)
        if (fits && e2vq::prefilter_lds_stage(s->NC)) {
            const size_t rows = (size_t)s->nblocks * 64, have = (size_t)T;
            if (adopt) {
                s->d_aos = (double*)const_cast<void*>(device_frames);
                *adopt = true;
                if (rows > have) HIPCHK(hipMemsetAsync(s->d_aos + have * s->NC, 0, (rows - have) * s->NC * 8, s->stream));
                HIPCHK(hipMemsetAsync(s->d_fg, 0, (size_t)s->nblocks * 64 * sizeof(float), s->stream));
            } else if (hipMalloc(&s->d_aos, rows * s->NC * 8 + 16) == hipSuccess) {  // (+ 16: the fused pass fetches rows in 16-byte pieces)
                HIPCHK(hipMemcpyAsync(s->d_aos, device_frames, have * s->NC * 8, hipMemcpyDeviceToDevice, s->stream));
                if (rows > have) HIPCHK(hipMemsetAsync(s->d_aos + have * s->NC, 0, (rows - have) * s->NC * 8, s->stream));
                HIPCHK(hipMemsetAsync(s->d_fg, 0, (size_t)s->nblocks * 64 * sizeof(float), s->stream));
            } else {
                (void)hipGetLastError();
                s->d_aos = nullptr;
            }
        }
        for (void** p : {(void**)&s->d_fimgF, (void**)&s->d_perm, (void**)&s->d_cand}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        s->perm_M = 0;
        s->two_stage_off_until_M = 0;
        s->sw_pending = false;
        if (fits) {
            e2vq::launch_prefilter_frames(s->d_blk, T, s->nblocks, s->NC, s->d_colmax, s->d_ea, s->d_fimg, s->d_fg,
                                          s->stream);
            HIPCHK(hipGetLastError());
            // round 5: the frame-major image the candidate sweep gathers from, the sorted list and the candidates (another
            // 264 B per frame at P = 36); without them recorded passes keep to round 4's fused kernel
            if (s->sweep2_enabled && s->d_aos && e2vq::sweep_supported(s->NC, 64)) {
                const size_t slots = (size_t)s->nblocks * 64;
                bool ok = hipMalloc(&s->d_fimgF, e2vq::sweep_frame_image_bytes(s->NC, s->nblocks)) == hipSuccess &&
                          hipMalloc(&s->d_perm, slots * sizeof(unsigned)) == hipSuccess &&
                          hipMalloc(&s->d_cand, slots * sizeof(unsigned)) == hipSuccess;
                if (ok && !s->d_sort) {
                    ok = hipMalloc(&s->d_sort, e2vq::sort_scratch_bytes()) == hipSuccess;
                    if (ok) HIPCHK(hipMemsetAsync(s->d_sort, 0, e2vq::sort_scratch_bytes(), s->stream));
                }
                if (ok) {
                    e2vq::launch_sweep_frames(s->d_aos, T, s->nblocks, s->NC, s->d_ea, s->d_fimgF, s->stream);
                    HIPCHK(hipGetLastError());
                } else {
                    (void)hipGetLastError();
                    for (void** p : {(void**)&s->d_fimgF, (void**)&s->d_perm, (void**)&s->d_cand}) {
                        if (*p) (void)hipFree(*p);
                        *p = nullptr;
                    }
                }
            }
        } else {
            (void)hipGetLastError();  // clear the out-of-memory status
            for (void** p : {(void**)&s->d_fimg, (void**)&s->d_fg, (void**)&s->d_fblist, (void**)&s->d_prev_sym, (void**)&s->d_aos}) {
                if (*p) (void)hipFree(*p);
                *p = nullptr;
            }
            if (!getenv("ECOZ2_VQ_QUIET"))
                fprintf(stderr, "ecoz2vq: no room for the prefilter images of %lld frames; using the plain FP64 sweep\n",
                        (long long)T);
        }
    }
    // The re-layout kernels read `device_frames` on the session's stream: the caller must have finished writing it
    // (or have written it on this stream).  They are complete when this returns, so the buffer may be reused or freed.
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int e2vq_set_frames_device(e2vq_session* s, const void* device_frames, int64_t T)
{
    return set_frames_device_impl(s, device_frames, T, nullptr);
}

extern "C" int e2vq_set_frames_host(e2vq_session* s, const double* frames, int64_t T)
{
    if (T < 1) return e2vq_set_error("empty training set");
    HIPCHK(hipSetDevice(s->device));
    double* tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, (size_t)T * s->NC * 8));
    HIPCHK(hipMemcpyAsync(tmp, frames, (size_t)T * s->NC * 8, hipMemcpyHostToDevice, s->stream));
    int rc = e2vq_set_frames_device(s, tmp, T);
    hipError_t e = hipStreamSynchronize(s->stream);
    (void)hipFree(tmp);
    if (rc) return rc;
    if (e != hipSuccess) return e2vq_set_error("upload failed: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int e2vq_prepare(e2vq_session* s)
{
    if (!s->d_blk) return e2vq_set_error("no training set");
    HIPCHK(hipSetDevice(s->device));
    const long count = (long)s->nblocks * s->NC * s->FB;
    if (!s->maxabs_scanned) {  // (the re-layout kernel of the MFMA path already scanned max |x|)
        HIPCHK(hipMemsetAsync(s->d_maxabs, 0, 8, s->stream));
        HIPCHK(hipMemsetAsync(s->d_flags, 0, 2 * sizeof(int), s->stream));
        e2vq::launch_maxabs(s->d_blk, count, s->d_maxabs, s->d_flags, s->stream);
    }
    s->maxabs_scanned = false;  // the all-reduce below overwrites the local maximum: rescan if prepare runs again
    if (reduce(s, s->d_maxabs, 1, 1)) return 1;
    e2vq::launch_finish_scalars(s->d_maxabs, s->d_sc, s->stream);
    HIPCHK(hipMemsetAsync(s->d_stats, 0, (size_t)(2 * s->NC + 3) * 8, s->stream));
    e2vq::launch_global_sums(s->d_blk, s->nblocks, s->NC, s->FB, s->d_sc, s->d_stats, s->stream);
    const i64 Tl = s->T;
    HIPCHK(hipMemcpyAsync(s->d_stats + 2 * s->NC + 2, &Tl, 8, hipMemcpyHostToDevice, s->stream));
    if (reduce(s, s->d_stats, 2 * s->NC + 3, 0)) return 1;
    // every rank learns of bad data in ANY shard (the two status words as one unsigned 64-bit maximum): all of them stop
    // here together instead of one rank leaving the others to a collective it will never join
    if (reduce(s, s->d_flags, 1, 1)) return 1;
    e2vq::launch_finish_q(s->d_stats, s->NC, s->d_sc, s->stream);
    int flags[2];
    i64 Ttot = 0;
    HIPCHK(hipMemcpyAsync(flags, s->d_flags, sizeof flags, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(&Ttot, s->d_stats + 2 * s->NC + 2, 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipMemcpyAsync(&s->h_sc, s->d_sc, sizeof(DevScalars), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipGetLastError());
    if (flags[0]) return e2vq_set_error("training set contains NaN or infinite values");
    if (!(s->h_sc.maxabs > 0.0)) return e2vq_set_error("training set is all zeros");
    s->T_total = Ttot;
    s->prepared = true;
    s->DDprv = DBL_MAX / 1e5;
    return 0;
}

// ---- codebook ----------------------------------------------------------------------------

// (callers that redefine the codebook's size or contents from outside: set / init / grow)
// zeroed_with_scale >= 0 (e2vq_grow): the L1 maximum and the scalars of limb image `zeroed_with_scale` have been zeroed by
// the kernel in front (k_grow), and the update kernel also finds the image's scale -- the image kernel of the first pass then
// needs neither a memset nor k_pre_cmax
static int codebook_prepare(e2vq_session* s, bool redefined = true, bool grown = false, int zeroed_with_scale = -1)
{
    if (redefined) s->incr_valid = false;
    if (!grown) s->fam_pending = false;  // (set / init: whatever e2vq_grow stashed belongs to another codebook)
    s->img_valid[0] = s->img_valid[1] = false;  // the codebook in d_cbq is a new one
    s->cb_version++;
    if (e2vq::has_cell_update(s->NC)) {
        const bool scale = zeroed_with_scale >= 0;  // (-2: only the L1 maximum has been zeroed)
        e2vq::launch_cell_update(nullptr, s->M, s->NC, s->d_sc, s->d_refl, nullptr, s->d_cbq, s->d_cbm, s->d_l1max,
                                 nullptr, nullptr, s->stream, /*zero_first=*/zeroed_with_scale == -1, scale ? s->d_ea : nullptr,
                                 scale ? e2vq::prefilter_codebook_scale(s->d_ps2[zeroed_with_scale]) : nullptr);
        if (scale) {
            s->scale_version = s->cb_version;
            s->scale_img = zeroed_with_scale;
        }
    } else
        e2vq::launch_codebook_prepare(s->d_refl, s->M, s->NC, s->d_cbq, s->d_l1max, s->d_cbm, s->stream);
    HIPCHK(hipGetLastError());
    s->stats_valid = false;
    s->rows_fresh = false;
    s->spec_valid = false;
    return 0;
}

extern "C" int e2vq_set_codebook(e2vq_session* s, const double* reflections, int M)
{
    if (M < 1 || M > 65536) return e2vq_set_error("codebook size %d out of range", M);
    HIPCHK(hipSetDevice(s->device));
    if (ensure_codebook_capacity(s, M)) return 1;
    HIPCHK(hipMemcpyAsync(s->d_refl, reflections, (size_t)M * s->NC * 8, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->M = M;
    return codebook_prepare(s);
}

extern "C" int e2vq_get_codebook(e2vq_session* s, double* reflections, int* M)
{
    HIPCHK(hipSetDevice(s->device));
    if (M) *M = s->M;
    if (reflections && s->M > 0) {
        HIPCHK(hipMemcpyAsync(reflections, s->d_refl, (size_t)s->M * s->NC * 8, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return 0;
}

extern "C" int e2vq_init_codebook(e2vq_session* s)
{
    if (!s->prepared) return e2vq_set_error("e2vq_prepare has not run");
    HIPCHK(hipSetDevice(s->device));
    if (ensure_codebook_capacity(s, 2)) return 1;
    e2vq::launch_init_codebook(s->d_stats, s->NC, s->d_sc, s->d_refl, s->d_flags + 1, s->stream);
    int st = 0;
    HIPCHK(hipMemcpyAsync(&st, s->d_flags + 1, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (st != 0) return e2vq_set_error("Levinson recursion failed on the global centroid (status %d)", st);
    s->M = 1;
    s->DDprv = DBL_MAX / 1e5;  // a fresh ladder: "e+303" in notes.md:128
    return codebook_prepare(s);
}

extern "C" int e2vq_grow(e2vq_session* s)
{
    if (s->M < 1) return e2vq_set_error("no codebook to grow");
    if (2 * s->M > 65536) return e2vq_set_error("codebook size limit (u16 symbols) reached");
    HIPCHK(hipSetDevice(s->device));
    // The rows and cells of the last pass over the codebook about to be split seed the first pass of the next size
    // (k_seed_family): they must be this rank's own sums, for the codebook as it stands, with every frame's cell recorded.
    const bool seed = s->fam_enabled && s->pre_enabled && s->d_aos && s->d_prev_sym && s->rows_fresh && s->rows_are_local &&
                      s->cells_M == s->M && 2 * s->M >= s->pre_min_M && 2 * s->M >= s->fam_min_M &&
                      e2vq::prefilter_supports(s->NC, 2 * s->M) &&
                      e2vq::prefilter_lds_stage(s->NC) && s->incr_enabled;
    if (seed) {
        if (s->fam_cap < s->M) {
            for (i64** p : {&s->d_rows_parent, &s->d_fam}) {
                if (*p) HIPCHK(hipFree(*p));
                *p = nullptr;
            }
            s->fam_cap = std::max(s->M, 1024);
            HIPCHK(hipMalloc(&s->d_rows_parent, (size_t)s->fam_cap * s->RS * 8));
            HIPCHK(hipMalloc(&s->d_fam, (size_t)s->fam_cap * s->RS * 8));
        }
        const i64* src = (s->d_rows_local && s->rows_local_is_current) ? s->d_rows_local : s->d_rows;
        HIPCHK(hipMemcpyAsync(s->d_rows_parent, src, (size_t)s->M * s->RS * 8, hipMemcpyDeviceToDevice, s->stream));
        s->fam_M = s->M;
    }
    if (ensure_codebook_capacity(s, 2 * s->M)) return 1;
    // one kernel zeroes what the kernels behind it accumulate into with atomicMax: the L1 maximum of the grown codebook and --
    // when its first pass will be a prefiltered one -- the scalars of the limb image that pass builds
    const int Mold = s->M;
    s->M = 2 * Mold;
    const bool fused = e2vq::has_cell_update(s->NC);
    const bool pre_next = fused && s->d_ea && s->d_ps2[s->img_cur] && use_prefilter(s, pass_mode(s));
    e2vq::ZeroList z{};
    int nz = 0;
    if (fused) {
        z.p[nz] = s->d_l1max;
        z.words[nz++] = 2;
        if (pre_next) {
            z.p[nz] = s->d_ps2[s->img_cur];
            z.words[nz++] = (int)(e2vq::prefilter_scalars_bytes() / 4);
        }
    }
    e2vq::launch_grow(s->d_refl, Mold, s->NC, s->d_refl_next, s->stream, &z);
    std::swap(s->d_refl, s->d_refl_next);
    if (codebook_prepare(s, true, /*grown=*/true, fused ? (pre_next ? s->img_cur : -2) : -1)) return 1;
    s->fam_pending = seed;
    return 0;
}

// ---- LBG iteration pieces ------------------------------------------------------------------

static int pass_mode(const e2vq_session* s)
{
    if (const char* f = getenv("ECOZ2_VQ_FORCE_MODE")) return atoi(f);  // diagnostics only (0 = assignment only)
    if (e2vq::uses_mfma(s->NC) && !e2vq::mfma_is_wide(s->NC)) {
        // all cells in the workgroup's LDS table while it fits beside the row images (NC = 37: M <= 128) ...
        const long images = 8L * 16 * (2 * s->NC + 5 + 3) * 4;
        if ((long)s->M * s->RS * 8 + images + 2048 <= E2VQ_LDS_BYTES && s->M <= 128) return 1;
        // ... then the hybrid, while its LDS share is worth having (atomic-bound levels)
        if (s->M <= 4 * e2vq::mfma_hybrid_cells(s->NC)) return 5;
        return 2;
    }
    return 2;  // generic kernel: global atomics
}

// adds the pass bracketed by ev0/ev1 to the running total; only called when those events have completed
static int fold_pending_timing(e2vq_session* s)
{
    if (!s->timing_pending) return 0;
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    s->timing_sum_ms += ms;
    s->timing_count += 1;
    if (s->timing_mid) HIPCHK(hipEventElapsedTime(&ms, s->ev0, s->ev_mid));
    s->timing_sweep_ms += ms;
    s->timing_mid = false;
    s->timing_pending = false;
    return 0;
}

// the prefiltered sweep serves the accumulate-by-global-atomics and assignment-only passes of large codebooks
static bool records_plan(const e2vq_session* s, int M, bool family, e2vq::PassRecords* plan, size_t* bytes);
static bool use_prefilter(const e2vq_session* s, int mode)
{
    if (!(s->pre_enabled && s->d_fimg && (mode == 1 || mode == 2 || mode == 5 || mode == 0) && s->M >= s->pre_min_M &&
          e2vq::prefilter_supports(s->NC, s->M)))
        return false;
    if (mode == 0) return true;
    // an accumulating prefiltered pass needs the row-major resident copy with LDS room for a block of it, and an accumulate
    // that takes its rows: records, or -- rows of at most 80 elements -- the burst of atomics.  Anything else runs the
    // plain FP64 sweep (round 2's accumulating kernel, which served those cases, left in round 5).
    return s->d_aos && e2vq::prefilter_lds_stage(s->NC) &&
           (e2vq::prefilter_burst_supported(s->NC) || records_plan(s, s->M, false, nullptr, nullptr));
}

// the recorded accumulate for accumulating prefiltered passes at this codebook size?  (plan: filled in but for the pointers)
static bool records_plan(const e2vq_session* s, int M, bool family, e2vq::PassRecords* plan, size_t* bytes)
{
    e2vq::PassRecords p{};
    size_t b = 0;
    if (!(s->rec_enabled && s->pre_enabled && s->incr_enabled && s->d_aos && s->d_prev_sym && M >= s->pre_min_M &&
          M >= s->rec_min_M && e2vq::prefilter_records_plan(s->NC, M, family, s->nblocks, &p, &b) && b <= s->rec_max_bytes))
        return false;
    if (plan) *plan = p;
    if (bytes) *bytes = b;
    return true;
}

static int ensure_codebook_image(e2vq_session* s)
{
    if (s->M <= s->cimg_cap) return 0;
    for (int k = 0; k < 2; ++k) {
        if (s->d_cimg2[k]) HIPCHK(hipFree(s->d_cimg2[k]));
        s->d_cimg2[k] = nullptr;
        s->img_valid[k] = false;
    }
    s->cimg_cap = std::max(s->M, 2048);
    for (int k = 0; k < 2; ++k)
        HIPCHK(hipMalloc(&s->d_cimg2[k], e2vq::prefilter_codebook_image_bytes(s->NC, s->cimg_cap)));
    return 0;
}

extern "C" int e2vq_pass(e2vq_session* s, void* device_sym, void* device_dmin)
{
    if (!s->prepared) return e2vq_set_error("e2vq_prepare has not run");
    if (s->M < 1) return e2vq_set_error("no codebook");
    HIPCHK(hipSetDevice(s->device));
    if (s->timing && s->timing_pending) {  // the previous timed pass has long finished (its statistics were read)
        HIPCHK(hipEventSynchronize(s->ev1));
        if (fold_pending_timing(s)) return 1;
    }
    const int mode = pass_mode(s);
    s->last_prefiltered = use_prefilter(s, mode);
    s->last_kind = s->last_prefiltered ? 1 : 0;  // (the split / fused branches below set 2 / 3)
    s->last_two_stage = false;
    const bool collective = s->allreduce && (s->world > 1 || s->ar_force || getenv("ECOZ2_VQ_FORCE_ALLREDUCE"));
    const bool keep = s->last_prefiltered && mode != 0 && s->incr_enabled;  // rows and cells persist for the next pass
    i64* rows = s->d_rows;
    if (keep && collective) {  // the all-reduce overwrites d_rows: accumulate into the rank's own copy
        if (s->rows_local_cap < s->M_cap) {
            if (s->d_rows_local) HIPCHK(hipFree(s->d_rows_local));
            s->d_rows_local = nullptr;
            HIPCHK(hipMalloc(&s->d_rows_local, (size_t)s->M_cap * s->RS * 8));
            s->rows_local_cap = s->M_cap;
            s->incr_valid = false;
        }
        rows = s->d_rows_local;
    }
    const bool incremental = keep && s->incr_valid && s->incr_M == s->M;
    // the first pass after a split, seeded with the parents' sums (e2vq_grow stashed them): k_seed_family
    bool family = s->fam_pending && keep && !incremental && s->d_aos && 2 * s->fam_M == s->M && mode != 0;
    s->fam_pending = false;
    // round 5: the frames are grouped by cell (a seeded first pass, or an incremental one) -> the fused sorted pass: sweep,
    // exact evaluation, outputs and the cell sums reduced in the block, one kernel (vq_sweep.hip); no records
    // (from fused_min_M codewords on: at M = 128 a pass is bound by reading its frames, which round 4's kernel does in
    // their natural order -- 0.35 against 0.40 ms on 2^21 frames; at 256 the two are level, beyond it the sorted pass wins)
    const bool fused = keep && mode != 0 && (family || incremental) && s->sweep2_enabled && s->fused_enabled && s->d_fimgF &&
                       s->d_aos && s->M >= s->fused_min_M && e2vq::sweep_supported(s->NC, s->M);
    // round 4: contributions recorded by the sweep, folded into the rows by k_reduce_records
    e2vq::PassRecords recplan{};
    bool records = false;
    if (fused) {
        s->last_recorded = false;
    } else if (keep && mode != 0) {
        size_t bytes = 0;
        records = records_plan(s, s->M, family, &recplan, &bytes);
        // few records on the last pass of this level: the rest of the level adds its contributions as a burst
        if (!incremental) {
            s->rec_level_burst = false;
            s->rec_last_total = -1;
        } else if (records && !(s->sweep2_enabled && s->d_fimgF && s->M >= s->fused_min_M) && s->rec_few_div > 0 && e2vq::prefilter_burst_supported(s->NC) &&
                   (s->rec_level_burst || (s->rec_last_total >= 0 && s->rec_last_total < s->T / s->rec_few_div))) {
            s->rec_level_burst = true;
            records = false;
        }
        if (records && bytes > s->recs_cap) {
            // (grown rarely: sized at once for a codebook four times this one's when the limit allows)
            size_t want = bytes;
            e2vq::PassRecords big{};
            size_t bb = 0;
            if (4 * s->M <= 4096 && e2vq::prefilter_records_plan(s->NC, 4 * s->M, true, s->nblocks, &big, &bb) && bb <= s->rec_max_bytes / 4)
                want = std::max(want, bb);
            if (s->d_recs) HIPCHK(hipFree(s->d_recs));
            s->d_recs = nullptr;
            s->recs_cap = 0;
            if (hipMalloc(&s->d_recs, want) == hipSuccess) {
                s->recs_cap = want;
            } else {
                (void)hipGetLastError();
                if (want > bytes && hipMalloc(&s->d_recs, bytes) == hipSuccess)
                    s->recs_cap = bytes;
                else
                    (void)hipGetLastError(), records = false;  // (no room: the burst of atomics instead)
            }
        }
        if (records && !s->d_rec_counts) HIPCHK(hipMalloc(&s->d_rec_counts, 256 * 64 * sizeof(int)));
        recplan.recs = s->d_recs;
        recplan.counts = s->d_rec_counts;
        s->last_recorded = records;
        if (records) {
            void* dt = nullptr;
            HIPCHK(hipHostGetDevicePointer(&dt, (void*)&s->h_stats->rec_total, 0));
            recplan.total_out = (long long*)dt;
            s->rec_pending = true;
        }
    }
    // (P = 40: rows of 83 elements are seeded only where the contributions are recorded -- the burst cannot add them)
    if (family && !records && !fused && !e2vq::prefilter_burst_supported(s->NC)) family = false;
    // The first pass of a level accumulates in full.  For the smallest prefiltered sizes that is cheaper on the plain
    // FP64 sweep with its workgroup-local LDS table (hybrid accumulate: 0.92 vs 1.5 ms at M = 256, where 2^21 frames
    // hammer 256 rows with global atomics); it records the cells for the incremental passes that follow.
    const bool plain_first = s->plain_first && keep && !incremental && !family && !records && !fused && mode == 5 && s->M <= 384;
    if (s->last_prefiltered && !plain_first && ensure_codebook_image(s)) return 1;
    {
        // one prologue launch: the rows (all of them, or the distortion columns of an incremental pass), the fallback
        // count of a codebook image that is already there, and what the speculative update after this pass
        // accumulates into with atomicMax (the shadow codebook's L1 max and the scalars of its limb image)
        e2vq::ZeroList z{};
        int nz = 0;
        if (s->last_prefiltered && !plain_first && s->img_valid[s->img_cur]) {
            z.p[nz] = (void*)e2vq::prefilter_fallback_count(s->d_ps2[s->img_cur]);
            z.words[nz++] = 1;
        }
        z.p[nz] = s->d_l1max_spec;
        z.words[nz++] = 2;
        if (s->d_ps2[1 - s->img_cur]) {
            z.p[nz] = s->d_ps2[1 - s->img_cur];
            z.words[nz++] = (int)(e2vq::prefilter_scalars_bytes() / 4);
        }
        // (the seeded first pass writes every word of the rows itself: k_seed_family takes the small words along)
        if (family)
            e2vq::launch_seed_family(s->d_rows_parent, rows, s->d_fam, s->fam_M, s->NC, s->stream, &z);
        else
            e2vq::launch_pass_prologue(rows, s->M, s->NC, incremental ? 2 : (mode != 0 ? 1 : 0), z, s->stream);
        s->spec_zeroed = true;
    }
    // a plain pass records every frame's cell when the next size could be seeded from it (the level below the first
    // prefiltered one)
    const bool record_cells = !s->last_prefiltered && !plain_first && mode != 0 && s->fam_enabled && s->pre_enabled &&
                              s->d_prev_sym && s->d_aos && !device_sym && 2 * s->M >= s->pre_min_M && 2 * s->M >= s->fam_min_M &&
                              e2vq::prefilter_supports(s->NC, 2 * s->M);
    if (plain_first) {
        unsigned short* sym_out = device_sym ? (unsigned short*)device_sym : s->d_prev_sym;
        if (s->timing) HIPCHK(hipEventRecord(s->ev0, s->stream));
        s->n_plain_launches++;
        e2vq::launch_pass(s->NC, mode, s->d_blk, s->T, s->nblocks, s->d_cbq, s->d_cbm ? s->d_cbm : s->d_cbT, s->M, s->d_sc, s->d_l1max, sym_out,
                          (double*)device_dmin, rows, s->stream);
        if (s->timing) {
            HIPCHK(hipEventRecord(s->ev1, s->stream));
            s->timed = true;
            s->timing_pending = true;
        }
        if (sym_out != s->d_prev_sym)
            HIPCHK(hipMemcpyAsync(s->d_prev_sym, sym_out, (size_t)s->T * sizeof(unsigned short), hipMemcpyDeviceToDevice,
                                  s->stream));
        s->last_prefiltered = false;
    } else if (s->last_prefiltered) {
        // f16 limb image of the current codebook, prefiltered sweep (exact evaluation of the certified top two),
        // then the full FP64 sweep of whatever it could not certify
        const int k = s->img_cur;
        if (!s->img_valid[k])  // (else: built ahead by e2vq_pass_stats for the codebook committed since; the prologue
                               // restarted its fallback count)
            e2vq::launch_prefilter_codebook(s->d_cbq, s->M, s->NC, s->d_ea, s->d_ps2[k], s->d_cimg2[k], s->stream,
                                            /*scale_ready=*/s->scale_version == s->cb_version && s->scale_img == k);
        s->img_valid[k] = true;
        s->img_last = k;
        void* const d_cimg = s->d_cimg2[k];
        void* const d_ps = s->d_ps2[k];
        if (s->timing) HIPCHK(hipEventRecord(s->ev0, s->stream));
        s->n_pre_launches++;
        if (fused) {
            // round 5, frames grouped: [sort] -> ONE kernel (two-stage sweep, exact evaluation, outputs, cell sums in the block)
            const int incr = family ? 2 : 1;
            if (incr == 2 || s->perm_M != s->M) {
                if (e2vq::launch_sort_by_cell(s->d_prev_sym, s->T, s->nblocks, incr == 2 ? s->M / 2 : s->M, s->d_sort, s->d_perm,
                                              s->stream))
                    return e2vq_set_error("sort by cell: unsupported size");
                s->perm_M = s->M;
            }
            // (two stages need tiles to skip: from eight tiles on; below, the home tile alone is a quarter or half of the codebook)
            const bool two = s->two_stage_enabled && s->M >= 256 && s->M > s->two_stage_off_until_M;
            // the flagged fraction is looked at once per level: on its first pass
            const bool count = two && incr == 2;
            s->last_kind = 3;
            s->last_two_stage = two;
            if (incr == 2) s->last_flagged_frac = -1.0;
            if (s->timing) HIPCHK(hipEventRecord(s->ev0, s->stream));  // (again: behind the sort)
            if (e2vq::launch_pass_sorted(s->NC, two, s->d_fimgF, s->d_perm, s->T, s->nblocks, d_cimg, d_ps, s->d_cbq, s->M, s->d_aos,
                                         s->d_sc, s->d_l1max, (unsigned short*)device_sym, (double*)device_dmin, rows,
                                         family ? s->d_fam : nullptr, s->d_fblist, s->d_prev_sym, incr,
                                         count ? e2vq::sweep_counters_of(s->d_sort) : nullptr, s->stream))
                return e2vq_set_error("fused sorted pass: unsupported configuration");
            if (s->timing) {
                HIPCHK(hipEventRecord(s->ev1, s->stream));
                s->timed = true;
                s->timing_pending = true;
            }
            if (count) {
                void* sw_host = nullptr;
                HIPCHK(hipHostGetDevicePointer(&sw_host, (void*)&s->h_stats->sw_flagged, 0));
                e2vq::launch_sweep_counters_out(e2vq::sweep_counters_of(s->d_sort), sw_host, s->stream);
                s->sw_pending = true;
            }
            e2vq::launch_pass_fallback(s->NC, true, s->d_blk, s->d_cbm, s->M, s->d_sc, s->d_l1max, (unsigned short*)device_sym,
                                       (double*)device_dmin, rows, s->d_fblist, e2vq::prefilter_fallback_count(d_ps), s->d_prev_sym,
                                       incr, s->stream);
            if (family) e2vq::launch_family_fixup(rows, s->d_fam, s->fam_M, s->NC, s->stream);
        } else if (records && s->sweep2_enabled && s->d_fimgF && s->M >= s->fused_min_M && e2vq::sweep_supported(s->NC, s->M)) {
            // round 5: [sort] -> candidate sweep -> finishing kernel (exact evaluation, outputs, records) -> reduce
            const int incr = family ? 2 : (incremental ? 1 : 0);
            if (incr != 0 && (incr == 2 || s->perm_M != s->M)) {
                if (e2vq::launch_sort_by_cell(s->d_prev_sym, s->T, s->nblocks, incr == 2 ? s->M / 2 : s->M, s->d_sort, s->d_perm,
                                              s->stream))
                    return e2vq_set_error("sort by cell: unsupported size");
                s->perm_M = s->M;
            }
            const bool sorted = incr != 0 && s->perm_M == s->M;
            // (two stages need tiles to skip: from eight tiles on; below, the home tile alone is a quarter or half of the codebook)
            const bool two = sorted && s->two_stage_enabled && s->M >= 256 && s->M > s->two_stage_off_until_M;
            s->last_kind = 2;
            s->last_two_stage = two;
            if (s->timing) HIPCHK(hipEventRecord(s->ev0, s->stream));  // (again: the sweep kernel alone is what ev0..ev_mid brackets)
            if (e2vq::launch_sweep_candidates(s->NC, two, s->d_fimgF, sorted ? s->d_perm : nullptr, s->T, s->nblocks, d_cimg, d_ps, s->M,
                                              sorted ? s->d_prev_sym : nullptr, sorted ? incr : 0, s->d_cand,
                                              two ? e2vq::sweep_counters_of(s->d_sort) : nullptr, s->stream))
                return e2vq_set_error("candidate sweep: unsupported configuration");
            if (s->timing) {
                HIPCHK(hipEventRecord(s->ev_mid, s->stream));
                s->timing_mid = true;
            }
            void* sw_host = nullptr;
            if (two) {
                HIPCHK(hipHostGetDevicePointer(&sw_host, (void*)&s->h_stats->sw_flagged, 0));
                s->sw_pending = true;
            }
            if (e2vq::launch_finish(s->NC, s->d_aos, s->T, s->nblocks, s->d_cand, d_ps, s->d_cbq, s->M, s->d_sc, s->d_l1max,
                                    (unsigned short*)device_sym, (double*)device_dmin, rows, s->d_fblist, s->d_prev_sym, incr,
                                    &recplan, two ? e2vq::sweep_counters_of(s->d_sort) : nullptr, sw_host, s->stream))
                return e2vq_set_error("finishing kernel: unsupported configuration");
            if (e2vq::launch_reduce_records(s->NC, s->d_aos, recplan, incremental, s->d_sc, rows, family ? s->d_fam : nullptr,
                                            s->stream))
                return e2vq_set_error("k_reduce_records: unsupported configuration");
            if (s->timing) {
                HIPCHK(hipEventRecord(s->ev1, s->stream));
                s->timed = true;
                s->timing_pending = true;
            }
            e2vq::launch_pass_fallback(s->NC, true, s->d_blk, s->d_cbm, s->M, s->d_sc, s->d_l1max, (unsigned short*)device_sym,
                                       (double*)device_dmin, rows, s->d_fblist, e2vq::prefilter_fallback_count(d_ps), s->d_prev_sym,
                                       incr, s->stream);
            if (family) e2vq::launch_family_fixup(rows, s->d_fam, s->fam_M, s->NC, s->stream);
        } else {
            if (e2vq::launch_pass_prefiltered(s->NC, mode != 0, s->d_blk, s->T, s->nblocks, s->d_fimg, s->d_fg, d_cimg, d_ps,
                                          s->d_cbq, s->M, s->d_sc, s->d_l1max, (unsigned short*)device_sym,
                                          (double*)device_dmin, rows, s->d_fblist, keep ? s->d_prev_sym : nullptr,
                                          incremental, s->stream,
                                          nullptr, nullptr, s->d_aos, family ? s->d_fam : nullptr, records ? &recplan : nullptr))
            return e2vq_set_error("prefiltered sweep: unsupported configuration");
        if (records && s->timing) {
            HIPCHK(hipEventRecord(s->ev_mid, s->stream));
            s->timing_mid = true;
        }
        if (records && e2vq::launch_reduce_records(s->NC, s->d_aos, recplan, incremental, s->d_sc, rows,
                                                   family ? s->d_fam : nullptr, s->stream))
            return e2vq_set_error("k_reduce_records: unsupported configuration");
        if (s->timing) {
            HIPCHK(hipEventRecord(s->ev1, s->stream));
            s->timed = true;
            s->timing_pending = true;
        }
        e2vq::launch_pass_fallback(s->NC, mode != 0, s->d_blk, s->d_cbm, s->M, s->d_sc, s->d_l1max,
                                   (unsigned short*)device_sym, (double*)device_dmin, rows, s->d_fblist,
                                   e2vq::prefilter_fallback_count(d_ps), keep ? s->d_prev_sym : nullptr,
                                   family ? 2 : (incremental ? 1 : 0), s->stream);
        if (family) e2vq::launch_family_fixup(rows, s->d_fam, s->fam_M, s->NC, s->stream);
        }
    } else {
        if (s->timing) HIPCHK(hipEventRecord(s->ev0, s->stream));
        s->n_plain_launches++;
        e2vq::launch_pass(s->NC, mode, s->d_blk, s->T, s->nblocks, s->d_cbq, s->d_cbm ? s->d_cbm : s->d_cbT, s->M, s->d_sc, s->d_l1max,
                          record_cells ? s->d_prev_sym : (unsigned short*)device_sym, (double*)device_dmin, rows, s->stream);
        if (s->timing) {
            HIPCHK(hipEventRecord(s->ev1, s->stream));
            s->timed = true;
            s->timing_pending = true;
        }
    }
    if (mode != 0) {
        s->incr_valid = keep;
        s->incr_M = s->M;
        // what a seeded first pass of the next size needs to know about this one
        s->cells_M = (keep || record_cells) ? s->M : 0;
        s->rows_local_is_current = rows != s->d_rows;
        s->rows_are_local = rows != s->d_rows || !collective;
    }
    if (rows != s->d_rows)
        HIPCHK(hipMemcpyAsync(s->d_rows, rows, (size_t)s->M * s->RS * 8, hipMemcpyDeviceToDevice, s->stream));
    HIPCHK(hipGetLastError());
    if (reduce(s, s->d_rows, (i64)s->M * s->RS, 0)) return 1;
    s->stats_valid = false;
    s->rows_fresh = true;
    s->spec_valid = false;
    s->img_valid[1 - s->img_cur] = false;
    return 0;
}

// did the last e2vq_pass record its contributions for k_reduce_records (1) or add them itself (0), and how many records the
// last recorded pass of this level wrote (-1: none yet; valid once that pass's statistics have been read)
extern "C" int e2vq_last_pass_records(e2vq_session* s, int* recorded, int64_t* records)
{
    if (recorded) *recorded = s->last_recorded ? 1 : 0;
    if (records) *records = s->rec_last_total;
    return 0;
}

extern "C" int e2vq_last_pass_sweep(e2vq_session* s, int* kind, int* two_stage, double* flagged_fraction)
{
    if (kind) *kind = s->last_prefiltered ? s->last_kind : 0;
    if (two_stage) *two_stage = s->last_two_stage ? 1 : 0;
    if (flagged_fraction) *flagged_fraction = s->last_flagged_frac;
    return 0;
}

extern "C" int e2vq_last_pass_info(e2vq_session* s, int* prefiltered, int64_t* fallback_frames)
{
    HIPCHK(hipSetDevice(s->device));
    if (prefiltered) *prefiltered = s->last_prefiltered ? 1 : 0;
    if (fallback_frames) {
        int n = 0;
        if (s->last_prefiltered) {
            HIPCHK(hipMemcpyAsync(&n, e2vq::prefilter_fallback_count(s->d_ps2[s->img_last]), sizeof(int),
                                  hipMemcpyDeviceToHost, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
        }
        *fallback_frames = n;
    }
    return 0;
}

// training-pass sweep launches so far, by kernel family: lets a profile of a whole run (rocprofv3 --kernel-trace)
// be cut to the dispatches of a timed region
extern "C" int e2vq_sweep_launch_counts(e2vq_session* s, int64_t* prefiltered, int64_t* plain)
{
    if (prefiltered) *prefiltered = s->n_pre_launches;
    if (plain) *plain = s->n_plain_launches;
    return 0;
}

extern "C" int e2vq_enable_timing(e2vq_session* s, int on)
{
    s->timing = on != 0;
    s->timed = false;
    s->timing_pending = false;
    s->timing_sum_ms = 0.0;
    s->timing_sweep_ms = 0.0;
    s->timing_mid = false;
    s->timing_count = 0;
    return 0;
}

// the sweep kernels alone: where a pass is a sweep + an accumulate kernel (recorded contributions + k_reduce_records),
// e2vq_timing_total covers both, this one the sweep
extern "C" int e2vq_timing_sweep_total(e2vq_session* s, double* total_ms, int64_t* passes)
{
    HIPCHK(hipSetDevice(s->device));
    if (s->timing_pending) {
        HIPCHK(hipEventSynchronize(s->ev1));
        if (fold_pending_timing(s)) return 1;
    }
    if (total_ms) *total_ms = s->timing_sweep_ms;
    if (passes) *passes = s->timing_count;
    return 0;
}

extern "C" int e2vq_timing_total(e2vq_session* s, double* total_ms, int64_t* passes)
{
    HIPCHK(hipSetDevice(s->device));
    if (s->timing_pending) {
        HIPCHK(hipEventSynchronize(s->ev1));
        if (fold_pending_timing(s)) return 1;
    }
    if (total_ms) *total_ms = s->timing_sum_ms;
    if (passes) *passes = s->timing_count;
    return 0;
}

extern "C" int e2vq_last_pass_kernel_ms(e2vq_session* s, float* ms)
{
    if (!s->timed) return e2vq_set_error("no timed pass recorded");
    HIPCHK(hipEventSynchronize(s->ev1));
    HIPCHK(hipEventElapsedTime(ms, s->ev0, s->ev1));
    return 0;
}

// Waits until the device has stored the current sequence number at *word (host-mapped memory; microseconds once the
// kernel runs).  No wall-clock limit: the wait also covers the sweep kernel queued ahead, which may legitimately take
// minutes (2^31 frames, generic prediction orders, ranks sharing a device).  What ends the wait without the number is the
// event recorded behind the kernel (a kernel that has finished without storing it: stream synchronisation, then an error)
// or a failed query; the publishing workgroup's own spin is bounded, so the kernel always ends, and the word it raises
// when a flag never arrived becomes an error here.
static int spin_for_sequence(e2vq_session* s, volatile u64* word, const char* what)
{
    // No wall-clock limit by default (the wait also covers the sweep queued ahead, which may legitimately take minutes);
    // ECOZ2_VQ_STATS_TIMEOUT_S sets one -- for hosts whose all-reduce hook can leave a collective pending for ever (a peer
    // process that died).  A rank of an in-process group also gives up as soon as the group has failed.
    static const double limit_s = getenv("ECOZ2_VQ_STATS_TIMEOUT_S") ? atof(getenv("ECOZ2_VQ_STATS_TIMEOUT_S")) : 0.0;
    const auto t_start = std::chrono::steady_clock::now();
    bool slow = false;  // after a few milliseconds: sleep between polls instead of burning a core
    for (unsigned long spins = 0; *word != s->stats_seq; ++spins) {
        if (slow) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 0xfff) == 0xfff || slow) {
            if (s->group_failed && *s->group_failed) return e2vq_set_error("%s: another rank of the in-process group failed", what);
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
            slow = waited > 5e-3;
            if (limit_s > 0.0 && waited > limit_s)
                return e2vq_set_error("%s: no statistics after %.1f s (ECOZ2_VQ_STATS_TIMEOUT_S)", what, waited);
            // (the safety net: the stream has drained and the number never came.  A stream query, not an event recorded
            // behind the kernel: the event's marker packet sat between the update and the next kernel of the stream and
            // cost ~5 us of idle GPU per pass)
            const hipError_t q = hipStreamQuery(s->stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) return e2vq_set_error("%s failed: %s", what, hipGetErrorString(q));
            (void)hipGetLastError();  // (hipErrorNotReady is sticky for hipGetLastError: nobody downstream should see it)
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    if (*word != s->stats_seq) {
        HIPCHK(hipStreamSynchronize(s->stream));
        if (*word != s->stats_seq)
            return e2vq_set_error("%s finished without publishing sequence %llu", what, (unsigned long long)s->stats_seq);
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (s->h_stats->err == s->stats_seq)
        return e2vq_set_error("%s: the publishing workgroup gave up waiting for a cell's flag (sequence %llu)", what,
                              (unsigned long long)s->stats_seq);
    return 0;
}

// ECOZ2_VQ_VERIFY_PUBLISH=1: everything the update kernel published through host-mapped memory -- level statistics,
// within-cell terms, the L1 maximum, the count of failed recursions -- is recomputed on the host from a copy of the
// accumulator rows (after a stream synchronisation) and compared bit for bit.  A lost or early publication would
// otherwise only show as a different convergence decision.
static int verify_published(e2vq_session* s, const i64 (&l)[8], double l1max)
{
    HIPCHK(hipStreamSynchronize(s->stream));
    const int NC = s->NC, RS = s->RS, M = s->M;
    std::vector<i64> rows((size_t)M * RS);
    std::vector<double> within((size_t)M);
    u64 l1bits = 0;
    HIPCHK(hipMemcpy(rows.data(), s->d_rows, rows.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(within.data(), s->d_within, (size_t)M * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&l1bits, s->d_l1max, 8, hipMemcpyDeviceToHost));
    i64 want[5] = {0, 0, 0, 0, 0}, failed = 0;
    std::vector<double> S((size_t)NC), rc((size_t)NC), a((size_t)NC);
    for (int m = 0; m < M; ++m) {
        const i64* row = rows.data() + (size_t)m * RS;
        for (int k = 0; k < 4; ++k) want[k] = (i64)((u64)want[k] + (u64)row[2 * NC + 1 + k]);
        const i64 cnt = row[2 * NC];
        double w = 0.0;
        if (cnt == 0) {
            want[4] += 1;
        } else {
            double ss = 0.0;
            for (int n = 0; n < NC; ++n) {
                S[(size_t)n] = e2vq::unfix(row[2 * n], row[2 * n + 1], s->h_sc.sh_r);
                ss += S[(size_t)n] * S[(size_t)n];
            }
            w = ss / (double)cnt;
            if (e2vq_io::lpca_r_host(s->P, S.data(), rc.data(), a.data()) != 0) ++failed;
        }
        u64 wb, hb, db;
        memcpy(&wb, &w, 8);
        memcpy(&hb, &s->h_within[m], 8);
        memcpy(&db, &within[(size_t)m], 8);
        if (wb != hb || wb != db)
            return e2vq_set_error("publish verification: within-cell term of cell %d: host %.17g, published %.17g, device %.17g "
                                  "(M = %d, sequence %llu)", m, w, s->h_within[m], within[(size_t)m], M, (unsigned long long)s->stats_seq);
    }
    for (int k = 0; k < 5; ++k)
        if (want[k] != l[k])
            return e2vq_set_error("publish verification: level statistic %d: rows give %lld, published %lld (M = %d, sequence %llu)",
                                  k, (long long)want[k], (long long)l[k], M, (unsigned long long)s->stats_seq);
    u64 pub_l1;
    memcpy(&pub_l1, &l1max, 8);
    if (pub_l1 != l1bits)
        return e2vq_set_error("publish verification: L1 maximum differs (M = %d, sequence %llu)", M, (unsigned long long)s->stats_seq);
    if (s->h_stats->seq2 != s->stats_seq || s->h_stats->failed != failed)
        return e2vq_set_error("publish verification: failed recursions: host %lld, published %lld (M = %d, sequence %llu / %llu)",
                              (long long)failed, (long long)s->h_stats->failed, M, (unsigned long long)s->h_stats->seq2,
                              (unsigned long long)s->stats_seq);
    s->verified_passes += 1;
    return 0;
}

// the count of failed recursions of the last fused update arrives at the kernel's end (PublishArgs::h_seq2)
static int resolve_failed_cells(e2vq_session* s)
{
    if (!s->failed_pending) return 0;
    if (spin_for_sequence(s, &s->h_stats->seq2, "update kernel")) return 1;
    s->last.failed_cells = s->h_stats->failed;
    if (s->failed_patch) s->failed_patch->failed_cells = s->h_stats->failed;
    s->failed_patch = nullptr;
    s->failed_pending = false;
    return 0;
}

// wait_failed = false (e2vq_learn): return as soon as the statistics the convergence rule needs are there;
// failed_cells of *out is then filled in by resolve_failed_cells later
static int pass_stats_impl(e2vq_session* s, e2vq_level_stats* out, bool wait_failed)
{
    HIPCHK(hipSetDevice(s->device));
    if (s->stats_valid) {
        if (wait_failed && resolve_failed_cells(s)) return 1;
        if (out) *out = s->last;
        return 0;
    }
    // (the distortion sums in the rows are fixed-point numbers scaled for the codebook the pass ran on: after an update
    // they cannot be read any more)
    if (!s->rows_fresh) return e2vq_set_error("no statistics: e2vq_pass has not run on the current codebook");
    if (resolve_failed_cells(s)) return 1;  // (of the pass before: long there)
    // (d_lstats is zero here -- zeroed at session start and by every publish kernel -- unless a separate centroid
    // kernel counted failed cells into it afterwards)
    if (s->lstats_dirty) HIPCHK(hipMemsetAsync(s->d_lstats, 0, 64 * 8 * 8, s->stream));
    s->lstats_dirty = false;
    const bool fused = e2vq::has_cell_update(s->NC);
    void *dl = nullptr, *dw = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dl, s->h_stats, 0));
    HIPCHK(hipHostGetDevicePointer(&dw, s->h_within, 0));
    auto* dstats = (e2vq_session::HostStats*)dl;
    if (fused) {
        // statistics + speculative update into the shadow codebook in ONE wave-per-cell kernel; its last workgroup
        // writes the statistics into host-mapped memory and then a sequence number.  The next pass will most likely
        // run on the shadow codebook: if that pass is going to be a prefiltered one, the kernel also finds the scale of
        // the shadow's limb image, and the image itself is built right behind it -- after the statistics went out,
        // i.e. during the host's round trip.
        const int k = 1 - s->img_cur;
        const bool image = s->d_cimg2[k] && use_prefilter(s, pass_mode(s)) && s->M <= s->cimg_cap;
        if (!s->spec_zeroed) {  // (no e2vq_pass in front: a repeated e2vq_pass_stats after an update)
            HIPCHK(hipMemsetAsync(s->d_l1max_spec, 0, sizeof(u64), s->stream));
            if (s->d_ps2[k]) HIPCHK(hipMemsetAsync(s->d_ps2[k], 0, e2vq::prefilter_scalars_bytes(), s->stream));
        }
        s->spec_zeroed = false;
        e2vq::PublishArgs pub{};
        pub.flags = (unsigned int*)(s->d_within + s->M_cap);
        pub.l1max_cur = s->d_l1max;
        pub.h_l = dstats->l;
        pub.h_l1 = &dstats->l1bits;
        pub.h_within = (double*)dw;
        pub.h_seq = (volatile u64*)&dstats->seq;
        pub.h_failed = &dstats->failed;
        pub.h_seq2 = (volatile u64*)&dstats->seq2;
        pub.h_err = (volatile u64*)&dstats->err;
        pub.seq = ++s->stats_seq;
        e2vq::launch_cell_update(s->d_rows, s->M, s->NC, s->d_sc, s->d_refl, s->d_refl_spec, s->d_cbq_spec,
                                 s->d_cbm_spec, s->d_l1max_spec, s->d_within, s->d_lstats, s->stream,
                                 /*zero_first=*/false, image ? s->d_ea : nullptr,
                                 image ? e2vq::prefilter_codebook_scale(s->d_ps2[k]) : nullptr, &pub);
        s->failed_pending = true;
        if (image) {
            e2vq::launch_prefilter_codebook(s->d_cbq_spec, s->M, s->NC, s->d_ea, s->d_ps2[k], s->d_cimg2[k], s->stream,
                                            /*scale_ready=*/true);
            s->img_valid[k] = true;
        }
    } else {
        // thread-per-cell path (P > 63): statistics, a one-block publish kernel, then the speculative centroid update
        // (keeps the GPU busy while the host decides)
        e2vq::launch_rows_stats(s->d_rows, s->M, s->NC, s->d_sc, s->d_S, s->d_within, s->d_lstats, s->stream);
        e2vq::launch_publish_stats(s->d_lstats, s->d_l1max, s->d_within, s->M, dstats->l, &dstats->l1bits, (double*)dw,
                                   (u64*)&dstats->seq, ++s->stats_seq, s->stream);
        e2vq::launch_centroids(s->d_rows, s->d_S, s->M, s->NC, s->d_refl, s->d_refl_spec, s->d_lstats, s->stream);
        s->lstats_dirty = true;
        e2vq::launch_codebook_prepare(s->d_refl_spec, s->M, s->NC, s->d_cbq_spec, s->d_l1max_spec, s->d_cbm_spec,
                                      s->stream);
    }
    HIPCHK(hipGetLastError());
    s->spec_valid = true;
    // spin on the sequence number (microseconds); the event is the safety net should the kernel never get there
    if (spin_for_sequence(s, &s->h_stats->seq, "statistics kernel")) return 1;
    if (s->rec_pending) {  // (stored by the reduce kernel, which ran ahead of the statistics kernel on the same queue)
        s->rec_last_total = s->h_stats->rec_total;
        s->rec_pending = false;
    }
    if (s->sw_pending) {  // (stored by the finishing kernel of a two-stage sweep, likewise ahead on the queue)
        const u64 fl = s->h_stats->sw_flagged, jobs = s->h_stats->sw_jobs;
        s->sw_pending = false;
        s->last_flagged_frac = jobs ? (double)fl / (double)jobs : -1.0;
        // most tiles flagged: the coarse stage is wasted on this data -- one stage for the rest of this level (the next level's
        // first pass tries again: finer cells, more tiles)
        if (jobs && s->last_flagged_frac > s->two_stage_max_frac) s->two_stage_off_until_M = s->M;
    }
    i64 l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int slot = 0; slot < 64; ++slot)
        for (int k = 0; k < 8; ++k) l[k] += s->h_stats->l[slot * 8 + k];
    double l1max;
    memcpy(&l1max, &s->h_stats->l1bits, 8);
    const int Ed = e2vq::dist_exponent(s->h_sc.maxabs, l1max);
    const double DD = e2vq::unfix(l[0], l[1], 30 - Ed);
    const double SS = e2vq::unfix(l[2], l[3], 30 - 2 * Ed);
    const double T = (double)s->T_total;
    const double avg = DD / T;
    const double q = SS / T;
    const double p = avg * avg;
    double v = q - p;
    if (!(v > 0.0)) v = 0.0;
    double w = 0.0;
    for (int m = 0; m < s->M; ++m) w += s->h_within[m];  // empty cells contribute +0.0
    s->last.M = s->M;
    s->last.DD = DD;
    s->last.avg_distortion = avg;
    s->last.sigma = sqrt(v);
    s->last.inertia = s->h_sc.Q - w;
    s->last.empty_cells = l[4];
    s->last.failed_cells = 0;
    if (fused && s->verify_publish && verify_published(s, l, l1max)) return 1;
    if (fused) {
        if (wait_failed && resolve_failed_cells(s)) return 1;
    } else {
        // thread-per-cell path (P > 63): k_centroids counted the failed recursions after the slots were published
        i64 slots[64 * 8];
        HIPCHK(hipMemcpyAsync(slots, s->d_lstats, sizeof slots, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        i64 f = 0;
        for (int slot = 0; slot < 64; ++slot) f += slots[slot * 8 + 5];
        s->last.failed_cells = f;
    }
    s->stats_valid = true;
    if (out) *out = s->last;
    return 0;
}

extern "C" int e2vq_pass_stats(e2vq_session* s, e2vq_level_stats* out) { return pass_stats_impl(s, out, true); }

// passes whose published statistics were verified against a host recomputation (ECOZ2_VQ_VERIFY_PUBLISH=1)
extern "C" int e2vq_verified_passes(e2vq_session* s, int64_t* passes)
{
    if (passes) *passes = s->verified_passes;
    return 0;
}

extern "C" int e2vq_update(e2vq_session* s)
{
    if (!s->stats_valid) {
        if (e2vq_pass_stats(s, nullptr)) return 1;
    }
    HIPCHK(hipSetDevice(s->device));
    s->cb_version++;
    s->rows_fresh = false;
    if (s->spec_valid) {  // commit the speculative update: no launch, just swap the codebook sets
        std::swap(s->d_refl, s->d_refl_spec);
        std::swap(s->d_cbq, s->d_cbq_spec);
        std::swap(s->d_cbm, s->d_cbm_spec);
        std::swap(s->d_l1max, s->d_l1max_spec);
        s->img_valid[s->img_cur] = false;  // (that codebook is the shadow now)
        s->img_cur = 1 - s->img_cur;
        s->spec_valid = false;
        s->stats_valid = false;
        return 0;
    }
    e2vq::launch_centroids(s->d_rows, s->d_S, s->M, s->NC, s->d_refl, s->d_refl, s->d_lstats, s->stream);
    s->lstats_dirty = true;
    HIPCHK(hipGetLastError());
    return codebook_prepare(s, false);
}

extern "C" int e2vq_iterate(e2vq_session* s, void* device_sym, void* device_dmin, e2vq_level_stats* out)
{
    if (e2vq_pass(s, device_sym, device_dmin)) return 1;
    if (e2vq_pass_stats(s, out)) return 1;
    return e2vq_update(s);
}

// One saved point of the ladder: the codebook, DDprv, and -- what the seeded first pass of the next size starts from --
// the accumulator rows and every frame's cell of the last pass.  e2vq_restore_state puts the session back there (device
// copies, microseconds), so that a caller can repeat a level exactly as the uninterrupted ladder runs it: bench.py times
// the M = 1024 level this way.  Call e2vq_save_state right after the pass (and statistics) that ended a level.
extern "C" int e2vq_save_state(e2vq_session* s)
{
    if (s->M < 1) return e2vq_set_error("no codebook to save");
    HIPCHK(hipSetDevice(s->device));
    auto& v = s->sv;
    v.valid = false;
    if (v.cap_M < s->M) {
        for (void** p : {(void**)&v.refl, (void**)&v.rows, (void**)&v.rows_local}) {
            if (*p) HIPCHK(hipFree(*p));
            *p = nullptr;
        }
        v.cap_M = s->M;
        HIPCHK(hipMalloc(&v.refl, (size_t)v.cap_M * s->NC * 8));
        HIPCHK(hipMalloc(&v.rows, (size_t)v.cap_M * s->RS * 8));
        HIPCHK(hipMalloc(&v.rows_local, (size_t)v.cap_M * s->RS * 8));
    }
    if (s->d_prev_sym && v.cap_T < s->nblocks * 64) {
        if (v.cells) HIPCHK(hipFree(v.cells));
        v.cells = nullptr;
        v.cap_T = s->nblocks * 64;
        HIPCHK(hipMalloc(&v.cells, (size_t)v.cap_T * 2 + 256));
    }
    HIPCHK(hipMemcpyAsync(v.refl, s->d_refl, (size_t)s->M * s->NC * 8, hipMemcpyDeviceToDevice, s->stream));
    HIPCHK(hipMemcpyAsync(v.rows, s->d_rows, (size_t)s->M * s->RS * 8, hipMemcpyDeviceToDevice, s->stream));
    if (s->d_rows_local && s->rows_local_cap >= s->M)
        HIPCHK(hipMemcpyAsync(v.rows_local, s->d_rows_local, (size_t)s->M * s->RS * 8, hipMemcpyDeviceToDevice, s->stream));
    if (s->d_prev_sym) HIPCHK(hipMemcpyAsync(v.cells, s->d_prev_sym, (size_t)s->nblocks * 64 * 2, hipMemcpyDeviceToDevice, s->stream));
    v.M = s->M;
    v.nblocks = s->nblocks;
    v.cells_M = s->cells_M;
    v.incr_M = s->incr_M;
    v.DDprv = s->DDprv;
    v.rows_fresh = s->rows_fresh;
    v.rows_are_local = s->rows_are_local;
    v.rows_local_is_current = s->rows_local_is_current;
    v.incr_valid = s->incr_valid;
    v.valid = true;
    return 0;
}

extern "C" int e2vq_restore_state(e2vq_session* s)
{
    auto& v = s->sv;
    if (!v.valid) return e2vq_set_error("no saved state");
    if (v.nblocks != s->nblocks) return e2vq_set_error("the saved state belongs to another training set");
    HIPCHK(hipSetDevice(s->device));
    if (ensure_codebook_capacity(s, v.M)) return 1;
    HIPCHK(hipMemcpyAsync(s->d_refl, v.refl, (size_t)v.M * s->NC * 8, hipMemcpyDeviceToDevice, s->stream));
    s->M = v.M;
    if (codebook_prepare(s)) return 1;  // (codeword images of the restored codebook; drops every derived flag)
    HIPCHK(hipMemcpyAsync(s->d_rows, v.rows, (size_t)v.M * s->RS * 8, hipMemcpyDeviceToDevice, s->stream));
    if (s->d_rows_local && s->rows_local_cap >= v.M)
        HIPCHK(hipMemcpyAsync(s->d_rows_local, v.rows_local, (size_t)v.M * s->RS * 8, hipMemcpyDeviceToDevice, s->stream));
    if (s->d_prev_sym && v.cells)
        HIPCHK(hipMemcpyAsync(s->d_prev_sym, v.cells, (size_t)s->nblocks * 64 * 2, hipMemcpyDeviceToDevice, s->stream));
    s->DDprv = v.DDprv;
    s->cells_M = v.cells_M;
    s->incr_M = v.incr_M;
    s->incr_valid = v.incr_valid;
    s->rows_fresh = v.rows_fresh;
    s->rows_are_local = v.rows_are_local;
    s->rows_local_is_current = v.rows_local_is_current;
    return 0;
}

// DDprv of the convergence rule (notes.md:128-153: it carries over between codebook sizes).  A caller that restores
// an earlier codebook with e2vq_set_codebook restores the matching DD with this, so that e2vq_learn repeats the
// level exactly as the uninterrupted ladder ran it (bench.py times the real M = 1024 level this way).
extern "C" int e2vq_set_prev_distortion(e2vq_session* s, double DDprv)
{
    s->DDprv = DDprv;
    return 0;
}

extern "C" int e2vq_get_prev_distortion(e2vq_session* s, double* DDprv)
{
    *DDprv = s->DDprv;
    return 0;
}

extern "C" int e2vq_row_stride(int prediction_order) { return e2vq::row_stride(prediction_order + 1); }

extern "C" int e2vq_get_rows(e2vq_session* s, int64_t* rows)
{
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipMemcpyAsync(rows, s->d_rows, (size_t)s->M * s->RS * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
}

// ---- LBG ladder ------------------------------------------------------------------------------

extern "C" int e2vq_learn(e2vq_session* s, double epsilon, int max_M, const char* class_name, const char* out_root,
                          void* target, ecoz2_vq_learn_callback_t callback, e2vq_level_stats* levels, int max_levels,
                          int* num_levels)
{
    if (num_levels) *num_levels = 0;
    if (!s->prepared) return e2vq_set_error("e2vq_prepare has not run");
    if (s->M < 1) return e2vq_set_error("no codebook: call e2vq_init_codebook or e2vq_set_codebook");
    const bool verbose = getenv("ECOZ2_VQ_QUIET") == nullptr && s->rank == 0;
    const bool write_files = out_root != nullptr && s->rank == 0;
    struct FileCloser {  // the report is closed on every return path
        FILE* f = nullptr;
        ~FileCloser() { if (f) fclose(f); }
    } rpt_guard;
    struct PatchGuard {  // no pointer into the caller's level records outlives the call
        e2vq_session* s;
        ~PatchGuard() { s->failed_patch = nullptr; }
    } patch_guard{s};
    FILE*& rpt = rpt_guard.f;
    char path[4096];
    if (write_files) {
        snprintf(path, sizeof path, "%s/data/codebooks/%s/eps_%g.rpt", out_root, class_name, epsilon);
        if (e2vq_io::mkdirs_for(path) == 0) rpt = fopen(path, "w");
        if (rpt)
            fprintf(rpt,
                    "# %lld training vectors, P=%d, eps=%g\n# M passes DD avg_distortion sigma inertia empty_cells\n",
                    (long long)s->T_total, s->P, epsilon);
        if (verbose) printf("Report: %s\n", path);
    }
    std::vector<double> refl;
    int nlev = 0;
    while (s->M < max_M) {
        if (e2vq_grow(s)) return 1;
        if (write_files) {
            snprintf(path, sizeof path, "%s/data/codebooks/%s/eps_%g_M_%04d.cbook", out_root, class_name, epsilon,
                     s->M);
            if (verbose) printf("%s\n", path);
        }
        e2vq_level_stats ls{};
        int pass = 0;
        for (;; ++pass) {
            if (e2vq_pass(s, nullptr, nullptr)) return 1;
            if (pass_stats_impl(s, &ls, /*wait_failed=*/false)) return 1;
            const double DD = ls.DD;
            const double ratio = (s->DDprv - DD) / DD;
            if (verbose) {
                printf("(%d)\tDP=%g\tDDprv=%g\tDD=%g\t%g\n", pass, ls.avg_distortion, s->DDprv, DD, ratio);
                if (ls.empty_cells > 0)
                    printf("WARN: review_cells: %lld empty cell(s) for codebook size %d)\n", (long long)ls.empty_cells,
                           s->M);
            }
            // pass 0 never ends a level; DDprv carries over between levels (notes.md:128-153)
            // (a level also ends after E2VQ_MAX_PASSES passes: eps <= 0 would otherwise never terminate)
            const bool converged = (pass > 0 && !(ratio >= epsilon)) || pass + 1 >= E2VQ_MAX_PASSES;
            s->DDprv = DD;
            if (converged) break;
            if (e2vq_update(s)) return 1;
        }
        ls.passes = pass + 1;
        if (write_files) {
            refl.resize((size_t)s->M * s->NC);
            if (e2vq_get_codebook(s, refl.data(), nullptr)) return 1;
            if (e2vq_cbook_write(path, class_name, s->P, s->M, refl.data())) return 1;
            if (rpt)
                fprintf(rpt, "%d %d %.17g %.17g %.17g %.17g %lld\n", s->M, ls.passes, ls.DD, ls.avg_distortion,
                        ls.sigma, ls.inertia, (long long)ls.empty_cells);
        }
        if (levels && nlev < max_levels) {
            levels[nlev] = ls;
            if (s->failed_pending) s->failed_patch = &levels[nlev];  // (filled in before the next statistics / on return)
        }
        ++nlev;
        if (callback && s->rank == 0) callback(target, s->M, ls.avg_distortion, ls.sigma, ls.inertia);
    }
    if (resolve_failed_cells(s)) return 1;
    if (num_levels) *num_levels = nlev;
    return 0;
}

// ---- quantize --------------------------------------------------------------------------------

// staging buffers of e2vq_quantize_host (row-major frames in, symbols / distortions out)
static int ensure_quantize_scratch(e2vq_session* s, i64 T, bool need_aos, bool need_out)
{
    HIPCHK(hipSetDevice(s->device));
    if (T > s->q_cap) {
        if (s->d_qaos) { (void)hipFree(s->d_qaos); s->d_qaos = nullptr; }
        if (s->d_qsym) { (void)hipFree(s->d_qsym); s->d_qsym = nullptr; }
        if (s->d_qdmin) { (void)hipFree(s->d_qdmin); s->d_qdmin = nullptr; }
        s->q_cap = T;
    }
    if (need_aos && !s->d_qaos) HIPCHK(hipMalloc(&s->d_qaos, (size_t)s->q_cap * s->NC * 8));
    if (need_out && !s->d_qsym) {
        HIPCHK(hipMalloc(&s->d_qsym, (size_t)s->q_cap * 2 + 64));
        HIPCHK(hipMalloc(&s->d_qdmin, (size_t)s->q_cap * 8));
    }
    return 0;
}

// re-layout buffer for the kernels that do not read the row-major payload directly (P != 36)
static int ensure_qblk(e2vq_session* s, i64 T)
{
    if (T <= s->qblk_cap && s->d_qblk) return 0;
    HIPCHK(hipSetDevice(s->device));
    if (s->d_qblk) (void)hipFree(s->d_qblk);
    s->d_qblk = nullptr;
    const i64 nb = (T + s->FB - 1) / s->FB;
    HIPCHK(hipMalloc(&s->d_qblk, (size_t)nb * s->NC * s->FB * 8));
    s->qblk_cap = T;
    return 0;
}

extern "C" int e2vq_quantize_device(e2vq_session* s, const void* device_frames, int64_t T, void* device_sym,
                                    void* device_dmin)
{
    if (s->M < 1) return e2vq_set_error("no codebook");
    if (T < 1) return 0;
    if (T > (int64_t)INT32_MAX - 64) return e2vq_set_error("%lld frames per quantize call exceed 2^31 - 65 (split the call)", (long long)T);
    HIPCHK(hipSetDevice(s->device));
    const i64 nb = (T + s->FB - 1) / s->FB;
    if (s->pre_enabled && s->M >= s->pre_min_M_quant && e2vq::prefilter_supports(s->NC, s->M)) {
        // prefiltered sweep.  Fused (every prefiltered order): the assignment-only kernel builds the f16 limb images of its frames from the
        // row-major payload itself and keeps the FP64 frames in LDS for the exact evaluation -- every frame is read once.
        // Otherwise one preparation pass over the payload writes the limb image and the tolerance terms first.  Either
        // way the FP64 sweep of whatever could not be certified reads the payload too (no blocked copy).
        const bool fused = e2vq::prefilter_fused_quantize(s->NC);
        if (s->qfb_cap < nb) {
            if (s->d_qfblist) HIPCHK(hipFree(s->d_qfblist));
            s->d_qfblist = nullptr;
            HIPCHK(hipMalloc(&s->d_qfblist, (size_t)nb * 64 * sizeof(int)));
            s->qfb_cap = nb;
        }
        if (!fused && s->qpre_cap < nb) {
            for (void* p : {(void*)s->d_qfimg, (void*)s->d_qfg})
                if (p) HIPCHK(hipFree(p));
            s->d_qfimg = nullptr;
            s->d_qfg = nullptr;
            HIPCHK(hipMalloc(&s->d_qfimg, e2vq::prefilter_frame_image_bytes(s->NC, nb)));
            HIPCHK(hipMalloc(&s->d_qfg, (size_t)nb * 64 * sizeof(float)));
            s->qpre_cap = nb;
        }
        if (!s->d_ea_q) HIPCHK(hipMalloc(&s->d_ea_q, (size_t)s->NC * sizeof(int)));
        if (s->qcimg_cap < s->M) {
            if (s->d_qcimg) HIPCHK(hipFree(s->d_qcimg));
            s->d_qcimg = nullptr;
            s->qcimg_cap = std::max(s->M, 2048);
            HIPCHK(hipMalloc(&s->d_qcimg, e2vq::prefilter_codebook_image_bytes(s->NC, s->qcimg_cap)));
            s->qimg_version = 0;
        }
        const double* aos = (const double*)device_frames;
        if (s->qimg_version != s->cb_version) {
            // scales and limb image of the codebook: once per codebook, not per call (a corpus is many short files)
            e2vq::launch_prefilter_quantize_scales(s->d_cbq, s->M, s->NC, s->d_ea_q, s->stream);
            e2vq::launch_prefilter_codebook(s->d_cbq, s->M, s->NC, s->d_ea_q, s->d_ps, s->d_qcimg, s->stream);
            s->qimg_version = s->cb_version;
        } else {
            HIPCHK(hipMemsetAsync((void*)e2vq::prefilter_fallback_count(s->d_ps), 0, sizeof(int), s->stream));
        }
        if (fused) {
            e2vq::launch_pass_prefiltered(s->NC, false, nullptr, T, nb, nullptr, nullptr, s->d_qcimg, s->d_ps, s->d_cbq,
                                          s->M, s->d_sc, s->d_l1max, (unsigned short*)device_sym, (double*)device_dmin,
                                          nullptr, s->d_qfblist, nullptr, false, s->stream, aos, s->d_ea_q);
        } else {
            e2vq::launch_prefilter_quantize_prep(aos, T, nb, s->NC, s->d_ea_q, nullptr, s->d_qfimg, s->d_qfg, s->stream);
            e2vq::launch_pass_prefiltered(s->NC, false, nullptr, T, nb, s->d_qfimg, s->d_qfg, s->d_qcimg, s->d_ps, s->d_cbq,
                                          s->M, s->d_sc, s->d_l1max, (unsigned short*)device_sym, (double*)device_dmin,
                                          nullptr, s->d_qfblist, nullptr, false, s->stream, aos);
        }
        e2vq::launch_pass_fallback(s->NC, false, aos, s->d_cbm, s->M, s->d_sc, s->d_l1max, (unsigned short*)device_sym,
                                   (double*)device_dmin, nullptr, s->d_qfblist, e2vq::prefilter_fallback_count(s->d_ps),
                                   nullptr, false, s->stream, /*rowmajor=*/true);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (e2vq::uses_mfma(s->NC) && !e2vq::mfma_is_wide(s->NC) && ((uintptr_t)device_frames & 15) == 0) {
        // P = 36: the sweep reads the row-major payload directly (coalesced staging through LDS)
        e2vq::launch_pass(s->NC, 4, (const double*)device_frames, T, nb, s->d_cbq, s->d_cbm, s->M, s->d_sc, s->d_l1max,
                          (unsigned short*)device_sym, (double*)device_dmin, nullptr, s->stream);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (ensure_qblk(s, T)) return 1;
    e2vq::launch_blockify((const double*)device_frames, T, s->NC, s->FB, s->d_qblk, nb, nullptr, nullptr, s->stream);
    e2vq::launch_pass(s->NC, 0, s->d_qblk, T, nb, s->d_cbq, s->d_cbm ? s->d_cbm : s->d_cbT, s->M, s->d_sc, s->d_l1max, (unsigned short*)device_sym,
                      (double*)device_dmin, nullptr, s->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int e2vq_quantize_host(e2vq_session* s, const double* frames, int64_t T, uint16_t* sym, double* dmin)
{
    if (s->M < 1) return e2vq_set_error("no codebook");
    HIPCHK(hipSetDevice(s->device));
    const i64 CH = 1 << 22;  // frames per chunk (1.2 GB of predictor vectors)
    for (i64 t0 = 0; t0 < T; t0 += CH) {
        const i64 n = std::min<i64>(CH, T - t0);
        if (ensure_quantize_scratch(s, n, true, true)) return 1;
        HIPCHK(hipMemcpyAsync(s->d_qaos, frames + (size_t)t0 * s->NC, (size_t)n * s->NC * 8, hipMemcpyHostToDevice,
                              s->stream));
        if (e2vq_quantize_device(s, s->d_qaos, n, s->d_qsym, dmin ? s->d_qdmin : nullptr)) return 1;
        HIPCHK(hipMemcpyAsync(sym + t0, s->d_qsym, (size_t)n * 2, hipMemcpyDeviceToHost, s->stream));
        if (dmin) HIPCHK(hipMemcpyAsync(dmin + t0, s->d_qdmin, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return 0;
}

extern "C" int e2vq_avg_distortion_host(e2vq_session* s, const double* frames, int64_t T, double* avg)
{
    if (T < 1) return e2vq_set_error("no frames");
    std::vector<uint16_t> sym((size_t)T);
    std::vector<double> dmin((size_t)T);
    if (e2vq_quantize_host(s, frames, T, sym.data(), dmin.data())) return 1;
    double e = 0.0;
    for (int64_t t = 0; t < T; ++t) e += dmin[(size_t)t] - 1.0;
    *avg = e / (double)T;
    return 0;
}


// ==========================================================================================
// In-process group: N sessions (one per GPU, one host thread each) behind the single-process entry points.
// The per-pass exchange of the int64 cell sums is a reduce-scatter + all-gather over peer-to-peer memory: rank r owns
// slice r of the buffer; ONE kernel per rank, all running at the same time, reads that slice from every rank's buffer
// (xGMI between the GPUs of a node), adds, and writes the sum back into every buffer.  Ordering is carried by events
// (producers done -> slice kernels -> consumers); the two host barriers only make sure an event has been recorded
// before another rank's stream is told to wait for it.  Integer sums: bit-identical for any N.
// Opt-in: ECOZ2_VQ_GPUS=N (ranks beyond the device count share devices, which is how the single-GPU tests run it).
// ==========================================================================================
namespace {

struct LocalGroup {
    int n = 1;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    volatile bool failed = false;
    // exchange state: every rank publishes its buffer and records its events, then waits on the others'
    e2vq::PeerBuffers bufs{};
    std::vector<hipEvent_t> ev_ready, ev_done;

    // Reusable barrier; returns false if the group has failed.  The ranks of a group run in lock step -- every collective is
    // a rendezvous of host threads that arrive within microseconds of each other -- so a rank first SPINS on the generation
    // counter (round 5: a condition-variable wake-up cost each of the two rendezvous of an exchange 20-50 us, most of what
    // the exchange took at the small levels) and only blocks when the others are far behind (~50 us).
    std::atomic<long> gen_spin{0};
    bool barrier()
    {
        long gen;
        {
            std::unique_lock<std::mutex> lk(mu);
            if (failed) return false;
            gen = generation;
            if (++arrived == n) {
                arrived = 0;
                ++generation;
                gen_spin.store(generation, std::memory_order_release);
                cv.notify_all();
                return true;
            }
        }
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            if (gen_spin.load(std::memory_order_acquire) != gen) return !failed;
            if (failed) return false;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            if ((spins & 0x3ff) == 0x3ff && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(50)) break;
        }
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return generation != gen || failed; });
        return !failed;
    }
    // first failing rank's message (g_err is thread-local: the workers' text would be lost with their threads)
    std::string first_error;
    // run once, by the first rank that fails: with RCCL it aborts every communicator of the group, so that a collective
    // some ranks have already enqueued -- and that the failed rank will never join -- ends instead of hanging their streams
    void (*on_fail)(void*) = nullptr;
    void* on_fail_arg = nullptr;
    void fail()
    {
        bool first = false;
        {
            std::lock_guard<std::mutex> lk(mu);
            first = !failed;
            if (first) first_error = g_err;
            failed = true;
            cv.notify_all();
        }
        if (first && on_fail) on_fail(on_fail_arg);
    }
};

struct LocalRank {
    LocalGroup* g;
    int rank;
    int device;
};

#define GRPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            e2vq_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            g->fail();                                                                            \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

int local_allreduce(void* user, void* buf, int64_t count, int op, void* stream_)
{
    LocalRank* lr = (LocalRank*)user;
    LocalGroup* g = lr->g;
    hipStream_t stream = (hipStream_t)stream_;
    const int r = lr->rank, n = g->n;
    GRPCHK(hipSetDevice(lr->device));
    g->bufs.p[r] = (long long*)buf;
    GRPCHK(hipEventRecord(g->ev_ready[r], stream));  // this rank's words are final once the stream gets here
    if (!g->barrier()) return e2vq_set_error("in-process group: another rank failed");  // A: buffers + ready events published
    for (int k = 0; k < n; ++k)
        if (k != r) GRPCHK(hipStreamWaitEvent(stream, g->ev_ready[k], 0));
    const long lo = (long)((int64_t)r * count / n), hi = (long)((int64_t)(r + 1) * count / n);
    e2vq::launch_reduce_slice_i64(g->bufs, n, lo, hi, op, stream);
    GRPCHK(hipGetLastError());
    GRPCHK(hipEventRecord(g->ev_done[r], stream));
    if (!g->barrier()) return e2vq_set_error("in-process group: another rank failed");  // B: every slice kernel is enqueued
    // nobody touches its buffer again (reads the sums, zeroes the rows) before every slice has been written everywhere
    for (int k = 0; k < n; ++k)
        if (k != r) GRPCHK(hipStreamWaitEvent(stream, g->ev_done[k], 0));
    return 0;
}

// ---- RCCL inside the library (north_star: "an RCCL all-reduce over xGMI of the per-cluster sums each LBG iteration") ----
// librccl.so is loaded on first use (dlopen: the library itself keeps linking against the HIP runtime only, and a
// process that never shards never loads RCCL).  One communicator per in-process rank (ncclCommInitAll over the ranks'
// devices, which must be distinct); every rank's host thread enqueues ncclAllReduce(buf, buf, count, ncclInt64 /
// ncclUint64, ncclSum / ncclMax) on its session's stream -- in place, exact integers, so any rank count gives the same bits.
struct Rccl {
    typedef int (*get_version_t)(int*);
    typedef int (*comm_init_all_t)(void**, int, const int*);
    typedef int (*all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
    typedef int (*comm_destroy_t)(void*);
    typedef int (*comm_abort_t)(void*);
    typedef const char* (*error_string_t)(int);
    void* handle = nullptr;
    get_version_t get_version = nullptr;
    comm_init_all_t comm_init_all = nullptr;
    all_reduce_t all_reduce = nullptr;
    comm_destroy_t comm_destroy = nullptr;
    comm_abort_t comm_abort = nullptr;  // (optional)
    error_string_t error_string = nullptr;
    std::string why;  // why it could not be loaded
    enum { Int64 = 4, Uint64 = 5, Sum = 0, Max = 2 };  // ncclDataType_t / ncclRedOp_t values of rccl.h (stable ABI)
};

Rccl* rccl_api()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // an RCCL that the process has loaded already (a host application's, PyTorch's) is the one to use: a second copy
        // of the library beside it fails to initialise ("unhandled cuda error")
        const char* names[] = {getenv("ECOZ2_VQ_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
        for (int pass = 0; pass < 2 && !r.handle; ++pass)
            for (const char* n : names) {
                if (!n || !*n) continue;
                r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (r.handle) break;
                if (pass == 1) r.why = dlerror();
            }
        if (!r.handle) return;
        r.get_version = (Rccl::get_version_t)dlsym(r.handle, "ncclGetVersion");
        r.comm_init_all = (Rccl::comm_init_all_t)dlsym(r.handle, "ncclCommInitAll");
        r.all_reduce = (Rccl::all_reduce_t)dlsym(r.handle, "ncclAllReduce");
        r.comm_destroy = (Rccl::comm_destroy_t)dlsym(r.handle, "ncclCommDestroy");
        r.comm_abort = (Rccl::comm_abort_t)dlsym(r.handle, "ncclCommAbort");
        r.error_string = (Rccl::error_string_t)dlsym(r.handle, "ncclGetErrorString");
        if (!r.comm_init_all || !r.all_reduce || !r.comm_destroy) {
            r.why = "librccl.so lacks ncclCommInitAll / ncclAllReduce / ncclCommDestroy";
            dlclose(r.handle);
            r.handle = nullptr;
        }
    });
    return r.handle ? &r : nullptr;
}

struct RcclComms;
struct RcclRank {
    LocalGroup* g;
    RcclComms* comms;
    int rank, device;
    long calls = 0, bytes = 0;
};

// the communicators of an in-process group; abort() is the group's on_fail hook.  A rank enqueues its collective under the
// shared lock and takes its communicator from here, not from a cached pointer: abort_all (exclusive) cannot free a
// communicator another rank's thread is about to hand to ncclAllReduce.
struct RcclComms {
    std::vector<void*> comms;
    std::shared_mutex mu;
    bool aborted = false;
    static void abort_all(void* self_)
    {
        RcclComms* self = (RcclComms*)self_;
        Rccl* api = rccl_api();
        std::unique_lock<std::shared_mutex> lk(self->mu);
        if (self->aborted || !api || !api->comm_abort) return;
        self->aborted = true;  // (ncclCommAbort releases the communicator: no ncclCommDestroy afterwards)
        for (void*& c : self->comms)
            if (c) {
                (void)api->comm_abort(c);
                c = nullptr;
            }
    }
};

int rccl_allreduce(void* user, void* buf, int64_t count, int op, void* stream_)
{
    RcclRank* rr = (RcclRank*)user;
    Rccl* api = rccl_api();
    if (!api) return e2vq_set_error("RCCL is not loaded");
    if (hipSetDevice(rr->device) != hipSuccess) {
        e2vq_set_error("hipSetDevice(%d) failed", rr->device);
        rr->g->fail();
        return 1;
    }
    // Host rendezvous: every rank of the group is alive and about to enqueue this collective.  ncclAllReduce itself only
    // enqueues; without the rendezvous a rank that failed earlier (a read error, bad data in its shard, no memory) would
    // leave the others with a collective that never completes -- blocked in the next stream synchronisation for good.
    if (!rr->g->barrier()) return e2vq_set_error("in-process group: another rank failed");
    (void)hipGetLastError();  // (see ncclCommInitAll below: hipErrorNotReady of a polled event must not reach RCCL)
    int rc;
    {
        std::shared_lock<std::shared_mutex> lk(rr->comms->mu);
        void* comm = rr->comms->aborted ? nullptr : rr->comms->comms[(size_t)rr->rank];
        if (!comm) return e2vq_set_error("in-process group: another rank failed (communicators aborted)");
        rc = api->all_reduce(buf, buf, (size_t)count, op == 0 ? Rccl::Int64 : Rccl::Uint64, op == 0 ? Rccl::Sum : Rccl::Max, comm,
                             (hipStream_t)stream_);
    }
    if (rc != 0) {
        e2vq_set_error("ncclAllReduce failed: %s", api->error_string ? api->error_string(rc) : "?");
        rr->g->fail();
        return 1;
    }
    rr->calls += 1;
    rr->bytes += (long)count * 8;
    return 0;
}


// ---- the in-process group as an object (round 4): what ecoz2_vq_learn builds for ECOZ2_VQ_GPUS > 1, exported so that a
// host -- bench.py --in-process -- can drive one session per rank from its own threads and time the library's OWN
// exchange (ncclAllReduce inside the library, or the peer-to-peer slice kernel), not a caller-supplied hook ----------------
struct GroupImpl {
    LocalGroup g;
    int world = 0;
    bool use_rccl = false;
    std::vector<int> devs;
    std::vector<LocalRank> ranks;
    RcclComms rc_comms;
    std::vector<RcclRank> rranks;
    std::string what;  // one line describing the exchange
    ~GroupImpl()
    {
        for (hipEvent_t ev : g.ev_ready)
            if (ev) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : g.ev_done)
            if (ev) (void)hipEventDestroy(ev);
        if (g.failed) RcclComms::abort_all(&rc_comms);  // (a failed group may hold a collective that cannot complete)
        std::unique_lock<std::shared_mutex> lk(rc_comms.mu);
        Rccl* api = rc_comms.comms.empty() ? nullptr : rccl_api();
        // (a failed group on an RCCL without ncclCommAbort: ncclCommDestroy could block on that collective for good --
        // the communicators are leaked instead)
        if (api && !(g.failed && !api->comm_abort))
            for (void* c : rc_comms.comms)
                if (c) (void)api->comm_destroy(c);
    }
};

// more than one HIP runtime mapped into the process (a host application's bundled ROCm beside /opt/rocm's)?  An RCCL
// initialised in that mix reports "no device" / "unhandled cuda error": say so instead of leaving the user with that.
std::string hip_runtime_copies()
{
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return "";
    std::vector<std::string> seen;
    char line[4096];
    while (fgets(line, sizeof line, f)) {
        const char* p = strstr(line, "libamdhip64");
        if (!p) continue;
        const char* path = strchr(line, '/');
        if (!path) continue;
        std::string sp(path);
        while (!sp.empty() && (sp.back() == '\n' || sp.back() == ' ')) sp.pop_back();
        if (std::find(seen.begin(), seen.end(), sp) == seen.end()) seen.push_back(sp);
    }
    fclose(f);
    if (seen.size() < 2) return "";
    std::string out = "; " + std::to_string(seen.size()) + " copies of the HIP runtime are mapped into this process (";
    for (size_t i = 0; i < seen.size(); ++i) out += (i ? ", " : "") + seen[i];
    out += "): RCCL must be the one built against the runtime this library uses -- load the library before the other copy, "
           "set ECOZ2_VQ_RCCL_LIB, or use ECOZ2_VQ_COLLECTIVE=p2p";
    return out;
}

// devices[r] = HIP device of rank r.  collective: "rccl", "p2p" or "" (RCCL when every rank has a device of its own and
// librccl.so loads, else the peer-to-peer kernel).  Returns null with the error message set.
GroupImpl* group_create(int world, const int* devices, const std::string& coll, bool verbose)
{
    if (world < 1 || world > e2vq::E2VQ_MAX_LOCAL_RANKS) {
        e2vq_set_error("%d in-process ranks: expected 1 .. %d", world, e2vq::E2VQ_MAX_LOCAL_RANKS);
        return nullptr;
    }
    if (!coll.empty() && coll != "rccl" && coll != "p2p") {
        e2vq_set_error("collective '%s': expected rccl or p2p", coll.c_str());
        return nullptr;
    }
    std::unique_ptr<GroupImpl> G(new GroupImpl());
    G->world = world;
    G->g.n = world;
    G->g.ev_ready.assign((size_t)world, nullptr);
    G->g.ev_done.assign((size_t)world, nullptr);
    G->devs.assign(devices, devices + world);
    G->ranks.resize((size_t)world);
    bool distinct = true;
    for (int r = 0; r < world; ++r) {
        G->ranks[(size_t)r] = LocalRank{&G->g, r, devices[r]};
        for (int q = 0; q < r; ++q) distinct = distinct && devices[q] != devices[r];
    }
    bool use_rccl = coll == "rccl" || (coll.empty() && distinct);
    if (use_rccl && !distinct) {
        if (verbose) printf("collective: ranks share a device: RCCL needs one device per rank, using the peer-to-peer exchange\n");
        use_rccl = false;
    }
    if (use_rccl && !rccl_api()) {
        if (coll == "rccl") {
            e2vq_set_error("collective rccl: librccl.so could not be loaded (dlopen failed)");
            return nullptr;
        }
        if (verbose) printf("collective: librccl.so not found, using the peer-to-peer exchange\n");
        use_rccl = false;
    }
    G->use_rccl = use_rccl;
    if (use_rccl) {
        Rccl* api = rccl_api();
        G->rc_comms.comms.assign((size_t)world, nullptr);
        // (RCCL reads the thread's last HIP error after some of its calls: one left behind by an earlier, handled
        // condition -- an event polled before it completed, a probe for free memory -- would fail the initialisation)
        (void)hipGetLastError();
        const int rc = api->comm_init_all(G->rc_comms.comms.data(), world, G->devs.data());
        if (rc != 0) {
            e2vq_set_error("ncclCommInitAll over %d device(s) failed: %s%s", world, api->error_string ? api->error_string(rc) : "?",
                           hip_runtime_copies().c_str());
            return nullptr;
        }
        int ver = 0;
        if (api->get_version) (void)api->get_version(&ver);
        char buf[160];
        snprintf(buf, sizeof buf, "RCCL %d.%d.%d, ncclAllReduce(int64 sum) per LBG iteration over %d rank(s)", ver / 10000,
                 (ver / 100) % 100, ver % 100, world);
        G->what = buf;
        G->g.on_fail = RcclComms::abort_all;
        G->g.on_fail_arg = &G->rc_comms;
        G->rranks.resize((size_t)world);
        for (int r = 0; r < world; ++r) G->rranks[(size_t)r] = RcclRank{&G->g, &G->rc_comms, r, devices[r]};
    } else {
        char buf[160];
        snprintf(buf, sizeof buf, "peer-to-peer reduce-scatter + all-gather kernel (int64 sum) per LBG iteration over %d rank(s)", world);
        G->what = buf;
        for (int r = 0; r < world; ++r) {
            if (hipSetDevice(devices[r]) != hipSuccess ||
                // (release-to-system events: a peer device waits on them before it reads this rank's words)
                hipEventCreateWithFlags(&G->g.ev_ready[(size_t)r], hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess ||
                hipEventCreateWithFlags(&G->g.ev_done[(size_t)r], hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess) {
                e2vq_set_error("in-process group: events on device %d could not be created", devices[r]);
                return nullptr;
            }
        }
        // every rank's slice kernel reads and writes every other rank's buffer: peer access between all pairs of distinct
        // devices ("already enabled" is the only tolerated failure)
        for (int a = 0; a < world; ++a)
            for (int b = 0; b < world; ++b) {
                const int from = devices[a], to = devices[b];
                if (from == to) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, from, to) != hipSuccess || !can) {
                    e2vq_set_error("device %d cannot access device %d (no peer path): the p2p collective needs P2P", from, to);
                    return nullptr;
                }
                if (hipSetDevice(from) != hipSuccess) {
                    e2vq_set_error("hipSetDevice(%d) failed", from);
                    return nullptr;
                }
                const hipError_t pe = hipDeviceEnablePeerAccess(to, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) {
                    e2vq_set_error("hipDeviceEnablePeerAccess(%d -> %d) failed: %s", from, to, hipGetErrorString(pe));
                    return nullptr;
                }
                (void)hipGetLastError();
            }
    }
    if (verbose) printf("collective: %s\n", G->what.c_str());
    return G.release();
}

// the exchange of rank r as a session hook
void group_hook(GroupImpl* G, int r, e2vq_allreduce_fn* fn, void** user, bool* force)
{
    if (G->use_rccl) {
        *fn = rccl_allreduce;
        *user = &G->rranks[(size_t)r];
    } else {
        *fn = local_allreduce;
        *user = &G->ranks[(size_t)r];
    }
    *force = G->use_rccl && G->world == 1;  // (a one-rank RCCL group exercises the plumbing on one GPU)
}

}  // namespace

struct e2vq_group {
    GroupImpl* impl;
};

extern "C" int e2vq_group_create(int num_ranks, const int* devices, const char* collective, e2vq_group** out)
{
    *out = nullptr;
    if (!devices) return e2vq_set_error("e2vq_group_create: no device list");
    const int ndev = e2vq_device_count();
    for (int r = 0; r < num_ranks; ++r)
        if (devices[r] < 0 || devices[r] >= ndev) return e2vq_set_error("rank %d: device %d not in [0, %d)", r, devices[r], ndev);
    GroupImpl* G = group_create(num_ranks, devices, collective ? collective : "", false);
    if (!G) return 1;
    *out = new e2vq_group{G};
    return 0;
}

extern "C" int e2vq_group_bind(e2vq_group* g, int rank, e2vq_session* s)
{
    if (!g || !s || rank < 0 || rank >= g->impl->world) return e2vq_set_error("e2vq_group_bind: bad arguments");
    if (s->device != g->impl->devs[(size_t)rank])
        return e2vq_set_error("rank %d of the group lives on device %d, the session on device %d", rank, g->impl->devs[(size_t)rank], s->device);
    e2vq_allreduce_fn fn = nullptr;
    void* user = nullptr;
    bool force = false;
    group_hook(g->impl, rank, &fn, &user, &force);
    if (e2vq_set_allreduce(s, fn, user, rank, g->impl->world)) return 1;
    s->ar_force = force;
    s->group_failed = &g->impl->g.failed;
    return 0;
}

extern "C" const char* e2vq_group_collective(e2vq_group* g) { return g ? g->impl->what.c_str() : ""; }
extern "C" int e2vq_group_uses_rccl(e2vq_group* g) { return g && g->impl->use_rccl ? 1 : 0; }
// a rank that fails outside the library's calls (its thread gives up) releases the others from their rendezvous
extern "C" void e2vq_group_fail(e2vq_group* g)
{
    if (g) g->impl->g.fail();
}
// every session bound to the group must have been destroyed (or have synchronised its stream) before
extern "C" void e2vq_group_destroy(e2vq_group* g)
{
    if (!g) return;
    delete g->impl;
    delete g;
}

// ==========================================================================================
// Part 1: the reference's entry points
// ==========================================================================================

static int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

static const char* env_str(const char* name, const char* dflt)
{
    const char* v = getenv(name);
    return v && *v ? v : dflt;
}

// the training set as a list of files: per file its vector count and the global index of its first vector
struct PrdSet {
    const char* const* files = nullptr;
    int n = 0, P = 0;
    std::vector<i64> first;  // n + 1 entries
    i64 T = 0;
};

// Pinned staging buffers cost page pinning both ways: ~0.2 ms per MB to make, ~0.13 ms per MB to release
// (tools/probe/alloc_cost.hip: 2 x 78 MB = 27-35 ms + 19-22 ms -- a fifth of a warm ecoz2_vq_learn over 10 M frames, a third
// of an ecoz2_vq_quantize).  The process keeps them for its next call instead: up to ECOZ2_VQ_PINNED_KEEP_MB (default 512; 0 =
// allocate and free every time) stay in this pool, portable across devices; whatever is pooled when the process ends is left to
// the operating system (the HIP runtime may already be gone when static destructors run).
namespace {
struct PinnedPool {
    struct Buf {
        void* p;
        size_t bytes;
    };
    std::mutex m;
    std::vector<Buf> idle;
    size_t kept = 0;
    static size_t cap()
    {
        static const size_t c = (size_t)(getenv("ECOZ2_VQ_PINNED_KEEP_MB") ? std::max(0, atoi(getenv("ECOZ2_VQ_PINNED_KEEP_MB"))) : 512) << 20;
        return c;
    }
    // a buffer of at least `bytes` (an idle one no larger than twice that, else a new one); null on failure
    void* acquire(size_t bytes, size_t* got)
    {
        // sizes in steps of 32 MB (1 MB below 16 MB): the staging buffers of learn, quantize and classify differ by a few
        // per cent and should be able to stand in for each other
        const size_t step = bytes >= ((size_t)16 << 20) ? (size_t)32 << 20 : (size_t)1 << 20;
        bytes = (bytes + step - 1) / step * step;
        {
            std::lock_guard<std::mutex> lk(m);
            int best = -1;
            for (int i = 0; i < (int)idle.size(); ++i)
                if (idle[(size_t)i].bytes >= bytes && idle[(size_t)i].bytes <= 2 * bytes &&
                    (best < 0 || idle[(size_t)i].bytes < idle[(size_t)best].bytes))
                    best = i;
            if (best >= 0) {
                const Buf b = idle[(size_t)best];
                idle.erase(idle.begin() + best);
                kept -= b.bytes;
                *got = b.bytes;
                return b.p;
            }
        }
        void* p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        *got = bytes;
        return p;
    }
    void release(void* p, size_t bytes)
    {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(m);
            if (kept + bytes <= cap()) {
                idle.push_back(Buf{p, bytes});
                kept += bytes;
                return;
            }
        }
        (void)hipHostFree(p);
    }
};
PinnedPool& pinned_pool()
{
    static PinnedPool* pool = new PinnedPool();  // (never destroyed: see above)
    return *pool;
}
}  // namespace

static int scan_predictors(const char* const* files, int n, int P_expected, PrdSet& ps)
{
    ps.files = files;
    ps.n = n;
    ps.first.assign(1, 0);
    int P = P_expected;
    for (int i = 0; i < n; ++i) {
        char cls[96];
        int p;
        int64_t t;
        if (e2vq_prd_info(files[i], cls, &p, &t)) return 1;
        if (P < 0) P = p;
        if (p != P) return e2vq_set_error("%s: prediction order %d, expected %d", files[i], p, P);
        ps.first.push_back(ps.first.back() + t);
    }
    ps.T = ps.first.back();
    ps.P = P;
    if (ps.T < 1) return e2vq_set_error("no training vectors");
    return 0;
}

// Frames [lo, hi) of the set (file order = frame order) into the session: each rank reads only its own range, in
// chunks through two pinned buffers, so that reading chunk k + 1 from the files overlaps the host-to-device copy of
// chunk k; the row-major device copy is then re-laid out by e2vq_set_frames_device.
static int upload_predictors(e2vq_session* s, const PrdSet& ps, i64 lo, i64 hi)
{
    const int NC = ps.P + 1;
    const i64 T = hi - lo;
    if (T < 1) return e2vq_set_error("empty training shard");
    HIPCHK(hipSetDevice(s->device));
    struct Res {
        double* d = nullptr;
        double* h[2] = {nullptr, nullptr};
        size_t hb[2] = {0, 0};
        hipEvent_t ev[2] = {nullptr, nullptr};
        hipStream_t st = nullptr;
        ~Res()
        {
            if (st) (void)hipStreamSynchronize(st);  // (no copy still reads a staging buffer that goes back to the pool)
            if (d) (void)hipFree(d);
            for (int k = 0; k < 2; ++k) {
                pinned_pool().release(h[k], hb[k]);
                if (ev[k]) (void)hipEventDestroy(ev[k]);
            }
            if (st) (void)hipStreamDestroy(st);
        }
    } r;
    static const bool timing = getenv("ECOZ2_VQ_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tl = now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const double t1 = now();
        fprintf(stderr, "[ecoz2 vq learn]   upload: %-22s %8.1f ms\n", what, (t1 - tl) * 1e3);
        tl = t1;
    };
    const i64 CH = std::min<i64>(T, 1 << 18);  // 78 MB of predictor vectors per chunk at P = 36
    HIPCHK(hipMalloc(&r.d, (size_t)((T + 63) / 64 * 64) * NC * 8 + 16));  // (whole blocks + 16 bytes: the session may keep the buffer)
    HIPCHK(hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        r.h[k] = (double*)pinned_pool().acquire((size_t)CH * NC * 8, &r.hb[k]);
        if (!r.h[k]) return e2vq_set_error("no pinned memory for the upload staging (%zu bytes)", (size_t)CH * NC * 8);
        HIPCHK(hipEventCreateWithFlags(&r.ev[k], hipEventDisableTiming));
    }
    lap("allocations");
    int file = (int)(std::upper_bound(ps.first.begin(), ps.first.end(), lo) - ps.first.begin()) - 1;
    int k = 0;
    for (i64 t0 = lo; t0 < hi; t0 += CH, k ^= 1) {
        const i64 n = std::min(CH, hi - t0);
        HIPCHK(hipEventSynchronize(r.ev[k]));  // (never recorded: returns at once) the copy out of this buffer is done
        for (i64 got = 0; got < n;) {          // a chunk may span several files
            while (ps.first[(size_t)file + 1] <= t0 + got) ++file;
            const i64 in_file = t0 + got - ps.first[(size_t)file];
            const i64 take = std::min(n - got, ps.first[(size_t)file + 1] - (t0 + got));
            if (e2vq_io::prd_read_range_mt(ps.files[file], ps.P, in_file, take, r.h[k] + (size_t)got * NC,
                                           e2vq_io::io_threads()))
                return 1;
            got += take;
        }
        HIPCHK(hipMemcpyAsync(r.d + (size_t)(t0 - lo) * NC, r.h[k], (size_t)n * NC * 8, hipMemcpyHostToDevice, r.st));
        HIPCHK(hipEventRecord(r.ev[k], r.st));
    }
    HIPCHK(hipStreamSynchronize(r.st));
    lap("read + H2D");
    bool adopted = false;
    const int rc = set_frames_device_impl(s, r.d, T, &adopted);  // (synchronises: the row-major copy can go, unless the session kept it)
    if (adopted) r.d = nullptr;
    lap("re-layout + images");
    return rc;
}

// one rank of a learn: session on `device`, frames [lo, hi) of the training set.
// Every failing path of a group rank marks the group failed, so the other ranks leave their barriers.
// how a rank of an in-process group exchanges its cell sums: the hook, its argument, and the group to mark failed
struct RankCtx {
    LocalGroup* g = nullptr;
    int rank = 0;
    e2vq_allreduce_fn fn = nullptr;
    void* user = nullptr;
    bool force = false;  // call the hook even in a group of one
};

static int learn_rank(int device, double eps, const char* class_name, const double* base_refl, int base_M,
                      const PrdSet& ps, i64 lo, i64 hi, const RankCtx* lr, int world, void* target,
                      ecoz2_vq_learn_callback_t cb)
{
    // ECOZ2_VQ_TIMING=1: wall time of the stages of a rank on stderr (diagnostics)
    static const bool timing = getenv("ECOZ2_VQ_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const double t1 = now();
        fprintf(stderr, "[ecoz2 vq learn, rank %d] %-28s %8.1f ms\n", lr ? lr->rank : 0, what, (t1 - t0) * 1e3);
        t0 = t1;
    };
    e2vq_session* s = nullptr;
    int rc = e2vq_session_create(device, ps.P, &s);
    if (!rc && lr) {
        rc = e2vq_set_allreduce(s, lr->fn, lr->user, lr->rank, world);
        s->ar_force = lr->force;
    }
    lap("session");
    if (!rc) rc = upload_predictors(s, ps, lo, hi);
    lap("read + upload + re-layout");
    if (!rc) rc = e2vq_prepare(s);
    if (!rc) rc = base_refl ? e2vq_set_codebook(s, base_refl, base_M) : e2vq_init_codebook(s);
    lap("statistics, first codebook");
    if (!rc)
        rc = e2vq_learn(s, eps, env_int("ECOZ2_VQ_MAX_CODEBOOK_SIZE", 2048), class_name,
                        env_str("ECOZ2_VQ_OUT_ROOT", "."), target, cb, nullptr, 0, nullptr);
    lap("LBG ladder (+ files)");
    if (rc && lr && lr->g) lr->g->fail();
    if (s) e2vq_session_destroy(s);
    return rc;
}

static int learn_common(int P, double eps, const char* class_name, const double* base_refl, int base_M,
                        const char* const* files, int n, void* target, ecoz2_vq_learn_callback_t cb)
{
    PrdSet ps;
    if (scan_predictors(files, n, P, ps)) return 1;
    const i64 T = ps.T;
    printf("Codebook generation:\n\n%lld training vectors (ε=%g)\n", (long long)T, eps);
    const int ndev = e2vq_device_count();
    if (ndev < 1) return e2vq_set_error("no HIP device available; this library has no CPU path");
    const int dev0 = env_int("ECOZ2_VQ_DEVICE", 0);
    int world = env_int("ECOZ2_VQ_GPUS", 1);
    if (world < 1) world = 1;
    if ((i64)world > T) world = (int)T;  // every rank needs at least one training vector
    // ECOZ2_VQ_COLLECTIVE = rccl | p2p (default: RCCL when every rank has a device of its own, else the peer-to-peer
    // slice kernel -- RCCL cannot place two ranks of a communicator on one device)
    const std::string coll = env_str("ECOZ2_VQ_COLLECTIVE", "");
    if (!coll.empty() && coll != "rccl" && coll != "p2p")
        return e2vq_set_error("ECOZ2_VQ_COLLECTIVE=%s: expected rccl or p2p", coll.c_str());
    if (world == 1 && coll != "rccl")
        return learn_rank(dev0, eps, class_name, base_refl, base_M, ps, 0, T, nullptr, 1, target, cb);

    // ---- in-process group: rank r on device (dev0 + r) % ndev, contiguous frame shards --------------------------
    printf("sharding over %d rank(s) on %d device(s)\n", world, ndev);
    if (world > e2vq::E2VQ_MAX_LOCAL_RANKS) return e2vq_set_error("ECOZ2_VQ_GPUS=%d exceeds %d in-process ranks", world, e2vq::E2VQ_MAX_LOCAL_RANKS);
    std::vector<int> devs((size_t)world);
    for (int r = 0; r < world; ++r) devs[(size_t)r] = (dev0 + r) % ndev;
    std::unique_ptr<GroupImpl> G(group_create(world, devs.data(), coll, true));  // (events and communicators go with it on every return path)
    if (!G) return 1;
    LocalGroup& g = G->g;
    const bool use_rccl = G->use_rccl;
    std::vector<LocalRank>& ranks = G->ranks;
    std::vector<RcclRank>& rranks = G->rranks;
    std::vector<RankCtx> ctx((size_t)world);
    for (int r = 0; r < world; ++r) {
        ctx[(size_t)r].g = &g;
        ctx[(size_t)r].rank = r;
        group_hook(G.get(), r, &ctx[(size_t)r].fn, &ctx[(size_t)r].user, &ctx[(size_t)r].force);
    }
    std::vector<int> rcs((size_t)world, 0);
    std::vector<std::thread> th;
    auto shard = [&](int r, i64* lo, i64* hi) {
        const i64 base = T / world, rem = T % world;
        *lo = r * base + std::min<i64>(r, rem);
        *hi = *lo + base + (r < rem ? 1 : 0);
    };
    for (int r = 1; r < world; ++r) {
        th.emplace_back([&, r]() {
            i64 lo, hi;
            shard(r, &lo, &hi);
            rcs[r] = learn_rank(ranks[r].device, eps, class_name, base_refl, base_M, ps, lo, hi, &ctx[r], world, nullptr, nullptr);
        });
    }
    {  // rank 0 runs on the calling thread: files, messages and the callback come from here
        i64 lo, hi;
        shard(0, &lo, &hi);
        rcs[0] = learn_rank(ranks[0].device, eps, class_name, base_refl, base_M, ps, lo, hi, &ctx[0], world, target, cb);
    }
    for (auto& t : th) t.join();
    if (use_rccl && !getenv("ECOZ2_VQ_QUIET"))
        printf("collective: rank 0 made %ld ncclAllReduce call(s), %ld bytes\n", rranks[0].calls, rranks[0].bytes);
    for (int rc : rcs)
        if (rc) {
            // the message of the rank that failed FIRST (the others only report the broken barrier)
            if (!g.first_error.empty()) snprintf(g_err, sizeof g_err, "%s", g.first_error.c_str());
            return rc;
        }
    return 0;
}

extern "C" int ecoz2_vq_learn(int prediction_order, double epsilon, const char* codebook_class_name,
                              const char* const* predictor_filenames, int num_predictors, void* target,
                              ecoz2_vq_learn_callback_t callback)
{
    if (!codebook_class_name || !predictor_filenames || num_predictors < 1)
        return e2vq_set_error("ecoz2_vq_learn: bad arguments");
    return learn_common(prediction_order, epsilon, codebook_class_name, nullptr, 0, predictor_filenames,
                        num_predictors, target, callback);
}

extern "C" int ecoz2_vq_learn_using_base_codebook(const char* base_codebook, double epsilon,
                                                  const char* const* predictor_filenames, int num_predictors,
                                                  void* target, ecoz2_vq_learn_callback_t callback)
{
    if (!base_codebook || !predictor_filenames || num_predictors < 1)
        return e2vq_set_error("ecoz2_vq_learn_using_base_codebook: bad arguments");
    char cls[96];
    int P, M;
    if (e2vq_cbook_info(base_codebook, cls, &P, &M)) return 1;
    std::vector<double> refl((size_t)M * (P + 1));
    if (e2vq_cbook_read(base_codebook, refl.data(), M)) return 1;
    printf("base codebook: %s (class '%s', P=%d, M=%d)\n", base_codebook, cls, P, M);
    return learn_common(P, epsilon, cls, refl.data(), M, predictor_filenames, num_predictors, target, callback);
}

// ---- vq quantize / vq classify: predictor files through the GPU with I/O, copies and sweeps overlapped ------------
namespace {

// ---- ecoz2_vq_quantize: units of at most CHUNK frames through fixed-size pinned staging ------------------------------
// A unit is a run of consecutive frames of the corpus (file order, frame order) made of segments (file, first frame,
// count): many short files are batched into one unit -- one upload, ONE sweep, one download for all of them (frames are
// independent, a 64-frame block may span files) --, a file longer than a chunk is split into several units that any
// worker may take.  Workers (ECOZ2_VQ_GPUS: one session + host thread each, device (dev0 + w) % ndev) pull units from a
// shared counter; each keeps two units in flight so that file reads, the host-to-device copy, the sweep and the .seq
// writes overlap.  Every worker allocates ONE pinned and ONE device block (2 slots x CHUNK frames), whatever the file
// sizes: round 2's whole-file slots cost 370 MB of pinned memory per slot at 1.25 M frames, and four workers sharing a
// device took 0.82 s where one took 0.23.
struct QSegment {
    int file;
    i64 t0, n;    // frames [t0, t0 + n) of the file
    i64 off;      // position of the segment's first frame in the unit
    bool whole;   // the segment is the whole file
};
struct QUnit {
    std::vector<QSegment> segs;
    i64 n = 0;
};
struct QFileResult {
    i64 T = 0;
    double e = 0.0;  // sum over the file's frames of (dmin - 1), frame order
    std::string cls, seq_path;
    std::string tmp_path;  // split files are written piecewise to <seq_path>.tmp and renamed once every range is stored
    // split files: chunks fold into `e` in frame order whatever order the workers finish them in
    std::mutex mu;
    i64 next_t = 0;
    std::vector<std::pair<i64, std::vector<double>>> pending;
};

struct QSlot {
    double* h_frames = nullptr;
    uint16_t* h_sym = nullptr;
    double* h_dmin = nullptr;
    double* d_frames = nullptr;
    unsigned short* d_sym = nullptr;
    double* d_dmin = nullptr;
    hipEvent_t done = nullptr;
    int unit = -1;  // index of the unit in flight, -1 = free
};

struct QShared {
    const char* const* files;
    int P, M;
    const char* root;
    i64 chunk;
    std::vector<QUnit> units;
    std::vector<QFileResult> results;
    std::atomic<int> next{0};
    std::atomic<bool> failed{false};
    QShared(int nfiles) : results((size_t)nfiles) {}
};

// folds the distortions of frames [t0, t0 + n) of a file into its sum, in frame order
void quantize_fold(QFileResult& r, i64 t0, const double* dmin, i64 n)
{
    std::lock_guard<std::mutex> lk(r.mu);
    if (t0 != r.next_t) {  // an earlier chunk of the file is still in flight: park this one
        r.pending.emplace_back(t0, std::vector<double>(dmin, dmin + n));
        return;
    }
    double e = r.e;
    for (i64 t = 0; t < n; ++t) e += dmin[t] - 1.0;
    r.next_t += n;
    for (bool again = true; again;) {
        again = false;
        for (size_t k = 0; k < r.pending.size(); ++k)
            if (r.pending[k].first == r.next_t) {
                for (double d : r.pending[k].second) e += d - 1.0;
                r.next_t += (i64)r.pending[k].second.size();
                r.pending.erase(r.pending.begin() + (long)k);
                again = true;
                break;
            }
    }
    r.e = e;
}

int quantize_worker(int device, QShared& sh, const double* refl)
{
    static const bool timing = getenv("ECOZ2_VQ_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    const int NC = sh.P + 1;
    e2vq_session* s = nullptr;
    if (e2vq_session_create(device, sh.P, &s)) return 1;
    hipStream_t st = nullptr;
    QSlot slots[2];
    char* h_block = nullptr;
    char* d_block = nullptr;
    int rc = e2vq_set_codebook(s, refl, sh.M);
    if (!rc && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) rc = e2vq_set_error("stream creation failed");
    if (!rc) rc = e2vq_set_stream(s, (void*)st);
    // one pinned and one device allocation, carved into the two slots (frames | distortions | symbols, 256-byte aligned)
    const size_t fb = ((size_t)sh.chunk * NC * 8 + 255) & ~(size_t)255, db = ((size_t)sh.chunk * 8 + 255) & ~(size_t)255,
                 sb = ((size_t)sh.chunk * 2 + 64 + 255) & ~(size_t)255, slot_bytes = fb + db + sb;
    size_t h_block_bytes = 0;
    if (!rc && !(h_block = (char*)pinned_pool().acquire(2 * slot_bytes, &h_block_bytes)))
        rc = e2vq_set_error("no pinned memory for the quantize staging (%zu bytes)", 2 * slot_bytes);
    if (!rc && hipMalloc((void**)&d_block, 2 * slot_bytes) != hipSuccess)
        rc = e2vq_set_error("no device memory for the quantize staging (%zu bytes)", 2 * slot_bytes);
    for (int k = 0; k < 2 && !rc; ++k) {
        QSlot& q = slots[k];
        q.h_frames = (double*)(h_block + k * slot_bytes);
        q.h_dmin = (double*)(h_block + k * slot_bytes + fb);
        q.h_sym = (uint16_t*)(h_block + k * slot_bytes + fb + db);
        q.d_frames = (double*)(d_block + k * slot_bytes);
        q.d_dmin = (double*)(d_block + k * slot_bytes + fb);
        q.d_sym = (unsigned short*)(d_block + k * slot_bytes + fb + db);
        if (hipEventCreateWithFlags(&q.done, hipEventDisableTiming) != hipSuccess) rc = e2vq_set_error("event creation failed");
    }
    const double t_setup = now();
    auto finish = [&](QSlot& q) -> int {  // results of the unit in flight in q: distortion sums + .seq files
        if (q.unit < 0) return 0;
        HIPCHK(hipEventSynchronize(q.done));
        const QUnit& u = sh.units[(size_t)q.unit];
        q.unit = -1;
        for (const QSegment& g : u.segs) {
            QFileResult& r = sh.results[(size_t)g.file];
            if (g.whole) {
                double e = 0.0;
                for (i64 t = 0; t < g.n; ++t) e += q.h_dmin[g.off + t] - 1.0;
                r.e = e;
                if (e2vq_seq_write(r.seq_path.c_str(), r.cls.c_str(), sh.M, q.h_sym + g.off, g.n)) return 1;
            } else {
                quantize_fold(r, g.t0, q.h_dmin + g.off, g.n);
                if (e2vq_io::seq_write_range(r.tmp_path.c_str(), g.t0, q.h_sym + g.off, g.n)) return 1;
            }
        }
        return 0;
    };
    int k = 0, done_units = 0;
    while (!rc && !sh.failed.load()) {
        const int ui = sh.next.fetch_add(1);
        if (ui >= (int)sh.units.size()) break;
        QSlot& q = slots[k & 1];
        ++k;
        rc = finish(q);
        if (rc) break;
        const QUnit& u = sh.units[(size_t)ui];
        bool finite = true;
        for (const QSegment& g : u.segs) {
            if (g.n < 1) continue;
            bool fin = true;
            rc = e2vq_io::prd_read_range_mt(sh.files[g.file], sh.P, g.t0, g.n, q.h_frames + (size_t)g.off * NC,
                                            e2vq_io::io_threads(), &fin);
            if (rc) break;
            if (!fin) {
                rc = e2vq_set_error("%s: contains NaN or infinite values", sh.files[g.file]);
                finite = false;
                break;
            }
        }
        if (rc || !finite) break;
        q.unit = ui;
        if (u.n > 0) {
            hipError_t e = hipMemcpyAsync(q.d_frames, q.h_frames, (size_t)u.n * NC * 8, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) rc = e2vq_quantize_device(s, q.d_frames, u.n, q.d_sym, q.d_dmin);
            if (e == hipSuccess && !rc) e = hipMemcpyAsync(q.h_sym, q.d_sym, (size_t)u.n * 2, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && !rc) e = hipMemcpyAsync(q.h_dmin, q.d_dmin, (size_t)u.n * 8, hipMemcpyDeviceToHost, st);
            if (e != hipSuccess) rc = e2vq_set_error("quantize: copy failed: %s", hipGetErrorString(e));
        }
        if (!rc && hipEventRecord(q.done, st) != hipSuccess) rc = e2vq_set_error("event record failed");
        ++done_units;
    }
    for (int j = 0; j < 2 && !rc; ++j) rc = finish(slots[(k + j) & 1]);  // oldest first
    if (rc) sh.failed.store(true);
    if (st) (void)hipStreamSynchronize(st);
    const double t_work = now();
    for (QSlot& q : slots)
        if (q.done) (void)hipEventDestroy(q.done);
    pinned_pool().release(h_block, h_block_bytes);  // (the stream was synchronised above)
    if (d_block) (void)hipFree(d_block);
    e2vq_session_destroy(s);
    if (st) (void)hipStreamDestroy(st);
    if (timing)
        fprintf(stderr, "[ecoz2 vq quantize, device %d] setup %.1f ms, %d unit(s) %.1f ms, teardown %.1f ms\n", device,
                (t_setup - t_start) * 1e3, done_units, (t_work - t_setup) * 1e3, (now() - t_work) * 1e3);
    return rc;
}

}  // namespace

// ECOZ2_VQ_GPUS = N workers (one session + host thread each; ranks beyond the device count share devices).  Frames are
// independent, so there is no collective; every .seq, and the totals (per file in frame order, files in list order, on
// the calling thread), are the same for any N.  ECOZ2_VQ_QUANTIZE_CHUNK: frames per unit (default 2^17 = 39 MB at P = 36).
extern "C" int ecoz2_vq_quantize(const char* nom_raas, const char* const* predictor_filenames, int num_predictors,
                                 int show_filenames)
{
    if (!nom_raas || !predictor_filenames || num_predictors < 0) return e2vq_set_error("ecoz2_vq_quantize: bad arguments");
    char cb_cls[96];
    int P, M;
    if (e2vq_cbook_info(nom_raas, cb_cls, &P, &M)) return 1;
    std::vector<double> refl((size_t)M * (P + 1));
    if (e2vq_cbook_read(nom_raas, refl.data(), M)) return 1;
    const int ndev = e2vq_device_count();
    if (ndev < 1) return e2vq_set_error("no HIP device available; this library has no CPU path");
    const int dev0 = env_int("ECOZ2_VQ_DEVICE", 0);
    const char* root = env_str("ECOZ2_VQ_OUT_ROOT", ".");
    QShared sh(num_predictors);
    // split files are written to <seq>.tmp and renamed at the end: whatever way this call ends short of that, the .tmp files
    // it has created so far go away (a later file's bad header, a failed worker, a failed rename)
    struct TmpGuard {
        QShared& sh;
        bool keep = false;
        ~TmpGuard()
        {
            if (!keep)
                for (const QFileResult& r : sh.results)
                    if (!r.tmp_path.empty()) (void)remove(r.tmp_path.c_str());
        }
    } tmp_guard{sh};
    sh.files = predictor_filenames;
    sh.P = P;
    sh.M = M;
    sh.root = root;
    sh.chunk = std::max(1024, env_int("ECOZ2_VQ_QUANTIZE_CHUNK", 1 << 17));
    // plan: headers of every file, then units of at most `chunk` frames
    {
        QUnit cur;
        auto flush = [&] {
            if (!cur.segs.empty()) sh.units.push_back(std::move(cur));
            cur = QUnit();
        };
        for (int i = 0; i < num_predictors; ++i) {
            char cls[96];
            int p;
            int64_t T;
            if (e2vq_prd_info(predictor_filenames[i], cls, &p, &T)) return 1;
            if (p != P) return e2vq_set_error("%s: prediction order %d differs from the codebook's %d", predictor_filenames[i], p, P);
            QFileResult& r = sh.results[(size_t)i];
            r.T = T;
            r.cls = cls;
            char path[4096];
            snprintf(path, sizeof path, "%s/data/sequences/M%d/%s/%s.seq", root, M, cls,
                     e2vq_io::basename_noext(predictor_filenames[i]).c_str());
            r.seq_path = path;
            if (T <= sh.chunk) {
                if (cur.n + T > sh.chunk) flush();
                cur.segs.push_back(QSegment{i, 0, T, cur.n, true});
                cur.n += T;
            } else {  // longer than a chunk: units of its own, any worker takes them; the .seq is written piecewise
                flush();
                // (not at the final path: a run that fails later must not leave a well-formed .seq of zeros behind, nor
                // overwrite an earlier good one)
                r.tmp_path = r.seq_path + ".tmp";
                if (e2vq_io::seq_create(r.tmp_path.c_str(), cls, M, T)) return 1;
                for (i64 t0 = 0; t0 < T; t0 += sh.chunk) {
                    const i64 n = std::min<i64>(sh.chunk, T - t0);
                    cur.segs.push_back(QSegment{i, t0, n, 0, false});
                    cur.n = n;
                    flush();
                }
            }
        }
        flush();
    }
    int W = std::max(1, env_int("ECOZ2_VQ_GPUS", 1));
    W = std::max(1, std::min(W, (int)sh.units.size()));
    // Workers that SHARE a device only pay when there is host work per file to spread (5 000 short files: 0.43 -> 0.19 s with
    // four of them); on a few long files they cost a session each and gain nothing (0.11 -> 0.13 s): beyond one worker
    // per distinct device, one more per 256 files
    {
        const int distinct = std::min(W, ndev);
        if (W > distinct) W = std::max(distinct, std::min(W, num_predictors / 256));
    }
    std::vector<int> rcs((size_t)W, 0);
    std::vector<std::string> errs((size_t)W);
    std::vector<std::thread> th;
    auto run = [&](int w) {
        rcs[(size_t)w] = quantize_worker((dev0 + w) % ndev, sh, refl.data());
        if (rcs[(size_t)w]) errs[(size_t)w] = g_err;
    };
    for (int w = 1; w < W; ++w) th.emplace_back(run, w);
    run(0);
    for (auto& t : th) t.join();
    for (int w = 0; w < W; ++w)
        if (rcs[(size_t)w]) {
            if (w > 0) snprintf(g_err, sizeof g_err, "%s", errs[(size_t)w].c_str());
            return rcs[(size_t)w];
        }
    for (QFileResult& r : sh.results)
        if (!r.tmp_path.empty()) {
            if (rename(r.tmp_path.c_str(), r.seq_path.c_str()) != 0)
                return e2vq_set_error("%s: cannot move the finished sequence into place: %s", r.seq_path.c_str(), strerror(errno));
            r.tmp_path.clear();  // (in place: no longer the guard's business)
        }
    tmp_guard.keep = true;
    double total_e = 0.0;
    i64 total_T = 0;
    for (int i = 0; i < num_predictors; ++i) {
        const QFileResult& r = sh.results[(size_t)i];
        total_e += r.e;
        total_T += r.T;
        if (show_filenames)
            printf("%s: '%s' T=%lld avg distortion=%g -> %s\n", predictor_filenames[i], r.cls.c_str(), (long long)r.T,
                   r.T ? r.e / (double)r.T : 0.0, r.seq_path.c_str());
    }
    printf("total: %d predictor file(s), %lld vectors, M=%d, avg distortion=%g\n", num_predictors, (long long)total_T, M,
           total_T ? total_e / (double)total_T : 0.0);
    return 0;
}

extern "C" int ecoz2_vq_classify(const char* const* cb_filenames, int num_codebooks, const char* const* prd_filenames,
                                 int num_predictors, int show_ranked)
{
    if (!cb_filenames || !prd_filenames || num_codebooks < 1 || num_predictors < 0)
        return e2vq_set_error("ecoz2_vq_classify: bad arguments");
    struct Cb {
        std::string cls;
        int P, M;
        std::vector<double> refl;
    };
    std::vector<Cb> cbs((size_t)num_codebooks);
    for (int i = 0; i < num_codebooks; ++i) {
        char cls[96];
        if (e2vq_cbook_info(cb_filenames[i], cls, &cbs[i].P, &cbs[i].M)) return 1;
        cbs[i].cls = cls;
        cbs[i].refl.resize((size_t)cbs[i].M * (cbs[i].P + 1));
        if (e2vq_cbook_read(cb_filenames[i], cbs[i].refl.data(), cbs[i].M)) return 1;
        if (cbs[i].P != cbs[0].P) return e2vq_set_error("%s: prediction order differs from the first codebook", cb_filenames[i]);
    }
    const int P = cbs[0].P, NC = P + 1;
    // Predictor files stream through fixed-size pinned staging in units of at most ECOZ2_VQ_QUANTIZE_CHUNK frames (short
    // files batched, long ones cut: as ecoz2_vq_quantize); every unit is uploaded ONCE and swept once per codebook where it
    // lies.  Host and device memory stay bounded whatever the corpus (round 2 held every frame in one std::vector and one
    // device allocation, copied with a pageable hipMemcpy).
    struct Prd {
        std::string cls;
        int64_t T = 0;
    };
    std::vector<Prd> prds((size_t)num_predictors);
    const i64 chunk = std::max(1024, env_int("ECOZ2_VQ_QUANTIZE_CHUNK", 1 << 17));
    std::vector<QUnit> units;
    {
        QUnit cur;
        auto flush = [&] {
            if (!cur.segs.empty()) units.push_back(std::move(cur));
            cur = QUnit();
        };
        for (int k = 0; k < num_predictors; ++k) {
            char cls[96];
            int p;
            if (e2vq_prd_info(prd_filenames[k], cls, &p, &prds[k].T)) return 1;
            if (p != P) return e2vq_set_error("%s: prediction order %d differs from the codebooks' %d", prd_filenames[k], p, P);
            prds[k].cls = cls;
            const i64 T = prds[k].T;
            if (T <= chunk) {
                if (cur.n + T > chunk) flush();
                cur.segs.push_back(QSegment{k, 0, T, cur.n, true});
                cur.n += T;
            } else {
                flush();
                for (i64 t0 = 0; t0 < T; t0 += chunk) {
                    const i64 n = std::min<i64>(chunk, T - t0);
                    cur.segs.push_back(QSegment{k, t0, n, 0, false});
                    cur.n = n;
                    flush();
                }
            }
        }
        flush();
    }
    e2vq_session* s = nullptr;
    const int device = env_int("ECOZ2_VQ_DEVICE", 0);
    if (e2vq_session_create(device, P, &s)) return 1;
    // sums of (dmin - 1) per (file, codebook), in frame order (units are processed in order, frames within a unit too)
    std::vector<double> esum((size_t)num_predictors * num_codebooks, 0.0);
    int rc = 0;
    hipStream_t st = nullptr;
    double *h_frames = nullptr, *h_dmin = nullptr, *d_frames = nullptr, *d_dmin = nullptr;
    unsigned short* d_sym = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) rc = e2vq_set_error("stream creation failed");
    if (!rc) rc = e2vq_set_stream(s, (void*)st);
    size_t h_frames_bytes = 0, h_dmin_bytes = 0;
    if (!rc && (!(h_frames = (double*)pinned_pool().acquire((size_t)chunk * NC * 8, &h_frames_bytes)) ||
                !(h_dmin = (double*)pinned_pool().acquire((size_t)chunk * 8, &h_dmin_bytes)) ||
                hipMalloc((void**)&d_frames, (size_t)chunk * NC * 8) != hipSuccess || hipMalloc((void**)&d_dmin, (size_t)chunk * 8) != hipSuccess ||
                hipMalloc((void**)&d_sym, (size_t)chunk * 2 + 64) != hipSuccess))
        rc = e2vq_set_error("no memory for the classify staging (%lld frames per unit)", (long long)chunk);
    for (size_t u = 0; u < units.size() && !rc; ++u) {
        const QUnit& un = units[u];
        if (un.n < 1) continue;
        for (const QSegment& g : un.segs) {
            if (g.n < 1) continue;
            bool fin = true;
            rc = e2vq_io::prd_read_range_mt(prd_filenames[g.file], P, g.t0, g.n, h_frames + (size_t)g.off * NC, e2vq_io::io_threads(), &fin);
            if (!rc && !fin) rc = e2vq_set_error("%s: contains NaN or infinite values", prd_filenames[g.file]);
            if (rc) break;
        }
        if (rc) break;
        if (hipMemcpyAsync(d_frames, h_frames, (size_t)un.n * NC * 8, hipMemcpyHostToDevice, st) != hipSuccess)
            rc = e2vq_set_error("upload of the predictor vectors failed");
        for (int i = 0; i < num_codebooks && !rc; ++i) {
            rc = e2vq_set_codebook(s, cbs[i].refl.data(), cbs[i].M);
            if (!rc) rc = e2vq_quantize_device(s, d_frames, un.n, d_sym, d_dmin);
            if (!rc && hipMemcpyAsync(h_dmin, d_dmin, (size_t)un.n * 8, hipMemcpyDeviceToHost, st) != hipSuccess)
                rc = e2vq_set_error("download of the distortions failed");
            if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = e2vq_set_error("classify: device work failed");
            if (rc) break;
            for (const QSegment& g : un.segs) {
                double e = esum[(size_t)g.file * num_codebooks + i];
                for (i64 t = 0; t < g.n; ++t) e += h_dmin[g.off + t] - 1.0;
                esum[(size_t)g.file * num_codebooks + i] = e;
            }
        }
    }
    if (st) (void)hipStreamSynchronize(st);
    pinned_pool().release(h_frames, h_frames_bytes);
    pinned_pool().release(h_dmin, h_dmin_bytes);
    if (d_frames) (void)hipFree(d_frames);
    if (d_dmin) (void)hipFree(d_dmin);
    if (d_sym) (void)hipFree(d_sym);
    e2vq_session_destroy(s);
    if (st) (void)hipStreamDestroy(st);
    if (rc) return rc;
    std::vector<double> score((size_t)num_predictors * num_codebooks, 0.0);
    for (int k = 0; k < num_predictors; ++k)
        for (int i = 0; i < num_codebooks; ++i)
            if (prds[k].T > 0) score[(size_t)k * num_codebooks + i] = esum[(size_t)k * num_codebooks + i] / (double)prds[k].T;
    int correct = 0, total_n = 0;
    std::vector<std::string> classes;
    std::vector<int> ok_by, n_by;
    for (int k = 0; k < num_predictors; ++k) {
        if (prds[k].T < 1) continue;
        const double* sc = &score[(size_t)k * num_codebooks];
        int best = 0;
        for (int i = 1; i < num_codebooks; ++i)
            if (sc[i] < sc[best]) best = i;
        const bool ok = cbs[best].cls == prds[k].cls;
        size_t ci = 0;
        for (; ci < classes.size(); ++ci)
            if (classes[ci] == prds[k].cls) break;
        if (ci == classes.size()) {
            classes.push_back(prds[k].cls);
            ok_by.push_back(0);
            n_by.push_back(0);
        }
        n_by[ci]++;
        ok_by[ci] += ok;
        total_n++;
        correct += ok;
        if (!ok && show_ranked) {
            std::vector<int> order((size_t)num_codebooks);
            for (int i = 0; i < num_codebooks; ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sc[a] < sc[b]; });
            printf("%s: '%s' classified as '%s'; ranked:", prd_filenames[k], prds[k].cls.c_str(), cbs[best].cls.c_str());
            for (int i : order) printf(" %s(%g)", cbs[i].cls.c_str(), sc[i]);
            printf("\n");
        }
    }
    printf("\n%-24s %8s %8s %8s\n", "class", "tests", "correct", "percent");
    for (size_t ci = 0; ci < classes.size(); ++ci)
        printf("%-24s %8d %8d %7.2f%%\n", classes[ci].c_str(), n_by[ci], ok_by[ci], 100.0 * ok_by[ci] / n_by[ci]);
    printf("%-24s %8d %8d %7.2f%%\n", "TOTAL", total_n, correct, total_n ? 100.0 * correct / total_n : 0.0);
    return 0;
}

extern "C" int ecoz2_vq_show(const char* codebook_filename, int from, int to)
{
    char cls[96];
    int P, M;
    if (e2vq_cbook_info(codebook_filename, cls, &P, &M)) return 1;
    std::vector<double> refl((size_t)M * (P + 1));
    if (e2vq_cbook_read(codebook_filename, refl.data(), M)) return 1;
    if (from < 0) from = 1;
    if (to < 0 || to > P) to = P;
    printf("# %s:\n# className='%s', M=%d, P=%d\n", codebook_filename, cls, M, P);
    for (int n = from; n <= to; ++n) printf("%sk%d", n == from ? "" : ",", n);
    printf("\n");
    for (int m = 0; m < M; ++m) {
        for (int n = from; n <= to; ++n) printf("%s%g", n == from ? "" : ",", refl[(size_t)m * (P + 1) + n]);
        printf("\n");
    }
    return 0;
}
