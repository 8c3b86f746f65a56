// vq_host.cpp -- the resident-data session of libecoz2vq.so (include/ecoz2_vq.h, part 2): life cycle, training set, codebook,
// saved points of the ladder, the LBG ladder itself (e2vq_learn) and quantize on resident data.  All arithmetic of the hot
// path runs in the HIP kernels; the host only sequences launches, reads a few scalars per pass to take the convergence
// decision the reference takes on the CPU (loop shape: /root/reference/notes.md:122-153), and does file I/O.
// One LBG iteration is vq_pass.cpp; the in-process group vq_group.cpp; the reference's entry points vq_entry.cpp.
#include "vq_session.h"

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
char* e2vq_err_buf() { return g_err; }

int e2vq_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(e2vq_err_buf(), 1024, fmt, ap);
    va_end(ap);
    fprintf(stderr, "ecoz2vq: ERROR: %s\n", g_err);
    return 1;
}

extern "C" const char* e2vq_last_error(void) { return g_err; }


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


int e2vq_ensure_codebook_capacity(e2vq_session* s, int M)
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
    // max |x| and the two status words side by side: e2vq_prepare reduces them over the ranks as ONE maximum of two words
    HIPCHK(hipMalloc(&s->d_maxabs, 16));
    s->d_flags = (int*)(s->d_maxabs + 1);
    HIPCHK(hipMalloc(&s->d_l1max, 8));
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
    if (const char* rc = getenv("ECOZ2_VQ_RECORDS_FEW_DIV")) s->rec_few_div = std::max(0, atoi(rc));
    // ECOZ2_VQ_ACCUMULATE: which kernels a prefiltered training pass runs -- one switch for the tests and the probes, which
    // have to reach every one of them; "auto" is the product
    {
        const char* am = getenv("ECOZ2_VQ_ACCUMULATE");
        const std::string acc = am ? am : "auto";
        if (acc == "sorted") {           // the fused sorted pass wherever the prefilter runs (default: from M = 256), two
            s->sweep_min_M = 64;         // blocks per turn whatever the shard's size (default: from 4 096 blocks on)
            s->two_blocks_always = true;
        } else if (acc == "sweep") {     // candidate sweep + finishing kernel + k_reduce_records, grouped or not
            s->fused_enabled = false;
            s->sweep_min_M = 64;
        } else if (acc == "records") {   // round 4: k_pass_pre_lds recording its contributions + k_reduce_records
            s->sweep2_enabled = false;
        } else if (acc == "burst") {     // round 3: k_pass_pre_lds with its burst of atomics
            s->sweep2_enabled = false;
            s->rec_enabled = false;
        } else if (acc == "full") {      // every pass accumulates every frame (no seeded first pass, no incremental ones)
            s->incr_enabled = false;
        } else if (acc != "auto") {
            return e2vq_set_error("ECOZ2_VQ_ACCUMULATE=%s: auto, sorted, sweep, records, burst or full", acc.c_str());
        }
    }
    // With the recorded accumulate the prefiltered pass also wins at M = 128 (0.36 vs 0.43 ms per pass on 2^21 frames; not
    // at 64: 0.30 vs 0.28)
    if (s->rec_enabled && e2vq::prefilter_lds_stage(s->NC) && !getenv("ECOZ2_VQ_PREFILTER_MIN_M")) s->pre_min_M = 128;
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
    void* ptrs[] = {s->d_cbT, s->sv.refl, s->sv.rows, s->sv.rows_local, s->sv.cells, s->d_rows_parent, s->d_fam, s->d_aos, s->d_refl_spec, s->d_cbq_spec, s->d_cbm_spec, s->d_l1max_spec, s->d_cbm, s->d_blk,   s->d_refl,  s->d_refl_next, s->d_cbq,  s->d_l1max, s->d_sc,   s->d_maxabs,
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

int e2vq_reduce(e2vq_session* s, void* buf, i64 count, int op)
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
int e2vq_set_frames_device_impl(e2vq_session* s, const void* device_frames, int64_t T, bool* adopt)
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
        // simply keeps to the plain FP64 sweep (same results): e2vq_use_prefilter() looks at d_fimg.
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
    return e2vq_set_frames_device_impl(s, device_frames, T, nullptr);
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
    // one exchange for both maxima: max |x|, and -- the two status words as one unsigned 64-bit word right behind it -- bad
    // data in ANY shard: every rank learns of it and all of them stop below together, instead of one rank leaving the others
    // to a collective it will never join (round 6: one call instead of two; the sums further down need the scale this
    // maximum defines, so they remain a second one)
    if (e2vq_reduce(s, s->d_maxabs, 2, 1)) return 1;
    e2vq::launch_finish_scalars(s->d_maxabs, s->d_sc, s->stream);
    HIPCHK(hipMemsetAsync(s->d_stats, 0, (size_t)(2 * s->NC + 3) * 8, s->stream));
    e2vq::launch_global_sums(s->d_blk, s->nblocks, s->NC, s->FB, s->d_sc, s->d_stats, s->stream);
    const i64 Tl = s->T;
    HIPCHK(hipMemcpyAsync(s->d_stats + 2 * s->NC + 2, &Tl, 8, hipMemcpyHostToDevice, s->stream));
    if (e2vq_reduce(s, s->d_stats, 2 * s->NC + 3, 0)) return 1;
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
int e2vq_codebook_prepare(e2vq_session* s, bool redefined, bool grown, int zeroed_with_scale)
{
    if (redefined) s->incr_valid = false;
    if (!grown) s->fam_pending = false;  // (set / init: whatever e2vq_grow stashed belongs to another codebook)
    if (redefined && !grown) {
        // a codebook from outside (set / init / restore): the sorted list groups the frames by the cells of ANOTHER codebook --
        // still a permutation, so a pass over it would be exact, but its two-stage sweep would flag most tiles -- and what
        // the flagged share of some earlier level said about two stages says nothing about this one
        s->perm_M = 0;
        s->two_stage_off_until_M = 0;
        s->last_flagged_frac = -1.0;
        s->pre_off_from_M = 0;
    }
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
    if (e2vq_ensure_codebook_capacity(s, M)) return 1;
    HIPCHK(hipMemcpyAsync(s->d_refl, reflections, (size_t)M * s->NC * 8, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->M = M;
    return e2vq_codebook_prepare(s);
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
    if (e2vq_ensure_codebook_capacity(s, 2)) return 1;
    e2vq::launch_init_codebook(s->d_stats, s->NC, s->d_sc, s->d_refl, s->d_flags + 1, s->stream);
    int st = 0;
    HIPCHK(hipMemcpyAsync(&st, s->d_flags + 1, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (st != 0) return e2vq_set_error("Levinson recursion failed on the global centroid (status %d)", st);
    s->M = 1;
    s->DDprv = DBL_MAX / 1e5;  // a fresh ladder: "e+303" in notes.md:128
    return e2vq_codebook_prepare(s);
}

extern "C" int e2vq_grow(e2vq_session* s)
{
    if (s->M < 1) return e2vq_set_error("no codebook to grow");
    if (2 * s->M > 65536) return e2vq_set_error("codebook size limit (u16 symbols) reached");
    HIPCHK(hipSetDevice(s->device));
    // The rows and cells of the last pass over the codebook about to be split seed the first pass of the next size
    // (k_seed_family): they must be this rank's own sums, for the codebook as it stands, with every frame's cell recorded.
    const bool seed = s->fam_enabled && s->pre_enabled && s->d_aos && s->d_prev_sym && s->rows_fresh && s->rows_are_local &&
                      s->cells_M == s->M && 2 * s->M >= s->pre_min_M &&
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
    if (e2vq_ensure_codebook_capacity(s, 2 * s->M)) return 1;
    // one kernel zeroes what the kernels behind it accumulate into with atomicMax: the L1 maximum of the grown codebook and --
    // when its first pass will be a prefiltered one -- the scalars of the limb image that pass builds
    const int Mold = s->M;
    s->M = 2 * Mold;
    const bool fused = e2vq::has_cell_update(s->NC);
    const bool pre_next = fused && s->d_ea && s->d_ps2[s->img_cur] && e2vq_use_prefilter(s, e2vq_pass_mode(s));
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
    if (e2vq_codebook_prepare(s, true, /*grown=*/true, fused ? (pre_next ? s->img_cur : -2) : -1)) return 1;
    s->fam_pending = seed;
    return 0;
}

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
    if (e2vq_ensure_codebook_capacity(s, v.M)) return 1;
    HIPCHK(hipMemcpyAsync(s->d_refl, v.refl, (size_t)v.M * s->NC * 8, hipMemcpyDeviceToDevice, s->stream));
    s->M = v.M;
    if (e2vq_codebook_prepare(s)) return 1;  // (codeword images of the restored codebook; drops every derived flag)
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
            if (e2vq_pass_stats_impl(s, &ls, /*wait_failed=*/false)) return 1;
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
    if (e2vq_resolve_failed_cells(s)) return 1;
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
