// vq_pass.cpp -- one LBG iteration of a session: e2vq_pass (which kernels serve the pass, their launches, the all-reduce),
// the statistics the update kernel publishes to the host, the speculative centroid update and its commit.
#include "vq_session.h"

// ---- LBG iteration pieces ------------------------------------------------------------------

int e2vq_pass_mode(const e2vq_session* s)
{
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
int e2vq_fold_pending_timing(e2vq_session* s)
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
bool e2vq_use_prefilter(const e2vq_session* s, int mode)
{
    if (!(s->pre_enabled && s->d_fimg && (mode == 1 || mode == 2 || mode == 5 || mode == 0) && s->M >= s->pre_min_M &&
          e2vq::prefilter_supports(s->NC, s->M)))
        return false;
    if (mode == 0) return true;
    if (s->pre_off_from_M && s->M >= s->pre_off_from_M) return false;  // (too many uncertified frames on this data: round 6)
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
        if (e2vq_fold_pending_timing(s)) return 1;
    }
    const int mode = e2vq_pass_mode(s);
    s->last_prefiltered = e2vq_use_prefilter(s, mode);
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
    s->last_first_of_level = !incremental;  // (the pass right behind a split: twin codewords, more uncertified frames)
    // round 5: the frames are grouped by cell (a seeded first pass, or an incremental one) -> the fused sorted pass: sweep,
    // exact evaluation, outputs and the cell sums reduced in the block, one kernel (vq_sweep.hip); no records
    // (from sweep_min_M codewords on: at M = 128 a pass is bound by reading its frames, which round 4's kernel does in
    // their natural order -- 0.35 against 0.40 ms on 2^21 frames; at 256 the two are level, beyond it the sorted pass wins)
    // round 6: ... where its two-stage sweep pays.  On data that flags most tiles (a level's first sorted pass measures it:
    // two_stage_off_until_M) the frames gain nothing from being grouped, and round 4's kernel -- frames in their natural
    // order, read sequentially -- is the faster one-stage pass: 1.05 against 1.2-1.3 ms at M = 1024 on 2^21 frames
    // (bench.py config.robustness).  ECOZ2_VQ_ACCUMULATE=sorted keeps the sorted pass, one stage or two (tests).
    const bool two_ok = s->M >= 256 && s->M > s->two_stage_off_until_M;
    const bool fused = keep && mode != 0 && (family || incremental) && s->sweep2_enabled && s->fused_enabled && s->d_fimgF &&
                       s->d_aos && s->M >= s->sweep_min_M && e2vq::sweep_fused_supported(s->NC, s->M) &&
                       (two_ok || s->two_blocks_always);
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
        } else if (records && !(s->sweep2_enabled && s->d_fimgF && s->M >= s->sweep_min_M) && s->rec_few_div > 0 && e2vq::prefilter_burst_supported(s->NC) &&
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
    if (s->last_prefiltered && ensure_codebook_image(s)) return 1;
    {
        // one prologue launch: the rows (all of them, or the distortion columns of an incremental pass), the fallback
        // count of a codebook image that is already there, and what the speculative update after this pass
        // accumulates into with atomicMax (the shadow codebook's L1 max and the scalars of its limb image)
        e2vq::ZeroList z{};
        int nz = 0;
        if (s->last_prefiltered && s->img_valid[s->img_cur]) {
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
    const bool record_cells = !s->last_prefiltered && mode != 0 && s->fam_enabled && s->pre_enabled &&
                              s->d_prev_sym && s->d_aos && !device_sym && 2 * s->M >= s->pre_min_M &&
                              e2vq::prefilter_supports(s->NC, 2 * s->M);
    // (the FP64 sweep of the uncertified frames: in the plain pass's shape when the pass before left more than 2 % of the
    // frames to it -- the list's length is not known on the host when the kernels are enqueued, the last one's is)
    const bool long_list = s->last_fb >= 0 && (double)s->last_fb > 0.02 * (double)s->T;
    if (s->last_prefiltered) {
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
            const bool resort = incr == 2 || s->perm_M != s->M;
            if (resort) {
                if (e2vq::launch_sort_by_cell(s->d_prev_sym, s->T, s->nblocks, incr == 2 ? s->M / 2 : s->M, s->d_sort, s->d_perm,
                                              s->stream))
                    return e2vq_set_error("sort by cell: unsupported size");
                s->perm_M = s->M;
            }
            // (two stages need tiles to skip: from eight tiles on; below, the home tile alone is a quarter or half of the codebook)
            const bool two = two_ok;
            // the host looks at the flagged fraction once per grouping of the frames: on the first pass behind every sort (a
            // level's seeded first pass; the first incremental pass over a codebook that was set or restored from outside).
            // The other two-stage passes add their jobs to a second pair of words (e2vq_sweep_executed reads them back).
            const bool count = two && resort;
            s->last_kind = 3;
            s->n_sweep_launches++;
            s->last_two_stage = two;
            if (resort) s->last_flagged_frac = -1.0;
            if (s->timing) HIPCHK(hipEventRecord(s->ev0, s->stream));  // (again: behind the sort)
            // (a turn of the kernel's loop takes two blocks of 64 slots where the shard has at least two per wave of the grid --
            // 256 workgroups of 8 waves --, one where it has not; ECOZ2_VQ_ACCUMULATE=sorted: two, whatever the size)
            const bool one_block = !s->two_blocks_always && s->nblocks < 2 * 256 * 8;
            if (e2vq::launch_pass_sorted(s->NC, two, one_block, s->d_fimgF, s->d_perm, s->T, s->nblocks, d_cimg, d_ps, s->d_cbq, s->M, s->d_aos,
                                         s->d_sc, s->d_l1max, (unsigned short*)device_sym, (double*)device_dmin, rows,
                                         family ? s->d_fam : nullptr, s->d_fblist, s->d_prev_sym, incr,
                                         count ? e2vq::sweep_counters_of(s->d_sort) : (two ? e2vq::sweep_totals_of(s->d_sort) : nullptr), s->stream))
                return e2vq_set_error("fused sorted pass: unsupported configuration");
            if (!two) s->sw_one_stage_jobs += 2ull * (u64)(s->M / 32) * (u64)s->nblocks;  // (every job with all its k-steps)
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
                                       incr, s->stream, false, nullptr, long_list);
            if (family) e2vq::launch_family_fixup(rows, s->d_fam, s->fam_M, s->NC, s->stream);
        } else if (records && s->sweep2_enabled && s->d_fimgF && s->M >= s->sweep_min_M && e2vq::sweep_supported(s->NC, s->M) &&
                   (two_ok || s->two_blocks_always || !s->fused_enabled || !(family || incremental))) {
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
            const bool two = sorted && two_ok;
            s->last_kind = 2;
            s->n_sweep_launches++;
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
                                       incr, s->stream, false, nullptr, long_list);
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
                                   family ? 2 : (incremental ? 1 : 0), s->stream, false, nullptr, long_list);
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
    if (e2vq_reduce(s, s->d_rows, (i64)s->M * s->RS, 0)) return 1;
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

// (tile, 32-frame column block) jobs of the two-stage sweeps since the last reset: how many there were and how many ran
// stage 2 -- every fused sorted pass counts (one atomic per wave), so that the k-steps a timed region EXECUTED can be stated:
// (jobs * 8 + flagged * 15 + one_stage_jobs * 15) MFMAs at P = 36; one_stage_jobs: the jobs of the fused passes that ran
// without a coarse stage.  Synchronises the stream.
extern "C" int e2vq_sweep_executed(e2vq_session* s, int64_t* flagged, int64_t* jobs, int64_t* one_stage_jobs, int reset)
{
    HIPCHK(hipSetDevice(s->device));
    u64 dev[2] = {0, 0};
    if (s->d_sort) {
        HIPCHK(hipMemcpyAsync(dev, e2vq::sweep_totals_of(s->d_sort), sizeof dev, hipMemcpyDeviceToHost, s->stream));
        if (reset) HIPCHK(hipMemsetAsync(e2vq::sweep_totals_of(s->d_sort), 0, sizeof dev, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    if (flagged) *flagged = (int64_t)(dev[0] + s->sw_host_flagged);
    if (jobs) *jobs = (int64_t)(dev[1] + s->sw_host_jobs);
    if (one_stage_jobs) *one_stage_jobs = (int64_t)s->sw_one_stage_jobs;
    if (reset) s->sw_host_flagged = s->sw_host_jobs = s->sw_one_stage_jobs = 0;
    return 0;
}

// The two switches the host makes from what a pass measured (both only choose kernels: results are the same bits either way).
//   two_stage_max_fraction  flagged share of a level's first sorted pass above which the rest of the level runs without the
//                           coarse stage (0 = never two stages beyond the measuring pass, 1 = always)
//   max_uncertified_fraction share of a pass's frames the prefiltered sweep may leave to the FP64 fallback sweep before the
//                           plain FP64 sweep takes over from that codebook size on (1 = never)
// A negative argument leaves the value as it is.  Also clears what earlier passes of this session decided.
extern "C" int e2vq_set_sweep_policy(e2vq_session* s, double two_stage_max_fraction, double max_uncertified_fraction)
{
    if (two_stage_max_fraction >= 0.0) s->two_stage_max_frac = two_stage_max_fraction;
    if (max_uncertified_fraction >= 0.0) s->pre_max_uncertified = max_uncertified_fraction;
    s->two_stage_off_until_M = 0;
    s->pre_off_from_M = 0;
    return 0;
}

// what the last passes decided: *one_stage_until_M = codebook sizes up to this run the sorted pass without a coarse stage
// (0: none), *plain_from_M = training passes from this codebook size on run the plain FP64 sweep (0: none),
// *uncertified = frames the last prefiltered pass left to the fallback sweep (-1: the last pass was not prefiltered)
extern "C" int e2vq_sweep_policy_state(e2vq_session* s, int* one_stage_until_M, int* plain_from_M, int64_t* uncertified)
{
    if (one_stage_until_M) *one_stage_until_M = s->two_stage_off_until_M;
    if (plain_from_M) *plain_from_M = s->pre_off_from_M;
    if (uncertified) *uncertified = s->last_fb;
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

extern "C" int e2vq_launch_counts_by_kernel(e2vq_session* s, int64_t* pass_pre_lds, int64_t* sweep_cand, int64_t* plain)
{
    if (pass_pre_lds) *pass_pre_lds = s->n_pre_launches - s->n_sweep_launches;
    if (sweep_cand) *sweep_cand = s->n_sweep_launches;
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
        if (e2vq_fold_pending_timing(s)) return 1;
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
        if (e2vq_fold_pending_timing(s)) return 1;
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
// expect_s (optional): in -- how long the same wait took for the previous pass; out -- how long this one took.  A wait of more
// than a few milliseconds sleeps between polls, but not through its end when that is known: a sleeping thread wakes 50-100 us
// late (later on a busy host), once per pass -- 1-2 % of a 5.5 ms pass over 2^24 frames.  From 0.85 of the expected time on
// the word is polled again (back to sleeping when the wait turns out much longer: a level's first pass, a larger codebook).
static int spin_for_sequence(e2vq_session* s, volatile u64* word, const char* what, double* expect_s = nullptr)
{
    // No wall-clock limit by default (the wait also covers the sweep queued ahead, which may legitimately take minutes);
    // ECOZ2_VQ_STATS_TIMEOUT_S sets one -- for hosts whose all-reduce hook can leave a collective pending for ever (a peer
    // process that died).  A rank of an in-process group also gives up as soon as the group has failed.
    static const double limit_s = getenv("ECOZ2_VQ_STATS_TIMEOUT_S") ? atof(getenv("ECOZ2_VQ_STATS_TIMEOUT_S")) : 0.0;
    const auto t_start = std::chrono::steady_clock::now();
    const double expect = expect_s ? *expect_s : 0.0;
    bool slow = false;  // after a few milliseconds: sleep between polls instead of burning a core
    for (unsigned long spins = 0; *word != s->stats_seq; ++spins) {
        if (slow) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if ((spins & 0xfff) == 0xfff || slow) {
            if (s->group_failed && *s->group_failed) return e2vq_set_error("%s: another rank of the in-process group failed", what);
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
            const bool near_end = expect > 0.0 && waited > 0.85 * expect - 1e-4 && waited < 1.5 * expect + 5e-3;
            slow = waited > 5e-3 && !near_end;
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
    if (expect_s) *expect_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
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
int e2vq_resolve_failed_cells(e2vq_session* s)
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
int e2vq_pass_stats_impl(e2vq_session* s, e2vq_level_stats* out, bool wait_failed)
{
    HIPCHK(hipSetDevice(s->device));
    if (s->stats_valid) {
        if (wait_failed && e2vq_resolve_failed_cells(s)) return 1;
        if (out) *out = s->last;
        return 0;
    }
    // (the distortion sums in the rows are fixed-point numbers scaled for the codebook the pass ran on: after an update
    // they cannot be read any more)
    if (!s->rows_fresh) return e2vq_set_error("no statistics: e2vq_pass has not run on the current codebook");
    if (e2vq_resolve_failed_cells(s)) return 1;  // (of the pass before: long there)
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
        const bool image = s->d_cimg2[k] && e2vq_use_prefilter(s, e2vq_pass_mode(s)) && s->M <= s->cimg_cap;
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
        pub.fb_count = s->last_prefiltered ? e2vq::prefilter_fallback_count(s->d_ps2[s->img_last]) : nullptr;
        pub.h_fb = (long long*)&dstats->fb;
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
    if (spin_for_sequence(s, &s->h_stats->seq, "statistics kernel", &s->stats_wait_s)) return 1;
    if (s->rec_pending) {  // (stored by the reduce kernel, which ran ahead of the statistics kernel on the same queue)
        s->rec_last_total = s->h_stats->rec_total;
        s->rec_pending = false;
    }
    if (fused) {  // (the publisher copied the pass's fallback count along)
        s->last_fb = s->last_prefiltered ? (i64)s->h_stats->fb : -1;
        // (a level's first pass runs on twins -- every codeword next to its sibling -- and certifies fewer frames than the
        // passes behind it: 0.27 against 0.09 on the continuum data of bench.py; it is given half as much rope again)
        const double limit = s->last_first_of_level ? std::min(1.0, 1.5 * s->pre_max_uncertified) : s->pre_max_uncertified;
        if (s->last_fb >= 0 && s->T > 0 && (double)s->last_fb > limit * (double)s->T &&
            (!s->pre_off_from_M || s->M < s->pre_off_from_M))
            s->pre_off_from_M = s->M;
    }
    if (s->sw_pending) {  // (stored by the finishing kernel of a two-stage sweep, likewise ahead on the queue)
        const u64 fl = s->h_stats->sw_flagged, jobs = s->h_stats->sw_jobs;
        s->sw_pending = false;
        s->sw_host_flagged += fl;
        s->sw_host_jobs += jobs;
        s->last_flagged_frac = jobs ? (double)fl / (double)jobs : -1.0;
        // most tiles flagged: the coarse stage is wasted on this data -- one stage for the rest of this level.  The switch is
        // per codebook: a larger one (the next level: finer cells, more tiles) tries again, and so does any codebook defined
        // from outside (e2vq_codebook_prepare clears it)
        if (jobs && s->last_flagged_frac > s->two_stage_max_frac) s->two_stage_off_until_M = s->M;
        // (nearly every job flagged: the next size will not look different -- its first pass is not spent on finding out; the
        // size after that tries again)
        if (jobs && s->last_flagged_frac > 0.8 && s->last_flagged_frac > s->two_stage_max_frac)
            s->two_stage_off_until_M = (s->last_flagged_frac > 0.98 ? 4 : 2) * s->M;
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
        if (wait_failed && e2vq_resolve_failed_cells(s)) return 1;
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

extern "C" int e2vq_pass_stats(e2vq_session* s, e2vq_level_stats* out) { return e2vq_pass_stats_impl(s, out, true); }

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
    return e2vq_codebook_prepare(s, false);
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
