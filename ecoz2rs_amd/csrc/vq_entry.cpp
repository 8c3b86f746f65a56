// vq_entry.cpp -- the reference's entry points for this path (include/ecoz2_vq.h, part 1): predictor files in, codebooks /
// sequences / reports out, on top of the session API.
#include "vq_group.h"

// ==========================================================================================
// Part 1: the reference's entry points
// ==========================================================================================

int e2vq_env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

const char* e2vq_env_str(const char* name, const char* dflt)
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
// of an ecoz2_vq_quantize).  The process keeps them for its next call instead: up to 512 MB stay in this pool, portable across devices; whatever is pooled when the process ends is left to
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
        return (size_t)512 << 20;
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
    const int rc = e2vq_set_frames_device_impl(s, r.d, T, &adopted);  // (synchronises: the row-major copy can go, unless the session kept it)
    if (adopted) r.d = nullptr;
    lap("re-layout + images");
    return rc;
}

// one rank of a learn: session on `device`, frames [lo, hi) of the training set.
// Every failing path of a group rank marks the group failed, so the other ranks leave their barriers.
// how a rank of an in-process group exchanges its cell sums: the hook, its argument, and the group to mark failed
struct RankCtx {
    E2Group* g = nullptr;
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
        rc = e2vq_learn(s, eps, e2vq_env_int("ECOZ2_VQ_MAX_CODEBOOK_SIZE", 2048), class_name,
                        e2vq_env_str("ECOZ2_VQ_OUT_ROOT", "."), target, cb, nullptr, 0, nullptr);
    lap("LBG ladder (+ files)");
    if (rc && lr && lr->g) e2g_fail(lr->g);
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
    const int dev0 = e2vq_env_int("ECOZ2_VQ_DEVICE", 0);
    int world = e2vq_env_int("ECOZ2_VQ_GPUS", 1);
    if (world < 1) world = 1;
    if ((i64)world > T) world = (int)T;  // every rank needs at least one training vector
    // ECOZ2_VQ_COLLECTIVE = rccl | p2p (default: RCCL when every rank has a device of its own, else the peer-to-peer
    // slice kernel -- RCCL cannot place two ranks of a communicator on one device)
    const std::string coll = e2vq_env_str("ECOZ2_VQ_COLLECTIVE", "");
    if (!coll.empty() && coll != "rccl" && coll != "p2p")
        return e2vq_set_error("ECOZ2_VQ_COLLECTIVE=%s: expected rccl or p2p", coll.c_str());
    if (world == 1 && coll != "rccl")
        return learn_rank(dev0, eps, class_name, base_refl, base_M, ps, 0, T, nullptr, 1, target, cb);

    // ---- in-process group: rank r on device (dev0 + r) % ndev, contiguous frame shards --------------------------
    printf("sharding over %d rank(s) on %d device(s)\n", world, ndev);
    if (world > e2vq::E2VQ_MAX_LOCAL_RANKS) return e2vq_set_error("ECOZ2_VQ_GPUS=%d exceeds %d in-process ranks", world, e2vq::E2VQ_MAX_LOCAL_RANKS);
    std::vector<int> devs((size_t)world);
    for (int r = 0; r < world; ++r) devs[(size_t)r] = (dev0 + r) % ndev;
    struct Closer {  // (events and communicators go with the group on every return path)
        E2Group* g;
        ~Closer() { e2g_destroy(g); }
    } G{e2g_create(world, devs.data(), coll, true)};
    if (!G.g) return 1;
    const bool use_rccl = e2g_uses_rccl(G.g);
    std::vector<RankCtx> ctx((size_t)world);
    for (int r = 0; r < world; ++r) {
        ctx[(size_t)r].g = G.g;
        ctx[(size_t)r].rank = r;
        e2g_hook(G.g, r, &ctx[(size_t)r].fn, &ctx[(size_t)r].user, &ctx[(size_t)r].force);
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
            rcs[r] = learn_rank(e2g_device(G.g, r), eps, class_name, base_refl, base_M, ps, lo, hi, &ctx[r], world, nullptr, nullptr);
        });
    }
    {  // rank 0 runs on the calling thread: files, messages and the callback come from here
        i64 lo, hi;
        shard(0, &lo, &hi);
        rcs[0] = learn_rank(e2g_device(G.g, 0), eps, class_name, base_refl, base_M, ps, lo, hi, &ctx[0], world, target, cb);
    }
    for (auto& t : th) t.join();
    if (use_rccl && !getenv("ECOZ2_VQ_QUIET")) {
        long calls = 0, bytes = 0;
        e2g_rccl_traffic(G.g, 0, &calls, &bytes);
        printf("collective: rank 0 made %ld ncclAllReduce call(s), %ld bytes\n", calls, bytes);
    }
    for (int rc : rcs)
        if (rc) {
            // the message of the rank that failed FIRST (the others only report the broken barrier)
            if (!e2g_first_error(G.g).empty()) snprintf(e2vq_err_buf(), 1024, "%s", e2g_first_error(G.g).c_str());
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
    // The reference hands this name over as `base_codebook.as_ptr()` of a Rust String (src/ecoz2_lib/mod.rs:295): bytes without
    // a terminator of their own -- what a C reader finds behind them is whatever the heap holds up to the next NUL.  The names
    // this path produces and consumes end in ".cbook" (src/vq/mod.rs:27-33: `-B <codebook>`): when the string as given names no
    // file but a prefix ending in ".cbook" does, that prefix is the name.
    std::string base_name(base_codebook);
    {
        FILE* probe = fopen(base_name.c_str(), "rb");
        if (probe) {
            fclose(probe);
        } else {
            const size_t at = base_name.find(".cbook");
            if (at != std::string::npos && at + 6 < base_name.size()) base_name.resize(at + 6);
        }
    }
    base_codebook = base_name.c_str();
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
    const int dev0 = e2vq_env_int("ECOZ2_VQ_DEVICE", 0);
    const char* root = e2vq_env_str("ECOZ2_VQ_OUT_ROOT", ".");
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
    sh.chunk = std::max(1024, e2vq_env_int("ECOZ2_VQ_QUANTIZE_CHUNK", 1 << 17));
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
    int W = std::max(1, e2vq_env_int("ECOZ2_VQ_GPUS", 1));
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
        if (rcs[(size_t)w]) errs[(size_t)w] = e2vq_err_buf();
    };
    for (int w = 1; w < W; ++w) th.emplace_back(run, w);
    run(0);
    for (auto& t : th) t.join();
    for (int w = 0; w < W; ++w)
        if (rcs[(size_t)w]) {
            if (w > 0) snprintf(e2vq_err_buf(), 1024, "%s", errs[(size_t)w].c_str());
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
    const i64 chunk = std::max(1024, e2vq_env_int("ECOZ2_VQ_QUANTIZE_CHUNK", 1 << 17));
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
    const int device = e2vq_env_int("ECOZ2_VQ_DEVICE", 0);
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
