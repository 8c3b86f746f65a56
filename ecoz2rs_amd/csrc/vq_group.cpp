// vq_group.cpp -- the in-process group of libecoz2vq.so: ranks as host threads of one process, one session each;
// the exchange is the library's peer-to-peer slice kernel or RCCL loaded with dlopen.
#include "vq_group.h"

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
            if (first) first_error = e2vq_err_buf();
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

// (GroupImpl is built in place by group_create: E2Group only gives it a name other files can hold)
E2Group* e2g_create(int world, const int* devices, const std::string& coll, bool verbose)
{
    return reinterpret_cast<E2Group*>(group_create(world, devices, coll, verbose));
}
static GroupImpl* impl_of(E2Group* g) { return reinterpret_cast<GroupImpl*>(g); }
void e2g_destroy(E2Group* g) { delete impl_of(g); }
void e2g_hook(E2Group* g, int r, e2vq_allreduce_fn* fn, void** user, bool* force) { group_hook(impl_of(g), r, fn, user, force); }
void e2g_fail(E2Group* g) { impl_of(g)->g.fail(); }
const volatile bool* e2g_failed_flag(E2Group* g) { return &impl_of(g)->g.failed; }
std::string e2g_first_error(E2Group* g) { return impl_of(g)->g.first_error; }
bool e2g_uses_rccl(E2Group* g) { return impl_of(g)->use_rccl; }
int e2g_world(E2Group* g) { return impl_of(g)->world; }
int e2g_device(E2Group* g, int r) { return impl_of(g)->devs[(size_t)r]; }
const char* e2g_what(E2Group* g) { return impl_of(g)->what.c_str(); }
void e2g_rccl_traffic(E2Group* g, int r, long* calls, long* bytes)
{
    GroupImpl* G = impl_of(g);
    *calls = G->use_rccl ? G->rranks[(size_t)r].calls : 0;
    *bytes = G->use_rccl ? G->rranks[(size_t)r].bytes : 0;
}

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
