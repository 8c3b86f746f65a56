// hmm_host.cpp -- host side of the HMM consumers of the VQ path: the reference's FFI symbols
// ecoz2_set_random_seed / ecoz2_hmm_learn / ecoz2_hmm_classify / ecoz2_hmm_classify_predictors / ecoz2_hmm_show
// (/root/reference/src/ecoz2_lib/mod.rs:75,134-167) over the kernels of hmm_device.hip.  The host loads files,
// draws the initial model, sequences the launches, takes the logarithm of the (mantissa, exponent) pairs the
// kernels return and the stopping decision, and prints the report; every sum over states, time or sequences that
// defines a model or a score runs on the GPU (no CPU fallback: without a HIP device the entry points fail).
// Definitions (file layout, generator, scaled Baum-Welch with exact fixed-point sums, stopping rule): this repo's
// own -- the reference's C bodies are absent -- written down in oracle/hmm_oracle.h and DESIGN.md.
#include "../../include/ecoz2_classify.h"
#include "../../include/ecoz2_vq.h"
#include "hmm_device.h"
#include "host_util.h"
#include "vq_io.h"

#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <string>
#include <thread>
#include <vector>

using namespace e2host;
using e2hmm::ModelDev;
typedef long long i64;

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return e2vq_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

namespace {

int env_device()
{
    const char* v = getenv("ECOZ2_VQ_DEVICE");
    return v && *v ? atoi(v) : 0;
}

// ECOZ2_VQ_GPUS = N: sequences (or predictor files) are dealt in contiguous shares to N workers, worker w on device
// (ECOZ2_VQ_DEVICE + w) % device count -- workers beyond the device count share devices, which is how the 1-GPU tests run
int env_workers()
{
    const char* v = getenv("ECOZ2_VQ_GPUS");
    const int n = v && *v ? atoi(v) : 1;
    return n < 1 ? 1 : (n > 64 ? 64 : n);
}
void share_of(int total, int workers, int w, int* lo, int* hi)
{
    const int base = total / workers, rem = total % workers;
    *lo = w * base + std::min(w, rem);
    *hi = *lo + base + (w < rem ? 1 : 0);
}
int device_of_worker(int w)
{
    int n = 1;
    (void)hipGetDeviceCount(&n);
    const char* v = getenv("ECOZ2_VQ_DEVICE");
    return ((v && *v ? atoi(v) : 0) + w) % std::max(n, 1);
}
// runs fn(w) for w = 0 .. workers - 1, worker 0 on the calling thread; the first failing worker's message is kept
template <typename Fn>
int run_workers(int workers, Fn fn)
{
    std::vector<int> rcs((size_t)workers, 0);
    std::vector<std::string> errs((size_t)workers);
    std::vector<std::thread> th;
    auto body = [&](int w) {
        rcs[(size_t)w] = fn(w);
        if (rcs[(size_t)w]) errs[(size_t)w] = e2vq_last_error();
    };
    for (int w = 1; w < workers; ++w) th.emplace_back(body, w);
    body(0);
    for (auto& t : th) t.join();
    for (int w = 0; w < workers; ++w)
        if (rcs[(size_t)w]) return w == 0 ? rcs[0] : e2vq_set_error("%s", errs[(size_t)w].c_str());
    return 0;
}

int require_device(int device)
{
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return e2vq_set_error("no HIP device available (%s); this library has no CPU path",
                              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return e2vq_set_error("device %d not in [0, %d)", device, n);
    HIPCHK(hipSetDevice(device));
    return 0;
}

// device buffer released on scope exit
template <typename T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t count)
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        HIPCHK(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
        return 0;
    }
    int upload(const T* src, size_t count, hipStream_t st)
    {
        if (alloc(count)) return 1;
        if (count) HIPCHK(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, st));
        return 0;
    }
};

struct Stream {
    hipStream_t s = nullptr;
    ~Stream() { if (s) (void)hipStreamDestroy(s); }
    int create()
    {
        HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        return 0;
    }
};

// ---- generator: ecoz2_set_random_seed (oracle: e2h_set_random_seed / splitmix64) ---------------------------
uint64_t g_rng = 0x9E3779B97F4A7C15ull;

uint64_t rng_next()
{
    uint64_t z = (g_rng += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ---- model ----------------------------------------------------------------------------------------------------
struct Hmm {
    std::string class_name;
    int N = 0, M = 0;
    std::vector<double> pi, A, B;
    void resize(int n, int m)
    {
        N = n;
        M = m;
        pi.assign((size_t)n, 0.0);
        A.assign((size_t)n * n, 0.0);
        B.assign((size_t)n * m, 0.0);
    }
};

// uniform draws in (0, 1] divided by their sequential sum
void random_row(double* row, int n)
{
    double s = 0.0;
    for (int k = 0; k < n; ++k) {
        row[k] = (double)((rng_next() >> 11) + 1) * 0x1.0p-53;
        s = s + row[k];
    }
    for (int k = 0; k < n; ++k) row[k] = row[k] / s;
}

// model types of `hmm learn -t` (src/hmm/mod.rs:49-55): 0 random, 1 uniform, 2 cascade-2, 3 cascade-3 (random B)
int hmm_init(Hmm& h, int type)
{
    const int N = h.N, M = h.M;
    if (type == 0) {
        random_row(h.pi.data(), N);
        for (int i = 0; i < N; ++i) random_row(&h.A[(size_t)i * N], N);
        for (int j = 0; j < N; ++j) random_row(&h.B[(size_t)j * M], M);
    } else if (type == 1) {
        for (int i = 0; i < N; ++i) h.pi[i] = 1.0 / (double)N;
        for (int i = 0; i < N * N; ++i) h.A[i] = 1.0 / (double)N;
        for (size_t k = 0; k < (size_t)N * M; ++k) h.B[k] = 1.0 / (double)M;
    } else if (type == 2 || type == 3) {
        const int width = type == 2 ? 2 : 3;
        for (int i = 0; i < N; ++i) h.pi[i] = i == 0 ? 1.0 : 0.0;
        for (int i = 0; i < N; ++i) {
            const int reach = std::min(N - i, width);
            for (int j = 0; j < N; ++j) h.A[(size_t)i * N + j] = (j >= i && j < i + reach) ? 1.0 / (double)reach : 0.0;
        }
        for (int j = 0; j < N; ++j) random_row(&h.B[(size_t)j * M], M);
    } else {
        return e2vq_set_error("model type %d not in 0..3", type);
    }
    return 0;
}

// .hmm: 16-byte ident "<hmm>", 96-byte class name (src/utl/mod.rs:19-20), u32 N, u32 M, pi, A, B as LE f64
int hmm_save(const std::string& path, const Hmm& h)
{
    std::vector<unsigned char> b(16 + 96 + 8, 0);
    memcpy(b.data(), "<hmm>", 5);
    memcpy(b.data() + 16, h.class_name.data(), std::min<size_t>(h.class_name.size(), 95));
    for (int k = 0; k < 4; ++k) {
        b[112 + k] = (unsigned char)((uint32_t)h.N >> (8 * k));
        b[116 + k] = (unsigned char)((uint32_t)h.M >> (8 * k));
    }
    auto put = [&](const std::vector<double>& v) {
        const unsigned char* p = (const unsigned char*)v.data();
        b.insert(b.end(), p, p + v.size() * 8);  // little-endian host (as the other writers of this library)
    };
    put(h.pi);
    put(h.A);
    put(h.B);
    return write_file(path, b);
}

int hmm_load(const char* path, Hmm& h)
{
    std::vector<unsigned char> raw;
    if (read_file(path, raw)) return 1;
    if (raw.size() < 120 || strncmp((const char*)raw.data(), "<hmm>", 5) != 0) return e2vq_set_error("%s: Not an HMM model", path);
    char cls[97] = {0};
    memcpy(cls, raw.data() + 16, 96);
    h.class_name = cls;
    uint32_t n = 0, m = 0;
    for (int k = 0; k < 4; ++k) {
        n |= (uint32_t)raw[112 + k] << (8 * k);
        m |= (uint32_t)raw[116 + k] << (8 * k);
    }
    if (n < 1 || n > (uint32_t)e2hmm::MAX_N || m < 1 || m > 65536) return e2vq_set_error("%s: implausible N=%u M=%u", path, n, m);
    const size_t need = 120 + ((size_t)n + (size_t)n * n + (size_t)n * m) * 8;
    if (raw.size() != need) return e2vq_set_error("%s: %zu bytes, expected %zu for N=%u M=%u", path, raw.size(), need, n, m);
    h.resize((int)n, (int)m);
    const unsigned char* p = raw.data() + 120;
    memcpy(h.pi.data(), p, h.pi.size() * 8);
    memcpy(h.A.data(), p + h.pi.size() * 8, h.A.size() * 8);
    memcpy(h.B.data(), p + (h.pi.size() + h.A.size()) * 8, h.B.size() * 8);
    return 0;
}

// natural log of mant * 2^exp2 (oracle: e2h_log_prob)
double log_prob(double mant, i64 exp2)
{
    if (!(mant > 0.0)) return -INFINITY;
    return log(mant) + (double)exp2 * M_LN2;
}

// ---- sequences ---------------------------------------------------------------------------------------------------
struct SeqSet {
    std::vector<std::string> files, classes;
    std::vector<uint16_t> sym;  // concatenated
    std::vector<i64> offs;      // S + 1
    int M = -1;                 // codebook size (all equal)
    int S() const { return (int)files.size(); }
};

int load_sequences(const char* const* files, unsigned n, SeqSet& ss)
{
    ss.offs.assign(1, 0);
    for (unsigned i = 0; i < n; ++i) {
        char cls[96];
        int M;
        int64_t T;
        if (e2vq_seq_info(files[i], cls, &M, &T)) return 1;
        if (ss.M < 0) ss.M = M;
        if (M != ss.M) return e2vq_set_error("%s: codebook size %d differs from the first sequence's %d", files[i], M, ss.M);
        const size_t at = ss.sym.size();
        ss.sym.resize(at + (size_t)T);
        if (T > 0 && e2vq_seq_read(files[i], ss.sym.data() + at, T)) return 1;
        ss.files.push_back(files[i]);
        ss.classes.push_back(cls);
        ss.offs.push_back((i64)ss.sym.size());
    }
    return 0;
}

// ---- device-side model set ----------------------------------------------------------------------------------------
struct DevModels {
    DevBuf<double> params;   // all pi | A | B, model after model
    DevBuf<ModelDev> table;
    std::vector<ModelDev> host;
    int maxN = 0;
    int upload(const std::vector<const Hmm*>& ms, hipStream_t st)
    {
        size_t total = 0;
        for (const Hmm* h : ms) total += h->pi.size() + h->A.size() + h->B.size();
        std::vector<double> flat;
        flat.reserve(total);
        std::vector<size_t> at;
        for (const Hmm* h : ms) {
            at.push_back(flat.size());
            flat.insert(flat.end(), h->pi.begin(), h->pi.end());
            flat.insert(flat.end(), h->A.begin(), h->A.end());
            flat.insert(flat.end(), h->B.begin(), h->B.end());
        }
        if (params.upload(flat.data(), flat.size(), st)) return 1;
        HIPCHK(hipStreamSynchronize(st));  // `flat` is a local
        host.clear();
        maxN = 0;
        for (size_t k = 0; k < ms.size(); ++k) {
            const double* base = params.p + at[k];
            host.push_back(ModelDev{ms[k]->N, ms[k]->M, base, base + ms[k]->N, base + ms[k]->N + (size_t)ms[k]->N * ms[k]->N});
            maxN = std::max(maxN, ms[k]->N);
        }
        if (table.upload(host.data(), host.size(), st)) return 1;
        HIPCHK(hipStreamSynchronize(st));
        return 0;
    }
};

// scores of S device-resident sequences under K models: log_probs[s * K + k] (natural log; -inf when the model cannot
// emit the sequence or a symbol is outside its alphabet)
int score_device(const std::vector<const Hmm*>& ms, const unsigned short* d_sym, const i64* d_offs, int S, hipStream_t st,
                 std::vector<double>& log_probs, std::vector<double>* mant_out = nullptr, std::vector<i64>* exp_out = nullptr,
                 std::vector<int>* status_out = nullptr)
{
    const int K = (int)ms.size();
    DevModels dm;
    if (dm.upload(ms, st)) return 1;
    DevBuf<double> d_mant;
    DevBuf<i64> d_exp;
    DevBuf<int> d_st;
    const size_t n = (size_t)S * K;
    if (d_mant.alloc(n) || d_exp.alloc(n) || d_st.alloc(n)) return 1;
    e2hmm::launch_score(dm.table.p, K, dm.maxN, d_sym, d_offs, S, d_mant.p, d_exp.p, d_st.p, st);
    HIPCHK(hipGetLastError());
    std::vector<double> mant(n);
    std::vector<i64> ex(n);
    std::vector<int> stat(n);
    if (n) {
        HIPCHK(hipMemcpyAsync(mant.data(), d_mant.p, n * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(ex.data(), d_exp.p, n * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(stat.data(), d_st.p, n * 4, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    log_probs.resize(n);
    for (size_t i = 0; i < n; ++i) log_probs[i] = stat[i] == 0 ? log_prob(mant[i], ex[i]) : -INFINITY;
    if (mant_out) *mant_out = mant;
    if (exp_out) *exp_out = ex;
    if (status_out) *status_out = stat;
    return 0;
}

// ---- Baum-Welch driver over device-resident sequences -----------------------------------------------------------
struct Trainer {
    int N, M, S;
    i64 total;
    hipStream_t st;
    DevBuf<double> d_params, d_alpha, d_c, d_mant;
    DevBuf<i64> d_acc, d_exp, d_scratch;
    DevBuf<int> d_status;
    std::vector<double> mant;
    std::vector<i64> ex;
    std::vector<int> stat;
    ModelDev md{};
    i64 W = 0;

    int setup(const Hmm& h, int S_, i64 total_, hipStream_t st_)
    {
        N = h.N; M = h.M; S = S_; total = total_; st = st_;
        W = e2hmm::acc_words(N, M);
        std::vector<double> flat;
        flat.insert(flat.end(), h.pi.begin(), h.pi.end());
        flat.insert(flat.end(), h.A.begin(), h.A.end());
        flat.insert(flat.end(), h.B.begin(), h.B.end());
        if (d_params.upload(flat.data(), flat.size(), st)) return 1;
        HIPCHK(hipStreamSynchronize(st));
        double* base = d_params.p;
        md = ModelDev{N, M, base, base + N, base + N + (size_t)N * N};
        if (d_alpha.alloc((size_t)total * N) || d_c.alloc((size_t)total) || d_acc.alloc((size_t)W) || d_mant.alloc((size_t)S) ||
            d_exp.alloc((size_t)S) || d_status.alloc((size_t)S))
            return 1;
        if (e2hmm::fb_scratch_words(N) > 0 && d_scratch.alloc((size_t)e2hmm::fb_scratch_words(N))) return 1;
        mant.resize((size_t)S);
        ex.resize((size_t)S);
        stat.resize((size_t)S);
        return 0;
    }
    // E-step, the device part: expected counts of this trainer's sequences into d_acc, P(O) of each into mant / ex / stat.
    // acc_out (optional): the count words copied to the host (several workers: summed there and handed back)
    int estep_counts(const unsigned short* d_sym, const i64* d_offs, std::vector<i64>* acc_out = nullptr)
    {
        HIPCHK(hipMemsetAsync(d_acc.p, 0, (size_t)W * 8, st));
        e2hmm::launch_fb(md, d_sym, d_offs, S, d_alpha.p, d_c.p, d_acc.p, d_mant.p, d_exp.p, d_status.p, st, d_scratch.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(mant.data(), d_mant.p, (size_t)S * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(ex.data(), d_exp.p, (size_t)S * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(stat.data(), d_status.p, (size_t)S * 4, hipMemcpyDeviceToHost, st));
        if (acc_out) {
            acc_out->resize((size_t)W);
            HIPCHK(hipMemcpyAsync(acc_out->data(), d_acc.p, (size_t)W * 8, hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipStreamSynchronize(st));
        return 0;
    }
    int acc_upload(const std::vector<i64>& acc)
    {
        HIPCHK(hipMemcpyAsync(d_acc.p, acc.data(), (size_t)W * 8, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        return 0;
    }
    // E-step: expected counts into d_acc; returns L = sequential sum of log P over the used sequences
    int estep(const unsigned short* d_sym, const i64* d_offs, double* L, i64* used, i64* skipped)
    {
        if (estep_counts(d_sym, d_offs)) return 1;
        double sum = 0.0;
        i64 u = 0;
        for (int s = 0; s < S; ++s)
            if (stat[(size_t)s] == 0) {
                sum = sum + log_prob(mant[(size_t)s], ex[(size_t)s]);
                ++u;
            }
        *L = sum;
        if (used) *used = u;
        if (skipped) *skipped = S - u;
        return 0;
    }
    int mstep(double epsilon)
    {
        double* base = d_params.p;
        e2hmm::launch_reestimate(N, M, d_acc.p, epsilon, base, base + N, base + N + (size_t)N * N, st);
        HIPCHK(hipGetLastError());
        return 0;
    }
    int download(Hmm& h)
    {
        std::vector<double> flat(h.pi.size() + h.A.size() + h.B.size());
        HIPCHK(hipMemcpyAsync(flat.data(), d_params.p, flat.size() * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        std::copy(flat.begin(), flat.begin() + h.pi.size(), h.pi.begin());
        std::copy(flat.begin() + h.pi.size(), flat.begin() + h.pi.size() + h.A.size(), h.A.begin());
        std::copy(flat.begin() + h.pi.size() + h.A.size(), flat.end(), h.B.begin());
        return 0;
    }
};

typedef void (*hmm_learn_callback_t)(char* variable, double value);
constexpr int MAX_ESTEPS = 1000;  // safety cap, same in the oracle (val_auto <= 0 with no iteration limit would never stop)

// the training loop of oracle/hmm_oracle.h (e2h_learn): returns the number of E-steps through *n_esteps
// The same loop with the sequences dealt to `workers` devices (independent sequences; the expected counts
// are exact int64 limb sums, so their sum over the workers -- taken on the host here: acc_words(N, M) words, 50 KB at
// N = 6, M = 1024 -- is the count of a single worker bit for bit).  Every worker then runs the M-step on the summed
// counts: identical parameters everywhere, no broadcast.  L is summed on the host over all sequences in sequence order.
int train_sharded(Hmm& h, const SeqSet& ss, double epsilon, double val_auto, int max_iterations,
                  hmm_learn_callback_t callback, std::vector<double>& hist, bool verbose, int workers)
{
    struct Worker {
        int device = 0, s0 = 0, s1 = 0;
        Stream st;
        DevBuf<unsigned short> d_sym;
        DevBuf<i64> d_offs;
        Trainer tr;
        std::vector<i64> acc;
    };
    std::vector<Worker> ws((size_t)workers);
    if (run_workers(workers, [&](int w) -> int {
            Worker& k = ws[(size_t)w];
            k.device = device_of_worker(w);
            share_of(ss.S(), workers, w, &k.s0, &k.s1);
            if (require_device(k.device) || k.st.create()) return 1;
            const i64 a = ss.offs[(size_t)k.s0], b = ss.offs[(size_t)k.s1];
            std::vector<i64> offs;
            for (int i = k.s0; i <= k.s1; ++i) offs.push_back(ss.offs[(size_t)i] - a);
            if (k.d_sym.upload(ss.sym.data() + a, (size_t)(b - a), k.st.s) || k.d_offs.upload(offs.data(), offs.size(), k.st.s)) return 1;
            HIPCHK(hipStreamSynchronize(k.st.s));  // (`offs` is a local)
            return k.tr.setup(h, k.s1 - k.s0, b - a, k.st.s);
        }))
        return 1;
    static char var[] = "sum_log_prob";
    int it = 0;
    double Lprev = 0.0;
    hist.clear();
    std::vector<i64> total;
    for (;;) {
        if ((max_iterations >= 0 && it >= max_iterations) || it >= MAX_ESTEPS) break;
        if (run_workers(workers, [&](int w) -> int {
                Worker& k = ws[(size_t)w];
                HIPCHK(hipSetDevice(k.device));
                return k.tr.estep_counts(k.d_sym.p, k.d_offs.p, &k.acc);
            }))
            return 1;
        total.assign(ws[0].acc.size(), 0);
        double L = 0.0;
        i64 skipped = 0;
        for (const Worker& k : ws) {
            for (size_t i = 0; i < total.size(); ++i) total[i] = (i64)((unsigned long long)total[i] + (unsigned long long)k.acc[i]);
            for (int q = 0; q < k.s1 - k.s0; ++q) {
                if (k.tr.stat[(size_t)q] == 0)
                    L = L + log_prob(k.tr.mant[(size_t)q], k.tr.ex[(size_t)q]);
                else
                    ++skipped;
            }
        }
        hist.push_back(L);
        if (verbose)
            printf("  it=%d  sum log(P) = %.10g%s\n", it, L,
                   skipped ? (" (" + std::to_string(skipped) + " sequence(s) the model cannot emit were skipped)").c_str() : "");
        if (callback) callback(var, L);
        if (it > 0 && L - Lprev <= val_auto) {
            ++it;
            break;
        }
        if (run_workers(workers, [&](int w) -> int {
                Worker& k = ws[(size_t)w];
                HIPCHK(hipSetDevice(k.device));
                if (k.tr.acc_upload(total)) return 1;
                if (k.tr.mstep(epsilon)) return 1;
                HIPCHK(hipStreamSynchronize(k.st.s));
                return 0;
            }))
            return 1;
        Lprev = L;
        ++it;
    }
    HIPCHK(hipSetDevice(ws[0].device));
    return ws[0].tr.download(h);
}

int train(Hmm& h, const SeqSet& ss, double epsilon, double val_auto, int max_iterations, hmm_learn_callback_t callback,
          std::vector<double>& hist, bool verbose)
{
    const int workers = std::min(env_workers(), std::max(1, ss.S()));
    if (workers > 1) return train_sharded(h, ss, epsilon, val_auto, max_iterations, callback, hist, verbose, workers);
    Stream st;
    if (st.create()) return 1;
    DevBuf<unsigned short> d_sym;
    DevBuf<i64> d_offs;
    if (d_sym.upload(ss.sym.data(), ss.sym.size(), st.s) || d_offs.upload(ss.offs.data(), ss.offs.size(), st.s)) return 1;
    Trainer tr;
    if (tr.setup(h, ss.S(), (i64)ss.sym.size(), st.s)) return 1;
    static char var[] = "sum_log_prob";
    int it = 0;
    double Lprev = 0.0;
    hist.clear();
    for (;;) {
        if ((max_iterations >= 0 && it >= max_iterations) || it >= MAX_ESTEPS) break;
        double L;
        i64 used, skipped;
        if (tr.estep(d_sym.p, d_offs.p, &L, &used, &skipped)) return 1;
        hist.push_back(L);
        if (verbose)
            printf("  it=%d  sum log(P) = %.10g%s\n", it, L,
                   skipped ? (" (" + std::to_string(skipped) + " sequence(s) the model cannot emit were skipped)").c_str() : "");
        if (callback) callback(var, L);
        if (it > 0 && L - Lprev <= val_auto) {
            ++it;
            break;
        }
        if (tr.mstep(epsilon)) return 1;
        Lprev = L;
        ++it;
    }
    return tr.download(h);
}

std::string fmt_g(double v)
{
    char b[64];
    snprintf(b, sizeof b, "%g", v);
    return b;
}

// classification report shared by ecoz2_hmm_classify / ecoz2_hmm_classify_predictors
int classify_report(const std::vector<Hmm>& models, const std::vector<std::string>& case_files,
                    const std::vector<std::string>& case_classes, const std::vector<double>& log_probs, int M,
                    bool show_ranked, const char* c12n_filename)
{
    const size_t K = models.size();
    std::vector<std::string> names;
    for (const Hmm& h : models) names.push_back(h.class_name);
    C12nResults c12n(names);
    std::string csv;
    size_t classified = 0;
    for (size_t s = 0; s < case_files.size(); ++s) {
        const auto it = std::find(names.begin(), names.end(), case_classes[s]);
        if (it == names.end()) continue;  // no model of that class
        const size_t class_id = (size_t)(it - names.begin());
        std::vector<double> probs(log_probs.begin() + (ptrdiff_t)(s * K), log_probs.begin() + (ptrdiff_t)((s + 1) * K));
        c12n.add_case(class_id, case_classes[s], probs, show_ranked,
                      [&] { return std::string("\n") + case_files[s] + ": '" + case_classes[s] + "'"; });
        // rank of the true class, from 1 (CHANGELOG.md:273-284)
        std::vector<std::pair<size_t, double>> ranked;
        for (size_t k = 0; k < K; ++k) ranked.emplace_back(k, probs[k]);
        std::stable_sort(ranked.begin(), ranked.end(), [](const auto& a, const auto& b) { return a.second < b.second; });
        size_t rank = 0;
        for (size_t i = 0; i < K; ++i)
            if (ranked[K - 1 - i].first == class_id) rank = i + 1;
        csv += case_files[s] + "," + case_classes[s] + "," + (rank == 1 ? "*" : "!") + "," + std::to_string(rank) + "\n";
        ++classified;
    }
    printf("\n");
    if (c12n.report_results(names, "", /*c_report=*/true)) return 1;
    if (c12n_filename && *c12n_filename) {
        std::string doc = "# num_models=" + std::to_string(K) + "  M=" + std::to_string(M) + "  num_seqs=" + std::to_string(classified) +
                          "\nseq_filename,seq_class_name,correct,rank\n" + csv;
        if (write_file(c12n_filename, std::vector<unsigned char>(doc.begin(), doc.end()))) return 1;
        printf("%s saved\n", c12n_filename);
    }
    return 0;
}

int load_models(const char* const* files, unsigned n, std::vector<Hmm>& models)
{
    models.resize(n);
    for (unsigned i = 0; i < n; ++i)
        if (hmm_load(files[i], models[i])) return 1;
    return 0;
}

}  // namespace

// ==========================================================================================
// the reference's FFI symbols
// ==========================================================================================

// fn ecoz2_set_random_seed(seed: c_long) -> c_ulong    src/ecoz2_lib/mod.rs:75; negative = time based (src/hmm/mod.rs:73-76)
extern "C" unsigned long ecoz2_set_random_seed(long seed)
{
    const uint64_t s = seed < 0 ? (uint64_t)time(nullptr) : (uint64_t)seed;
    g_rng = s;
    return (unsigned long)s;
}

// fn ecoz2_hmm_learn(N, model_type, sequence_filenames, num_sequences, hmm_epsilon, val_auto, max_iterations, use_par,
//                    callback: extern "C" fn(*mut c_char, c_double))                     src/ecoz2_lib/mod.rs:134-145
// use_par is accepted and ignored (the E-step always runs one wavefront per sequence on the GPU).
extern "C" int ecoz2_hmm_learn(int N, int model_type, const char* const* sequence_filenames, unsigned num_sequences,
                               double hmm_epsilon, double val_auto, int max_iterations, int use_par,
                               hmm_learn_callback_t callback)
{
    FlushStdout flush_on_return;
    (void)use_par;
    if (!sequence_filenames || num_sequences < 1) return e2vq_set_error("ecoz2_hmm_learn: no sequences");
    if (N < 1 || N > e2hmm::MAX_N) return e2vq_set_error("number of states %d not in [1, %d]", N, e2hmm::MAX_N);
    if (require_device(env_device())) return 1;
    SeqSet ss;
    if (load_sequences(sequence_filenames, num_sequences, ss)) return 1;
    // the name of the trained model is taken from the first training sequence (CHANGELOG.md:174-176)
    for (int i = 1; i < ss.S(); ++i)
        if (ss.classes[(size_t)i] != ss.classes[0])
            return e2vq_set_error("conformity error: class_name: %s != %s", ss.classes[0].c_str(), ss.classes[(size_t)i].c_str());
    for (uint16_t v : ss.sym)
        if ((int)v >= ss.M) return e2vq_set_error("symbol %u outside the codebook size %d", v, ss.M);
    i64 maxT = 0;
    for (int i = 0; i < ss.S(); ++i) maxT = std::max(maxT, ss.offs[(size_t)i + 1] - ss.offs[(size_t)i]);
    Hmm h;
    h.class_name = ss.classes[0];
    h.resize(N, ss.M);
    if (hmm_init(h, model_type)) return 1;
    printf("\nHMM learn: class '%s'  N=%d M=%d type=%d  #sequences = %d  max_T=%lld\n", h.class_name.c_str(), N, ss.M,
           model_type, ss.S(), (long long)maxT);
    printf("  epsilon=%g  val_auto=%g  max_iterations=%d\n", hmm_epsilon, val_auto, max_iterations);
    std::vector<double> hist;
    if (train(h, ss, hmm_epsilon, val_auto, max_iterations, callback, hist, getenv("ECOZ2_VQ_QUIET") == nullptr)) return 1;
    // data/hmms/N<N>__M<M>_t<type>__a<val_auto>[_I<max_iterations>]/<class>.hmm   (CHANGELOG.md:460)
    std::string dir = std::string(out_root()) + "/data/hmms/N" + std::to_string(N) + "__M" + std::to_string(ss.M) + "_t" +
                      std::to_string(model_type) + "__a" + fmt_g(val_auto);
    if (max_iterations >= 0) dir += "_I" + std::to_string(max_iterations);
    const std::string path = dir + "/" + h.class_name + ".hmm";
    if (hmm_save(path, h)) return 1;
    // the training measure per iteration (CHANGELOG.md:288 "generates csv with hmm training measure")
    std::string csv = "# class=" + h.class_name + " N=" + std::to_string(N) + " M=" + std::to_string(ss.M) + "\nI,sum_log_prob\n";
    for (size_t i = 0; i < hist.size(); ++i) {
        char b[64];
        snprintf(b, sizeof b, "%zu,%.17g\n", i, hist[i]);
        csv += b;
    }
    if (write_file(dir + "/" + h.class_name + ".csv", std::vector<unsigned char>(csv.begin(), csv.end()))) return 1;
    printf("%zu E-step(s); model saved: %s\n", hist.size(), path.c_str());
    return 0;
}

// fn ecoz2_hmm_classify(model_filenames, num_models, sequence_filenames, num_sequences, show_ranked,
//                       classification_filename)                                       src/ecoz2_lib/mod.rs:147-154
extern "C" int ecoz2_hmm_classify(const char* const* model_filenames, unsigned num_models,
                                  const char* const* sequence_filenames, unsigned num_sequences, int show_ranked,
                                  const char* classification_filename)
{
    FlushStdout flush_on_return;
    if (!model_filenames || num_models < 1 || !sequence_filenames) return e2vq_set_error("ecoz2_hmm_classify: bad arguments");
    if (require_device(env_device())) return 1;
    std::vector<Hmm> models;
    if (load_models(model_filenames, num_models, models)) return 1;
    SeqSet ss;
    if (load_sequences(sequence_filenames, num_sequences, ss)) return 1;
    std::vector<const Hmm*> ms;
    for (const Hmm& h : models) ms.push_back(&h);
    // ECOZ2_VQ_GPUS workers, each scoring a contiguous share of the sequences under every model (independent: the
    // scores are the single worker's bit for bit)
    const int workers = std::min(env_workers(), std::max(1, ss.S()));
    const size_t K = ms.size();
    std::vector<double> lp((size_t)ss.S() * K);
    if (run_workers(workers, [&](int w) -> int {
            int s0, s1;
            share_of(ss.S(), workers, w, &s0, &s1);
            if (require_device(device_of_worker(w))) return 1;
            Stream st;
            if (st.create()) return 1;
            const i64 a = ss.offs[(size_t)s0], b = ss.offs[(size_t)s1];
            std::vector<i64> offs;
            for (int i = s0; i <= s1; ++i) offs.push_back(ss.offs[(size_t)i] - a);
            DevBuf<unsigned short> d_sym;
            DevBuf<i64> d_offs;
            if (d_sym.upload(ss.sym.data() + a, (size_t)(b - a), st.s) || d_offs.upload(offs.data(), offs.size(), st.s)) return 1;
            std::vector<double> part;
            if (score_device(ms, d_sym.p, d_offs.p, s1 - s0, st.s, part)) return 1;
            std::copy(part.begin(), part.end(), lp.begin() + (ptrdiff_t)((size_t)s0 * K));
            return 0;
        }))
        return 1;
    return classify_report(models, ss.files, ss.classes, lp, ss.M < 0 ? models[0].M : ss.M, show_ranked != 0,
                           classification_filename);
}

// fn ecoz2_hmm_classify_predictors(model_filenames, num_models: c_uint, cb_filenames, num_codebooks: c_int,
//        prd_filenames, num_predictors: c_int, show_ranked, classification_filename)   src/ecoz2_lib/mod.rs:156-165
// Every .prd is quantised on the GPU against the codebook of each model (one codebook for all models, or one per
// class, matched by class name) and the symbol sequences are scored where they are: frames in, log-probabilities out.
extern "C" int ecoz2_hmm_classify_predictors(const char* const* model_filenames, unsigned num_models,
                                             const char* const* cb_filenames, int num_codebooks,
                                             const char* const* prd_filenames, int num_predictors, int show_ranked,
                                             const char* classification_filename)
{
    FlushStdout flush_on_return;
    if (!model_filenames || num_models < 1 || !cb_filenames || num_codebooks < 1 || !prd_filenames || num_predictors < 0)
        return e2vq_set_error("ecoz2_hmm_classify_predictors: bad arguments");
    const int device = env_device();
    if (require_device(device)) return 1;
    std::vector<Hmm> models;
    if (load_models(model_filenames, num_models, models)) return 1;
    // codebooks
    struct Cb {
        std::string cls;
        int P = 0, M = 0;
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
    const int P = cbs[0].P;
    // which codebook feeds which model
    std::vector<int> cb_of((size_t)num_models, 0);
    for (unsigned k = 0; k < num_models; ++k) {
        int found = num_codebooks == 1 ? 0 : -1;
        for (int i = 0; i < num_codebooks && found < 0; ++i)
            if (cbs[i].cls == models[k].class_name) found = i;
        if (found < 0) return e2vq_set_error("no codebook of class '%s' for model %s", models[k].class_name.c_str(), model_filenames[k]);
        if (cbs[found].M != models[k].M)
            return e2vq_set_error("%s: model has M=%d but codebook %s has M=%d", model_filenames[k], models[k].M, cb_filenames[found], cbs[found].M);
        cb_of[k] = found;
    }
    // predictors: headers only here (file order = case order); the frames are streamed below
    std::vector<std::string> files, classes;
    std::vector<i64> offs(1, 0);
    i64 max_T = 0;
    for (int f = 0; f < num_predictors; ++f) {
        char cls[96];
        int p;
        int64_t T;
        if (e2vq_prd_info(prd_filenames[f], cls, &p, &T)) return 1;
        if (p != P) return e2vq_set_error("%s: prediction order %d differs from the codebooks' %d", prd_filenames[f], p, P);
        files.push_back(prd_filenames[f]);
        classes.push_back(cls);
        offs.push_back(offs.back() + T);
        max_T = std::max<i64>(max_T, T);
    }
    const i64 total = offs.back();
    const int S = (int)files.size();
    printf("number of HMM models: %u  number of codebooks: %d  number of predictor files: %d (%lld vectors)\n", num_models,
           num_codebooks, S, (long long)total);
    // Bounded memory (round 4): the corpus is cut into UNITS of whole files holding at most CHUNK frames together
    // (ECOZ2_VQ_CLASSIFY_CHUNK, default 2^18 = 78 MB at P = 36; at most 65 536 files), which the ECOZ2_VQ_GPUS workers pull
    // from a shared counter.  A unit's frames go through one of the worker's two pinned slots to the device, are quantised
    // against each codebook in turn and scored at once under that codebook's models: only the unit's symbols are ever
    // resident besides its frames.  Reading unit u + 1 from the files overlaps the device's work on unit u.  A file longer
    // than a chunk is a unit of its own: its frames stream through the slot piece by piece into the symbol buffer (once per
    // codebook), then its one sequence is scored.  Files are independent: the same scores for any chunk size and any
    // number of workers.
    const char* chv = getenv("ECOZ2_VQ_CLASSIFY_CHUNK");
    const i64 CHUNK = std::max<i64>(64, chv && *chv ? atoll(chv) : (1 << 18));
    constexpr int MAX_UNIT_FILES = 65536;
    struct Unit {
        int f0, f1;
    };
    std::vector<Unit> units;
    for (int f = 0; f < S;) {
        int g = f;
        i64 n = 0;
        while (g < S && g - f < MAX_UNIT_FILES && (g == f || n + (offs[(size_t)g + 1] - offs[(size_t)g]) <= CHUNK)) {
            n += offs[(size_t)g + 1] - offs[(size_t)g];
            ++g;
            if (n > CHUNK) break;  // (a single file longer than a chunk)
        }
        units.push_back(Unit{f, g});
        f = g;
    }
    int max_files = 1;
    for (const Unit& u : units) max_files = std::max(max_files, u.f1 - u.f0);
    // which models each codebook feeds
    struct Group {
        int cb;
        std::vector<const Hmm*> ms;
        std::vector<unsigned> idx;
    };
    std::vector<Group> groups;
    for (int c = 0; c < num_codebooks; ++c) {
        Group gr;
        gr.cb = c;
        for (unsigned k = 0; k < num_models; ++k)
            if (cb_of[k] == c) {
                gr.ms.push_back(&models[k]);
                gr.idx.push_back(k);
            }
        if (!gr.ms.empty()) groups.push_back(std::move(gr));
    }
    std::vector<double> lp((size_t)S * num_models, -INFINITY);
    std::atomic<int> next_unit{0};
    std::atomic<bool> failed{false};
    const int workers = std::min(env_workers(), std::max(1, (int)units.size()));
    const int NC = P + 1;
    if (run_workers(workers, [&](int w) -> int {
            // (any way out of this worker but the last line stops the others at their next unit)
            struct FailGuard {
                std::atomic<bool>& f;
                bool ok = false;
                ~FailGuard()
                {
                    if (!ok) f.store(true);
                }
            } fail_guard{failed};
            const int dev = workers == 1 ? device : device_of_worker(w);
            if (require_device(dev)) return 1;
            Stream st;
            if (st.create()) return 1;
            // one quantize session per codebook (its codeword images are built once), all on the worker's stream
            struct Sessions {
                std::vector<e2vq_session*> v;
                ~Sessions()
                {
                    for (e2vq_session* s : v)
                        if (s) e2vq_session_destroy(s);
                }
            } sessions;
            std::vector<DevModels> dms(groups.size());
            for (size_t g = 0; g < groups.size(); ++g) {
                e2vq_session* vq = nullptr;
                if (e2vq_session_create(dev, P, &vq)) return 1;
                sessions.v.push_back(vq);
                if (e2vq_set_stream(vq, (void*)st.s)) return 1;  // quantize and scoring are ordered on one stream
                if (e2vq_set_codebook(vq, cbs[(size_t)groups[g].cb].refl.data(), cbs[(size_t)groups[g].cb].M)) return 1;
                if (dms[g].upload(groups[g].ms, st.s)) return 1;
            }
            struct Slot {
                double* h_frames = nullptr;
                i64* h_offs = nullptr;
                double* h_mant = nullptr;
                i64* h_exp = nullptr;
                int* h_st = nullptr;
                DevBuf<double> d_frames, d_mant;
                DevBuf<unsigned short> d_sym;
                DevBuf<i64> d_offs, d_exp;
                DevBuf<int> d_st;
                hipEvent_t done = nullptr;
                int unit = -1;
                ~Slot()
                {
                    for (void* p : {(void*)h_frames, (void*)h_offs, (void*)h_mant, (void*)h_exp, (void*)h_st})
                        if (p) (void)hipHostFree(p);
                    if (done) (void)hipEventDestroy(done);
                }
            } slots[2];
            const size_t res_cap = (size_t)max_files * num_models;
            const size_t sym_cap = (size_t)std::max<i64>(CHUNK, max_T) + 64;
            // (the staging slots hold a unit's frames: never more than the whole corpus has)
            const i64 STAGE = std::max<i64>(64, std::min<i64>(CHUNK, offs[(size_t)S]));
            for (Slot& q : slots) {
                HIPCHK(hipHostMalloc((void**)&q.h_frames, (size_t)STAGE * NC * 8, hipHostMallocDefault));
                HIPCHK(hipHostMalloc((void**)&q.h_offs, (size_t)(max_files + 1) * 8, hipHostMallocDefault));
                HIPCHK(hipHostMalloc((void**)&q.h_mant, res_cap * 8, hipHostMallocDefault));
                HIPCHK(hipHostMalloc((void**)&q.h_exp, res_cap * 8, hipHostMallocDefault));
                HIPCHK(hipHostMalloc((void**)&q.h_st, res_cap * 4, hipHostMallocDefault));
                if (q.d_frames.alloc((size_t)STAGE * NC) || q.d_sym.alloc(sym_cap) || q.d_offs.alloc((size_t)max_files + 1) ||
                    q.d_mant.alloc(res_cap) || q.d_exp.alloc(res_cap) || q.d_st.alloc(res_cap))
                    return 1;
                HIPCHK(hipEventCreateWithFlags(&q.done, hipEventDisableTiming));
            }
            // results of the unit in flight in a slot -> lp (layout on the device: group after group, [sequence][model of the group])
            auto harvest = [&](Slot& q) -> int {
                if (q.unit < 0) return 0;
                HIPCHK(hipEventSynchronize(q.done));
                const Unit& u = units[(size_t)q.unit];
                const int Su = u.f1 - u.f0;
                size_t base = 0;
                for (const Group& gr : groups) {
                    const size_t K = gr.idx.size();
                    for (int sq = 0; sq < Su; ++sq)
                        for (size_t j = 0; j < K; ++j) {
                            const size_t o = base + (size_t)sq * K + j;
                            lp[(size_t)(u.f0 + sq) * num_models + gr.idx[j]] =
                                q.h_st[o] == 0 ? log_prob(q.h_mant[o], q.h_exp[o]) : -INFINITY;
                        }
                    base += (size_t)Su * K;
                }
                q.unit = -1;
                return 0;
            };
            auto score_groups = [&](Slot& q, int Su, size_t g0, size_t g1, size_t base) -> int {  // groups [g0, g1) on q.d_sym
                for (size_t g = g0; g < g1; ++g) {
                    const int K = (int)groups[g].idx.size();
                    e2hmm::launch_score(dms[g].table.p, K, dms[g].maxN, q.d_sym.p, q.d_offs.p, Su, q.d_mant.p + base,
                                        q.d_exp.p + base, q.d_st.p + base, st.s);
                    HIPCHK(hipGetLastError());
                    base += (size_t)Su * K;
                }
                return 0;
            };
            int turn = 0;
            while (!failed.load()) {
                const int ui = next_unit.fetch_add(1);
                if (ui >= (int)units.size()) break;
                Slot& q = slots[turn & 1];
                ++turn;
                if (harvest(q)) return 1;
                const Unit& u = units[(size_t)ui];
                const int Su = u.f1 - u.f0;
                const i64 n_fr = offs[(size_t)u.f1] - offs[(size_t)u.f0];
                for (int f = u.f0; f <= u.f1; ++f) q.h_offs[f - u.f0] = offs[(size_t)f] - offs[(size_t)u.f0];
                HIPCHK(hipMemcpyAsync(q.d_offs.p, q.h_offs, (size_t)(Su + 1) * 8, hipMemcpyHostToDevice, st.s));
                size_t n_res = 0;
                for (const Group& gr : groups) n_res += (size_t)Su * gr.idx.size();
                if (n_fr <= CHUNK) {
                    for (int f = u.f0; f < u.f1; ++f) {
                        const i64 T = offs[(size_t)f + 1] - offs[(size_t)f];
                        bool fin = true;
                        if (T > 0 && e2vq_io::prd_read_range_mt(files[(size_t)f].c_str(), P, 0, T,
                                                                q.h_frames + (size_t)(offs[(size_t)f] - offs[(size_t)u.f0]) * NC,
                                                                e2vq_io::io_threads(), &fin))
                            return 1;
                        if (!fin) return e2vq_set_error("%s: contains NaN or infinite values", files[(size_t)f].c_str());
                    }
                    if (n_fr > 0) HIPCHK(hipMemcpyAsync(q.d_frames.p, q.h_frames, (size_t)n_fr * NC * 8, hipMemcpyHostToDevice, st.s));
                    size_t base = 0;
                    for (size_t g = 0; g < groups.size(); ++g) {
                        if (n_fr > 0 && e2vq_quantize_device(sessions.v[g], q.d_frames.p, n_fr, q.d_sym.p, nullptr)) return 1;
                        if (score_groups(q, Su, g, g + 1, base)) return 1;
                        base += (size_t)Su * groups[g].idx.size();
                    }
                } else {
                    // one file longer than a chunk: piece by piece into the symbol buffer, once per codebook (synchronous:
                    // the one staging buffer is refilled for every piece)
                    size_t base = 0;
                    for (size_t g = 0; g < groups.size(); ++g) {
                        for (i64 t0 = 0; t0 < n_fr; t0 += CHUNK) {
                            const i64 n = std::min(CHUNK, n_fr - t0);
                            bool fin = true;
                            HIPCHK(hipStreamSynchronize(st.s));  // (the previous piece has left the staging buffer)
                            if (e2vq_io::prd_read_range_mt(files[(size_t)u.f0].c_str(), P, t0, n, q.h_frames, e2vq_io::io_threads(), &fin))
                                return 1;
                            if (!fin) return e2vq_set_error("%s: contains NaN or infinite values", files[(size_t)u.f0].c_str());
                            HIPCHK(hipMemcpyAsync(q.d_frames.p, q.h_frames, (size_t)n * NC * 8, hipMemcpyHostToDevice, st.s));
                            if (e2vq_quantize_device(sessions.v[g], q.d_frames.p, n, q.d_sym.p + t0, nullptr)) return 1;
                        }
                        if (score_groups(q, Su, g, g + 1, base)) return 1;
                        base += (size_t)Su * groups[g].idx.size();
                    }
                }
                if (n_res) {
                    HIPCHK(hipMemcpyAsync(q.h_mant, q.d_mant.p, n_res * 8, hipMemcpyDeviceToHost, st.s));
                    HIPCHK(hipMemcpyAsync(q.h_exp, q.d_exp.p, n_res * 8, hipMemcpyDeviceToHost, st.s));
                    HIPCHK(hipMemcpyAsync(q.h_st, q.d_st.p, n_res * 4, hipMemcpyDeviceToHost, st.s));
                }
                HIPCHK(hipEventRecord(q.done, st.s));
                q.unit = ui;
            }
            for (int k = 0; k < 2; ++k)
                if (harvest(slots[(turn + k) & 1])) return 1;
            HIPCHK(hipStreamSynchronize(st.s));
            fail_guard.ok = !failed.load();
            return 0;
        }))
        return 1;
    return classify_report(models, files, classes, lp, models[0].M, show_ranked != 0, classification_filename);
}

// fn ecoz2_hmm_show(hmm_filename, format)        src/ecoz2_lib/mod.rs:167; default format "%Lg " (src/hmm/mod.rs:153-154)
extern "C" int ecoz2_hmm_show(const char* hmm_filename, const char* format)
{
    FlushStdout flush_on_return;
    Hmm h;
    if (hmm_load(hmm_filename, h)) return 1;
    const std::string fmt = format && *format ? format : "%Lg ";
    // the format is applied to a long double when it asks for one ("%Lg": prob_t was long double originally,
    // notes.md:17-21), to a double otherwise; exactly one conversion is accepted
    size_t pct = 0, convs = 0;
    for (size_t i = 0; i + 1 < fmt.size(); ++i)
        if (fmt[i] == '%') {
            if (fmt[i + 1] == '%') { ++i; continue; }
            ++convs;
            pct = i;
        }
    if (convs != 1) return e2vq_set_error("format '%s' must hold exactly one floating-point conversion", fmt.c_str());
    size_t e = pct + 1;
    while (e < fmt.size() && strchr("-+ #0123456789.", fmt[e])) ++e;
    const bool is_long = e < fmt.size() && fmt[e] == 'L';
    if (is_long) ++e;
    if (e >= fmt.size() || !strchr("eEfFgGaA", fmt[e])) return e2vq_set_error("format '%s' is not a floating-point format", fmt.c_str());
    auto put = [&](double v) {
        if (is_long) printf(fmt.c_str(), (long double)v);
        else printf(fmt.c_str(), v);
    };
    printf("# %s:\n# className='%s', N=%d, M=%d\n", hmm_filename, h.class_name.c_str(), h.N, h.M);
    printf("pi = ");
    for (int i = 0; i < h.N; ++i) put(h.pi[(size_t)i]);
    printf("\nA =\n");
    for (int i = 0; i < h.N; ++i) {
        printf(" [%d]: ", i);
        for (int j = 0; j < h.N; ++j) put(h.A[(size_t)i * h.N + j]);
        printf("\n");
    }
    printf("B =\n");
    for (int j = 0; j < h.N; ++j) {
        printf(" [%d]: ", j);
        for (int k = 0; k < h.M; ++k) put(h.B[(size_t)j * h.M + k]);
        printf("\n");
    }
    return 0;
}

// ==========================================================================================
// array-level entry points (tests, bench, Python mirror): same kernels, no files
// ==========================================================================================
extern "C" int e2vq_hmm_init(int N, int M, int model_type, double* pi, double* A, double* B)
{
    if (N < 1 || N > e2hmm::MAX_N || M < 1 || M > 65536) return e2vq_set_error("e2vq_hmm_init: N=%d M=%d out of range", N, M);
    Hmm h;
    h.resize(N, M);
    if (hmm_init(h, model_type)) return 1;
    memcpy(pi, h.pi.data(), h.pi.size() * 8);
    memcpy(A, h.A.data(), h.A.size() * 8);
    memcpy(B, h.B.data(), h.B.size() * 8);
    return 0;
}

static int model_from_arrays(int N, int M, const double* pi, const double* A, const double* B, Hmm& h)
{
    if (N < 1 || N > e2hmm::MAX_N || M < 1 || M > 65536) return e2vq_set_error("HMM with N=%d M=%d out of range", N, M);
    h.resize(N, M);
    memcpy(h.pi.data(), pi, h.pi.size() * 8);
    memcpy(h.A.data(), A, h.A.size() * 8);
    memcpy(h.B.data(), B, h.B.size() * 8);
    return 0;
}

extern "C" int e2vq_hmm_save(const char* path, const char* class_name, int N, int M, const double* pi, const double* A,
                             const double* B)
{
    Hmm h;
    if (model_from_arrays(N, M, pi, A, B, h)) return 1;
    h.class_name = class_name ? class_name : "";
    return hmm_save(path, h);
}

extern "C" int e2vq_hmm_info(const char* path, char class_name[96], int* N, int* M)
{
    Hmm h;
    if (hmm_load(path, h)) return 1;
    memset(class_name, 0, 96);
    memcpy(class_name, h.class_name.data(), std::min<size_t>(h.class_name.size(), 95));
    *N = h.N;
    *M = h.M;
    return 0;
}

extern "C" int e2vq_hmm_load(const char* path, double* pi, double* A, double* B)
{
    Hmm h;
    if (hmm_load(path, h)) return 1;
    memcpy(pi, h.pi.data(), h.pi.size() * 8);
    memcpy(A, h.A.data(), h.A.size() * 8);
    memcpy(B, h.B.data(), h.B.size() * 8);
    return 0;
}

// scaled forward scores of S host sequences (concatenated symbols + S+1 offsets) under K models given as arrays:
// Ns[k], shared M, pis[k] / As[k] / Bs[k].  Outputs [s * K + k]: P(O) = mant * 2^exp2, status, natural-log probability.
extern "C" int e2vq_hmm_score(int device, int K, const int* Ns, int M, const double* const* pis, const double* const* As,
                              const double* const* Bs, const uint16_t* sym, const int64_t* offs, int S, double* mant,
                              int64_t* exp2, int* status, double* log_probs)
{
    if (K < 1 || S < 0) return e2vq_set_error("e2vq_hmm_score: bad arguments");
    if (require_device(device)) return 1;
    std::vector<Hmm> models((size_t)K);
    std::vector<const Hmm*> ms;
    for (int k = 0; k < K; ++k) {
        if (model_from_arrays(Ns[k], M, pis[k], As[k], Bs[k], models[(size_t)k])) return 1;
        ms.push_back(&models[(size_t)k]);
    }
    Stream st;
    if (st.create()) return 1;
    DevBuf<unsigned short> d_sym;
    DevBuf<i64> d_offs;
    if (d_sym.upload(sym, (size_t)offs[S], st.s) || d_offs.upload((const i64*)offs, (size_t)S + 1, st.s)) return 1;
    std::vector<double> lp, mt;
    std::vector<i64> ex;
    std::vector<int> stt;
    if (score_device(ms, d_sym.p, d_offs.p, S, st.s, lp, &mt, &ex, &stt)) return 1;
    const size_t n = (size_t)S * K;
    for (size_t i = 0; i < n; ++i) {
        if (mant) mant[i] = mt[i];
        if (exp2) exp2[i] = ex[i];
        if (status) status[i] = stt[i];
        if (log_probs) log_probs[i] = lp[i];
    }
    return 0;
}

extern "C" int64_t e2vq_hmm_acc_words(int N, int M) { return e2hmm::acc_words(N, M); }

// one Baum-Welch E-step on the GPU: the exact expected-count accumulators (e2vq_hmm_acc_words int64 words) and the
// per-sequence P(O) / status
extern "C" int e2vq_hmm_estep(int device, int N, int M, const double* pi, const double* A, const double* B,
                              const uint16_t* sym, const int64_t* offs, int S, int64_t* acc, double* mant, int64_t* exp2,
                              int* status)
{
    if (require_device(device)) return 1;
    Hmm h;
    if (model_from_arrays(N, M, pi, A, B, h)) return 1;
    Stream st;
    if (st.create()) return 1;
    DevBuf<unsigned short> d_sym;
    DevBuf<i64> d_offs;
    if (d_sym.upload(sym, (size_t)offs[S], st.s) || d_offs.upload((const i64*)offs, (size_t)S + 1, st.s)) return 1;
    Trainer tr;
    if (tr.setup(h, S, offs[S], st.s)) return 1;
    double L;
    if (tr.estep(d_sym.p, d_offs.p, &L, nullptr, nullptr)) return 1;
    HIPCHK(hipMemcpy(acc, tr.d_acc.p, (size_t)tr.W * 8, hipMemcpyDeviceToHost));
    for (int s = 0; s < S; ++s) {
        if (mant) mant[s] = tr.mant[(size_t)s];
        if (exp2) exp2[s] = tr.ex[(size_t)s];
        if (status) status[s] = tr.stat[(size_t)s];
    }
    return 0;
}

// whole training on arrays (in place): the loop of ecoz2_hmm_learn without files
extern "C" int e2vq_hmm_train(int device, int N, int M, double* pi, double* A, double* B, const uint16_t* sym,
                              const int64_t* offs, int S, double epsilon, double val_auto, int max_iterations,
                              double* sum_log_prob, int cap, int* num_esteps)
{
    if (require_device(device)) return 1;
    Hmm h;
    if (model_from_arrays(N, M, pi, A, B, h)) return 1;
    SeqSet ss;
    ss.M = M;
    ss.sym.assign(sym, sym + offs[S]);
    ss.offs.assign((const i64*)offs, (const i64*)offs + S + 1);
    ss.files.assign((size_t)S, "");
    ss.classes.assign((size_t)S, "");
    std::vector<double> hist;
    if (train(h, ss, epsilon, val_auto, max_iterations, nullptr, hist, false)) return 1;
    memcpy(pi, h.pi.data(), h.pi.size() * 8);
    memcpy(A, h.A.data(), h.A.size() * 8);
    memcpy(B, h.B.data(), h.B.size() * 8);
    for (size_t i = 0; i < hist.size() && (int)i < cap; ++i) sum_log_prob[i] = hist[i];
    if (num_esteps) *num_esteps = (int)hist.size();
    return 0;
}
