// hmm_device.hip -- HIP kernels (gfx950) of the HMM consumers of the VQ path:
//   k_hmm_score   scaled forward pass of every (sequence, model) pair: one wavefront per pair, lane = state
//   k_hmm_fb      E-step of Baum-Welch: forward, backward and the expected counts of one sequence per wavefront,
//                 added to exact fixed-point accumulators (int64 limb sums: order-free, hence deterministic and
//                 shardable across GPUs with an integer all-reduce, like the VQ cell sums)
//   k_hmm_reestimate / k_hmm_adjustb   M-step
// Arithmetic: IEEE f64 multiply / fma / add / divide in the order oracle/hmm_oracle.h defines -- the kernels are
// bit-exact against the oracle.  No transcendental runs on the device: P(O) leaves as (mantissa, exponent) and the
// host takes the logarithm.
//
// Wave layout (N <= 64 states): lane j owns state j.  The sums over the other state index are lane-uniform loops that
// broadcast one lane's value with v_readlane (SGPR operand of the fma) and read the transition matrix from LDS --
// column access A[i][j] with j = lane for the forward pass, and a transposed copy AT[j][i] with i = lane for the
// backward pass, so both are conflict-free.  Each step is a dependent chain (N fmas, N adds, one divide), so the
// parallelism is across waves: S x K pairs for scoring, one wave per training sequence for the E-step.
//
// More than 64 states (the reference's -N is free; N <= 512 here): k_hmm_score_wg / k_hmm_fb_wg give a sequence a whole
// workgroup, thread j = state j; the values of the other states travel through LDS arrays with a barrier per step, every
// thread adds them up itself in the oracle's order (so all threads hold the same bits and take the same branches), A is
// read from global memory, and the xi sums of a workgroup live in a global scratch table whose column j only thread j
// ever touches (no atomics until the final flush).  Same arithmetic, same order, same bits as the wave kernels; built to
// work, not to be fast.
#include "hmm_device.h"
#include "vq_fixed.h"

namespace e2hmm {

typedef long long i64;
typedef unsigned long long u64;

__device__ __forceinline__ double bcast(double x, int lane)
{
    // `lane` is wave-uniform: two v_readlane_b32
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// P = p * 2^E with p in [0.5, 1): one more factor c (frexp is exact; the product rounds once)
__device__ __forceinline__ void scale_step(double c, double& p, i64& E)
{
    int e, e2;
    const double m = frexp(c, &e);
    p = frexp(p * m, &e2);
    E += (i64)e + (i64)e2;
}

constexpr int SCORE_WAVES = 4;

// grid: (ceil(S / SCORE_WAVES), K).  models[k]: N, M, pi, A, B.  out index: s * K + k.
__global__ __launch_bounds__(64 * SCORE_WAVES) void k_hmm_score(const ModelDev* __restrict__ models, int K, int k0,
                                                                 const unsigned short* __restrict__ sym,
                                                                 const i64* __restrict__ offs, int S,
                                                                 double* __restrict__ mant, i64* __restrict__ exp2,
                                                                 int* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* As = (double*)smem;  // [N][N], row i = from-state
    const int k = k0 + (int)blockIdx.y;  // (models beyond 65535 come in further launches)
    const ModelDev md = models[k];
    const int N = md.N, M = md.M;
    for (int x = threadIdx.x; x < N * N; x += blockDim.x) As[x] = md.A[x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int s = blockIdx.x * SCORE_WAVES + wib;
    if (s >= S) return;
    const i64 base = offs[s];
    const i64 T = offs[s + 1] - base;
    const bool act = lane < N;
    const double pij = act ? md.pi[lane] : 0.0;
    const double* Brow = md.B + (size_t)(act ? lane : 0) * M;
    double al = 0.0, p = 0.5;
    i64 E = 1;
    int st = 0;
    for (i64 t0 = 0; t0 < T && st == 0; t0 += 64) {
        // this chunk's symbols: one per lane, handed out by readlane
        const int n = (int)((T - t0) < 64 ? (T - t0) : 64);
        const int mysym = lane < n ? (int)sym[base + t0 + lane] : 0;
        int o = __builtin_amdgcn_readlane(mysym, 0);
        double b = (act && o < M) ? Brow[o] : 0.0;
        for (int q = 0; q < n; ++q) {
            const double bq = b;
            const int oq = o;
            if (q + 1 < n) {  // next step's emission probability is requested before this step's chain runs
                o = __builtin_amdgcn_readlane(mysym, q + 1);
                b = (act && o < M) ? Brow[o] : 0.0;
            }
            if (oq >= M) {  // symbol outside the model's alphabet
                st = 2;
                break;
            }
            double nx;
            if (t0 + q == 0) {
                nx = pij * bq;
            } else {
                double acc = 0.0;
                for (int i = 0; i < N; ++i) acc = fma(bcast(al, i), act ? As[i * N + lane] : 0.0, acc);
                nx = acc * bq;
            }
            if (!act) nx = 0.0;
            double c = 0.0;
            for (int j = 0; j < N; ++j) c = c + bcast(nx, j);
            if (!(c > 0.0)) {
                st = 1;
                break;
            }
            al = nx / c;
            scale_step(c, p, E);
        }
    }
    // (an empty sequence scores P = 1 = 0.5 * 2^1 with status 0: the empty product; the E-step reports the same P but
    // status 1 -- "not used": an empty sequence has no counts to give and must not enter the pi denominator)
    if (lane == 0) {
        const size_t idx = (size_t)s * K + k;
        mant[idx] = st == 0 ? p : 0.0;
        exp2[idx] = st == 0 ? E : 0;
        status[idx] = st;
    }
}

__device__ __forceinline__ void acc_add(i64* cell, double x)
{
    int hi, lo;
    e2vq::fix2(x, ACC_SHIFT, hi, lo);
    atomicAdd((u64*)&cell[0], (u64)(i64)hi);
    atomicAdd((u64*)&cell[1], (u64)(i64)lo);
}

constexpr int FB_WAVES = 4;

__device__ __forceinline__ void acc_local(i64 (&cell)[2], double x)
{
    int hi, lo;
    e2vq::fix2(x, ACC_SHIFT, hi, lo);
    cell[0] += (i64)hi;
    cell[1] += (i64)lo;
}

// E-step: one wave per sequence of ONE model; persistent workgroups stride over the sequences.
// alpha_buf: [total symbols][N] scratch, c_buf: [total symbols].
// Where the expected counts go (all of them exact integer limb sums, so the routes are interchangeable bit for bit):
//   xi  -> AN : a workgroup-wide LDS table (ds_add_u64), flushed with N^2 global atomics per workgroup at the end --
//               per-step global atomics on the N^2 hot words serialise at the memory side (measured: 40 ms per
//               E-step for 600 k symbols at N = 5, all of it same-address contention)
//   gamma -> AD, BD, PI : per-lane registers over the wave's sequences, one global atomic per lane at the end
//   gamma -> BN[j][o_t] : global atomics (N x M words: spread out)
__global__ __launch_bounds__(64 * FB_WAVES) void k_hmm_fb(ModelDev md, const unsigned short* __restrict__ sym,
                                                           const i64* __restrict__ offs, int S,
                                                           double* __restrict__ alpha_buf, double* __restrict__ c_buf,
                                                           i64* __restrict__ acc, double* __restrict__ mant,
                                                           i64* __restrict__ exp2, int* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N = md.N, M = md.M;
    double* As = (double*)smem;       // [i][j]
    double* ATs = As + N * N;         // [j][i]
    i64* ANs = (i64*)(ATs + N * N);   // [i][j][hi, lo]
    for (int x = threadIdx.x; x < N * N; x += blockDim.x) {
        const double v = md.A[x];
        As[x] = v;
        ATs[(x % N) * N + (x / N)] = v;
        ANs[2 * x] = 0;
        ANs[2 * x + 1] = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    i64* PI = acc;
    i64* AN = PI + 2 * N;
    i64* AD = AN + 2 * (i64)N * N;
    i64* BN = AD + 2 * N;
    i64* BD = BN + 2 * (i64)N * M;
    i64* counts = BD + 2 * N;
    const bool act = lane < N;
    const int ln = act ? lane : 0;
    const double* Brow = md.B + (size_t)ln * M;
    const double pij = act ? md.pi[lane] : 0.0;
    i64 ad[2] = {0, 0}, bd[2] = {0, 0}, pic[2] = {0, 0};
    int used = 0, skipped = 0;
    for (int s = blockIdx.x * FB_WAVES + wib; s < S; s += gridDim.x * FB_WAVES) {
        const i64 base = offs[s];
        const i64 T = offs[s + 1] - base;
        double* alpha = alpha_buf + (size_t)base * N;
        double* cs = c_buf + base;
        // ---- forward (as k_hmm_score), keeping alpha^_t and c_t --------------------------------------------
        double al = 0.0, p = 0.5;
        i64 E = 1;
        int st = T < 1 ? 1 : 0;
        for (i64 t = 0; t < T && st == 0; ++t) {
            const int o = (int)sym[base + t];  // wave-uniform (scalar load)
            if (o >= M) {
                st = 2;
                break;
            }
            const double b = act ? Brow[o] : 0.0;
            double nx;
            if (t == 0) {
                nx = pij * b;
            } else {
                double a = 0.0;
                for (int i = 0; i < N; ++i) a = fma(bcast(al, i), As[i * N + ln], a);
                nx = a * b;
            }
            if (!act) nx = 0.0;
            double c = 0.0;
            for (int j = 0; j < N; ++j) c = c + bcast(nx, j);
            if (!(c > 0.0)) {
                st = 1;
                break;
            }
            al = nx / c;
            if (act) alpha[(size_t)t * N + lane] = al;
            if (lane == 0) cs[t] = c;
            scale_step(c, p, E);
        }
        if (lane == 0) {
            mant[s] = st == 0 ? p : (T < 1 ? 0.5 : 0.0);
            exp2[s] = st == 0 ? E : (T < 1 ? 1 : 0);
            status[s] = st;
        }
        if (st != 0) {  // (wave-uniform) the sequence contributes nothing
            ++skipped;
            continue;
        }
        ++used;
        // ---- backward + expected counts -----------------------------------------------------------------------
        // `al` is alpha^_{T-1} already; every later alpha^_t was written by this very lane, so no fence is needed
        double beta = 1.0;
        for (i64 t = T - 1; t >= 0; --t) {
            if (t < T - 1) {
                al = act ? alpha[(size_t)t * N + lane] : 0.0;
                const int o1 = (int)sym[base + t + 1];
                // c_{t+1}: lane 0 re-reads its own store (same-thread order) and hands it to the wave
                const double c1 = bcast(lane == 0 ? cs[t + 1] : 0.0, 0);
                const double u = act ? (Brow[o1] * beta) / c1 : 0.0;  // u_j, j = lane
                // xi_t(i, j) = (alpha^_t(i) * A_ij) * u_j   -- lane j, all i
                for (int i = 0; i < N; ++i) {
                    const double x = (bcast(al, i) * As[i * N + ln]) * u;
                    if (act) {
                        int hi, lo;
                        e2vq::fix2(x, ACC_SHIFT, hi, lo);
                        atomicAdd((u64*)&ANs[2 * (i * N + lane)], (u64)(i64)hi);
                        atomicAdd((u64*)&ANs[2 * (i * N + lane) + 1], (u64)(i64)lo);
                    }
                }
                // beta^_t(i) = chain_j fma(A_ij, u_j)        -- lane i, all j (transposed copy)
                double a = 0.0;
                for (int j = 0; j < N; ++j) a = fma(ATs[j * N + ln], bcast(u, j), a);
                beta = a;
            }
            if (act) {
                const double g = al * beta;
                const int o = (int)sym[base + t];
                if (t < T - 1) acc_local(ad, g);
                acc_add(BN + 2 * ((i64)lane * M + o), g);
                acc_local(bd, g);
                if (t == 0) acc_local(pic, g);
            }
        }
    }
    // ---- flush: per-lane sums, the workgroup's AN table, the sequence counts ---------------------------------------
    if (act) {
        if (ad[0]) atomicAdd((u64*)&AD[2 * lane], (u64)ad[0]);
        if (ad[1]) atomicAdd((u64*)&AD[2 * lane + 1], (u64)ad[1]);
        if (bd[0]) atomicAdd((u64*)&BD[2 * lane], (u64)bd[0]);
        if (bd[1]) atomicAdd((u64*)&BD[2 * lane + 1], (u64)bd[1]);
        if (pic[0]) atomicAdd((u64*)&PI[2 * lane], (u64)pic[0]);
        if (pic[1]) atomicAdd((u64*)&PI[2 * lane + 1], (u64)pic[1]);
    }
    if (lane == 0) {
        if (used) atomicAdd((u64*)&counts[0], (u64)used);
        if (skipped) atomicAdd((u64*)&counts[1], (u64)skipped);
    }
    __syncthreads();
    for (int x = threadIdx.x; x < 2 * N * N; x += blockDim.x) {
        const i64 v = ANs[x];
        if (v != 0) atomicAdd((u64*)&AN[x], (u64)v);
    }
}

// ---- more than 64 states: one workgroup per sequence, thread j = state j -------------------------------------------
// LDS: xs[N] (the values the next sum runs over), ys[N].  grid: (S, models of this launch) for scoring.
// LDSA (round 4, the E-step kernel k_hmm_fb_wg at N <= WG_LDS_N = 141): the transition matrix is staged in LDS with an odd
// leading dimension, so that both the forward pass (fixed i, consecutive j across the threads) and the backward pass (row j:
// stride ld across the threads) read it without bank conflicts -- one read of A per persistent workgroup instead of three
// per symbol.  Same values, same order of operations: same bits.  (E+M step at N = 141 / 142: 182 / 277 ms per 1 024 x 300 symbols.)
__host__ __device__ constexpr int wg_lda(int N) { return N | 1; }
__host__ __device__ constexpr size_t wg_lds_bytes(int N, bool ldsa) { return (size_t)2 * N * 8 + (ldsa ? (size_t)N * wg_lda(N) * 8 : 0); }
constexpr int WG_LDS_N = 141;  // 141 x 141 doubles + 2 x 141 = 161 292 B of the CU's 163 840

// (k_hmm_score_wg keeps reading A from global memory / L2: staged in LDS a workgroup -- one (sequence, model) pair -- would
// copy up to 159 KB for ~300 steps and be alone on its CU; measured at N = 141: 43 ms against 34 for 1 024 x 8 pairs)
__global__ void k_hmm_score_wg(const ModelDev* __restrict__ models, int K, int k0, const unsigned short* __restrict__ sym,
                               const i64* __restrict__ offs, int S, double* __restrict__ mant, i64* __restrict__ exp2,
                               int* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* als = (double*)smem;
    const int k = k0 + (int)blockIdx.y, s = (int)blockIdx.x;
    const ModelDev md = models[k];
    const int N = md.N, M = md.M;
    double* nxs = als + N;
    const int j = threadIdx.x;
    const bool act = j < N;
    const int jj = act ? j : 0;
    const i64 base = offs[s];
    const i64 T = offs[s + 1] - base;
    double p = 0.5;
    i64 E = 1;
    int st = 0;
    for (i64 t = 0; t < T; ++t) {
        const int o = (int)sym[base + t];
        if (o >= M) {
            st = 2;
            break;
        }
        const double b = md.B[(size_t)jj * M + o];
        double nx;
        if (t == 0) {
            nx = md.pi[jj] * b;
        } else {
            double acc = 0.0;
            for (int i = 0; i < N; ++i) acc = fma(als[i], md.A[(size_t)i * N + jj], acc);
            nx = acc * b;
        }
        if (act) nxs[j] = nx;
        __syncthreads();
        double c = 0.0;
        for (int i = 0; i < N; ++i) c = c + nxs[i];
        if (!(c > 0.0)) {  // (every thread holds the same c)
            st = 1;
            break;
        }
        if (act) als[j] = nx / c;
        __syncthreads();
        scale_step(c, p, E);
    }
    if (j == 0) {
        const size_t idx = (size_t)s * K + k;
        mant[idx] = st == 0 ? p : 0.0;
        exp2[idx] = st == 0 ? E : 0;
        status[idx] = st;
    }
}

// E-step.  scratch: gridDim.x tables of 2 N^2 words (zeroed here, flushed to AN with atomics at the end).
template <bool LDSA>
__global__ void k_hmm_fb_wg(ModelDev md, const unsigned short* __restrict__ sym, const i64* __restrict__ offs, int S,
                            double* __restrict__ alpha_buf, double* __restrict__ c_buf, i64* __restrict__ acc,
                            double* __restrict__ mant, i64* __restrict__ exp2, int* __restrict__ status,
                            i64* __restrict__ scratch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N = md.N, M = md.M;
    double* xs = (double*)smem;  // alpha^ of the step (forward: of the previous step)
    double* ys = xs + N;         // nx (forward) / u (backward)
    const int j = threadIdx.x;
    const bool act = j < N;
    const int jj = act ? j : 0;
    // LDSA: the transition matrix in LDS, odd leading dimension (wg_lda)
    const int ld = LDSA ? wg_lda(N) : N;
    double* As = ys + N;
    const double* Ag = md.A;
    if constexpr (LDSA) {
        for (int x = threadIdx.x; x < N * N; x += blockDim.x) As[(x / N) * ld + (x % N)] = md.A[x];
        __syncthreads();
    }
    auto Aat = [&](int r, int c) -> double { return LDSA ? As[r * ld + c] : Ag[(size_t)r * N + c]; };
    i64* PI = acc;
    i64* AN = PI + 2 * N;
    i64* AD = AN + 2 * (i64)N * N;
    i64* BN = AD + 2 * N;
    i64* BD = BN + 2 * (i64)N * M;
    i64* counts = BD + 2 * N;
    i64* ANw = scratch + (size_t)blockIdx.x * 2 * N * N;  // [i][j][hi, lo]: column j belongs to thread j
    if (act)
        for (int i = 0; i < N; ++i) ANw[2 * ((size_t)i * N + j)] = ANw[2 * ((size_t)i * N + j) + 1] = 0;
    i64 ad[2] = {0, 0}, bd[2] = {0, 0}, pic[2] = {0, 0};
    int used = 0, skipped = 0;
    for (int s = blockIdx.x; s < S; s += gridDim.x) {
        const i64 base = offs[s];
        const i64 T = offs[s + 1] - base;
        double* alpha = alpha_buf + (size_t)base * N;
        double* cs = c_buf + base;
        double al = 0.0, p = 0.5;
        i64 E = 1;
        int st = T < 1 ? 1 : 0;
        __syncthreads();  // (the previous sequence's readers of xs / ys are done)
        for (i64 t = 0; t < T && st == 0; ++t) {
            const int o = (int)sym[base + t];
            if (o >= M) {
                st = 2;
                break;
            }
            const double b = md.B[(size_t)jj * M + o];
            double nx;
            if (t == 0) {
                nx = md.pi[jj] * b;
            } else {
                double a = 0.0;
                for (int i = 0; i < N; ++i) a = fma(xs[i], Aat(i, jj), a);
                nx = a * b;
            }
            if (act) ys[j] = nx;
            __syncthreads();
            double c = 0.0;
            for (int i = 0; i < N; ++i) c = c + ys[i];
            if (!(c > 0.0)) {
                st = 1;
                break;
            }
            al = nx / c;
            if (act) {
                xs[j] = al;
                alpha[(size_t)t * N + j] = al;
            }
            if (j == 0) cs[t] = c;
            __syncthreads();
            scale_step(c, p, E);
        }
        if (j == 0) {
            mant[s] = st == 0 ? p : (T < 1 ? 0.5 : 0.0);
            exp2[s] = st == 0 ? E : (T < 1 ? 1 : 0);
            status[s] = st;
        }
        if (st != 0) {  // (workgroup-uniform)
            ++skipped;
            continue;
        }
        ++used;
        // backward: al is alpha^_{T-1}(j); cs[] was written by thread 0 -- the barriers above order it for the workgroup
        double beta = 1.0;
        for (i64 t = T - 1; t >= 0; --t) {
            if (t < T - 1) {
                al = act ? alpha[(size_t)t * N + j] : 0.0;
                const int o1 = (int)sym[base + t + 1];
                const double c1 = cs[t + 1];
                const double u = (md.B[(size_t)jj * M + o1] * beta) / c1;  // u_j
                __syncthreads();  // (the previous step's readers of xs / ys are done)
                if (act) {
                    xs[j] = al;
                    ys[j] = u;
                }
                __syncthreads();
                if (act) {
                    // xi_t(i, j) = (alpha^_t(i) A_ij) u_j: column j of the workgroup's table
                    for (int i = 0; i < N; ++i) {
                        const double x = (xs[i] * Aat(i, j)) * u;
                        int hi, lo;
                        e2vq::fix2(x, ACC_SHIFT, hi, lo);
                        ANw[2 * ((size_t)i * N + j)] += (i64)hi;
                        ANw[2 * ((size_t)i * N + j) + 1] += (i64)lo;
                    }
                }
                // beta^_t(j) = chain_i fma(A_ji, u_i)  (row j of A)
                double a = 0.0;
                for (int i = 0; i < N; ++i) a = fma(Aat(jj, i), ys[i], a);
                beta = a;
            }
            if (act) {
                const double g = al * beta;
                const int o = (int)sym[base + t];
                if (t < T - 1) acc_local(ad, g);
                acc_add(BN + 2 * ((i64)j * M + o), g);
                acc_local(bd, g);
                if (t == 0) acc_local(pic, g);
            }
        }
    }
    if (act) {
        if (ad[0]) atomicAdd((u64*)&AD[2 * j], (u64)ad[0]);
        if (ad[1]) atomicAdd((u64*)&AD[2 * j + 1], (u64)ad[1]);
        if (bd[0]) atomicAdd((u64*)&BD[2 * j], (u64)bd[0]);
        if (bd[1]) atomicAdd((u64*)&BD[2 * j + 1], (u64)bd[1]);
        if (pic[0]) atomicAdd((u64*)&PI[2 * j], (u64)pic[0]);
        if (pic[1]) atomicAdd((u64*)&PI[2 * j + 1], (u64)pic[1]);
        for (int i = 0; i < N; ++i) {
            const i64 h = ANw[2 * ((size_t)i * N + j)], l = ANw[2 * ((size_t)i * N + j) + 1];
            if (h) atomicAdd((u64*)&AN[2 * ((size_t)i * N + j)], (u64)h);
            if (l) atomicAdd((u64*)&AN[2 * ((size_t)i * N + j) + 1], (u64)l);
        }
    }
    if (j == 0) {
        if (used) atomicAdd((u64*)&counts[0], (u64)used);
        if (skipped) atomicAdd((u64*)&counts[1], (u64)skipped);
    }
}

// M-step: one thread per parameter
__global__ void k_hmm_reestimate(int N, int M, const i64* __restrict__ acc, double* __restrict__ pi,
                                 double* __restrict__ A, double* __restrict__ B)
{
    const i64* PI = acc;
    const i64* AN = PI + 2 * N;
    const i64* AD = AN + 2 * (i64)N * N;
    const i64* BN = AD + 2 * N;
    const i64* BD = BN + 2 * (i64)N * M;
    const i64 used = BD[2 * N];
    const i64 x = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < N) {
        if (used > 0) pi[x] = e2vq::unfix(PI[2 * x], PI[2 * x + 1], ACC_SHIFT) / (double)used;
    } else if (x < N + (i64)N * N) {
        const i64 e = x - N;
        const int i = (int)(e / N);
        const double den = e2vq::unfix(AD[2 * i], AD[2 * i + 1], ACC_SHIFT);
        if (den > 0.0) A[e] = e2vq::unfix(AN[2 * e], AN[2 * e + 1], ACC_SHIFT) / den;
    } else if (x < N + (i64)N * N + (i64)N * M) {
        const i64 e = x - N - (i64)N * N;
        const int j = (int)(e / M);
        const double den = e2vq::unfix(BD[2 * j], BD[2 * j + 1], ACC_SHIFT);
        if (den > 0.0) B[e] = e2vq::unfix(BN[2 * e], BN[2 * e + 1], ACC_SHIFT) / den;
    }
}

// hmm_adjustb: floor at epsilon, then divide the row by its sequential sum (one thread per state; M <= 65536)
__global__ void k_hmm_adjustb(int N, int M, double epsilon, double* __restrict__ B)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    double* row = B + (size_t)j * M;
    double s = 0.0;
    for (int k = 0; k < M; ++k) {
        double v = row[k];
        if (v < epsilon) {
            v = epsilon;
            row[k] = v;
        }
        s = s + v;
    }
    for (int k = 0; k < M; ++k) row[k] = row[k] / s;
}

// ---- launchers ------------------------------------------------------------------------------------------------
i64 acc_words(int N, int M) { return 2 * ((i64)N + (i64)N * N + N + (i64)N * M + N) + 2; }

constexpr int FB_WG_GRID = 64;  // workgroups of the big-N E-step (each owns a 2 N^2-word table of the scratch)

i64 fb_scratch_words(int N) { return N > WAVE_N ? (i64)FB_WG_GRID * 2 * N * N : 0; }

void launch_score(const ModelDev* models, int K, int maxN, const unsigned short* sym, const i64* offs, int S, double* mant,
                  i64* exp2, int* status, hipStream_t st)
{
    if (S < 1 || K < 1) return;
    // grid.y carries the models: at most 65535 per launch (HIP's limit for that dimension), more in further launches
    for (int k0 = 0; k0 < K; k0 += 65535) {
        const int kn = K - k0 < 65535 ? K - k0 : 65535;
        if (maxN > WAVE_N) {  // some model has more states than a wave has lanes: a workgroup per pair, for all of them
            for (int s0 = 0; s0 < S; s0 += 1 << 20) {  // (grid.x is ample, but keep launches bounded)
                const int sn = S - s0 < (1 << 20) ? S - s0 : (1 << 20);
                hipLaunchKernelGGL(k_hmm_score_wg, dim3((unsigned)sn, (unsigned)kn), dim3((unsigned)((maxN + 63) & ~63)),
                                   (size_t)2 * maxN * 8, st, models, K, k0, sym, offs + s0, sn, mant + (size_t)s0 * K,
                                   exp2 + (size_t)s0 * K, status + (size_t)s0 * K);
            }
            continue;
        }
        const dim3 grid((unsigned)((S + SCORE_WAVES - 1) / SCORE_WAVES), (unsigned)kn);
        hipLaunchKernelGGL(k_hmm_score, grid, dim3(64 * SCORE_WAVES), (size_t)maxN * maxN * 8, st, models, K, k0, sym, offs, S,
                           mant, exp2, status);
    }
}

void launch_fb(const ModelDev& md, const unsigned short* sym, const i64* offs, int S, double* alpha_buf, double* c_buf,
               i64* acc, double* mant, i64* exp2, int* status, hipStream_t st, i64* scratch)
{
    if (S < 1) return;
    if (md.N > WAVE_N) {
        const int grid = S < FB_WG_GRID ? S : FB_WG_GRID;
        if (md.N <= WG_LDS_N) {
            (void)hipFuncSetAttribute((const void*)k_hmm_fb_wg<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(k_hmm_fb_wg<true>, dim3((unsigned)grid), dim3((unsigned)((md.N + 63) & ~63)), wg_lds_bytes(md.N, true),
                               st, md, sym, offs, S, alpha_buf, c_buf, acc, mant, exp2, status, scratch);
        } else {
            hipLaunchKernelGGL(k_hmm_fb_wg<false>, dim3((unsigned)grid), dim3((unsigned)((md.N + 63) & ~63)), wg_lds_bytes(md.N, false),
                               st, md, sym, offs, S, alpha_buf, c_buf, acc, mant, exp2, status, scratch);
        }
        return;
    }
    const size_t lds = (size_t)md.N * md.N * (2 * 8 + 16);  // A, A^T, and the workgroup's AN limb table: 128 KB at N = 64
    // (a refusal shows up as a launch failure below, which the caller's hipGetLastError reports; say why here)
    if (hipFuncSetAttribute((const void*)k_hmm_fb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        fprintf(stderr, "ecoz2vq: the device refuses 160 KB of dynamic LDS for the E-step kernel (N = %d needs %zu bytes)\n", md.N, lds);
    const int blocks = (S + FB_WAVES - 1) / FB_WAVES;
    const int grid = blocks < 2048 ? blocks : 2048;  // persistent: each workgroup flushes its AN table once
    hipLaunchKernelGGL(k_hmm_fb, dim3((unsigned)grid), dim3(64 * FB_WAVES), lds, st, md, sym, offs, S, alpha_buf, c_buf, acc,
                       mant, exp2, status);
}

void launch_reestimate(int N, int M, const i64* acc, double epsilon, double* pi, double* A, double* B, hipStream_t st)
{
    const i64 total = (i64)N + (i64)N * N + (i64)N * M;
    hipLaunchKernelGGL(k_hmm_reestimate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, N, M, acc, pi, A, B);
    if (epsilon > 0.0) hipLaunchKernelGGL(k_hmm_adjustb, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, st, N, M, epsilon, B);
}

}  // namespace e2hmm
