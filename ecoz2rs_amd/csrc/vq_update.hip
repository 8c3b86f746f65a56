// vq_update.hip -- everything between two sweeps of an LBG iteration (gfx950 / CDNA4): the per-level statistics of the
// (all-reduced) rows, the centroid update K3 (Levinson-Durbin per cell, src/lpc/lpca_r_rs.rs:8-43) and the codebook
// images K4 of the next pass -- fused in k_cell_update, which publishes the statistics by itself --, the codebook's
// initialisation and M -> 2M split, the seeding of a level's first pass from its parents' sums, the pass prologue, and the
// slice reduction of the in-process group's peer-to-peer exchange.  (Split from vq_device.hip in round 5: that file keeps the
// frames' layouts and the FP64 sweeps.)
//
// Compiled with -ffp-contract=off: every FMA below is explicit.
#include "vq_accum.h"
#include "vq_device.h"
#include "vq_fixed.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace e2vq {

static inline int grid_for(long work_items, int per_block, int cap)
{
    long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

__device__ __forceinline__ i64 wave_sum_i64(i64 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------
// per-level statistics from the (all-reduced) rows
// ------------------------------------------------------------------------------------------
// lstats (i64): [0] dist_hi [1] dist_lo [2] dist2_hi [3] dist2_lo [4] empty cells [5] failed cells
__global__ void k_rows_stats(const i64* __restrict__ rows, int M, int NC, const DevScalars* __restrict__ sc,
                             double* __restrict__ S, double* __restrict__ within, i64* __restrict__ lstats)
{
    const int RS = (2 * NC + 5 + 7) & ~7;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const i64* row = rows + (long)m * RS;
    const int sh_r = sc->sh_r;
    atomicAdd((u64*)&lstats[0], (u64)row[2 * NC + 1]);
    atomicAdd((u64*)&lstats[1], (u64)row[2 * NC + 2]);
    atomicAdd((u64*)&lstats[2], (u64)row[2 * NC + 3]);
    atomicAdd((u64*)&lstats[3], (u64)row[2 * NC + 4]);
    const i64 cnt = row[2 * NC];
    if (cnt == 0) {
        atomicAdd((u64*)&lstats[4], 1ull);
        within[m] = 0.0;
        return;
    }
    double ss = 0.0;
    for (int n = 0; n < NC; ++n) {
        const double s = unfix(row[2 * n], row[2 * n + 1], sh_r);
        S[(long)m * NC + n] = s;
        ss += s * s;
    }
    within[m] = ss / (double)cnt;
}

// Per-thread arrays live in LDS as columns (element i of thread t at [i*64 + t]): dynamic indexing without
// scratch memory, conflict-free, ~64-cycle access instead of a global round trip.
struct Col {
    double* p;
    int stride;  // threads per block
    __device__ __forceinline__ double& operator[](int i) const { return p[i * stride]; }
};

// Levinson-Durbin from autocorrelation; src/lpc/lpca_r_rs.rs:8-43.  rc and a are LDS columns.
__device__ int lpca_r(int P, Col r, Col rc, Col a)
{
    const double r0 = r[0];
    if (0.0 == r0) return 1;
    double pe = r0;
    a[0] = 1.0;
    for (int k = 1; k <= P; ++k) {
        double sum = 0.0;
        for (int i = 1; i <= k; ++i) sum -= a[k - i] * r[i];
        const double akk = sum / pe;
        rc[k] = akk;
        a[k] = akk;
        for (int i = 1; i <= (k >> 1); ++i) {
            const double ai = a[i];
            const double aj = a[k - i];
            a[i] = ai + akk * aj;
            a[k - i] = aj + akk * ai;
        }
        pe *= 1.0 - akk * akk;
        if (pe <= 0.0) return 2;
    }
    return 0;
}

// K3: cell sums -> reflections.  Non-empty cells whose recursion succeeds get new reflections, every other
// cell keeps its codeword (refl_out may alias refl_in, or be a shadow buffer for the speculative update).
__global__ void k_centroids(const i64* __restrict__ rows, const double* __restrict__ S, int M, int NC,
                            const double* refl_in, double* refl_out, i64* __restrict__ lstats)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* base = (double*)smem;
    const int W = blockDim.x;
    const Col r{base + threadIdx.x, W}, rc{base + NC * W + threadIdx.x, W}, a{base + 2 * NC * W + threadIdx.x, W};
    const int RS = (2 * NC + 5 + 7) & ~7;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int P = NC - 1;
    double* dst = refl_out + (long)m * NC;
    const double* src = refl_in + (long)m * NC;
    bool fresh = rows[(long)m * RS + 2 * NC] != 0;
    if (fresh) {
        for (int n = 0; n < NC; ++n) r[n] = S[(long)m * NC + n];
        if (lpca_r(P, r, rc, a) != 0) {
            atomicAdd((u64*)&lstats[5], 1ull);
            fresh = false;
        }
    }
    if (fresh) {
        dst[0] = 0.0;
        for (int n = 1; n <= P; ++n) dst[n] = rc[n];
    } else if (dst != src) {
        for (int n = 0; n < NC; ++n) dst[n] = src[n];
    }
}


// ------------------------------------------------------------------------------------------
// K3+K4 fused, one wave per cell (NC <= 64): lane n owns coefficient n.
//   rows != nullptr : limbs -> S -> statistics -> Levinson (lpca_r) -> reflections -> codeword images
//   rows == nullptr : reflections -> codeword images only (after set_codebook / grow)
// Every sequential sum of the oracle (the Levinson inner product, sum S^2, raas, L1 norm) is still evaluated
// term by term in the canonical order: the lanes produce the terms in parallel and a lane-uniform loop adds them,
// each term broadcast from its lane with v_readlane (an SGPR operand of the add: no LDS round trip, the chain is
// one v_add_f64 per term), so all lanes carry the same running value.  Element-wise updates
// (a[i] += akk*a[k-i]) are independent per i and run across lanes as they are; the reversed operand a[k - lane]
// is kept as a second per-lane array that follows the same recursion and moves up one lane per step (DPP
// wave_shr:1).  The whole wave stays active throughout (DPP and readlane under a partial EXEC mask would read
// stale lanes): `act` only selects values.
// One extra workgroup publishes the level statistics to the host as soon as they are complete: see PublishArgs.
// ------------------------------------------------------------------------------------------
constexpr int CU_WAVES = 4;  // waves (cells) per workgroup of k_cell_update

__device__ __forceinline__ double lane_bcast(double x, int l)  // l must be wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
    return __hiloint2double(hi, lo);
}
// lane i takes the value of lane i - 1 (lane 0 reads 0) / of lane i + 1 (lane 63 reads 0)
__device__ __forceinline__ double lane_shr1(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_shl1(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// ((0.0 + x[first]) + x[first+1]) + ... + x[last], the terms taken from lanes first..last (wave-uniform bounds)
template <bool CONST_BOUNDS = false>
__device__ __forceinline__ double lane_ordered_sum(double x, int first, int last)
{
    double sum = 0.0;
    if constexpr (CONST_BOUNDS) {  // (first / last are constants after the caller's loop has been unrolled)
#pragma unroll
        for (int i = first; i <= last; ++i) sum += lane_bcast(x, i);
        return sum;
    }
    int i = first;
    for (; i + 3 <= last; i += 4) {
        const double t0 = lane_bcast(x, i), t1 = lane_bcast(x, i + 1), t2 = lane_bcast(x, i + 2),
                     t3 = lane_bcast(x, i + 3);
        sum += t0;
        sum += t1;
        sum += t2;
        sum += t3;
    }
    for (; i <= last; ++i) sum += lane_bcast(x, i);
    return sum;
}
__device__ __forceinline__ bool wave_uniform(bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; }

// step-up of the predictor, one order: a[i] += akk * a[k-i] (0 < i < k), a[k] = akk; arev[i] = a[k - i] follows the
// same recursion (arev[i] += akk * a[i], arev[0] = akk) and is then moved up one lane: arev[i] = a[(k+1) - i]
__device__ __forceinline__ void step_up(int lane, int k, double akk, double& a, double& arev)
{
    const double an = a + akk * arev, rn = arev + akk * a;
    const bool mid = lane >= 1 && lane < k;
    a = lane == k ? akk : (mid ? an : a);
    arev = lane == 0 ? akk : (mid ? rn : arev);
    arev = lane_shr1(arev);
}

// NCT > 0 (round 6): the order is a compile-time constant -- every loop below is unrolled, every broadcast comes from a
// CONSTANT lane, and the 666 + 3 x 37 dependent additions of a cell are straight-line code: two v_readlane (independent of
// the chain: issued ahead) and one v_add_f64 per term, no loop counter, no branch.  With run-time bounds (NCT = 0: the other
// orders) a term cost ~75 cycles -- scalar loop control and a lane select from an SGPR in front of every group of four --;
// the order of the additions, and with it every bit of the result, is the same either way (src/lpc/lpca_r_rs.rs:17-24).
template <int NCT>
__global__ __launch_bounds__(64 * CU_WAVES) void k_cell_update(
    const i64* __restrict__ rows, int M, int NC_rt, const DevScalars* __restrict__ sc, const double* refl_in,
    double* refl_out, double* __restrict__ cbq, double* __restrict__ cbm, int MT, u64* __restrict__ l1max_bits,
    double* __restrict__ within, i64* __restrict__ lstats, const int* __restrict__ ea, int* __restrict__ eC_biased,
    PublishArgs pub)
{
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int m = blockIdx.x * CU_WAVES + wib;
    const int NC = NCT ? NCT : NC_rt;
    const int P = NC - 1, RS = (2 * NC + 5 + 7) & ~7, NPAD = (NC + 7) & ~7;
    const int nb = (M + CU_WAVES - 1) / CU_WAVES;  // workgroups that own cells
    if (pub.flags && (int)blockIdx.x == nb) {      // the publisher (dispatched last: every cell workgroup is under way)
        // (every datum read below was written device-coherently -- memory-side atomics, agent-scope atomic stores -- and
        // is read with agent-scope atomic loads: no cache write-back or invalidate is involved anywhere)
        // The cells' workgroups are dispatched before this one (lower indices: the order HIP launches in, though it
        // promises none), so the flags polled here are on their way.  Should one never arrive -- a workgroup that was
        // not scheduled, a lost store -- the spin gives up after ~0.5 s, publishes what is there and raises *h_err:
        // the host turns that into an error instead of hanging.
        const unsigned int want = (unsigned int)pub.seq;
        constexpr int SPIN_CAP = 1 << 22;
        bool late = false;
        for (int i = threadIdx.x; i < M; i += blockDim.x) {
            int spins = 0;
            while (__hip_atomic_load(&pub.flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want && spins < SPIN_CAP) {
                __builtin_amdgcn_s_sleep(4);
                ++spins;
            }
            late |= spins >= SPIN_CAP;
        }
        if (late && pub.h_err) *pub.h_err = pub.seq;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * 8; i += blockDim.x) {
            if ((i & 7) == 5) continue;  // (failed recursions: counted in part 2, published at the end)
            pub.h_l[i] = (i64)__hip_atomic_load((u64*)&lstats[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lstats[i] = 0;  // ready for the next pass
        }
        for (int i = threadIdx.x; i < M; i += blockDim.x)
            pub.h_within[i] = __longlong_as_double(
                (i64)__hip_atomic_load((const u64*)&within[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (threadIdx.x == 0) *pub.h_l1 = __hip_atomic_load(pub.l1max_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the frames the pass's prefiltered sweep could not certify: the host weighs them against the plain sweep)
        if (threadIdx.x == 1 && pub.h_fb)
            *pub.h_fb = pub.fb_count ? (long long)__hip_atomic_load(pub.fb_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1ll;
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            *pub.h_seq = pub.seq;
            __threadfence_system();
        }
        late = false;
        for (int i = threadIdx.x; i < M; i += blockDim.x) {
            int spins = 0;
            while (__hip_atomic_load(&pub.flags[M + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want && spins < SPIN_CAP) {
                __builtin_amdgcn_s_sleep(4);
                ++spins;
            }
            late |= spins >= SPIN_CAP;
        }
        if (late && pub.h_err) *pub.h_err = pub.seq;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __syncthreads();
        if (threadIdx.x < 64) {
            i64 f = (i64)__hip_atomic_load((u64*)&lstats[threadIdx.x * 8 + 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lstats[threadIdx.x * 8 + 5] = 0;
            for (int d = 32; d >= 1; d >>= 1) f += __shfl_xor(f, d, 64);
            if (threadIdx.x == 0) {
                *pub.h_failed = f;
                __threadfence_system();
                *pub.h_seq2 = pub.seq;
                __threadfence_system();
            }
        }
        return;
    }
    const bool act = lane < NC;
    const bool cell = m < M;  // wave-uniform
    i64* ls = lstats + (blockIdx.x & 63) * 8;

    // ---- part 1: everything the host's convergence decision needs (level statistics, within-cell terms) ---------
    double S = 0.0;
    i64 cnt = 0;
    if (cell && rows) {
        const i64* row = rows + (long)m * RS;
        cnt = row[2 * NC];
        // level statistics: 64 slots of 8 words (same-address atomics serialise at the memory side; the host
        // adds the slots -- integers, so still exact)
        if (lane < 4) atomicAdd((u64*)&ls[lane], (u64)row[2 * NC + 1 + lane]);
        if (lane == 4 && cnt == 0) atomicAdd((u64*)&ls[4], 1ull);
        double w = 0.0;
        if (cnt != 0) {  // wave-uniform
            S = act ? unfix(row[2 * lane], row[2 * lane + 1], sc->sh_r) : 0.0;
            // within-cell term: ss = sum_n S_n^2 (ascending n), / count
            w = lane_ordered_sum<(NCT > 0)>(S * S, 0, P) / (double)cnt;
        }
        if (lane == 0)
            __hip_atomic_store((u64*)&within[m], (u64)__double_as_longlong(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pub.flags) {
            // this cell's statistics are out: once the atomics and the store above have been performed the publisher may
            // count the cell in.  The wait is spelled out: a workgroup-scope release fence emits none (stores of one wave
            // to different addresses may land in any order at the memory side), and an agent-scope one would also write
            // the whole L2 back -- a thousand of those queue up for tens of microseconds
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store(&pub.flags[m], (unsigned int)pub.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- part 2: the new codeword of the cell and its images ------------------------------------------------------
    if (cell) {
        const double* src = refl_in + (long)m * NC;
        double a = lane == 0 ? 1.0 : 0.0;     // predictor coefficient a[lane]
        double arev = lane == 1 ? 1.0 : 0.0;  // a[k - lane] for the order k about to be computed (k = 1)
        double rcn = 0.0;                     // reflection rc[lane]
        bool have_a = false;                  // a[] already holds the step-up of the final reflections
        bool fresh = false;

        if (rows && cnt != 0) {  // wave-uniform
            // ---- lpca_r (src/lpc/lpca_r_rs.rs:8-43) on S ------------------------------------------------
            const double r0 = lane_bcast(S, 0);
            int status = 0;
            if (wave_uniform(0.0 == r0)) {
                status = 1;
            } else {
                double pe = r0;
#pragma unroll
                for (int k = 1; k <= P; ++k) {
                    // sum = ((0 - a[k-1] r[1]) - a[k-2] r[2]) - ... - a[0] r[k]   (x - y == x + (-y), exactly)
                    const double sum = lane_ordered_sum<(NCT > 0)>(-(arev * S), 1, k);
                    const double akk = sum / pe;
                    if (lane == k) rcn = akk;
                    step_up(lane, k, akk, a, arev);
                    pe *= 1.0 - akk * akk;
                    if (wave_uniform(pe <= 0.0)) {
                        status = 2;
                        break;
                    }
                }
            }
            if (status == 0) {
                fresh = true;
                have_a = true;
            } else if (lane == 0) {
                atomicAdd((u64*)&ls[5], 1ull);
            }
        }
        if (pub.flags && rows) {  // whether this cell's recursion failed is known (and counted)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store(&pub.flags[M + m], (unsigned int)pub.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }

        if (!fresh) rcn = act ? src[lane] : 0.0;  // keep the codeword
        if (lane == 0) rcn = 0.0;
        if (refl_out && act && (fresh || refl_out != refl_in)) refl_out[(long)m * NC + lane] = rcn;

        if (!have_a) {  // step-up from the reflections (same element-wise updates as inside lpca_r)
            a = lane == 0 ? 1.0 : 0.0;
            arev = lane == 1 ? 1.0 : 0.0;
#pragma unroll
            for (int k = 1; k <= P; ++k) step_up(lane, k, lane_bcast(rcn, k), a, arev);
        }
        if (!act) a = 0.0;

        // ---- raas: raa[n] = sum_{i=0}^{P-n} a[i]*a[i+n]  (ascending i) -> cq ------------------------------
        double raa = 0.0, ash = a;  // ash = a[lane + i]
#pragma unroll
        for (int i = 0; i <= P; ++i) {
            const double t = raa + lane_bcast(a, i) * ash;
            raa = i + lane <= P ? t : raa;
            ash = lane_shl1(ash);
        }
        const double c = !act ? 0.0 : (lane == 0 ? raa : 2.0 * raa);
        const double l1 = lane_ordered_sum<(NCT > 0)>(fabs(c), 0, P);
        if (lane == 0) {  // monotone max: skip the atomic unless it can still raise the value
            const u64 bits = (u64)__double_as_longlong(l1);
            if (bits > __hip_atomic_load(l1max_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(l1max_bits, bits);
        }
        if (eC_biased) {  // scale of the limb image of this codebook: max ilogb(c a) + 1 (k_pre_cmax)
            int e = (act && c != 0.0) ? ilogb(c) + ea[lane] + 1 + E2VQ_PRE_EBIAS : 0;
            for (int d = 32; d >= 1; d >>= 1) {
                const int o = __shfl_xor(e, d, 64);
                e = o > e ? o : e;
            }
            if (lane == 0 && e > __hip_atomic_load(eC_biased, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(eC_biased, e);
        }
        if (lane < NPAD) cbq[(long)m * NPAD + lane] = c;
        if (cbm) {
            const int NS = (NC + 3) >> 2, NP = (NS + 1) >> 1;
            // this cell's slot, plus (cell 0 only) the padding slots of the last tile: copies of codeword 0
            const int first = m, last = (m == 0) ? 16 * MT : m + 1;
            for (int mm = first; mm < last; mm = (mm == first && m == 0) ? M : mm + 1) {
                if (mm >= 16 * MT) break;
                double* mt = cbm + (long)(mm >> 4) * NP * 128;
                const int jm = mm & 15;
                if (lane < 8 * NP)
                    mt[(((lane >> 3) * 64) + ((lane & 3) * 16 + jm)) * 2 + ((lane >> 2) & 1)] = c;
                if (lane == P) cbm[(long)MT * NP * 128 + (long)(mm >> 4) * 16 + (jm & 3) * 4 + (jm >> 2)] = c;
            }
        }
    }

}

// the M = 1 codeword from the global sums
__global__ void k_init_codebook(const i64* __restrict__ stats, int NC,
                                                      const DevScalars* __restrict__ sc,
                                                      double* __restrict__ reflections, int* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* base = (double*)smem;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int W = blockDim.x;
    const Col S{base, W}, rc{base + NC * W, W}, a{base + 2 * NC * W, W};
    for (int n = 0; n < NC; ++n) S[n] = unfix(stats[2 * n], stats[2 * n + 1], sc->sh_r);
    const int st = lpca_r(NC - 1, S, rc, a);
    *status = st;
    if (st != 0) return;
    reflections[0] = 0.0;
    for (int n = 1; n < NC; ++n) reflections[n] = rc[n];
}

// sum-of-squares limbs -> Q (runs after the SUM all-reduce of the data statistics)
__global__ void k_finish_q(const i64* __restrict__ stats, int NC, DevScalars* __restrict__ sc)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) sc->Q = unfix(stats[2 * NC], stats[2 * NC + 1], sc->sh_q);
}

// K4a: M -> 2M split, in place from the top (new[2i] = old[i]*0.99, new[2i+1] = old[i]*1.01)
// (+ up to two small blocks of words zeroed on the way -- the L1 maximum and the limb-image scalars that the kernels behind it
// accumulate into with atomicMax: a memset each, with its own gap in the queue, otherwise)
__global__ void k_grow(const double* __restrict__ old_refl, int M, int NC, double* __restrict__ new_refl, ZeroList z)
{
    if (blockIdx.x == 0)
        for (int k = 0; k < 3; ++k)
            for (int j = threadIdx.x; j < z.words[k]; j += blockDim.x) ((unsigned int*)z.p[k])[j] = 0u;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * NC) return;
    const int m = i / NC, n = i - m * NC;
    const double v = old_refl[i];
    new_refl[(long)(2 * m) * NC + n] = n == 0 ? 0.0 : v * 0.99;
    new_refl[(long)(2 * m + 1) * NC + n] = n == 0 ? 0.0 : v * 1.01;
}

// K4b: reflections -> predictor (step-up) -> raas -> pre-doubled padded codeword rows; L1 max
__global__ void k_codebook_prepare(const double* __restrict__ reflections, int M, int NC, double* __restrict__ cbq,
                                   u64* __restrict__ l1max_bits, double* __restrict__ cbm, int MT)
{
    const int NPAD = (NC + 7) & ~7;
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const Col a{(double*)smem + threadIdx.x, (int)blockDim.x};
    const bool real = m < M;
    if (!real && !(cbm && m < 16 * MT)) return;
    const int P = NC - 1;
    const double* rc = reflections + (long)(real ? m : 0) * NC;  // tile padding repeats codeword 0
    a[0] = 1.0;
    for (int k = 1; k <= P; ++k) {
        const double akk = rc[k];
        a[k] = akk;
        for (int i = 1; i <= (k >> 1); ++i) {
            const double ai = a[i];
            const double aj = a[k - i];
            a[i] = ai + akk * aj;
            a[k - i] = aj + akk * ai;
        }
    }
    double* dst = cbq + (long)m * NPAD;
    const int NS = (NC + 3) >> 2, NP = (NS + 1) >> 1;
    double* mt = cbm ? cbm + (long)(m >> 4) * NP * 128 : nullptr;  // [p][lane = 16q + j][e]
    const int jm = m & 15;
    double l1 = 0.0, lastc = 0.0;
    for (int n = 0; n <= P; ++n) {
        double s = 0.0;
        for (int i = 0; i <= P - n; ++i) s += a[i] * a[i + n];
        const double c = n == 0 ? s : 2.0 * s;
        lastc = c;
        if (real) dst[n] = c;
        if (mt) mt[(((n >> 3) * 64) + ((n & 3) * 16 + jm)) * 2 + ((n >> 2) & 1)] = c;
        l1 += fabs(c);
    }
    if (real)
        for (int n = NC; n < NPAD; ++n) dst[n] = 0.0;
    if (mt) {
        for (int n = NC; n < 8 * NP; ++n) mt[(((n >> 3) * 64) + ((n & 3) * 16 + jm)) * 2 + ((n >> 2) & 1)] = 0.0;
        // trailing-coefficient table [tile][q][rg] for the VALU term: codeword jm = 4*rg + q
        cbm[(long)MT * NP * 128 + (long)(m >> 4) * 16 + (jm & 3) * 4 + (jm >> 2)] = lastc;
    }
    if (real) atomicMax(l1max_bits, (u64)__double_as_longlong(l1));
}

// ---- the first pass after a split, seeded (round 3) -----------------------------------------------------------------------
// The children of codeword i are 2 i and 2 i + 1, and after the split 90-97 % of the frames of cell i land in one of
// the two (tools/probe/family_moves.py).  So the level's first pass need not accumulate in full: the rows start as
// "every frame in the even child of its old cell" (rows[2 i] = the parent's exact sums, rows[2 i + 1] = 0), a frame that
// lands in 2 i adds nothing, a frame that lands in 2 i + 1 adds its limbs ONCE -- to row i of a side table X --, and only
// a frame that leaves its family is moved with a subtraction and an addition.  k_family_fixup then moves X: rows[2 i + 1]
// += X[i], rows[2 i] -= X[i].  Exact 64-bit integers throughout: the rows equal a full accumulation bit for bit, at
// ~0.55-0.65 of its atomic traffic (1 add for ~46 % of the frames, 2 for the 3-10 % that leave, none for the rest).
__global__ void k_seed_family(const i64* __restrict__ parent, i64* __restrict__ rows, i64* __restrict__ X, int Mold, int NC,
                              int RS, ZeroList z)
{
    if (blockIdx.x == 0)  // (the pass's small words to zero: this kernel stands in for the prologue of a seeded pass)
        for (int k = 0; k < 3; ++k)
            for (int j = threadIdx.x; j < z.words[k]; j += blockDim.x) ((unsigned int*)z.p[k])[j] = 0u;
    const long n = (long)Mold * RS;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) {
        const long i = o / RS;
        const int e = (int)(o - i * RS);
        const i64 v = e <= 2 * NC ? parent[o] : 0;  // limb pairs and count; the distortion columns start at zero
        rows[(2 * i) * RS + e] = v;
        rows[(2 * i + 1) * RS + e] = 0;
        X[o] = 0;
    }
}

__global__ void k_family_fixup(i64* __restrict__ rows, const i64* __restrict__ X, int Mold, int NC, int RS)
{
    const long n = (long)Mold * RS;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long)gridDim.x * blockDim.x) {
        const long i = o / RS;
        const int e = (int)(o - i * RS);
        if (e > 2 * NC) continue;
        const i64 x = X[o];
        if (x == 0) continue;
        rows[(2 * i + 1) * RS + e] = (i64)((u64)rows[(2 * i + 1) * RS + e] + (u64)x);
        rows[(2 * i) * RS + e] = (i64)((u64)rows[(2 * i) * RS + e] - (u64)x);
    }
}

void launch_seed_family(const i64* parent, i64* rows, i64* X, int Mold, int NC, hipStream_t s, const ZeroList* zero)
{
    const int RS = row_stride(NC);
    ZeroList z{};
    if (zero) z = *zero;
    hipLaunchKernelGGL(k_seed_family, dim3(grid_for((long)Mold * RS, 256, 1024)), dim3(256), 0, s, parent, rows, X, Mold, NC, RS, z);
}

void launch_family_fixup(i64* rows, const i64* X, int Mold, int NC, hipStream_t s)
{
    const int RS = row_stride(NC);
    hipLaunchKernelGGL(k_family_fixup, dim3(grid_for((long)Mold * RS, 256, 1024)), dim3(256), 0, s, rows, X, Mold, NC, RS);
}

// incremental accumulation: the distortion elements of every row are rebuilt each pass, the cell sums persist
__global__ void k_zero_dist(i64* __restrict__ rows, int M, int NC, int RS)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 * M) rows[(long)(i >> 2) * RS + 2 * NC + 1 + (i & 3)] = 0;
}

void launch_zero_distortion_columns(i64* rows, int M, int NC, hipStream_t s)
{
    hipLaunchKernelGGL(k_zero_dist, dim3((4 * M + 255) / 256), dim3(256), 0, s, rows, M, NC, row_stride(NC));
}

// Level statistics -> host-mapped (fine-grained) pinned memory, then a sequence number: the host spins on the number
// instead of waiting for three small D2H copies and an event (tens of microseconds per pass, 45 passes per ladder).
__global__ void k_publish_stats(i64* __restrict__ lstats, const u64* __restrict__ l1max_bits,
                                const double* __restrict__ within, int M, i64* __restrict__ h_l, u64* __restrict__ h_l1,
                                double* __restrict__ h_within, volatile u64* __restrict__ h_seq, u64 seq)
{
    for (int i = threadIdx.x; i < 64 * 8; i += blockDim.x) {
        h_l[i] = lstats[i];
        lstats[i] = 0;  // ready for the next pass (saves a memset per pass)
    }
    for (int i = threadIdx.x; i < M; i += blockDim.x) h_within[i] = within[i];
    if (threadIdx.x == 0) *h_l1 = *l1max_bits;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        *h_seq = seq;
        __threadfence_system();
    }
}

void launch_publish_stats(i64* lstats, const u64* l1max_bits, const double* within, int M, i64* h_l, u64* h_l1,
                          double* h_within, u64* h_seq, u64 seq, hipStream_t s)
{
    hipLaunchKernelGGL(k_publish_stats, dim3(1), dim3(256), 0, s, lstats, l1max_bits, within, M, h_l, h_l1, h_within,
                       (volatile u64*)h_seq, seq);
}

// threads per block of the per-cell kernels: 3 LDS columns of NC doubles per thread must fit 64 KB
static inline int small_tpb(int NC) { return NC <= 42 ? 64 : (NC <= 84 ? 32 : 8); }

void launch_rows_stats(const i64* rows, int M, int NC, const DevScalars* sc, double* S, double* within, i64* lstats,
                       hipStream_t s)
{
    hipLaunchKernelGGL(k_rows_stats, dim3((M + 63) / 64), dim3(64), 0, s, rows, M, NC, sc, S, within, lstats);
}

void launch_centroids(const i64* rows, const double* S, int M, int NC, const double* refl_in, double* refl_out,
                      i64* lstats, hipStream_t s)
{
    const int tpb = small_tpb(NC);
    hipLaunchKernelGGL(k_centroids, dim3((M + tpb - 1) / tpb), dim3(tpb), (size_t)3 * NC * tpb * 8, s, rows, S, M, NC,
                       refl_in, refl_out, lstats);
}

void launch_finish_q(const i64* stats, int NC, DevScalars* sc, hipStream_t s)
{
    hipLaunchKernelGGL(k_finish_q, dim3(1), dim3(64), 0, s, stats, NC, sc);
}

// in-process multi-GPU exchange (reduce-scatter + all-gather in one kernel per rank): the launching rank owns words
// [lo, hi); it reads them from every rank's buffer (peer access over xGMI), combines (64-bit integer sum or unsigned
// max: exact, order-free) and writes the result back into every rank's buffer
// (round 4: every word of another rank's buffer is read and written with SYSTEM-scope atomic accesses -- they bypass this
// device's caches, so a peer device's earlier writes are seen and this kernel's results are visible to it without relying on
// what a kernel boundary writes back or invalidates across devices; ordering between the ranks' kernels is carried by the
// release-to-system events of local_allreduce.  Same-device ranks pay a few percent of a ~10 us kernel for it.)
__global__ void k_reduce_slice_i64(PeerBuffers bufs, int n, long lo, long hi, int op)
{
    for (long i = lo + (long)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (long)gridDim.x * blockDim.x) {
        i64 v = (i64)__hip_atomic_load((u64*)&bufs.p[0][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int k = 1; k < n; ++k) {
            const i64 o = (i64)__hip_atomic_load((u64*)&bufs.p[k][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            v = op == 0 ? (i64)((u64)v + (u64)o) : ((u64)o > (u64)v ? o : v);
        }
        for (int k = 0; k < n; ++k) __hip_atomic_store((u64*)&bufs.p[k][i], (u64)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

void launch_reduce_slice_i64(const PeerBuffers& bufs, int n, long lo, long hi, int op, hipStream_t s)
{
    if (hi <= lo) return;
    hipLaunchKernelGGL(k_reduce_slice_i64, dim3(grid_for(hi - lo, 256, 256)), dim3(256), 0, s, bufs, n, lo, hi, op);
}

bool has_cell_update(int NC) { return NC <= 64; }

// fused per-cell update (rows != nullptr) or codeword preparation only (rows == nullptr).  *l1max_bits (and
// *eC_biased, if given) must be zero when the kernel starts: zero_first adds a memset for callers that have no
// launch_pass_prologue in front.
void launch_cell_update(const i64* rows, int M, int NC, const DevScalars* sc, const double* refl_in, double* refl_out,
                        double* cbq, double* cbm, u64* l1max_bits, double* within, i64* lstats, hipStream_t s,
                        bool zero_first, const int* ea, int* eC_biased, const PublishArgs* pub)
{
    if (zero_first) (void)hipMemsetAsync(l1max_bits, 0, sizeof(u64), s);
    const int MT = (M + 15) / 16;
    PublishArgs p{};
    if (pub) p = *pub;
    const dim3 grid((M + CU_WAVES - 1) / CU_WAVES + (p.flags ? 1 : 0)), block(64 * CU_WAVES);
    if (NC == 37)  // P = 36, the reference's default order (src/lpc/mod.rs:17-74) and BASELINE's: straight-line code
        hipLaunchKernelGGL(k_cell_update<37>, grid, block, 0, s, rows, M, NC, sc, refl_in, refl_out, cbq, cbm, MT, l1max_bits, within,
                           lstats, ea, eC_biased, p);
    else
        hipLaunchKernelGGL(k_cell_update<0>, grid, block, 0, s, rows, M, NC, sc, refl_in, refl_out, cbq, cbm, MT, l1max_bits, within,
                           lstats, ea, eC_biased, p);
}

// One launch in front of a pass instead of up to four memsets: the rows (what = 1: every word, 2: the four
// distortion columns of every row, 0: nothing) and up to three small blocks of 4-byte words.
__global__ __launch_bounds__(256) void k_pass_prologue(i64* __restrict__ rows, int M, int NC, int RS, int what,
                                                       ZeroList z)
{
    if (what == 1) {
        const long n2 = (long)M * RS / 2;  // RS is a multiple of 8 words
        ulonglong2* r2 = (ulonglong2*)rows;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256)
            r2[i] = make_ulonglong2(0ull, 0ull);
    } else if (what == 2) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 4L * M; i += (long)gridDim.x * 256)
            rows[(i >> 2) * RS + 2 * NC + 1 + (i & 3)] = 0;
    }
    if (blockIdx.x == 0)
        for (int k = 0; k < 3; ++k)
            for (int j = threadIdx.x; j < z.words[k]; j += 256) ((unsigned int*)z.p[k])[j] = 0u;
}

void launch_pass_prologue(i64* rows, int M, int NC, int what, const ZeroList& z, hipStream_t s)
{
    const int RS = row_stride(NC);
    const long items = what == 1 ? (long)M * RS / 2 : (what == 2 ? 4L * M : 1);
    hipLaunchKernelGGL(k_pass_prologue, dim3(grid_for(items, 256, 1024)), dim3(256), 0, s, rows, M, NC, RS, what, z);
}

void launch_init_codebook(const i64* stats, int NC, const DevScalars* sc, double* reflections, int* status,
                          hipStream_t s)
{
    const int tpb = small_tpb(NC);
    hipLaunchKernelGGL(k_init_codebook, dim3(1), dim3(tpb), (size_t)3 * NC * tpb * 8, s, stats, NC, sc, reflections,
                       status);
}

void launch_grow(const double* old_refl, int M, int NC, double* new_refl, hipStream_t s, const ZeroList* zero)
{
    ZeroList z{};
    if (zero) z = *zero;
    hipLaunchKernelGGL(k_grow, dim3((M * NC + 255) / 256), dim3(256), 0, s, old_refl, M, NC, new_refl, z);
}

void launch_codebook_prepare(const double* reflections, int M, int NC, double* cbq, u64* l1max_bits, double* cbm,
                             hipStream_t s)
{
    (void)hipMemsetAsync(l1max_bits, 0, sizeof(u64), s);
    const int MT = (M + 15) / 16;
    const int n = cbm ? 16 * MT : M;
    const int tpb = small_tpb(NC);
    hipLaunchKernelGGL(k_codebook_prepare, dim3((n + tpb - 1) / tpb), dim3(tpb), (size_t)NC * tpb * 8, s, reflections, M, NC,
                       cbq, l1max_bits, cbm, MT);
}

}  // namespace e2vq
