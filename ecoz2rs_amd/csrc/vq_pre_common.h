// vq_pre_common.h -- what the prefiltered kernels share (vq_prefilter.hip: the fused quantize sweep and the round-2 / round-3
// accumulating kernels; vq_sweep.hip: the candidate sweep + finishing kernel of round 5): the K-slot packing of the f16 limb
// images (PrePack), the per-pass scalars, the key epilogue of a job, the LDS staging of a block of FP64 frames, the records.
// Device-internal; the derivation of the error bound is at the top of vq_prefilter.hip.
#pragma once
#include "vq_accum.h"
#include "vq_device.h"
#include "vq_fixed.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace e2vq {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));

template <int NC>
struct PrePack {
    static constexpr int NL = 3;                                  // limbs per value
    static constexpr int G = NC / 16, R = NC % 16;
    static constexpr int TAILP = (NL * R + 15) / 16;              // pairs holding the limbs' tails back to back
    static constexpr int PAIRS = NL * G + TAILP;                  // frame granule pairs (32 B each per frame)
    // tail pairs that contain slots of limbs 0..l (the tails are laid out limb after limb)
    __host__ __device__ static constexpr int tail_pairs_upto(int l) { return ((l + 1) * R + 15) / 16; }
    // k-steps of weight level lv = frame limb + codeword limb: every pair holding a frame limb <= lv
    __host__ __device__ static constexpr int level_steps(int lv) { return (lv + 1) * G + tail_pairs_upto(lv); }
    static constexpr int NSTEP = level_steps(0) + level_steps(1) + level_steps(2);
    // Unique codeword granules (round 4).  The granule a k-step needs holds codeword limb cl = lv - fl of the coefficients
    // of its pair: for the full pairs that depends on (cl, coefficient group g) only -- the same granule serves level cl
    // against frame limb 0, level cl + 1 against frame limb 1, ... --, so a tile image holds NL * G of those plus the tail
    // granules of each level (those do differ by level): NU granules instead of NSTEP.  NC = 37: 9 instead of 15 -- 9 KB
    // per codeword tile and wave through L2 and the texture path instead of 15, 36 operand registers instead of 60.
    __host__ __device__ static constexpr int tail_base(int lv)
    {
        return lv == 0 ? 0 : (lv == 1 ? tail_pairs_upto(0) : tail_pairs_upto(0) + tail_pairs_upto(1));
    }
    static constexpr int NU = NL * G + tail_base(2) + tail_pairs_upto(2);
    static constexpr int TILE_E = NU * 64;                        // h8 granules per 32-codeword tile image
    static constexpr int NCX = (NC + 1) & ~1;
    static_assert(NC >= 2 && 5 * NC * 65536 < (1 << 24), "partial sums must stay below 2^24");
    __host__ __device__ static constexpr int step_level(int s)
    {
        return s < level_steps(0) ? 0 : (s < level_steps(0) + level_steps(1) ? 1 : 2);
    }
    __host__ __device__ static constexpr int level_first(int lv)
    {
        return lv == 0 ? 0 : (lv == 1 ? level_steps(0) : level_steps(0) + level_steps(1));
    }
    // frame granule pair used by k-step s: the full pairs of limbs 0..lv first, then the tail pairs
    __host__ __device__ static constexpr int step_pair(int s)
    {
        const int lv = step_level(s), k = s - level_first(lv);
        return k < (lv + 1) * G ? k : NL * G + (k - (lv + 1) * G);
    }
    // unique granule (index into a tile image) used by k-step s
    __host__ __device__ static constexpr int step_unique(int s)
    {
        const int lv = step_level(s), pr = step_pair(s);
        if (pr < NL * G) {
            const int fl = pr / (G > 0 ? G : 1);
            return (lv - fl) * G + (pr - fl * G);
        }
        return NL * G + tail_base(lv) + (pr - NL * G);
    }
    // Order of the k-steps of a job: position i -> k-step.  (Any order gives the same accumulators: the partial sums are
    // exact integers.)  Level-major, as the steps are numbered: the accumulators of the higher levels come to life late in
    // the job, while the previous job's accumulators die value by value under the key epilogue -- granule-major (all uses of
    // a granule consecutive) starts all three levels at once and needs ~20 registers more, which k_pass_pre_lds does not have.
#ifndef E2VQ_PRE_ORDER
#define E2VQ_PRE_ORDER 0
#endif
    __host__ __device__ static constexpr int ord(int i)
    {
        if (E2VQ_PRE_ORDER == 0) return i;
        int c = 0;
        for (int u = 0; u < NU; ++u)
            for (int s = 0; s < NSTEP; ++s)
                if (step_unique(s) == u) {
                    if (c == i) return s;
                    ++c;
                }
        return -1;
    }
    // position of the last use of granule u in a job, and u's rank in the order of those positions: the order in which the
    // rotating operand loads of k_pass_pre_lds are issued (each behind the last reader of its registers)
    __host__ __device__ static constexpr int last_pos(int u)
    {
        int p = -1;
        for (int i = 0; i < NSTEP; ++i)
            if (step_unique(ord(i)) == u) p = i;
        return p;
    }
    __host__ __device__ static constexpr int issue_rank(int u)
    {
        int r = 0;
        for (int v = 0; v < NU; ++v)
            if (last_pos(v) < last_pos(u)) ++r;
        return r;
    }
    __host__ __device__ static constexpr bool pos_first_of_level(int i)
    {
        for (int j = 0; j < i; ++j)
            if (step_level(ord(j)) == step_level(ord(i))) return false;
        return true;
    }
    __host__ __device__ static constexpr bool pos_first_use(int i)
    {
        for (int j = 0; j < i; ++j)
            if (step_unique(ord(j)) == step_unique(ord(i))) return false;
        return true;
    }
    __host__ __device__ static constexpr bool pos_last_use(int i)
    {
        for (int j = i + 1; j < NSTEP; ++j)
            if (step_unique(ord(j)) == step_unique(ord(i))) return false;
        return true;
    }
    // content of unique granule u, element e of lane half h: codeword limb cl and coefficient n, or n = -1 (zero)
    __host__ __device__ static __forceinline__ void unique_slot(int u, int h, int e, int& cl, int& n)
    {
        if (u < NL * G) {
            cl = u / (G > 0 ? G : 1);
            n = 16 * (u - cl * G) + 8 * h + e;
        } else {
            const int t = u - NL * G;
            const int lv = t < tail_base(1) ? 0 : (t < tail_base(2) ? 1 : 2);
            int fl = 0;
            slot(NL * G + (t - tail_base(lv)), h, e, fl, n);
            cl = lv - fl;
            if (cl < 0 || cl > 2) n = -1;
        }
    }
    // element e of the granule (pair p, lane half h) of a frame: limb index fl and coefficient n, or n = -1 (zero)
    __host__ __device__ static __forceinline__ void slot(int p, int h, int e, int& fl, int& n)
    {
        if (p < NL * G) {
            fl = p / (G > 0 ? G : 1);
            n = 16 * (p - fl * G) + 8 * h + e;
        } else {
            const int k = 16 * (p - NL * G) + 8 * h + e;  // position in the run of tails
            fl = R > 0 ? k / R : 0;
            n = (R > 0 && k < NL * R) ? 16 * G + (k - fl * R) : -1;
        }
    }
};

// calls fn(integral_constant<p>) for p = 2 it + half, it = 0, 1, ... (p < PAIRS): the pair index is a compile-time
// constant inside fn although it depends on the runtime bit `half`
template <int PAIRS, int IT = 0, typename Fn>
__device__ __forceinline__ void pre_for_pairs(int half, Fn& fn)
{
    if constexpr (2 * IT < PAIRS) {
        if (half == 0)
            fn(std::integral_constant<int, 2 * IT>{});
        else if constexpr (2 * IT + 1 < PAIRS)
            fn(std::integral_constant<int, 2 * IT + 1>{});
        pre_for_pairs<PAIRS, IT + 1>(half, fn);
    }
}

// x in [-1, 1] -> the three integer limbs
__device__ __forceinline__ void pre_split(double x, int (&L)[3])
{
    const double s1 = x * 512.0, l1 = __builtin_rint(s1);
    const double s2 = (s1 - l1) * 512.0, l2 = __builtin_rint(s2);
    const double s3 = (s2 - l2) * 512.0, l3 = __builtin_rint(s3);
    L[0] = (int)l1;
    L[1] = (int)l2;
    L[2] = (int)l3;
}

// per-pass scalars, zeroed by one memset before the codebook image is built
struct PreScalars {
    int eC_biased;  // codebook scale C = 2^eC > max |c[m][n]| a_n, stored as eC + PRE_EBIAS (0 = empty)
    int ymax_bits;  // max_m sum_n |eta[m][n]| as float bits (positive -> ordered like ints)
    int fb_count;   // frames handed to the fallback sweep
};
constexpr int PRE_EBIAS = E2VQ_PRE_EBIAS;

__device__ __forceinline__ float med3f(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// ---- the tile loop's building blocks (shared by k_pass_pre and k_pass_pre_lds) --------------------------------------
// One *job* = the NSTEP MFMAs of (codeword tile, 32-frame column block) interleaved with the key epilogue of the
// previous job: per value two fmas (the three weight levels -> v), v_and_or (the codeword index into the low mantissa
// bits) and three v_med3 (running min / 2nd / 3rd of the keys): 96 VALU operations per 15 MFMAs at NC = 37 --
// 1 MFMA (32 matrix-pipe cycles, 8 of them blocking issue) : 6 VALU operations, a balanced stream for two waves per SIMD.
// The order is spelled out (MFMA, its share of the epilogue, sched_barrier): with the codeword tiles loaded by inline asm
// in other basic blocks, the pipeline solver behind sched_group_barrier left whole jobs unpinned (round 3).
//
// The k-steps run in granule-major order (PrePack::ord): the partial sums are exact integers, so any order gives the same
// accumulators, and this one makes every operand granule's uses consecutive.
// the share of the previous job's key epilogue that rides behind the MFMA at position I
template <int NC, int I>
__device__ __forceinline__ void pre_epilogue_slice(const f16v (&PREV)[3], int ptile, float& k1, float& k2, float& k3, int maskv,
                                                   float ninf)
{
    typedef PrePack<NC> PK;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r * PK::NSTEP / 16 != I) continue;
        // (an opaque scalar: seen as tile * 32 | constant the compiler splits the v_and_or into v_and + v_or3 -- a seventh
        // VALU operation per value)
        int sidx = ptile * 32 + 8 * (r >> 2) + (r & 3);
        asm("" : "+s"(sidx));
        const float v = __builtin_fmaf(PREV[0][r], 262144.f, __builtin_fmaf(PREV[1][r], 512.f, PREV[2][r]));
        const float key = __int_as_float((__float_as_int(v) & maskv) | sidx);
        k3 = med3f(k2, k3, key);
        k2 = med3f(k1, k2, key);
        k1 = med3f(k1, key, ninf);
    }
}
// A lower bound of a frame's FOURTH-smallest key from the three smallest each lane half kept for it (a1 <= a2 <= a3 of the one
// half's codewords, b1 <= b2 <= b3 of the other's): every key a half did not keep is >= its third, so the fourth smallest of
// all is at least the fourth smallest of {a1, a2, a3, a3, b1, b2, b3, b3} = min_i max(A_i, B_(4-i)) = min(a3, b3, max(a2, b2)).
// (The three smallest of all are exact: a key that was not kept has three kept ones of its own half below it.)
__device__ __forceinline__ float pre_fourth_bound(float a2, float a3, float b2, float b3)
{
    return __builtin_fminf(__builtin_fminf(a3, b3), __builtin_fmaxf(a2, b2));
}
// the whole epilogue of a job at once (behind the last tile of a block)
template <int NC>
__device__ __forceinline__ void pre_epilogue(const f16v (&PREV)[3], int ptile, float& k1, float& k2, float& k3, int maskv, float ninf)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // (an opaque scalar: seen as tile * 32 | constant the compiler splits the v_and_or into v_and + v_or3 -- a seventh
        // VALU operation per value)
        int sidx = ptile * 32 + 8 * (r >> 2) + (r & 3);
        asm("" : "+s"(sidx));
        const float v = __builtin_fmaf(PREV[0][r], 262144.f, __builtin_fmaf(PREV[1][r], 512.f, PREV[2][r]));
        const float key = __int_as_float((__float_as_int(v) & maskv) | sidx);
        k3 = med3f(k2, k3, key);
        k2 = med3f(k1, k2, key);
        k1 = med3f(k1, key, ninf);
    }
}

// WAIT: 0 = the tile's operands are known to be there; N > 0: granule u -- the r-th (r = PrePack::issue_rank(u)) of its
// tile's NU operand loads, all issued by an earlier job -- is waited for in front of its first use with s_waitcnt vmcnt(N - 1 - r):
// the vector-memory counter is in-order, so N - 1 - u younger requests may still be out (N = NU: nothing younger than the
// tile's own loads; N = 2 NU: the next tile's NU loads as well), and anything else in flight -- the previous block's
// atomics, the LDS-DMA of this block's frames -- is older and only makes the wait longer, never too short.
// LOADS: behind the last use of granule u in this job -- the last reader of its registers in the tile -- the same
// registers are requested for the tile whose image starts at `next` (uniform), lane offset `lo`.
// (s_nop 4 in front of every asm load: the base address may have been restored from a spilled SGPR by v_readlane just
// before -- a VALU write of an SGPR needs five wait states before a vector-memory instruction reads it, the hardware does
// not interlock that, and the compiler's hazard recogniser does not look into inline asm: without the nops the first load
// of a tile went to a garbage address whenever register pressure had put the tile pointer into a VGPR lane.)
template <int NC, int WAIT, bool LOADS, int I = 0>
__device__ __forceinline__ void pre_job(f16v (&ACC)[3], const h8 (&BC)[PrePack<NC>::PAIRS], h8 (&A)[PrePack<NC>::NU],
                                        const f16v (&PREV)[3], int ptile, float& k1, float& k2, float& k3, int maskv, float ninf,
                                        const char* next, unsigned lo)
{
    typedef PrePack<NC> PK;
    // (a recursive template, not a generic lambda over the positions: clang rejects captured variables as asm operands)
    if constexpr (I < PK::NSTEP) {
        constexpr int S = PK::ord(I), LV = PK::step_level(S), PR = PK::step_pair(S), U = PK::step_unique(S);
        constexpr bool FIRST = PK::pos_first_of_level(I);
        const f16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (WAIT > 0 && PK::pos_first_use(I))
            asm volatile("s_waitcnt vmcnt(%1)" : "+v"(A[U]) : "n"(WAIT - 1 - PK::issue_rank(U)) : "memory");
        ACC[LV] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[U], BC[PR], FIRST ? zero : ACC[LV], 0, 0, 0);
        if constexpr (LOADS && PK::pos_last_use(I)) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3"
                         : "=&v"(A[U])
                         : "v"(lo), "s"(next + (U >> 2) * 4096), "n"((U & 3) * 1024)
                         : "memory");
        }
        pre_epilogue_slice<NC, I>(PREV, ptile, k1, k2, k3, maskv, ninf);
        __builtin_amdgcn_sched_barrier(0);
        pre_job<NC, WAIT, LOADS, I + 1>(ACC, BC, A, PREV, ptile, k1, k2, k3, maskv, ninf, next, lo);
    }
}

// the compiler-scheduled form of a job (k_pass_pre: its operand loads are plain C++ and the interleave is pinned with
// sched_group_barrier: one MFMA, then its share of the 96 epilogue operations)
template <int NC>
__device__ __forceinline__ void pre_job_pinned(f16v (&ACC)[3], const h8 (&BC)[PrePack<NC>::PAIRS], const h8 (&A)[PrePack<NC>::NU],
                                               const f16v (&PREV)[3], int ptile, float& k1, float& k2, float& k3, int maskv,
                                               float ninf)
{
    typedef PrePack<NC> PK;
    const f16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < PK::NSTEP; ++s) {
        const int lv = PK::step_level(s), pr = PK::step_pair(s);
        const bool first = s == PK::level_first(lv);
        ACC[lv] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[PK::step_unique(s)], BC[pr], first ? zero : ACC[lv], 0, 0, 0);
    }
    pre_epilogue<NC>(PREV, ptile, k1, k2, k3, maskv, ninf);
#pragma unroll
    for (int s = 0; s < PK::NSTEP; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 96 / PK::NSTEP, 0);
    }
}

template <int NC>
struct PreLds {
    static constexpr int STAGE_BYTES = 64 * NC * 8;  // the block's frames, row-major
    static constexpr int AUX_BYTES = 256 + 256;      // tolerance terms (64 floats); cells of the previous pass (64 u16, read as 64 dwords)
    static constexpr int WAVE_BYTES = STAGE_BYTES + AUX_BYTES;
    static constexpr int NE = 2 * NC + 1;            // elements of a frame's contribution: limb pairs + count
    // waves per workgroup: eight (two per SIMD) while their regions fit; P = 40 (21 KB per wave) runs seven
    static constexpr int FIT = (E2VQ_LDS_BYTES - 512) / WAVE_BYTES;
    static constexpr int WAVES = FIT >= 8 ? 8 : FIT;
    static constexpr bool OK = WAVES >= 6 && NC < 64;
    // the burst of atomics (ACC = 1) and k_accum_ranges add a row with one 64-lane instruction + one carrying four 16-lane
    // tails: rows of at most 80 elements (P <= 39); longer rows are recorded (ACC = 2) or take the round-2 kernel
    static constexpr bool BURST_OK = OK && NE <= 80;
};

// requests block b of the row-major frames (padded with zero rows to whole blocks), its tolerance terms and the cells
// of the previous pass into the wave's LDS region: LDS-DMA, 1 KB / 256 B per instruction, no registers
template <int NC, bool WITH_FG = true>
__device__ __forceinline__ void pre_lds_request(const double* __restrict__ aos, const float* __restrict__ fg,
                                                const unsigned short* __restrict__ prev_sym, long b, int lane,
                                                unsigned char* wbase)
{
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int BYTES = 64 * NC * 8, K16 = BYTES / 1024, K4 = (BYTES - K16 * 1024) / 256;
    static_assert(K16 * 1024 + K4 * 256 == BYTES, "a block of frames is a whole number of 256-byte pieces");
    const char* g = (const char*)(aos + b * (long)(64 * NC));
#pragma unroll
    for (int k = 0; k < K16; ++k)
        __builtin_amdgcn_global_load_lds((gptr_t)(g + k * 1024 + lane * 16), (lptr_t)(wbase + k * 1024), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < K4; ++k)
        __builtin_amdgcn_global_load_lds((gptr_t)(g + K16 * 1024 + k * 256 + lane * 4),
                                         (lptr_t)(wbase + K16 * 1024 + k * 256), 4, 0, 0);
    if constexpr (WITH_FG)
        __builtin_amdgcn_global_load_lds((gptr_t)((const char*)(fg + b * 64) + lane * 4), (lptr_t)(wbase + BYTES), 4, 0, 0);
    if (prev_sym)  // (64 dwords: the block's 64 cells and 128 bytes beyond them, which the array is padded for)
        __builtin_amdgcn_global_load_lds((gptr_t)((const char*)(prev_sym + b * 64) + lane * 4),
                                         (lptr_t)(wbase + BYTES + 256), 4, 0, 0);
}

// the lane index straight from the hardware, in a form the compiler can neither hoist nor share between uses
__device__ __forceinline__ int pre_fresh_lane()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)::"memory");
    return l;
}

// a whole codeword tile requested at once by inline asm, and the wait for it (the simple tile loop of odd tile counts)
template <int NC, int U = 0>
__device__ __forceinline__ void pre_load_tile_issue(h8 (&A)[PrePack<NC>::NU], const char* tile, unsigned lo)
{
    if constexpr (U < PrePack<NC>::NU) {
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3"
                     : "=&v"(A[U])
                     : "v"(lo), "s"(tile + (U >> 2) * 4096), "n"((U & 3) * 1024)
                     : "memory");
        pre_load_tile_issue<NC, U + 1>(A, tile, lo);
    }
}
template <int NC, int U = 0>
__device__ __forceinline__ void pre_load_tile_wait(h8 (&A)[PrePack<NC>::NU])
{
    if constexpr (U < PrePack<NC>::NU) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[U])::"memory");  // (cheap: the counter is zero after the first)
        pre_load_tile_wait<NC, U + 1>(A);
    }
}
template <int NC>
__device__ __forceinline__ void pre_load_tile_asm(h8 (&A)[PrePack<NC>::NU], const char* tile, unsigned lo)
{
    pre_load_tile_issue<NC>(A, tile, lo);
    pre_load_tile_wait<NC>(A);
}

// records of the ACC = 2 pass (vq_device.h: PassRecords), as the kernels see them
struct PreRec {
    uint2* recs;
    int* counts;
    int nbins, nbins_rows, bin_cells, cap;
    unsigned magic;
    long long* total_out;
};

}  // namespace e2vq
