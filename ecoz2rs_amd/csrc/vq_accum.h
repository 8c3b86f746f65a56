// vq_accum.h -- the accumulate half of a pass (K2), shared by the FP64 sweep kernel (vq_device.hip) and the
// prefiltered sweep kernel (vq_prefilter.hip): a wave's 64 frames, resident in registers in the FP64 MFMA operand
// layout, are added to their cells as exact 64-bit integers.
#pragma once
#include "vq_device.h"
#include "vq_fixed.h"

#include <hip/hip_runtime.h>

namespace e2vq {

typedef long long i64;
typedef unsigned long long u64;
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int IMG_STRIDE_PAD = 3;  // image row stride 2*NC+5+pad: 82 dwords for NC=37 (conflict-free b64 writes)

// cells of the hybrid (MODE 5) LDS table: what fits beside the eight per-wave row images, multiple of 8
__host__ __device__ constexpr int mfma_hyb_cells(int NC)
{
    return (int)((E2VQ_LDS_BYTES - 10240 - 8 * 16 * (2 * NC + 5 + IMG_STRIDE_PAD) * 4) / (((2 * NC + 5 + 7) & ~7) * 8)) & ~7;
}

// offset (in doubles) of r[t][n] in the blocked FP64 MFMA frame layout written by k_blockify_mfma
__host__ __device__ __forceinline__ long mfma_blk_offset(int NC, long t, int n)
{
    const int NS = (NC + 3) >> 2;
    const long b = t >> 6;
    const int w = (int)(t & 63), u = w >> 5, h = (w >> 4) & 1, j = w & 15;
    const int st = n >> 2 < NS - 1 ? n >> 2 : NS - 1;
    const int x = st * 128 + ((n - 4 * st) * 16 + j) * 2 + h;
    return b * (long)NC * 64 + (long)u * NC * 32 + x;
}

// The 64 frames of block b from the blocked layout into the FP64 MFMA B operands: Bf[ft][st], lane (q, j), holds
// r[frame 16 ft + j][4 st + q]; with NC = 4k+1 every q lane holds r[NC-1] in the last slot (it is applied on the VALU).
// (one half of the block: frames 32 u .. 32 u + 31 -> Bf[2u], Bf[2u + 1])
template <int NC>
__device__ __forceinline__ void load_block_frames_half(const double* __restrict__ blk, long b, int lane, int u,
                                                       double (&Bf)[4][2 * ((((NC + 3) / 4) + 1) / 2)])
{
    constexpr int NS = (NC + 3) / 4, REM = NC - 4 * (NS - 1);
    const int q = lane >> 4, j = lane & 15;
    const double* base = blk + b * (long)(NC * 64) + u * (NC * 32);
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) {
        const double2 v = *(const double2*)(base + (st * 64 + lane) * 2);
        Bf[2 * u][st] = v.x;
        Bf[2 * u + 1][st] = v.y;
    }
    double2 v = make_double2(0.0, 0.0);
    if (REM == 1)  // every lane of frame j keeps r[NC-1] (the four q lanes read the same 16 B)
        v = *(const double2*)(base + (NS - 1) * 128 + j * 2);
    else if (q < REM)
        v = *(const double2*)(base + (NS - 1) * 128 + (q * 16 + j) * 2);
    Bf[2 * u][NS - 1] = v.x;
    Bf[2 * u + 1][NS - 1] = v.y;
}

template <int NC>
__device__ __forceinline__ void load_block_frames(const double* __restrict__ blk, long b, int lane,
                                                  double (&Bf)[4][2 * ((((NC + 3) / 4) + 1) / 2)])
{
    load_block_frames_half<NC>(blk, b, lane, 0, Bf);
    load_block_frames_half<NC>(blk, b, lane, 1, Bf);
}

// the same operands straight from the row-major .prd payload (quantize: no blocked copy is made); frames >= T read as 0
template <int NC>
__device__ __forceinline__ void load_block_frames_rowmajor(const double* __restrict__ aos, long b, long T, int lane,
                                                           double (&Bf)[4][2 * ((((NC + 3) / 4) + 1) / 2)])
{
    constexpr int NS = (NC + 3) / 4, REM = NC - 4 * (NS - 1);
    const int q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
        const long t = b * 64 + 16 * ft + j;
        const bool ok = t < T;
        const double* row = aos + (ok ? t : 0) * NC;
#pragma unroll
        for (int st = 0; st < NS - 1; ++st) Bf[ft][st] = ok ? row[4 * st + q] : 0.0;
        Bf[ft][NS - 1] = !ok ? 0.0 : (REM == 1 ? row[NC - 1] : (q < REM ? row[4 * (NS - 1) + q] : 0.0));
#pragma unroll
        for (int st = NS; st < 2 * ((NS + 1) / 2); ++st) Bf[ft][st] = 0.0;
    }
}

// the same operands from a wave's 64 row-major frames staged in LDS (fused quantize; frames >= T were staged as zeros)
template <int NC>
__device__ __forceinline__ void load_block_frames_stage(const double* stage, int lane,
                                                        double (&Bf)[4][2 * ((((NC + 3) / 4) + 1) / 2)])
{
    constexpr int NS = (NC + 3) / 4, REM = NC - 4 * (NS - 1);
    const int q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
        const double* row = stage + (16 * ft + j) * NC;
#pragma unroll
        for (int st = 0; st < NS - 1; ++st) Bf[ft][st] = row[4 * st + q];
        Bf[ft][NS - 1] = REM == 1 ? row[NC - 1] : (q < REM ? row[4 * (NS - 1) + q] : 0.0);
#pragma unroll
        for (int st = NS; st < 2 * ((NS + 1) / 2); ++st) Bf[ft][st] = 0.0;
    }
}

// Bf[ft][st]: lane (q, j) holds r[frame 16 ft + j][4 st + q] (with NC = 4k+1 every q lane holds r[NC-1] in the
// last slot); best / idx: min distortion and cell of frame 16 ft + j, in all four q lanes.  img: this wave's 16 row
// images in LDS.  MODE 1: all cells in the LDS table lacc; 5: cells < lds_cells there; 2: global atomics; 3: none.
// SKIP: frames flagged in skip[] (handled by the fallback launch) contribute an all-zero image (their idx must be 0).
// NFT: 16-frame tiles per block (4 everywhere but in the one-tile-per-wave fallback sweep).
// INCR (MODE 2 only): the rows persist from the previous pass over the same frames and codebook size.  When `incr`
// is set, a frame whose cell did not change (oldidx == idx) adds only its distortion elements (those are rebuilt
// every pass); a frame that moved is subtracted from its old cell and added to the new one.  64-bit integer sums
// are exactly invertible, so the rows equal those of a full accumulation bit for bit -- at a fraction of the atomics.
// DSEP (prefiltered sweep): the four distortion elements of a frame are NOT part of its row image -- the caller sums
// them per wave in registers and adds them to the distortion columns once at the end (only the column totals of
// those elements are ever used: the level statistics); a tile of an incremental pass in which no frame moved then has
// nothing to add at all.
template <int NC, int MODE, bool SKIP = false, int NFT = 4, bool INCR = false, bool DSEP = false>
__device__ __forceinline__ void accumulate_block(const double (&Bf)[4][2 * ((((NC + 3) / 4) + 1) / 2)],
                                                 const double (&best)[4], const int (&idx)[4], int* __restrict__ img,
                                                 i64* __restrict__ lacc, i64* __restrict__ rows, int lds_cells, int sh_r,
                                                 int sh_d, int sh_d2, long b, long T, int lane, const bool (&skip)[4],
                                                 bool incr = false, const int (&oldidx)[4] = {0, 0, 0, 0})
{
    static_assert(!INCR || MODE == 2, "incremental accumulation goes through global atomics");
    constexpr int NS = (NC + 3) / 4, REM = NC - 4 * (NS - 1);
    constexpr int RS = (2 * NC + 5 + 7) & ~7;
    constexpr int NE = DSEP ? 2 * NC + 1 : 2 * NC + 5;  // elements of a row image that are added
    constexpr int IMG = 2 * NC + 5 + IMG_STRIDE_PAD;
    constexpr int HYB_CELLS = mfma_hyb_cells(NC);
    const int q = lane >> 4, j = lane & 15;
    // (fix2_mul: same limbs as fix2 at a fraction of its cost, whenever 2^sh_r is a normal double)
    const bool fast_fix = sh_r >= -1000 && sh_r <= 1000;
    const double scale_r = ldexp(1.0, fast_fix ? sh_r : 0);
#pragma unroll
    for (int ft = 0; ft < NFT; ++ft) {
        int* my = img + j * IMG;
        // incremental pass: the limbs are needed only if a frame of this tile changed cell (wave-uniform test)
        const bool limbs = !INCR || !incr || __ballot(oldidx[ft] != idx[ft]) != 0;
        if (DSEP && !limbs) continue;  // (wave-uniform) nothing moves, and the distortions are the caller's business
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            if (limbs && (st < NS - 1 || q < REM)) {  // with TAILV only q = 0 writes r[NC-1] (all q lanes hold it)
                int hi, lo;
                if (fast_fix)
                    fix2_mul(Bf[ft][st], scale_r, hi, lo);
                else
                    fix2(Bf[ft][st], sh_r, hi, lo);
                if (SKIP && skip[ft]) hi = lo = 0;
                *(int2*)&my[2 * (4 * st + q)] = make_int2(hi, lo);
            }
        }
        if (q == 0 && DSEP) my[2 * NC] = (SKIP && skip[ft]) ? 0 : 1;
        if (q == 0 && !DSEP) {
            const double e = (SKIP && skip[ft]) ? 0.0 : best[ft] - 1.0;
            int hi, lo;
            my[2 * NC] = (SKIP && skip[ft]) ? 0 : 1;
            fix2(e, sh_d, hi, lo);
            my[2 * NC + 1] = hi;
            my[2 * NC + 2] = lo;
            fix2(e * e, sh_d2, hi, lo);
            my[2 * NC + 3] = hi;
            my[2 * NC + 4] = lo;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // padding frames (t >= T) are never counted; nv is wave-uniform
        const long left = T - (b * (16 * NFT) + ft * 16);
        const int nv = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
        if (nv == 16 && NE <= 80) {
            // full tile, 4 frames per step: four adds of elements 0..63 of each frame's row and, when the
            // row is longer (64 < NE <= 80), ONE add carrying the four row tails (lanes 16k.. -> frame k),
            // so LDS reads and atomics of different frames overlap and no lane-divergent branch remains.
            constexpr bool HAS_TAIL = NE > 64;  // (rows longer than 80 elements take the per-frame loop below)
            const int tq = lane >> 4, te = lane & 15;
#pragma unroll
            for (int j0 = 0; j0 < 16; j0 += 4) {
                int v[4], cell[4], old[4] = {0, 0, 0, 0};
                bool chg[4] = {true, true, true, true};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    cell[k] = __builtin_amdgcn_readlane(idx[ft], j0 + k);
                    v[k] = (HAS_TAIL || lane < NE) ? img[(j0 + k) * IMG + lane] : 0;
                    if constexpr (INCR) {
                        old[k] = __builtin_amdgcn_readlane(oldidx[ft], j0 + k);
                        chg[k] = old[k] != cell[k];  // wave-uniform
                    }
                }
                const int tv = (HAS_TAIL && te < NE - 64) ? img[(j0 + tq) * IMG + 64 + te] : 0;
                const int tcell = tq == 0 ? cell[0] : tq == 1 ? cell[1] : tq == 2 ? cell[2] : cell[3];
                const int told = tq == 0 ? old[0] : tq == 1 ? old[1] : tq == 2 ? old[2] : old[3];
                const bool tchg = tq == 0 ? chg[0] : tq == 1 ? chg[1] : tq == 2 ? chg[2] : chg[3];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!HAS_TAIL && lane >= NE) continue;  // short rows: lanes beyond the row sit out
                    if constexpr (MODE == 3) {
                        asm volatile("" ::"v"(v[k]), "s"(cell[k]));
                    } else if constexpr (MODE == 1) {
                        atomicAdd((u64*)&lacc[cell[k] * RS + lane], (u64)(i64)v[k]);
                    } else if constexpr (MODE == 5) {
                        if (cell[k] < HYB_CELLS)  // wave-uniform
                            atomicAdd((u64*)&lacc[cell[k] * RS + lane], (u64)(i64)v[k]);
                        else
                            atomicAdd((u64*)&rows[(long)cell[k] * RS + lane], (u64)(i64)v[k]);
                    } else if constexpr (INCR) {
                        const bool mov = lane <= 2 * NC;  // limbs and count move with the frame; distortions do not
                        if (!incr || chg[k] || !mov) atomicAdd((u64*)&rows[(long)cell[k] * RS + lane], (u64)(i64)v[k]);
                        if (incr && chg[k] && mov) atomicAdd((u64*)&rows[(long)old[k] * RS + lane], (u64)(-(i64)v[k]));
                    } else {
                        atomicAdd((u64*)&rows[(long)cell[k] * RS + lane], (u64)(i64)v[k]);
                    }
                }
                if (HAS_TAIL && te < NE - 64) {
                    if constexpr (MODE == 3) {
                        asm volatile("" ::"v"(tv), "v"(tcell));
                    } else if constexpr (MODE == 1) {
                        atomicAdd((u64*)&lacc[tcell * RS + 64 + te], (u64)(i64)tv);
                    } else if constexpr (MODE == 5) {
                        if (tcell < HYB_CELLS)  // per 16-lane group
                            atomicAdd((u64*)&lacc[tcell * RS + 64 + te], (u64)(i64)tv);
                        else
                            atomicAdd((u64*)&rows[(long)tcell * RS + 64 + te], (u64)(i64)tv);
                    } else if constexpr (INCR) {
                        const bool mov = 64 + te <= 2 * NC;
                        if (!incr || tchg || !mov) atomicAdd((u64*)&rows[(long)tcell * RS + 64 + te], (u64)(i64)tv);
                        if (incr && tchg && mov) atomicAdd((u64*)&rows[(long)told * RS + 64 + te], (u64)(-(i64)tv));
                    } else {
                        atomicAdd((u64*)&rows[(long)tcell * RS + 64 + te], (u64)(i64)tv);
                    }
                }
            }
        } else {
            for (int jj = 0; jj < nv; ++jj) {
                const int cell = __builtin_amdgcn_readlane(idx[ft], jj);
                const int* im = img + jj * IMG;
                if constexpr (INCR) {
                    const int old = __builtin_amdgcn_readlane(oldidx[ft], jj);
                    const bool chg = old != cell;
                    i64* row = rows + (long)cell * RS;
                    i64* orow = rows + (long)old * RS;
#pragma unroll
                    for (int e0 = 0; e0 < NE; e0 += 64) {
                        const int e = e0 + lane;
                        if (e < NE) {
                            const bool mov = e <= 2 * NC;
                            if (!incr || chg || !mov) atomicAdd((u64*)&row[e], (u64)(i64)im[e]);
                            if (incr && chg && mov) atomicAdd((u64*)&orow[e], (u64)(-(i64)im[e]));
                        }
                    }
                } else if (cell < lds_cells) {  // wave-uniform (two branches: an LDS and a global address are not mixed in one pointer)
                    i64* row = lacc + cell * RS;
#pragma unroll
                    for (int e0 = 0; e0 < NE; e0 += 64) {  // (rows of any length: P = 80 has 167 elements)
                        const int e = e0 + lane;
                        if (e < NE) atomicAdd((u64*)&row[e], (u64)(i64)im[e]);
                    }
                } else {
                    i64* row = rows + (long)cell * RS;
#pragma unroll
                    for (int e0 = 0; e0 < NE; e0 += 64) {
                        const int e = e0 + lane;
                        if (e < NE) atomicAdd((u64*)&row[e], (u64)(i64)im[e]);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

}

// ---- register accumulate (MODE 7: M <= 16, rows of at most 80 elements) ------------------------------------------------
// The sums of a wave's frames per cell are a product (cells x frames) . (frames x row elements) with a one-hot left factor,
// and the row elements are 32-bit integers: written as four signed byte digits they go through the i8 matrix unit exactly
// (i32 accumulators), 32 frames x 16 elements x 16 cells per v_mfma_i32_16x16x32_i8, and the accumulators stay in registers
// until the wave has swept all its blocks -- no atomic of any kind inside the block loop (the LDS table of MODE 1 costs 80
// ds_add_u64 per 64 frames, two thirds of that kernel's time at M <= 16).  v = d0 + 256 d1 + 65536 d2 + 2^24 d3 with
// d0..d2 in [-128, 127]: the bytes of (v + 0x808080) ^ 0x808080 (|v| <= 2^30, so the sum cannot wrap).  |digit sum| <=
// 128 * frames of the wave: the caller flushes before 2^23 frames.
// Image of a half block: img[f][RegAcc::STRIDE] ints, f = 0..31 (frame 32 u + f of the block), digits already applied;
// cellb[f]: cell of frame f, 0xFF for padding frames (no cell: the frame's column of the one-hot factor is empty).
template <int NC>
struct RegAcc {
    static constexpr int NE = 2 * NC + 5;
    static constexpr int NET = (NE + 15) / 16;      // element tiles of 16
    static constexpr int STRIDE = 16 * NET + 2;     // = 2 mod 16: conflict-free limb writes and digit reads
    static constexpr int WAVE_INTS = 32 * STRIDE + 8;  // + 32 cell bytes
    static constexpr bool OK = NE <= 80;
};

typedef int i4 __attribute__((ext_vector_type(4)));

// fix2 by one multiplication whenever 2^sh is a normal double (always, for data a .prd file can hold), else by ldexp
struct FixScale {
    int sh;
    double scale;
    bool fast;
};
__device__ __forceinline__ FixScale fix_scale(int sh)
{
    FixScale f;
    f.sh = sh;
    f.fast = sh >= -1000 && sh <= 1000;
    f.scale = ldexp(1.0, f.fast ? sh : 0);
    return f;
}
template <bool FAST>
__device__ __forceinline__ void fix2_by(double x, const FixScale& f, int& hi, int& lo)
{
    if (FAST)
        fix2_mul(x, f.scale, hi, lo);
    else
        fix2(x, f.sh, hi, lo);
}

// frames 32 u .. 32 u + 31 of block b: H[h][st], lane (q, j), holds r[frame 32 u + 16 h + j][4 st + q]
template <int NC>
__device__ __forceinline__ void load_half_frames(const double* __restrict__ blk, long b, int u, int lane,
                                                 double (&H)[2][2 * ((((NC + 3) / 4) + 1) / 2)])
{
    constexpr int NS = (NC + 3) / 4, REM = NC - 4 * (NS - 1);
    const int q = lane >> 4, j = lane & 15;
    const double* base = blk + b * (long)(NC * 64) + u * (NC * 32);
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) {
        const double2 v = *(const double2*)(base + (st * 64 + lane) * 2);
        H[0][st] = v.x;
        H[1][st] = v.y;
    }
    double2 v = make_double2(0.0, 0.0);
    if (REM == 1)
        v = *(const double2*)(base + (NS - 1) * 128 + j * 2);
    else if (q < REM)
        v = *(const double2*)(base + (NS - 1) * 128 + (q * 16 + j) * 2);
    H[0][NS - 1] = v.x;
    H[1][NS - 1] = v.y;
#pragma unroll
    for (int st = NS; st < 2 * ((NS + 1) / 2); ++st) H[0][st] = H[1][st] = 0.0;
}

__device__ __forceinline__ int digit_bytes(int v) { return (v + 0x00808080) ^ 0x00808080; }

// H[h][st]: lane (q, j) holds r[frame 16 h + j of the half block][4 st + q]; best / idx as in accumulate_block; t0: index of
// the half block's first frame
template <int NC, bool FAST>
__device__ __forceinline__ void regacc_stage_half(const double (&H)[2][2 * ((((NC + 3) / 4) + 1) / 2)],
                                                  const double (&best)[2], const int (&idx)[2], int* __restrict__ img,
                                                  const FixScale& fr, const FixScale& fd, const FixScale& fd2, long t0, long T,
                                                  int lane)
{
    constexpr int NS = (NC + 3) / 4, REM = NC - 4 * (NS - 1);
    constexpr int STRIDE = RegAcc<NC>::STRIDE;
    const int q = lane >> 4, j = lane & 15;
    unsigned char* cellb = (unsigned char*)(img + 32 * STRIDE);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int* my = img + (16 * h + j) * STRIDE;
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            if (st < NS - 1 || q < REM) {
                int hi, lo;
                fix2_by<FAST>(H[h][st], fr, hi, lo);
                *(int2*)&my[2 * (4 * st + q)] = make_int2(digit_bytes(hi), digit_bytes(lo));
            }
        }
        if (q == 0) {
            const double e = best[h] - 1.0;
            int hi, lo;
            my[2 * NC] = 1;
            fix2_by<FAST>(e, fd, hi, lo);
            my[2 * NC + 1] = digit_bytes(hi);
            my[2 * NC + 2] = digit_bytes(lo);
            fix2_by<FAST>(e * e, fd2, hi, lo);
            my[2 * NC + 3] = digit_bytes(hi);
            my[2 * NC + 4] = digit_bytes(lo);
            cellb[16 * h + j] = (t0 + 16 * h + j < T) ? (unsigned char)idx[h] : (unsigned char)0xFF;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 0x01 in every byte of x that equals the byte c (c replicated four times in c4), 0 elsewhere
__device__ __forceinline__ unsigned bytes_equal(unsigned x, unsigned c4)
{
    const unsigned y = x ^ c4;
    const unsigned t = (y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return (~(t | y | 0x7F7F7F7Fu)) >> 7;
}

template <int NC>
__device__ __forceinline__ void regacc_add_half(const int* __restrict__ img, int lane, i4 (&racc)[RegAcc<NC>::NET][4])
{
    constexpr int STRIDE = RegAcc<NC>::STRIDE, NET = RegAcc<NC>::NET;
    const int c = lane & 15, kg = lane >> 4;
    // one-hot factor: row = cell c, k-slots 8 kg .. 8 kg + 7 = frames 8 kg + i of the half block
    const uint2 cb = *(const uint2*)((const unsigned char*)(img + 32 * STRIDE) + 8 * kg);
    const unsigned c4 = (unsigned)c * 0x01010101u;
    const long hot = (long)(((u64)bytes_equal(cb.y, c4) << 32) | (u64)bytes_equal(cb.x, c4));
    const int* col = img + (8 * kg) * STRIDE + c;
#pragma unroll
    for (int et = 0; et < NET; ++et) {
        unsigned x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (unsigned)col[i * STRIDE + 16 * et];
        // 4 x 4 byte transposes: plane p of frames 0..3 / 4..7 (v_perm_b32: bytes 0-3 = second operand, 4-7 = first)
        unsigned pl[4][2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const unsigned t0 = __builtin_amdgcn_perm(x[4 * m + 1], x[4 * m], 0x05010400u);
            const unsigned t1 = __builtin_amdgcn_perm(x[4 * m + 1], x[4 * m], 0x07030602u);
            const unsigned t2 = __builtin_amdgcn_perm(x[4 * m + 3], x[4 * m + 2], 0x05010400u);
            const unsigned t3 = __builtin_amdgcn_perm(x[4 * m + 3], x[4 * m + 2], 0x07030602u);
            pl[0][m] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
            pl[1][m] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
            pl[2][m] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
            pl[3][m] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long bp = (long)(((u64)pl[p][1] << 32) | (u64)pl[p][0]);
            racc[et][p] = __builtin_amdgcn_mfma_i32_16x16x32_i8(hot, bp, racc[et][p], 0, 0, 0);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // (the next half overwrites the images)
}

// the wave's register sums into the workgroup's LDS table (rows of RS int64), registers back to zero
template <int NC>
__device__ __forceinline__ void regacc_flush(i4 (&racc)[RegAcc<NC>::NET][4], i64* __restrict__ lacc, int M, int lane)
{
    constexpr int NE = RegAcc<NC>::NE, NET = RegAcc<NC>::NET, RS = (NE + 7) & ~7;
    const int el = lane & 15, rg = lane >> 4;
#pragma unroll
    for (int et = 0; et < NET; ++et) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cell = 4 * rg + i, e = 16 * et + el;
            const i64 v = (i64)racc[et][0][i] + ((i64)racc[et][1][i] << 8) + ((i64)racc[et][2][i] << 16) +
                          ((i64)racc[et][3][i] << 24);
            if (cell < M && e < NE && v != 0) atomicAdd((u64*)&lacc[cell * RS + e], (u64)v);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) racc[et][p] = (i4){0, 0, 0, 0};
    }
}

}  // namespace e2vq
