// vq_device.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the ecoz2 VQ hot path.
//
// Replaces the arithmetic of the reference's absent C library behind
// ecoz2_vq_learn / ecoz2_vq_quantize (/root/reference/src/ecoz2_lib/mod.rs:96-122):
//   K1  sweep      per frame argmin_m d(r, c_m),  d = r0*c0 + 2*sum r[n]*c[n]   
//   K2  accumulate per cell exact fixed-point sums of member vectors (order-free)
//   K3  centroids  Levinson-Durbin per cell (lpca_r, src/lpc/lpca_r_rs.rs:8-43)
//   K4  codebook   reflections -> raas -> pre-doubled codewords; M -> 2M split
//
// Design notes (measured on MI355X):
//  * FP64-FMA bound at M >= 64.  The canonical distortion chain
//        acc = +0.0; for n = 0..P: acc = fma(r[n], c[n], acc)
//    is bit-for-bit what v_mfma_f64_16x16x4_f64 computes along k (verified on hardware:
//    tools/probe/mfma64.hip), and on real data the FP64 matrix pipe sustains ~71 TFLOP/s at
//    2.39 GHz where a v_fma_f64 stream power-throttles to ~56-60.  P = 36 therefore runs the
//    chain on the matrix pipe: 16 codewords x 16 frames per MFMA, frames resident in VGPRs,
//    codeword tiles streamed from L2, per-lane running argmin on the VALU (which idles
//    otherwise).  The template is instantiated for every P = 4 .. 40; larger P run k_pass_generic on the VALU.
//  * Training frames are resident in HBM in the operand layout of their kernel, so a wave's
//    loads are fully coalesced 16 B/lane.
//  * All sums are exact integers (two signed 32-bit limbs per value, 64-bit
//    accumulators), so LDS/global atomics in any order, any grid, any GPU count give
//    bit-identical cell sums; RCCL all-reduce runs on int64.
//
// Compiled with -ffp-contract=off: every FMA below is explicit.
#include "vq_accum.h"
#include "vq_device.h"
#include "vq_fixed.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef const double __attribute__((address_space(4))) cdouble_k;

namespace e2vq {

// diagnostics (-DE2VQ_MFMA_STAMP, tools/probe/mfma_stamps.py): cycles of a k_pass_mfma wave by phase; never in the product
#ifdef E2VQ_MFMA_STAMP
__device__ unsigned long long g_mfma_stamps[16];
#define E2VQ_MSTAMP_DECL unsigned long long ms_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ms_t = 0, ms_n = 0;
#define E2VQ_MSTAMP_START ms_t = __builtin_amdgcn_s_memtime();
#define E2VQ_MSTAMP(i)                                                   \
    {                                                                    \
        const unsigned long long ms_now = __builtin_amdgcn_s_memtime();  \
        ms_acc[i] += ms_now - ms_t;                                      \
        ms_t = ms_now;                                                   \
    }
#else
#define E2VQ_MSTAMP_DECL
#define E2VQ_MSTAMP_START
#define E2VQ_MSTAMP(i)
#endif

__device__ __forceinline__ i64 wave_sum_i64(i64 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ------------------------------------------------------------------------------------------
// layout: AoS frames [t][NC] -> blocked [b][n][64*F]  (tail block zero padded)
// ------------------------------------------------------------------------------------------
__global__ void k_blockify(const double* __restrict__ aos, long T, int NC, int FB, double* __restrict__ blk,
                           long nblocks)
{
    const long total = nblocks * (long)NC * FB;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const long b = o / ((long)NC * FB);
        const int rem = (int)(o - b * (long)NC * FB);
        const int n = rem / FB, l = rem - n * FB;
        const long t = b * FB + l;
        blk[o] = t < T ? aos[t * NC + n] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// data statistics: max |x| (bit pattern max), then global sums + sum of squares (exact)
// ------------------------------------------------------------------------------------------
__global__ void k_maxabs(const double* __restrict__ blk, long count, u64* __restrict__ out_bits, int* __restrict__ bad)
{
    u64 m = 0;
    int b = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const double v = fabs(blk[i]);
        if (!(v <= 1.7976931348623157e308)) b = 1;
        const u64 bits = (u64)__double_as_longlong(v);
        m = bits > m ? bits : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, m);
    if (b) atomicOr(bad, 1);
}

// stats layout (i64): [2*n+limb] n<NC global cell sums, [2*NC], [2*NC+1] = sum of squares limbs
__global__ void k_global_sums(const double* __restrict__ blk, long nblocks, int NC, int FB,
                              const DevScalars* __restrict__ sc, i64* __restrict__ stats)
{
    const int sh_r = sc->sh_r, sh_q = sc->sh_q;
    // one (block, n) row of FB contiguous doubles per wave iteration
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long rows = nblocks * NC;
    for (long rrow = wave; rrow < rows; rrow += nwaves) {
        const int n = (int)(rrow % NC);
        const double* p = blk + rrow * FB;
        i64 sh = 0, sl = 0, qh = 0, ql = 0;
        for (int l = lane; l < FB; l += 64) {
            const double x = p[l];
            int hi, lo;
            fix2(x, sh_r, hi, lo);
            sh += hi;
            sl += lo;
            fix2(x * x, sh_q, hi, lo);
            qh += hi;
            ql += lo;
        }
        sh = wave_sum_i64(sh);
        sl = wave_sum_i64(sl);
        qh = wave_sum_i64(qh);
        ql = wave_sum_i64(ql);
        if (lane == 0) {
            atomicAdd((u64*)&stats[2 * n], (u64)sh);
            atomicAdd((u64*)&stats[2 * n + 1], (u64)sl);
            atomicAdd((u64*)&stats[2 * NC], (u64)qh);
            atomicAdd((u64*)&stats[2 * NC + 1], (u64)ql);
        }
    }
}

// maxabs bits -> shifts (runs after the MAX all-reduce)
__global__ void k_finish_scalars(const u64* __restrict__ maxabs_bits, DevScalars* __restrict__ sc)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double maxabs = __longlong_as_double((i64)*maxabs_bits);
        sc->maxabs = maxabs;
        const int e = ilogb(maxabs);
        sc->sh_r = 29 - e;
        sc->sh_q = 28 - 2 * e;
    }
}

// ------------------------------------------------------------------------------------------
// K1 + K2: sweep + accumulate
// ------------------------------------------------------------------------------------------
//   MODE  0: assignment only (quantize)   1: LDS accumulator table (small M)   2: global atomics (large M)
//         5: hybrid LDS/global (mid M)    3: diagnostics (MODE 2 without the atomics)
[[maybe_unused]] constexpr int TPB = 256;

// ------------------------------------------------------------------------------------------
// MFMA path (P = 36): operand layouts
//   frames : block of 64 frames = 2 super-tiles of 32 frames; super-tile =
//            [s < NS-1][lane 0..63][h 0..1] then the partial last k-step [q < REM][j 0..15][h],
//            value = r[t0 + 16h + j][4s + q]  (lane = 16q + j): exactly NC*32 doubles, no padding
//   codebook: cbm[tile][p][lane][2] = cq[16*tile + j][4*(2p+e) + q], zero beyond n = P,
//            codewords beyond M are copies of codeword 0 (they can never win a tie)
// ------------------------------------------------------------------------------------------

// (also scans max |x| and non-finite values on the way: the data statistics need no separate pass)
__global__ void k_blockify_mfma(const double* __restrict__ aos, long T, int NC, double* __restrict__ blk,
                                long nblocks, u64* __restrict__ maxabs_bits, int* __restrict__ bad)
{
    const int NS = (NC + 3) >> 2;
    const long total = nblocks * (long)NC * 64;
    u64 mx = 0;
    int isbad = 0;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const long b = o / ((long)NC * 64);
        const int w = (int)(o - b * (long)NC * 64);
        const int u = w / (NC * 32), x = w - u * (NC * 32);
        int n, h, j;
        if (x < (NS - 1) * 128) {
            const int st = x >> 7, y = x & 127, l = y >> 1;
            h = y & 1;
            j = l & 15;
            n = 4 * st + (l >> 4);
        } else {
            const int y = x - (NS - 1) * 128, z = y >> 1;
            h = y & 1;
            j = z & 15;
            n = 4 * (NS - 1) + (z >> 4);
        }
        const long t = b * 64 + u * 32 + h * 16 + j;
        const double v = t < T ? aos[t * NC + n] : 0.0;
        blk[o] = v;
        const double av = fabs(v);
        if (!(av <= 1.7976931348623157e308)) isbad = 1;
        const u64 bits = (u64)__double_as_longlong(av);
        mx = bits > mx ? bits : mx;
    }
    if (maxabs_bits) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const u64 ot = __shfl_xor(mx, off, 64);
            mx = ot > mx ? ot : mx;
        }
        if ((threadIdx.x & 63) == 0 && mx > __hip_atomic_load(maxabs_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(maxabs_bits, mx);
        if (isbad) atomicOr(bad, 1);
    }
}

// AOS = true (quantize): `blk` is the row-major .prd payload [t][NC]; each wave stages its 64 rows through LDS
// with coalesced 16-B loads and picks its B operands from there, so no re-layout pass is needed.
// SRC = 2 (fallback of the prefiltered pass): the frames are those listed in fb_list[0 .. *fb_count), read from the
// blocked layout one coefficient at a time; T and nblocks come from the device-side count.
// SRC = 3 (round 6): the same list swept like the plain pass -- four frame tiles per wave, every wave the whole codebook --
// for data on which the list is LONG (tens of per cent of the frames: distortions that are small differences of large terms
// certify poorly): the short-list kernel loads every codeword tile for 16 frames and ran at 0.6 of the plain
// sweep's rate per frame there.
template <int NC, int MODE, int TPBM, int SRC = 0>
__global__ __launch_bounds__(TPBM, 2) void k_pass_mfma(const double* __restrict__ blk, long T, long nblocks,
                                                       const double* __restrict__ cbm, int MT, int M,
                                                       const DevScalars* __restrict__ sc,
                                                       const u64* __restrict__ l1max_bits,
                                                       unsigned short* __restrict__ sym, double* __restrict__ dmin,
                                                       i64* __restrict__ rows, int stagger,
                                                       const int* __restrict__ fb_list = nullptr,
                                                       const int* __restrict__ fb_count = nullptr,
                                                       unsigned short* __restrict__ prev_sym = nullptr, int incr = 0,
                                                       int list_rowmajor = 0, unsigned short* cells_out = nullptr)
{
    constexpr bool AOS = SRC == 1;
    constexpr bool LIST = SRC == 2 || SRC == 3;
    // frame tiles (of 16) per wave: the fallback list is short, so its waves take one tile each -- four times as
    // many waves, each a quarter of the latency of a full 64-frame sweep.  Prediction orders 41 .. 80 (round 4): two
    // tiles -- a wave takes one 32-frame half of a block, whose operands (up to 88 registers) fit beside two sets of
    // codeword-tile operands where those of four tiles would not
    constexpr bool WIDE = NC > 41;
    constexpr int NFT = SRC == 2 ? 1 : (WIDE ? 2 : 4);
    constexpr int FPB = 16 * NFT;
    static_assert(!WIDE || (SRC == 0 && (MODE == 0 || MODE == 2)), "wide orders: assignment and global-atomic accumulate only");
    if constexpr (WIDE) nblocks *= 2;  // (half blocks; frames beyond T in the last one are zero padding, as ever)
    // ... and the waves of a workgroup split the codebook of ONE tile between them (the list is latency-bound: a
    // single wave walking all M / 16 codeword tiles takes ~50 us at M = 1024), then combine through LDS
    constexpr int SPLIT = SRC == 2 ? TPBM / 64 : 1;
    if constexpr (LIST) {
        T = *fb_count;
        nblocks = (T + FPB - 1) / FPB;
    }
    // NS k-steps of 4 cover n < 4*NS; with NC = 4*NSF + 1 the last coefficient (n = NC-1) is not padded to
    // a fifth MFMA k-step but applied as one VALU fma after the MFMA chain: same ascending order, same roundings.
    constexpr int NS = (NC + 3) / 4, NP = (NS + 1) / 2, REM = NC - 4 * (NS - 1);
    constexpr bool PRIO = true;  // (s_setprio 1 around the MFMA cluster: -3.5 % on the plain sweep)
    constexpr bool TAILV = REM == 1;           // single trailing coefficient -> VALU
    constexpr int NSM = TAILV ? NS - 1 : NS;   // k-steps run on the matrix pipe
    constexpr int RS = (2 * NC + 5 + 7) & ~7;
    constexpr int NE = 2 * NC + 5;
    constexpr int IMG = NE + IMG_STRIDE_PAD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int q = lane >> 4, j = lane & 15;
    const int wib = threadIdx.x >> 6;
    const long wave = SPLIT > 1 ? (long)blockIdx.x : (long)blockIdx.x * (TPBM >> 6) + wib;
    const long nwaves = SPLIT > 1 ? (long)gridDim.x : (long)gridDim.x * (TPBM >> 6);

    int sh_r = 0, sh_d = 0, sh_d2 = 0;
    if constexpr (MODE != 0) {
        sh_r = sc->sh_r;
        const int Ed = dist_exponent(sc->maxabs, __longlong_as_double((i64)*l1max_bits));
        sh_d = 30 - Ed;
        sh_d2 = 30 - 2 * Ed;
    }
    // LDS accumulator table shared by the workgroup: all cells (MODE 1, M <= 128) or the first HYB_CELLS cells
    // (MODE 5, 128 < M <= 512: the other cells take global atomics, whose traffic drops accordingly)
    constexpr int HYB_CELLS = mfma_hyb_cells(NC);  // NC = 37: 176 x 640 B + 42 KB of images = 151 KB of the 160 KB LDS
    const int lds_cells = MODE == 1 ? M : (MODE == 5 ? HYB_CELLS : 0);
    i64* lacc = (i64*)smem;
    int* img = (int*)(smem + (size_t)lds_cells * RS * 8) + wib * (16 * IMG);
    if constexpr (MODE == 1 || MODE == 5) {
        for (int i = threadIdx.x; i < lds_cells * RS; i += TPBM) lacc[i] = 0;
        __syncthreads();
    }
    // Two waves share a SIMD (waves w and w+4 of an 8-wave workgroup).  Left alone they run in lockstep and
    // reach their accumulate phase -- atomic-latency bound, no MFMA -- together, idling the matrix pipe.
    // Delaying waves 4..7 by about half a sweep keeps one partner sweeping while the other accumulates.
    // The accumulate traffic (632 B of atomics per frame) is absorbed at the memory side at ~1 TB/s.  If all
    // waves arrive there together, each block round ends in a chip-wide atomic burst that every wave waits out
    // (its next frame loads queue behind its own atomics in vmcnt order).  Spreading the start phases over one
    // block period turns the bursts into a steady stream; partners differ by half a period.
    if constexpr (MODE != 0 && TPBM == 512) {
        if (nblocks >= 4 * nwaves && stagger && MT >= 16) {
            int phase = (((int)blockIdx.x + 4 * (wib & 3)) & 7) + 8 * (wib >> 2);  // 0..15 of 16
            if (stagger == 2) phase >>= 1;                                            // half amplitude
            if (stagger == 3) phase = 8 * (wib >> 2);                                 // partners only
            const int naps = (phase * (MT * NSM * 4 * 64 / 16)) >> 13;  // s_sleep(127) ~ 8k cycles
            for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }

    // MODE 1 (small codebooks, HBM-bound): the frames of the wave's next block are requested before the accumulate
    // phase of the current one, which hides their HBM latency
    constexpr bool PREFETCH = MODE == 1 && SRC == 0;
    double Bn[PREFETCH ? 4 : 1][2 * NP];
    if constexpr (PREFETCH) {
        if (wave < nblocks) load_block_frames<NC>(blk, wave, lane, Bn);
    }
    E2VQ_MSTAMP_DECL
    for (long b = wave; b < nblocks; b += nwaves) {
        E2VQ_MSTAMP_START
        // ---- frames -> B operands (resident for the whole sweep) ---------------------------
        double Bf[4][2 * NP];
        const double* fb = blk + b * (long)(NC * 64);
        if constexpr (PREFETCH) {
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
#pragma unroll
                for (int st = 0; st < 2 * NP; ++st) Bf[ft][st] = Bn[ft][st];
        } else if constexpr (AOS) {
            static_assert(MODE == 0, "the AoS path serves the assignment-only kernel");
            double* stage = (double*)smem + wib * (NC * 64);  // this wave's 64 rows
            const long remaining = (T - b * 64) * NC;         // doubles left in the payload from this block on
            constexpr int CHUNKS = NC * 64 / 2;               // 16-B chunks per block
#pragma unroll
            for (int c0 = 0; c0 < CHUNKS; c0 += 64) {
                const int c = c0 + lane;
                if (c < CHUNKS) {
                    double2 v = make_double2(0.0, 0.0);
                    if (2 * c + 1 < remaining)
                        v = *(const double2*)(fb + 2 * c);
                    else if (2 * c < remaining)
                        v.x = fb[2 * c];
                    *(double2*)(stage + 2 * c) = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft) {
                const double* row = stage + (16 * ft + j) * NC;
#pragma unroll
                for (int st = 0; st < NS - 1; ++st) Bf[ft][st] = row[4 * st + q];
                Bf[ft][NS - 1] = TAILV ? row[NC - 1] : (q < REM ? row[4 * (NS - 1) + q] : 0.0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else if constexpr (LIST) {
#pragma unroll
            for (int ft = NFT; ft < 4; ++ft)
#pragma unroll
                for (int st = 0; st < 2 * NP; ++st) Bf[ft][st] = 0.0;
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft) {
                const long slot = b * FPB + 16 * ft + j;
                const long t = slot < T ? fb_list[slot] : -1;
#pragma unroll
                for (int st = 0; st < NS; ++st) {
                    const int n = st < NS - 1 ? 4 * st + q : (TAILV ? NC - 1 : (q < REM ? 4 * (NS - 1) + q : -1));
                    // (list_rowmajor: the listed frames are rows of the row-major .prd payload -- quantize)
                    Bf[ft][st] = (t >= 0 && n >= 0) ? blk[list_rowmajor ? t * NC + n : mfma_blk_offset(NC, t, n)] : 0.0;
                }
            }
        } else if constexpr (WIDE) {
            double Hf[2][2 * NP];
            load_half_frames<NC>(blk, b >> 1, (int)(b & 1), lane, Hf);
#pragma unroll
            for (int st = 0; st < 2 * NP; ++st) {
                Bf[0][st] = Hf[0][st];
                Bf[1][st] = Hf[1][st];
                Bf[2][st] = Bf[3][st] = 0.0;
            }
        } else {
            load_block_frames<NC>(blk, b, lane, Bf);
        }

        E2VQ_MSTAMP(0)  // frames into registers (prefetched: a copy; else the load is issued here and waited for in the sweep)
        // ---- sweep: 16 codewords x 16 frames per MFMA, k ascending = canonical chain -------
        double best[4];
        int code[4];
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            best[ft] = __builtin_inf();
            code[ft] = 0;
        }
        const double* ctail = cbm + (long)MT * NP * 128;  // [tile][q][rg]: cq[16*tile + 4*rg + q][NC-1]
        // one codeword tile: 36 MFMAs (+ the trailing coefficient on the VALU), then the per-lane running argmin.
        // The A image is ping-ponged between two register sets (tile loop unrolled by two) so the prefetch of the
        // next tile lands in the other set and no register copies are needed.
        auto load_tile = [&](int t, double2 (&An)[NP], d4& Tn) {
#pragma unroll
            for (int p = 0; p < NP; ++p) An[p] = *(const double2*)(cbm + (((long)t * NP + p) * 64 + lane) * 2);
            if (TAILV) Tn = *(const d4*)(ctail + (long)t * 16 + q * 4);
        };
        auto do_tile = [&](int ct, const double2 (&Ac)[NP], const d4& Tc) {
            d4 acc[4];
            if (PRIO) __builtin_amdgcn_s_setprio(1);  // the wave feeding the matrix pipe wins issue arbitration
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft)
                acc[ft] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ac[0].x, Bf[ft][0], (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
#pragma unroll
            for (int st = 1; st < NSM; ++st)
#pragma unroll
                for (int ft = 0; ft < NFT; ++ft)
                    acc[ft] = __builtin_amdgcn_mfma_f64_16x16x4f64((st & 1) ? Ac[st >> 1].y : Ac[st >> 1].x, Bf[ft][st],
                                                                   acc[ft], 0, 0, 0);
            if (PRIO) __builtin_amdgcn_s_setprio(0);
            if (TAILV) {
#pragma unroll
                for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) acc[ft][rg] = __builtin_fma(Bf[ft][NS - 1], Tc[rg], acc[ft][rg]);
            }
            // lane (q, j) sees codewords 16ct + 4rg + q of frame j: ascending in (ct, rg)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int cval = __builtin_amdgcn_readfirstlane(ct * 4 + rg);  // wave-uniform: stays in an SGPR
#pragma unroll
                for (int ft = 0; ft < NFT; ++ft) {
                    const double v = acc[ft][rg];
                    const bool lt = v < best[ft];
                    code[ft] = lt ? cval : code[ft];
                    // raw v_min_f64: min() of finite values needs no NaN canonicalisation (inputs are validated)
                    asm("v_min_f64 %0, %1, %2" : "=v"(best[ft]) : "v"(best[ft]), "v"(v));
                }
            }
        };
        {
            // this wave's share of the codeword tiles (all of them unless the workgroup splits the codebook)
            const int ct0 = SPLIT > 1 ? (int)((long)wib * MT / SPLIT) : 0;
            const int ct1 = SPLIT > 1 ? (int)((long)(wib + 1) * MT / SPLIT) : MT;
            double2 A0[NP], A1[NP];
            d4 T0 = {0.0, 0.0, 0.0, 0.0}, T1 = {0.0, 0.0, 0.0, 0.0};
            int ct = ct0;
            if (ct < ct1) load_tile(ct, A0, T0);
            for (; ct + 1 < ct1; ct += 2) {
                load_tile(ct + 1, A1, T1);
                do_tile(ct, A0, T0);
                load_tile(ct + 2 < ct1 ? ct + 2 : ct + 1, A0, T0);
                do_tile(ct + 1, A1, T1);
            }
            if (ct < ct1) do_tile(ct, A0, T0);
        }

        E2VQ_MSTAMP(1)  // sweep
        // ---- combine the four lanes (q = 0..3) that hold one frame: min value, lowest index ---
        int idx[4] = {0, 0, 0, 0};
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            idx[ft] = ((code[ft] >> 2) << 4) + ((code[ft] & 3) << 2) + q;
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const double ob = __shfl_xor(best[ft], off, 64);
                const int oi = __shfl_xor(idx[ft], off, 64);
                const bool take = ob < best[ft] || (ob == best[ft] && oi < idx[ft]);
                best[ft] = take ? ob : best[ft];
                idx[ft] = take ? oi : idx[ft];
            }
        }

        if constexpr (SPLIT > 1) {
            // codebook split: wave 0 collects the partial (min, index) pairs of its partners (double-buffered by
            // iteration parity, so one barrier per block suffices), the others go on to the next block
            const long it = (b - wave) / nwaves;
            double* xb = (double*)(smem + (size_t)SPLIT * 16 * IMG * 4) + (it & 1) * (SPLIT * 64);
            int* xi = (int*)(smem + (size_t)SPLIT * 16 * IMG * 4 + (size_t)2 * SPLIT * 64 * 8) + (it & 1) * (SPLIT * 64);
            xb[wib * 64 + lane] = best[0];
            xi[wib * 64 + lane] = idx[0];
            __syncthreads();
            if (wib != 0) continue;
#pragma unroll
            for (int w = 1; w < SPLIT; ++w) {
                const double ob = xb[w * 64 + lane];
                const int oi = xi[w * 64 + lane];
                const bool take = ob < best[0] || (ob == best[0] && oi < idx[0]);
                best[0] = take ? ob : best[0];
                idx[0] = take ? oi : idx[0];
            }
        }

        if constexpr (PREFETCH) {
            if (b + nwaves < nblocks) load_block_frames<NC>(blk, b + nwaves, lane, Bn);
        }

        // ---- outputs: lane 16q + j owns frame b*64 + lane ---------------------------------------
        {
            const double bs = q == 0 ? best[0] : q == 1 ? best[1] : q == 2 ? best[2] : best[3];
            const int is = q == 0 ? idx[0] : q == 1 ? idx[1] : q == 2 ? idx[2] : idx[3];
            long t = b * FPB + lane;
            const bool live = t < T && lane < FPB;
            if constexpr (LIST) t = live ? fb_list[t] : 0;
            if (live) {
                if (sym) sym[t] = (unsigned short)is;
                if (dmin) dmin[t] = bs;
            }
        }

        E2VQ_MSTAMP(2)  // combine, prefetch request, outputs
        // ---- accumulate: int32 row images [frame][2n+limb | count, d, d2] -> exact 64-bit adds ----
        if constexpr (MODE == 2 && LIST) {
            // fallback of a prefiltered pass: incremental like the pass it completes (vq_accum.h)
            // (incr == 2: the first pass after a split, rows seeded with the parents' sums -- the frame counts as sitting
            // in the even child of its old cell; see k_seed_family)
            int oldidx[4] = {0, 0, 0, 0};
            long tq = -1;  // the listed frame of THIS lane's slot b * FPB + lane (frame tile q, column j)
#pragma unroll
            for (int ft = 0; ft < NFT; ++ft) {
                const long slot = b * FPB + 16 * ft + j;
                const long t = slot < T ? fb_list[slot] : -1;
                oldidx[ft] = (incr && t >= 0) ? (incr == 2 ? 2 : 1) * (int)prev_sym[t] : 0;
                if (NFT == 1 || ft == q) tq = t;
            }
            accumulate_block<NC, MODE, false, NFT, true>(Bf, best, idx, img, lacc, rows, lds_cells, sh_r, sh_d, sh_d2, b, T,
                                                         lane, {false, false, false, false}, incr != 0, oldidx);
            // (the new cell goes to cells_out when the caller keeps two cell arrays and swaps them: k_accum_ranges)
            const int newc = NFT == 1 ? idx[0] : (q == 0 ? idx[0] : q == 1 ? idx[1] : q == 2 ? idx[2] : idx[3]);
            if ((NFT > 1 || q == 0) && tq >= 0) {
                if (cells_out)
                    cells_out[tq] = (unsigned short)newc;
                else if (prev_sym)
                    prev_sym[tq] = (unsigned short)newc;
            }
        } else if constexpr (MODE != 0)
            accumulate_block<NC, MODE, false, NFT>(Bf, best, idx, img, lacc, rows, lds_cells, sh_r, sh_d, sh_d2, b, T, lane,
                                                   {false, false, false, false});
        E2VQ_MSTAMP(3)  // accumulate
#ifdef E2VQ_MFMA_STAMP
        ms_n += 1;
#endif
    }

    if constexpr (MODE == 1 || MODE == 5) {
        E2VQ_MSTAMP_START
        __syncthreads();
        for (int i = threadIdx.x; i < lds_cells * RS; i += TPBM) {
            const i64 v = lacc[i];
            if (v != 0) atomicAdd((u64*)&rows[i], (u64)v);
        }
        E2VQ_MSTAMP(4)  // wait for the workgroup, flush the table
    }
#ifdef E2VQ_MFMA_STAMP
    if (lane == 0 && MODE != 0 && SRC == 0) {
        for (int k = 0; k < 8; ++k) atomicAdd(&g_mfma_stamps[k], ms_acc[k]);
        atomicAdd(&g_mfma_stamps[8], ms_n);
        atomicAdd(&g_mfma_stamps[9], 1ull);
    }
#endif
}


// ------------------------------------------------------------------------------------------
// K1+K2 for M <= 16 (the first four levels of every ladder: 24 of the 45 passes of a 2..1024 ladder): the FP64 sweep of
// k_pass_mfma over the ONE codeword tile, 32 frames (a half block) at a time, and the cell sums as per-wave register
// accumulators fed by the i8 matrix unit (vq_accum.h: RegAcc) -- no atomic inside the loop.  The codeword tile stays in
// registers for the whole kernel; the next half block is in flight while the current one is swept, staged and added.
// ------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(512, 2) void k_pass_small(const double* __restrict__ blk, long T, long nblocks,
                                                       const double* __restrict__ cbm, int M,
                                                       const DevScalars* __restrict__ sc, const u64* __restrict__ l1max_bits,
                                                       unsigned short* __restrict__ sym, double* __restrict__ dmin,
                                                       i64* __restrict__ rows, unsigned flush_mask)
{
    constexpr int NS = (NC + 3) / 4, NP = (NS + 1) / 2, REM = NC - 4 * (NS - 1);
    constexpr bool TAILV = REM == 1;
    constexpr int NSM = TAILV ? NS - 1 : NS;
    constexpr int RS = (2 * NC + 5 + 7) & ~7;
    constexpr int NET = RegAcc<NC>::NET;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int q = lane >> 4;
    const int wib = threadIdx.x >> 6;
    const long wave = (long)blockIdx.x * 8 + wib, nwaves = (long)gridDim.x * 8;

    const FixScale fr = fix_scale(sc->sh_r);
    const int Ed = dist_exponent(sc->maxabs, __longlong_as_double((i64)*l1max_bits));
    const FixScale fd = fix_scale(30 - Ed), fd2 = fix_scale(30 - 2 * Ed);
    const bool fast = fr.fast && fd.fast && fd2.fast;

    // LDS: the cell table | the codeword tile in operand order, [p][lane] pairs + the trailing coefficient of the four
    // codewords 4 rg + q (read again for every half block: 28 registers that the accumulators need more) | per-wave images
    i64* lacc = (i64*)smem;
    double* ctile = (double*)(smem + (size_t)M * RS * 8);
    int* img = (int*)(ctile + NP * 128 + 16) + wib * RegAcc<NC>::WAVE_INTS;
    for (int i = threadIdx.x; i < M * RS; i += 512) lacc[i] = 0;
    for (int i = threadIdx.x; i < NP * 128 + 16; i += 512)
        ctile[i] = i < NP * 128 ? cbm[i] : (TAILV ? cbm[(long)((M + 15) / 16) * NP * 128 + (i - NP * 128)] : 0.0);
    __syncthreads();

    i4 racc[NET][4];
#pragma unroll
    for (int et = 0; et < NET; ++et)
#pragma unroll
        for (int p = 0; p < 4; ++p) racc[et][p] = (i4){0, 0, 0, 0};

    double Hn[2][2 * NP];
    if (wave < nblocks) load_half_frames<NC>(blk, wave, 0, lane, Hn);
    long done = 0;
    for (long b = wave; b < nblocks; b += nwaves) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            double H[2][2 * NP];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int st = 0; st < 2 * NP; ++st) H[h][st] = Hn[h][st];
            {
                // (unconditional: a conditional load keeps its registers live across the branch; the last request re-reads
                // the wave's own half)
                const long bn = u == 0 ? b : (b + nwaves < nblocks ? b + nwaves : b);
                load_half_frames<NC>(blk, bn, u ^ 1, lane, Hn);
            }
            // ---- sweep: two chains of NSM FP64 MFMAs (16 codewords x 16 frames each), k ascending ----
            int aoff = lane * 2;
            asm volatile("" : "+v"(aoff));  // (loop-variant as far as the compiler knows: the tile is not hoisted into registers)
            double2 A[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) A[p] = *(const double2*)(ctile + p * 128 + aoff);
            d4 acc[2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
                acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[0].x, H[h][0], (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
#pragma unroll
            for (int st = 1; st < NSM; ++st)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64((st & 1) ? A[st >> 1].y : A[st >> 1].x, H[h][st], acc[h], 0, 0, 0);
            double best[2];
            int idx[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (TAILV) {
                    const d4 Tc = *(const d4*)(ctile + NP * 128 + (aoff >> 5) * 4);  // (aoff >> 5 = q)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) acc[h][rg] = __builtin_fma(H[h][NS - 1], Tc[rg], acc[h][rg]);
                }
                // lane (q, j): codewords 4 rg + q of frame j, ascending in rg; then the four q lanes of the frame
                best[h] = acc[h][0];
                int code = 0;
#pragma unroll
                for (int rg = 1; rg < 4; ++rg) {
                    const bool lt = acc[h][rg] < best[h];
                    code = lt ? rg : code;
                    best[h] = lt ? acc[h][rg] : best[h];
                }
                idx[h] = (code << 2) + q;
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    const double ob = __shfl_xor(best[h], off, 64);
                    const int oi = __shfl_xor(idx[h], off, 64);
                    const bool take = ob < best[h] || (ob == best[h] && oi < idx[h]);
                    best[h] = take ? ob : best[h];
                    idx[h] = take ? oi : idx[h];
                }
            }
            // ---- outputs: lane 16 h + j (q = h < 2) owns frame t0 + lane ----
            const long t0 = b * 64 + 32 * u;
            {
                const long t = t0 + lane;
                if (lane < 32 && t < T) {
                    if (sym) sym[t] = (unsigned short)(q == 0 ? idx[0] : idx[1]);
                    if (dmin) dmin[t] = q == 0 ? best[0] : best[1];
                }
            }
            // ---- accumulate ----
            if (fast)  // (kernel-uniform)
                regacc_stage_half<NC, true>(H, best, idx, img, fr, fd, fd2, t0, T, lane);
            else
                regacc_stage_half<NC, false>(H, best, idx, img, fr, fd, fd2, t0, T, lane);
            regacc_add_half<NC>(img, lane, racc);
        }
        // digit sums stay below 2^31 for 2^24 frames per wave; flush long before (flush_mask = 0xFFFF: every 2^22 frames;
        // the tests set ECOZ2_VQ_SMALL_FLUSH_MASK=0 to take this path after every block)
        if ((++done & flush_mask) == 0) regacc_flush<NC>(racc, lacc, M, lane);
    }
    regacc_flush<NC>(racc, lacc, M, lane);
    __syncthreads();
    for (int i = threadIdx.x; i < M * RS; i += 512) {
        const i64 v = lacc[i];
        if (v != 0) atomicAdd((u64*)&rows[i], (u64)v);
    }
}

// global sums for the MFMA frame layout: per-lane 64-bit accumulators for its coefficients
// n = 4s + q, reduced over the 16 lanes (frames) that share q, one atomic per (n, limb) per wave.
template <int NC>
__global__ __launch_bounds__(256) void k_global_sums_mfma(const double* __restrict__ blk, long nblocks,
                                                          const DevScalars* __restrict__ sc, i64* __restrict__ stats)
{
    constexpr int NS = (NC + 3) / 4, REM = NC - 4 * (NS - 1);
    const int sh_r = sc->sh_r, sh_q = sc->sh_q;
    const int lane = threadIdx.x & 63, q = lane >> 4, j = lane & 15;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    i64 sh[NS], sl[NS], qh = 0, ql = 0;
#pragma unroll
    for (int st = 0; st < NS; ++st) sh[st] = sl[st] = 0;
    for (long u = wave; u < 2 * nblocks; u += nwaves) {  // super-tiles of 32 frames
        const double* base = blk + u * (long)(NC * 32);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            double2 v = make_double2(0.0, 0.0);
            if (st < NS - 1)
                v = *(const double2*)(base + (st * 64 + lane) * 2);
            else if (q < REM)
                v = *(const double2*)(base + (NS - 1) * 128 + (q * 16 + j) * 2);
            int hi, lo;
            fix2(v.x, sh_r, hi, lo);
            sh[st] += hi;
            sl[st] += lo;
            fix2(v.y, sh_r, hi, lo);
            sh[st] += hi;
            sl[st] += lo;
            fix2(v.x * v.x, sh_q, hi, lo);
            qh += hi;
            ql += lo;
            fix2(v.y * v.y, sh_q, hi, lo);
            qh += hi;
            ql += lo;
        }
    }
#pragma unroll
    for (int st = 0; st < NS; ++st) {
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            sh[st] += __shfl_xor(sh[st], off, 64);
            sl[st] += __shfl_xor(sl[st], off, 64);
        }
        const int n = 4 * st + q;
        if (j == 0 && n < NC) {
            atomicAdd((u64*)&stats[2 * n], (u64)sh[st]);
            atomicAdd((u64*)&stats[2 * n + 1], (u64)sl[st]);
        }
    }
    qh = wave_sum_i64(qh);
    ql = wave_sum_i64(ql);
    if (lane == 0) {
        atomicAdd((u64*)&stats[2 * NC], (u64)qh);
        atomicAdd((u64*)&stats[2 * NC + 1], (u64)ql);
    }
}

// Prediction orders beyond the MFMA instantiations (P > 40; the reference allows up to 200), round 3: the frames of a
// block are staged ONCE in the wave's LDS ([n][64]: lane = frame, conflict-free 8-byte rows) and the codewords come
// eight at a time from a transposed copy cbT[n][m] (8 consecutive doubles = one scalar load per coefficient), so a
// coefficient costs one LDS read for eight FMAs (round 2's generic kernel re-read every frame coefficient from global memory
// per codeword: ~2 TFLOP/s; it left the tree in round 5).  Same chain: acc = +0.0, fma(r[n], c[n], acc) for ascending n; ascending
// codeword index with strict < (padding codewords repeat codeword 0 and can never win).  One wave per workgroup; LDS =
// 520 (P + 1) bytes, so P = 200 still fits (one workgroup per CU).
constexpr int GEN_G = 8;    // codewords per group
constexpr int GEN_LD = 65;  // leading dimension (doubles) of the staged block


__global__ void k_transpose_codebook(const double* __restrict__ cbq, int M, int NC, int NPAD, double* __restrict__ cbT, int Mpad)
{
    const long total = (long)NC * Mpad;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int n = (int)(o / Mpad), m = (int)(o - (long)n * Mpad);
        cbT[o] = cbq[(long)(m < M ? m : 0) * NPAD + n];
    }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_pass_generic_lds(const double* __restrict__ blk, long T, long nblocks, int NC,
                                                          const double* __restrict__ cbT, int M, int Mpad,
                                                          const DevScalars* __restrict__ sc, const u64* __restrict__ l1max_bits,
                                                          unsigned short* __restrict__ sym, double* __restrict__ dmin,
                                                          i64* __restrict__ rows)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [n][GEN_LD]: lane = frame.  The odd leading dimension (65 doubles) lets the accumulate read ONE frame's limbs across
    // the lanes (element e = 2 n + limb at dword (n * 65 + f) * 2 + limb: banks 2 n + 2 f + limb, distinct for 32 n)
    constexpr int LD = GEN_LD;
    double* fr = (double*)smem;
    const int* fri = (const int*)smem;
    const int RS = (2 * NC + 5 + 7) & ~7;
    const int lane = threadIdx.x;
    int sh_r = 0, sh_d = 0, sh_d2 = 0;
    if (MODE != 0) {
        sh_r = sc->sh_r;
        const int Ed = dist_exponent(sc->maxabs, __longlong_as_double((i64)*l1max_bits));
        sh_d = 30 - Ed;
        sh_d2 = 30 - 2 * Ed;
    }
    for (long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const double* fb = blk + b * (long)(NC * 64);
        for (int n = 0; n < NC; ++n) fr[n * LD + lane] = fb[n * 64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double best = __builtin_inf();
        int bi = 0;
        for (int m0 = 0; m0 < Mpad; m0 += GEN_G) {
            double acc[GEN_G];
#pragma unroll
            for (int g = 0; g < GEN_G; ++g) acc[g] = 0.0;
            const double* c = cbT + m0;  // wave-uniform: scalar loads
#pragma unroll 4  // (four coefficients' loads in flight: one wave per SIMD has nobody else to hide their latency)
            for (int n = 0; n < NC; ++n) {
                const double r = fr[n * LD + lane];
#pragma unroll
                for (int g = 0; g < GEN_G; ++g) acc[g] = __builtin_fma(r, c[(long)n * Mpad + g], acc[g]);
            }
#pragma unroll
            for (int g = 0; g < GEN_G; ++g) {
                const bool lt = acc[g] < best;
                best = lt ? acc[g] : best;
                bi = lt ? m0 + g : bi;
            }
        }
        const long t = b * 64 + lane;
        const bool live = t < T;
        if (live) {
            if (sym) sym[t] = (unsigned short)bi;
            if (dmin) dmin[t] = best;
        }
        if (MODE != 0) {
            // every lane converts its frame's coefficients to limb pairs in place, then the wave adds frame after frame to
            // its cell ROW-wise (lanes = consecutive row elements: 512 contiguous bytes per atomic instruction; per-lane
            // atomics scattered over 64 rows run 17 x slower); the count and the four distortion limbs ride in lanes 0..4
            // of one more instruction
            for (int n = 0; n < NC; ++n) {
                int hi, lo;
                fix2(fr[n * LD + lane], sh_r, hi, lo);
                *(int2*)&fr[n * LD + lane] = make_int2(hi, lo);
            }
            int d[4] = {0, 0, 0, 0};
            {
                const double e = best - 1.0;
                fix2(e, sh_d, d[0], d[1]);
                fix2(e * e, sh_d2, d[2], d[3]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const long left = T - b * 64;
            const int nv = left >= 64 ? 64 : (int)left;
            for (int f = 0; f < nv; ++f) {
                const int cell = __builtin_amdgcn_readlane(bi, f);
                i64* row = rows + (long)cell * RS;
                for (int e0 = 0; e0 < 2 * NC; e0 += 64) {
                    const int e = e0 + lane;
                    if (e < 2 * NC) atomicAdd((u64*)&row[e], (u64)(i64)fri[((e >> 1) * LD + f) * 2 + (e & 1)]);
                }
                const int d0 = __builtin_amdgcn_readlane(d[0], f), d1 = __builtin_amdgcn_readlane(d[1], f),
                          d2 = __builtin_amdgcn_readlane(d[2], f), d3 = __builtin_amdgcn_readlane(d[3], f);
                if (lane < 5) {
                    const int v = lane == 0 ? 1 : lane == 1 ? d0 : lane == 2 ? d1 : lane == 3 ? d2 : d3;
                    atomicAdd((u64*)&row[2 * NC + lane], (u64)(i64)v);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (the next block overwrites the staged frames)
    }
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------

static inline int grid_for(long work_items, int per_block, int cap)
{
    long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// Every prediction order P = 4 .. 40 (NC = 5 .. 41) has an instantiation of the MFMA sweep: NC = 4k+1 runs the
// trailing coefficient on the VALU, the others zero-pad the last k-step.  Larger P take k_pass_generic.
// (overridable on the command line: tests/test_isa_guards.py compiles the P = 36 instantiations alone)
#ifndef E2VQ_MFMA_NC_LIST
#define E2VQ_MFMA_NC_LIST(X) \
    X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) \
    X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) X(41)
#endif
// ... and, round 4, P = 41 .. 80 (NC = 42 .. 81) with half blocks per wave (assignment and global-atomic accumulate)
#ifndef E2VQ_MFMA_WIDE_NC_LIST
#define E2VQ_MFMA_WIDE_NC_LIST(X) \
    X(42) X(43) X(44) X(45) X(46) X(47) X(48) X(49) X(50) X(51) X(52) X(53) X(54) X(55) X(56) X(57) X(58) X(59) X(60) X(61) \
    X(62) X(63) X(64) X(65) X(66) X(67) X(68) X(69) X(70) X(71) X(72) X(73) X(74) X(75) X(76) X(77) X(78) X(79) X(80) X(81)
#endif
bool uses_mfma(int NC) { return NC >= 5 && NC <= 81; }
bool mfma_is_wide(int NC) { return NC > 41 && NC <= 81; }
int mfma_hybrid_cells(int NC) { return mfma_hyb_cells(NC); }

// maxabs_bits / bad (optional): when given and the MFMA layout is used, the scan of max |x| rides along; returns
// true in that case (the caller then skips launch_maxabs)
bool launch_blockify(const double* aos, long T, int NC, int FB, double* blk, long nblocks, u64* maxabs_bits, int* bad,
                     hipStream_t s)
{
    if (uses_mfma(NC)) {
        hipLaunchKernelGGL(k_blockify_mfma, dim3(grid_for(nblocks * NC * 64, 256, 8192)), dim3(256), 0, s, aos, T, NC,
                           blk, nblocks, maxabs_bits, bad);
        return maxabs_bits != nullptr;
    }
    hipLaunchKernelGGL(k_blockify, dim3(grid_for(nblocks * NC * FB, 256, 8192)), dim3(256), 0, s, aos, T, NC, FB, blk,
                       nblocks);
    return false;
}

void launch_maxabs(const double* blk, long count, u64* out_bits, int* bad, hipStream_t s)
{
    hipLaunchKernelGGL(k_maxabs, dim3(grid_for(count, 256 * 8, 4096)), dim3(256), 0, s, blk, count, out_bits, bad);
}

void launch_finish_scalars(const u64* maxabs_bits, DevScalars* sc, hipStream_t s)
{
    hipLaunchKernelGGL(k_finish_scalars, dim3(1), dim3(64), 0, s, maxabs_bits, sc);
}

void launch_global_sums(const double* blk, long nblocks, int NC, int FB, const DevScalars* sc, i64* stats,
                        hipStream_t s)
{
    const dim3 g(grid_for(2 * nblocks, 4, 2048));
    switch (NC) {
#define X(N) \
    case N: hipLaunchKernelGGL((k_global_sums_mfma<N>), g, dim3(256), 0, s, blk, nblocks, sc, stats); return;
        E2VQ_MFMA_NC_LIST(X)
        E2VQ_MFMA_WIDE_NC_LIST(X)
#undef X
        default: break;
    }
    hipLaunchKernelGGL(k_global_sums, dim3(grid_for(nblocks * NC, 4, 4096)), dim3(256), 0, s, blk, nblocks, NC, FB, sc,
                       stats);
}

template <int NC>
static int launch_pass_mfma(int mode, const double* blk, long T, long nblocks, const double* cbm, int M,
                            const DevScalars* sc, const u64* l1max_bits, unsigned short* sym, double* dmin, i64* rows,
                            hipStream_t s)
{
    constexpr int RS = (2 * NC + 5 + 7) & ~7;
    constexpr int IMG = 2 * NC + 5 + IMG_STRIDE_PAD;
    const int MT = (M + 15) / 16;
    if (mode == 0) {
        const int grid = grid_for(nblocks, 4, 512);
        hipLaunchKernelGGL((k_pass_mfma<NC, 0, 256>), dim3(grid), dim3(256), 0, s, blk, T, nblocks, cbm, MT, M, sc,
                           l1max_bits, sym, dmin, rows, 0);
    } else if (mode == 4) {  // assignment only, frames in row-major (.prd) layout
        const size_t lds = (size_t)4 * NC * 64 * 8;
        (void)hipFuncSetAttribute((const void*)k_pass_mfma<NC, 0, 256, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  E2VQ_LDS_BYTES);
        const int grid = grid_for(nblocks, 4, 512);
        hipLaunchKernelGGL((k_pass_mfma<NC, 0, 256, 1>), dim3(grid), dim3(256), lds, s, blk, T, nblocks, cbm, MT, M, sc,
                           l1max_bits, sym, dmin, rows, 0);
    } else if (mode == 1 && M <= 16 && RegAcc<NC>::OK) {
        const size_t lds = (size_t)M * RS * 8 + (size_t)((((NC + 3) / 4 + 1) / 2) * 128 + 16) * 8 + (size_t)8 * RegAcc<NC>::WAVE_INTS * 4;
        (void)hipFuncSetAttribute((const void*)k_pass_small<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, E2VQ_LDS_BYTES);
        const int grid = grid_for(nblocks, 8, 256);
        const unsigned flush_mask = getenv("ECOZ2_VQ_SMALL_FLUSH_MASK") ? (unsigned)atoi(getenv("ECOZ2_VQ_SMALL_FLUSH_MASK")) : 0xFFFFu;
        hipLaunchKernelGGL((k_pass_small<NC>), dim3(grid), dim3(512), lds, s, blk, T, nblocks, cbm, M, sc, l1max_bits, sym,
                           dmin, rows, flush_mask);
    } else if (mode == 1) {
        const size_t lds = (size_t)M * RS * 8 + (size_t)8 * 16 * IMG * 4;
        // every launch: the attribute is per device, and sessions may live on several devices of one process
        (void)hipFuncSetAttribute((const void*)k_pass_mfma<NC, 1, 512>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  E2VQ_LDS_BYTES);
        const int grid = grid_for(nblocks, 8, 256);  // one persistent 8-wave workgroup per CU
        hipLaunchKernelGGL((k_pass_mfma<NC, 1, 512>), dim3(grid), dim3(512), lds, s, blk, T, nblocks, cbm, MT, M, sc,
                           l1max_bits, sym, dmin, rows, 0);
    } else if (mode == 5) {  // hybrid: cells < 176 in the LDS table, the rest by global atomics
        const size_t lds = (size_t)mfma_hyb_cells(NC) * RS * 8 + (size_t)8 * 16 * IMG * 4;
        (void)hipFuncSetAttribute((const void*)k_pass_mfma<NC, 5, 512>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  E2VQ_LDS_BYTES);
        const int grid = grid_for(nblocks, 8, 256);
        const int stagger = 1;
        hipLaunchKernelGGL((k_pass_mfma<NC, 5, 512>), dim3(grid), dim3(512), lds, s, blk, T, nblocks, cbm, MT, M, sc,
                           l1max_bits, sym, dmin, rows, stagger);
    } else if (mode == 3) {  // diagnostics: MODE 2 without the atomics
        const size_t lds = (size_t)8 * 16 * IMG * 4;
        const int grid = grid_for(nblocks, 8, 256);
        hipLaunchKernelGGL((k_pass_mfma<NC, 3, 512>), dim3(grid), dim3(512), lds, s, blk, T, nblocks, cbm, MT, M, sc,
                           l1max_bits, sym, dmin, rows, 0);
    } else {
        const size_t lds = (size_t)8 * 16 * IMG * 4;
        const int grid = grid_for(nblocks, 8, 256);  // one 8-wave workgroup per CU: partner waves are w, w+4
        const int stagger = 1;  // A/B: ~1 % faster on
        hipLaunchKernelGGL((k_pass_mfma<NC, 2, 512>), dim3(grid), dim3(512), lds, s, blk, T, nblocks, cbm, MT, M, sc,
                           l1max_bits, sym, dmin, rows, stagger);
    }
    return 0;
}


// prediction orders 41 .. 80: assignment only (mode 0) or accumulate with global atomics (any other mode)
template <int NC>
static int launch_pass_mfma_wide(int mode, const double* blk, long T, long nblocks, const double* cbm, int M,
                                 const DevScalars* sc, const u64* l1max_bits, unsigned short* sym, double* dmin, i64* rows,
                                 hipStream_t s)
{
    constexpr int IMG = 2 * NC + 5 + IMG_STRIDE_PAD;
    const int MT = (M + 15) / 16;
    if (mode == 0) {
        const int grid = grid_for(2 * nblocks, 4, 1024);
        hipLaunchKernelGGL((k_pass_mfma<NC, 0, 256>), dim3(grid), dim3(256), 0, s, blk, T, nblocks, cbm, MT, M, sc,
                           l1max_bits, sym, dmin, rows, 0);
    } else {
        const size_t lds = (size_t)8 * 16 * IMG * 4;
        (void)hipFuncSetAttribute((const void*)k_pass_mfma<NC, 2, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, E2VQ_LDS_BYTES);
        const int grid = grid_for(2 * nblocks, 8, 256);
        hipLaunchKernelGGL((k_pass_mfma<NC, 2, 512>), dim3(grid), dim3(512), lds, s, blk, T, nblocks, cbm, MT, M, sc,
                           l1max_bits, sym, dmin, rows, 0);
    }
    return 0;
}

int launch_pass(int NC, int mode, const double* blk, long T, long nblocks, const double* cbq, const double* cbm,
                int M, const DevScalars* sc, const u64* l1max_bits, unsigned short* sym, double* dmin, i64* rows,
                hipStream_t s)
{
    switch (NC) {
#define X(N) \
    case N: return launch_pass_mfma<N>(mode, blk, T, nblocks, cbm, M, sc, l1max_bits, sym, dmin, rows, s);
        E2VQ_MFMA_NC_LIST(X)
#undef X
#define X(N) \
    case N: return launch_pass_mfma_wide<N>(mode, blk, T, nblocks, cbm, M, sc, l1max_bits, sym, dmin, rows, s);
        E2VQ_MFMA_WIDE_NC_LIST(X)
#undef X
        default: break;
    }
    // generic orders (P > 40).  With a scratch buffer for the transposed codebook (generic_scratch_doubles(NC, M) doubles, handed
    // over in the `cbm` slot) the LDS-staged kernel runs; without one, the plain kernel below
    if (cbm && (size_t)NC * GEN_LD * 8 <= (size_t)E2VQ_LDS_BYTES - 1024) {
        const int Mpad = (M + GEN_G - 1) / GEN_G * GEN_G;
        double* cbT = const_cast<double*>(cbm);
        hipLaunchKernelGGL(k_transpose_codebook, dim3(grid_for((long)NC * Mpad, 256, 1024)), dim3(256), 0, s, cbq, M, NC,
                           (NC + 7) & ~7, cbT, Mpad);
        const size_t lds = (size_t)NC * GEN_LD * 8;
        const int g2 = grid_for(nblocks, 1, 256 * (int)((size_t)E2VQ_LDS_BYTES / lds > 8 ? 8 : (size_t)E2VQ_LDS_BYTES / lds));
        if (mode == 0) {
            (void)hipFuncSetAttribute((const void*)k_pass_generic_lds<0>, hipFuncAttributeMaxDynamicSharedMemorySize, E2VQ_LDS_BYTES);
            hipLaunchKernelGGL((k_pass_generic_lds<0>), dim3(g2), dim3(64), lds, s, blk, T, nblocks, NC, cbT, M, Mpad, sc, l1max_bits,
                               sym, dmin, rows);
        } else {
            (void)hipFuncSetAttribute((const void*)k_pass_generic_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, E2VQ_LDS_BYTES);
            hipLaunchKernelGGL((k_pass_generic_lds<2>), dim3(g2), dim3(64), lds, s, blk, T, nblocks, NC, cbT, M, Mpad, sc, l1max_bits,
                               sym, dmin, rows);
        }
        return 0;
    }
    return 1;  // (no scratch for the transposed codebook: the caller always provides one)
}

// full FP64 sweep of the frames a prefiltered pass could not certify (vq_prefilter.hip): fb_list[0 .. *fb_count)
int launch_pass_fallback(int NC, bool accumulate, const double* blk, const double* cbm, int M, const DevScalars* sc,
                         const u64* l1max_bits, unsigned short* sym, double* dmin, i64* rows, const int* fb_list,
                         const int* fb_count, unsigned short* prev_sym, int incremental, hipStream_t s, bool rowmajor,
                         unsigned short* cells_out, bool long_list)
{
    const int MT = (M + 15) / 16;
    if (long_list && !mfma_is_wide(NC)) {
        // round 6: the list is expected to be long (the caller saw the count of the pass before): four frame tiles per wave,
        // every wave the whole codebook -- the plain sweep's shape and rate
        switch (NC) {
#define X(N)                                                                                                           \
    case N: {                                                                                                          \
        const size_t lds = (size_t)8 * 16 * (2 * N + 5 + IMG_STRIDE_PAD) * 4;                                          \
        if (accumulate)                                                                                                \
            hipLaunchKernelGGL((k_pass_mfma<N, 2, 512, 3>), dim3(256), dim3(512), lds, s, blk, 0L, 0L, cbm, MT, M, sc,  \
                               l1max_bits, sym, dmin, rows, 0, fb_list, fb_count, prev_sym, incremental,              \
                               rowmajor ? 1 : 0, cells_out);                                                           \
        else                                                                                                           \
            hipLaunchKernelGGL((k_pass_mfma<N, 0, 512, 3>), dim3(256), dim3(512), lds, s, blk, 0L, 0L, cbm, MT, M, sc,  \
                               l1max_bits, sym, dmin, rows, 0, fb_list, fb_count, (unsigned short*)nullptr, 0,          \
                               rowmajor ? 1 : 0);                                                                      \
        return 0;                                                                                                      \
    }
            E2VQ_PRE_NC_LIST(X)
#undef X
            default: break;
        }
    }
    switch (NC) {
#define X(N)                                                                                                           \
    case N: {                                                                                                          \
        /* one 4-wave workgroup per 16 listed frames at a time; LDS: the waves' row images + the (min, index) exchange */ \
        const size_t lds = (size_t)4 * 16 * (2 * N + 5 + IMG_STRIDE_PAD) * 4 + (size_t)2 * 4 * 64 * (8 + 4);          \
        if (accumulate)                                                                                                \
            hipLaunchKernelGGL((k_pass_mfma<N, 2, 256, 2>), dim3(1024), dim3(256), lds, s, blk, 0L, 0L, cbm, MT, M, sc, \
                               l1max_bits, sym, dmin, rows, 0, fb_list, fb_count, prev_sym, incremental,              \
                               rowmajor ? 1 : 0, cells_out);                                                           \
        else                                                                                                           \
            hipLaunchKernelGGL((k_pass_mfma<N, 0, 256, 2>), dim3(1024), dim3(256), lds, s, blk, 0L, 0L, cbm, MT, M, sc, \
                               l1max_bits, sym, dmin, rows, 0, fb_list, fb_count, (unsigned short*)nullptr, 0,          \
                               rowmajor ? 1 : 0);                                                                      \
        return 0;                                                                                                      \
    }
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return 1;
    }
}

}  // namespace e2vq

#ifdef E2VQ_MFMA_STAMP
extern "C" int e2vq_debug_mfma_stamps(unsigned long long* out16, int reset)
{
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(e2vq::g_mfma_stamps), 16 * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(e2vq::g_mfma_stamps), z, 16 * 8) != hipSuccess) return 1;
    }
    return 0;
}
#endif
