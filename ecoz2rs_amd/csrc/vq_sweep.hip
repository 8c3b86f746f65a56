// vq_sweep.hip -- round 5: the accumulating prefiltered pass over frames grouped by cell
//
//   k_sort_*        once per level: the frames' numbers grouped by the cell they had when the level began (a counting sort of
//                   2-byte keys: `perm`, 4 bytes per frame).  Nothing but this list is moved: every kernel below addresses
//                   frames by their number.
//   k_sweep_cand<NC, TWO, FUSE = true>   THE pass over grouped frames (seeded first pass, incremental passes): one kernel.
//                   A turn of a wave = two blocks of 64 *slots* of the sorted list; the B operands -- the f16 limb images of
//                   their frames -- are gathered through `perm` from a frame-major image (256 bytes per frame at P = 36, two
//                   cache lines).  Two-stage sweep (below), then per block: the FP64 rows by LDS-DMA, the canonical FP64
//                   chain for the one or two candidates (lane = slot), symbol / distortion / new cell out, and the block's
//                   contributions to the cell sums reduced in the block (a handful of rows of atomics: the frames of a block
//                   share their cells).  Bound by the CU's vector-memory path (profiles/r05_sweep_experiments.txt), which is
//                   why a loaded codeword tile serves the four column blocks of a turn.
//   k_sweep_cand<NC, TWO, FUSE = false> + k_finish + k_reduce_records   the same pass for frames that are NOT grouped (the
//                   first pass after e2vq_set_codebook): the sweep emits per frame the two codewords that can be the nearest
//                   one and whether that is certain (4 bytes); k_finish takes every frame once, in its natural order (the
//                   canonical FP64 chain, lane = frame, rows staged in LDS by LDS-DMA; symbol / distortion out; the frame's
//                   contribution to the cell sums as 8-byte records); k_reduce_records (vq_prefilter.hip) folds the records
//                   into the rows, as in round 4.  The round's first version ran grouped frames through this chain too
//                   (0.80 ms at M = 1024 against the fused kernel's 0.68).
//
// Why the sort: two-stage keys.  With the frames of a block coming from one cell, the codewords that can win for any of
// them sit in one or two of the codebook's 32-codeword tiles (children of neighbouring cells are neighbours in the index).
// The sweep therefore runs the limb products in two stages, exactly:
//   stage 1, every tile: weight levels 0 and 1 only -- W0 = sum X1 Y1, W1 = sum X1 Y2 + X2 Y1: 8 of the 15 k-steps at
//     P = 36 -- and the coarse key v2 = 512 W0 + W1 = 2^27 sum xi eta - (W2 2^-9 + remainders).  With g = sum |xi| of the
//     frame and y = max_m sum |eta_m|:
//         |2^27 sum xi eta - v2|  <=  E2 = 257 (g + y) + 129 NC + 2   +  2^-23 |v2|
//     (|W2| <= sum |X1 Y3| + |X2 Y2| + |X3 Y1| <= 2^17 (g + y) + (2^16 + 2^8) NC with |X1| <= 512 |xi| + 1/2, |Y3| <= 256, ...;
//      the three-limb remainder 2^8 (g + y + NC + 4) of vq_prefilter.hip; one f32 rounding of the fma; all over 2^9.)
//     Each lane keeps U = the smallest coarse key it has seen for its frame.  A codeword whose coarse key exceeds
//     U (1 + 2^-20) + 2.54 E2 has a larger distortion -- in real numbers, by more than any rounding of the FP64 chain --
//     than the codeword that gave U: it is not the nearest.  A (tile, 32-frame column block) in which EVERY value passes
//     that test is done; any other is flagged.
//   stage 2, flagged tiles only: all 15 k-steps and the key epilogue of round 1 (top three keys per frame, certification
//     of the top two) -- over a subset of the codebook that provably contains the nearest codeword of every frame of the
//     block, so the certification argument of vq_prefilter.hip holds unchanged for it.
// On the bench data 5 % (M = 1024) to 12 % (M = 256) of the (block, tile) pairs are flagged when the frames are grouped,
// 93-99 % when they are not (profiles/r05_skip_feasibility.txt): the sweep issues 0.58 of the limb products and 1.5
// instead of 6 VALU operations per value in stage 1 (the key epilogue's VALU operations bound round 4's tile loop).  Nothing is decided by a key: a frame whose top two cannot be certified goes to the FP64
// fallback sweep as before; data without such structure only flags more tiles (the host watches the flagged fraction and
// drops stage 1 when it does not pay).
#include "vq_pre_common.h"

namespace e2vq {

template <int NC>
struct SweepImg {
    typedef PrePack<NC> PK;
    static constexpr int FS = PK::PAIRS * 32 + 32;  // bytes per frame: PAIRS x (two lane halves x 8 halves), then float g + pad
    static constexpr int NSTEP_C = PK::level_steps(0) + PK::level_steps(1);  // k-steps of the coarse stage
    __host__ __device__ static constexpr bool coarse_unique(int u)
    {
        for (int s = 0; s < NSTEP_C; ++s)
            if (PK::step_unique(s) == u) return true;
        return false;
    }
    __host__ __device__ static constexpr bool coarse_pair(int p)
    {
        for (int s = 0; s < NSTEP_C; ++s)
            if (PK::step_pair(s) == p) return true;
        return false;
    }
    // registers of the coarse stage with FOUR column blocks resident (two blocks of 64 slots per turn of the fused kernel):
    // their coarse pairs, two sets of coarse granules, the accumulators of two jobs -- beside ~55 of everything else
    __host__ __device__ static constexpr int count_coarse_pairs()
    {
        int n = 0;
        for (int p = 0; p < PK::PAIRS; ++p) n += coarse_pair(p) ? 1 : 0;
        return n;
    }
    __host__ __device__ static constexpr int count_coarse_unique()
    {
        int n = 0;
        for (int u = 0; u < PK::NU; ++u) n += coarse_unique(u) ? 1 : 0;
        return n;
    }
    static constexpr bool TWO_BLOCKS_FIT = 16 * count_coarse_pairs() + 8 * count_coarse_unique() + 64 <= 200;
};

// ---- frame-major limb image: the same limbs as k_pre_frames (same scales, same pre_split), one frame's granules together --
template <int NC>
__global__ __launch_bounds__(128) void k_frames_fm(const double* __restrict__ aos, long T, long nframes, const int* __restrict__ ea,
                                                   unsigned char* __restrict__ img)
{
    typedef PrePack<NC> PK;
    constexpr int FS = SweepImg<NC>::FS;
    __shared__ int eas[NC];
    for (int n = threadIdx.x; n < NC; n += 128) eas[n] = ea[n];
    __syncthreads();
    for (long i = (long)blockIdx.x * 128 + threadIdx.x; i < 2 * nframes; i += (long)gridDim.x * 128) {
        const long t = i >> 1;
        const int h = (int)(i & 1);
        const double* row = aos + (t < T ? t : 0) * NC;
        const bool live = t < T;
        int eA = -100000;
        if (live)
            for (int n = 0; n < NC; ++n) {
                const double v = row[n];
                if (v != 0.0) {
                    const int e = ilogb(v) - eas[n] + 1;
                    eA = e > eA ? e : eA;
                }
            }
        if (eA == -100000) eA = 0;
        unsigned char* dst = img + (size_t)t * FS;
#pragma unroll
        for (int p = 0; p < PK::PAIRS; ++p) {
            h8 out = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int fl, n;
                PK::slot(p, h, e, fl, n);
                if (n >= 0 && live) {
                    int L[3];
                    pre_split(ldexp(row[n], -eas[n] - eA), L);
                    out[e] = (_Float16)L[fl];
                }
            }
            *(h8*)(dst + p * 32 + h * 16) = out;
        }
        if (h == 0) {
            double g = 0.0;
            if (live)
                for (int n = 0; n < NC; ++n) g += fabs(ldexp(row[n], -eas[n] - eA));
            float4 aux = make_float4((float)g * 1.000001f, 0.f, 0.f, 0.f);
            *(float4*)(dst + PK::PAIRS * 32) = aux;
        } else {
            *(float4*)(dst + PK::PAIRS * 32 + 16) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// ---- counting sort of the frames by a 2-byte key (their cell): perm[slot] = frame ----------------------------------------
constexpr int SORT_TPB = 1024;
constexpr int SORT_MAX_BINS = 8192;

__global__ __launch_bounds__(SORT_TPB) void k_sort_hist(const unsigned short* __restrict__ key, long T, long chunk, int nbins,
                                                        int* __restrict__ hist)
{
    extern __shared__ int lh[];
    for (int i = threadIdx.x; i < nbins; i += SORT_TPB) lh[i] = 0;
    __syncthreads();
    const long t0 = (long)blockIdx.x * chunk, t1 = t0 + chunk < T ? t0 + chunk : T;
    for (long t = t0 + threadIdx.x; t < t1; t += SORT_TPB) {
        const int k = key[t];
        atomicAdd(&lh[k < nbins ? k : nbins - 1], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += SORT_TPB)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// exclusive prefix of the histogram -> cursor; the histogram is zeroed for its next use
__global__ __launch_bounds__(SORT_TPB) void k_sort_base(int* __restrict__ hist, int nbins, int* __restrict__ cursor)
{
    __shared__ int part[SORT_TPB];
    constexpr int PER = SORT_MAX_BINS / SORT_TPB;
    int v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int b = threadIdx.x * PER + k;
        v[k] = b < nbins ? hist[b] : 0;
        sum += v[k];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < SORT_TPB; d <<= 1) {
        const int o = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += o;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int b = threadIdx.x * PER + k;
        if (b < nbins) {
            cursor[b] = run;
            hist[b] = 0;
        }
        run += v[k];
    }
}

__global__ __launch_bounds__(SORT_TPB) void k_sort_scatter(const unsigned short* __restrict__ key, long T, long chunk, int nbins,
                                                           int* __restrict__ cursor, unsigned* __restrict__ perm, long nslots)
{
    extern __shared__ int ls[];  // [nbins] counts, then ranks; [nbins] bases
    int* lc = ls;
    int* lb = ls + nbins;
    for (int i = threadIdx.x; i < nbins; i += SORT_TPB) lc[i] = 0;
    __syncthreads();
    const long t0 = (long)blockIdx.x * chunk, t1 = t0 + chunk < T ? t0 + chunk : T;
    for (long t = t0 + threadIdx.x; t < t1; t += SORT_TPB) {
        const int k = key[t];
        atomicAdd(&lc[k < nbins ? k : nbins - 1], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += SORT_TPB) {
        const int c = lc[i];
        lb[i] = c ? atomicAdd(&cursor[i], c) : 0;
        lc[i] = 0;
    }
    __syncthreads();
    for (long t = t0 + threadIdx.x; t < t1; t += SORT_TPB) {
        int k = key[t];
        k = k < nbins ? k : nbins - 1;
        perm[lb[k] + atomicAdd(&lc[k], 1)] = (unsigned)t;
    }
    // the slots behind the last frame of a partial last block: a frame that exists (the sweep stores nothing for them)
    if (blockIdx.x == 0)
        for (long i = T + threadIdx.x; i < nslots; i += SORT_TPB) perm[i] = (unsigned)(T - 1);
}

// ---- k_sweep_cand ---------------------------------------------------------------------------------------------------------
// cand[f] = c1 | c2 << 13 | amb << 26 | cert << 27   (codebooks of up to 8192 codewords: prefilter_supports)
constexpr unsigned CAND_AMB = 1u << 26, CAND_CERT = 1u << 27;

template <int NC>
__device__ __forceinline__ void sweep_load_tile(h8 (&A)[PrePack<NC>::NU], const h8* __restrict__ cimg, int tile, int lane, bool coarse_only)
{
    typedef PrePack<NC> PK;
    const h8* src = cimg + (size_t)tile * PK::TILE_E + lane;
#pragma unroll
    for (int u = 0; u < PK::NU; ++u)
        if (!coarse_only || SweepImg<NC>::coarse_unique(u)) A[u] = src[u * 64];
}

// the coarse stage of one (tile, column block): NSTEP_C MFMAs, then the smallest coarse key of each lane's 16 values
template <int NC>
__device__ __forceinline__ float sweep_coarse_job(const h8 (&A)[PrePack<NC>::NU], const h8 (&BC)[PrePack<NC>::PAIRS])
{
    typedef PrePack<NC> PK;
    const f16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f16v acc[2];
#pragma unroll
    for (int s = 0; s < SweepImg<NC>::NSTEP_C; ++s) {
        const int lv = PK::step_level(s);
        acc[lv] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[PK::step_unique(s)], BC[PK::step_pair(s)],
                                                         s == PK::level_first(lv) ? zero : acc[lv], 0, 0, 0);
    }
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = __builtin_fmaf(acc[0][r], 512.f, acc[1][r]);
    float m = __builtin_fminf(v[0], v[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) m = __builtin_fminf(m, __builtin_fminf(v[r], v[r + 1]));  // (v_min3_f32)
    return m;
}

// ... software-pipelined: the NSTEP_C MFMAs of this job interleaved 1 : 4 with the epilogue of the PREVIOUS job (its 16 key
// fmas and the minimum tree), so that a wave keeps issuing to the matrix pipe while it digests the last job's values -- run
// one after the other (round 5, first version) a wave spent ~975 cycles per job where the pipe needs 256
template <int NC>
__device__ __forceinline__ float sweep_coarse_job_pinned(f16v (&ACC)[2], const h8 (&A)[PrePack<NC>::NU],
                                                         const h8 (&BC)[PrePack<NC>::PAIRS], const f16v (&PREV)[2])
{
    typedef PrePack<NC> PK;
    const f16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < SweepImg<NC>::NSTEP_C; ++s) {
        const int lv = PK::step_level(s);
        ACC[lv] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[PK::step_unique(s)], BC[PK::step_pair(s)],
                                                         s == PK::level_first(lv) ? zero : ACC[lv], 0, 0, 0);
    }
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = __builtin_fmaf(PREV[0][r], 512.f, PREV[1][r]);
    float m = __builtin_fminf(v[0], v[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) m = __builtin_fminf(m, __builtin_fminf(v[r], v[r + 1]));  // (v_min3_f32)
#pragma unroll
    for (int s = 0; s < SweepImg<NC>::NSTEP_C; ++s) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    }
    return m;
}

template <int NC>
__device__ __forceinline__ void sweep_full_job(const h8 (&A)[PrePack<NC>::NU], const h8 (&BC)[PrePack<NC>::PAIRS], int tile,
                                               float& k1, float& k2, float& k3, int maskv, float ninf)
{
    typedef PrePack<NC> PK;
    const f16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f16v acc[3];
#pragma unroll
    for (int s = 0; s < PK::NSTEP; ++s) {
        const int lv = PK::step_level(s);
        acc[lv] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[PK::step_unique(s)], BC[PK::step_pair(s)],
                                                         s == PK::level_first(lv) ? zero : acc[lv], 0, 0, 0);
    }
    pre_epilogue<NC>(acc, tile, k1, k2, k3, maskv, ninf);
}

// ---- diagnostics (-DE2VQ_SWEEP_STAMP, tools/probe/sweep_stamps.py): where a sweeping wave's cycles go, phase by phase --------
// s_memtime deltas summed per wave, added to a global table at the end; -DE2VQ_SWEEP_STAMP=1 also drains the vector-memory
// counter at the phase ends so that a phase pays for the loads it waits on.  Never defined in the product build.
#ifdef E2VQ_SWEEP_STAMP
__device__ unsigned long long g_sweep_stamps[16];
// experiments on the stamped build only (tools/probe/sweep_stamps.py; results are WRONG under them, timing is the point):
// 1 = stage 1 does not reload its codeword tiles, 2 = stage 2 runs the home tile only, 4 = no rows request,
// 8 = the cell sums' atomics are skipped, 16 = no codeword-row gathers in the evaluation, 32 = rows request of even frames
// only (16-byte aligned pieces), 64 = rows request without the frame-number shuffle
__device__ int g_sweep_exp;
#define SW_EXP(bit) ((sw_exp & (bit)) != 0)
#define SW_STAMP_DECL unsigned long long sw_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sw_conv = 0, sw_t = __builtin_amdgcn_s_memtime(), sw_n = 0; \
    const int sw_exp = __builtin_amdgcn_readfirstlane(g_sweep_exp);
#define SW_STAMP(i)                                                                          \
    {                                                                                        \
        if (E2VQ_SWEEP_STAMP == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
        const unsigned long long sw_now = __builtin_amdgcn_s_memtime();                     \
        sw_acc[i] += sw_now - sw_t;                                                          \
        sw_t = sw_now;                                                                       \
    }
#else
#define SW_STAMP_DECL
#define SW_STAMP(i)
#define SW_EXP(bit) false
#endif

struct SweepCounters {
    unsigned long long flagged;  // (tile, column block) jobs that ran stage 2
    unsigned long long jobs;     // (tile, column block) jobs in all
};

// FUSE (the frames are grouped): the same kernel also finishes its frames -- what k_finish and k_reduce_records do for a
// sweep over ungrouped frames, at a fraction of their cost because a block's frames share their cells:
//   * the block's FP64 rows are gathered into the wave's LDS region by LDS-DMA when the block begins (one frame at a time:
//     2 NC dwords from a uniform base -- they land during stage 1);
//   * behind the certification each lane evaluates ITS frame's candidates with the canonical chain -- the codeword rows
//     gathered from L2, where most lanes of a block ask for the same row --, writes symbol / distortion / the frame's new
//     cell (or lists the frame for the fallback sweep), and keeps the distortion sums in registers;
//   * the contributions to the cell sums -- '+' new cell (the family side table's row for the odd child of a seeded pass),
//     '-' old cell of a mover -- are reduced IN THE BLOCK: per distinct row among them the wave sums the limbs of the
//     contributing frames' LDS rows (lane = coefficient) and adds the 2 NC + 1 totals to the row with one atomic each.
//     A handful of rows per block where the frames are grouped (any number where they are not: still exact, only slow).
// Exact 64-bit integers: the rows equal those of every other accumulate bit for bit.
struct SweepFuse {
    const double* aos;          // row-major resident frames, padded with zero rows to whole blocks
    const double* cbq;
    const DevScalars* sc;
    const u64* l1max_bits;
    unsigned short* sym;        // (optional) outputs by frame
    double* dmin;
    i64* rows;
    i64* fam;                   // side table of a seeded pass (incr == 2)
    int* fb_list;
    unsigned short* cells;      // every frame's cell: read (old) and written (new)
    int incr;                   // 0 full, 1 incremental, 2 seeded
    int M;
};

template <int NC>
struct SweepLds {
    // a frame's row in the wave's LDS region: RP pieces of 16 bytes -- the row's NC * 8 bytes rounded up to whole pieces, plus one
    // more piece where that makes the stride 4 * odd dwords (an 8-byte read per lane then meets a 4-way bank conflict at
    // worst; 4 * even would be 8-way or more).  The LDS-DMA that fills it fetches whole 16-byte pieces: the last piece of a row
    // may reach 8 bytes into the next row (the resident copy is padded by 16 bytes for the last one).
    static constexpr int RP0 = (NC * 8 + 15) / 16;
    static constexpr int RP = (RP0 & 1) ? RP0 : RP0 + 1;
    static constexpr int ROW_BYTES = RP * 16;
    static constexpr int ROWS_BYTES = 64 * ROW_BYTES;
    static constexpr int WAVE_BYTES_FUSE = ROWS_BYTES + 512;  // + the list of flagged tiles (16 bits each)
    static constexpr int FIT = (E2VQ_LDS_BYTES - 256) / WAVE_BYTES_FUSE;
    static constexpr int WAVES_FUSE = FIT >= 8 ? 8 : FIT;
    static constexpr bool FUSE_OK = WAVES_FUSE >= 6 && NC < 64;
};

template <int NC, bool TWO, bool FUSE, bool ONE_BLOCK = false>
__global__ __launch_bounds__((FUSE ? SweepLds<NC>::WAVES_FUSE : 8) * 64, 2) void k_sweep_cand(
    const unsigned char* __restrict__ fimg, const unsigned* __restrict__ perm, long T, long nblocks, const h8* __restrict__ cimg,
    PreScalars* __restrict__ ps, int MT, int idxmask, const unsigned short* __restrict__ prev_sym, int home_mul,
    unsigned* __restrict__ cand, SweepCounters* __restrict__ counters, SweepFuse fz)
{
    typedef PrePack<NC> PK;
    constexpr int NU = PK::NU, FS = SweepImg<NC>::FS;
    typedef SweepLds<NC> SL;
    constexpr int WAVES = FUSE ? SweepLds<NC>::WAVES_FUSE : 8;
    constexpr int WAVE_LDS = FUSE ? SweepLds<NC>::WAVE_BYTES_FUSE : 512;
    constexpr int RS = (2 * NC + 5 + 7) & ~7, NPAD = (NC + 7) & ~7, NH = (NC + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long wave = (long)blockIdx.x * WAVES + wib;
    const long nwaves = (long)gridDim.x * WAVES;
    unsigned char* const wlds = smem + (size_t)wib * WAVE_LDS;
    unsigned short* tlist = (unsigned short*)(wlds + (FUSE ? SweepLds<NC>::ROWS_BYTES : 0));  // flagged tiles: tile | column-block bits << 8 (MT <= 256)
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const unsigned lds_rows = (unsigned)(unsigned long long)(lptr_t)wlds;  // (the region's LDS address, for the asm reads)
    // FUSE: fixed-point scales of the cell sums and of the distortion sums (as k_pass_pre_lds)
    int sh_r = 0, sh_d = 0, sh_d2 = 0;
    bool fast_fix = true, fast_d = true;
    double scale_r = 1.0, scale_d = 1.0, scale_d2 = 1.0;
    i64 dacc0 = 0, dacc1 = 0, dacc2 = 0, dacc3 = 0;
    if constexpr (FUSE) {
        sh_r = fz.sc->sh_r;
        const int Ed = dist_exponent(fz.sc->maxabs, __longlong_as_double((i64)*fz.l1max_bits));
        sh_d = 30 - Ed;
        sh_d2 = 30 - 2 * Ed;
        auto pow2 = [](int e) { return __longlong_as_double((long long)(1023 + e) << 52); };  // |e| <= 1000
        fast_fix = sh_r >= -1000 && sh_r <= 1000;
        fast_d = sh_d >= -1000 && sh_d <= 1000 && sh_d2 >= -1000 && sh_d2 <= 1000;
        scale_r = pow2(fast_fix ? sh_r : 0);
        scale_d = pow2(fast_d ? sh_d : 0);
        scale_d2 = pow2(fast_d ? sh_d2 : 0);
    }
    const float ymax1 = __int_as_float(ps->ymax_bits);
    const float relk = __int_as_float((127 + __builtin_popcount(~idxmask) - 21) << 23);  // 2 rho, rho = 2^-(22-idxbits)
    const float pinf = __builtin_inff();
    const int col = lane & 31, h = lane >> 5;
    unsigned long long nflag = 0, njobs = 0;
    SW_STAMP_DECL

    // A turn of the loop = NB blocks of 64 slots = NCB column blocks of 32.  The fused two-stage kernel takes TWO blocks per
    // turn: what bounds its stage 1 is not the matrix pipe but what the codeword tiles' operand loads cost the CU's
    // vector-memory path (profiles/r05_sweep_experiments.txt: 6 KB per tile and wave; without the reloads the stage runs at
    // the pipe's rate) -- with four column blocks a loaded tile serves four jobs instead of two.  The coarse stage needs the
    // pairs holding frame limbs 0 and 1 only (5 of 7 at P = 36: 20 registers per column block); the others are fetched
    // behind it, block by block, for stage 2.
#ifndef E2VQ_SWEEP_ONE_BLOCK  // (A/B switch of the probes: tools/probe/ab/build_variant.sh)
#define E2VQ_SWEEP_ONE_BLOCK 0
#endif
    // (ONE_BLOCK: shards of fewer blocks than two per wave of the grid -- a turn of two would leave waves without work)
    constexpr int NB = (TWO && FUSE && SweepImg<NC>::TWO_BLOCKS_FIT && !ONE_BLOCK && !E2VQ_SWEEP_ONE_BLOCK) ? 2 : 1;
    constexpr int NCB = 2 * NB;
    const long nturns = (nblocks + NB - 1) / NB;

    // slots -> frames; lane (h, col) holds lane half h of the granules of slots col and 32 + col of each block.  Loop-carried:
    // the frames, the home tile and the B operands of a turn are requested while the turn before it is still at work
    // (three dependent global latencies -- list, home cell, granules -- would otherwise open every turn)
    auto slot_frames = [&](long blk, unsigned& fa, unsigned& fb) {
        int cq = col;  // (afresh at every call: see load_pairs)
        asm volatile("" : "+v"(cq));
        long s0 = blk * 64 + cq, s1 = s0 + 32;
        s0 = s0 < T ? s0 : T - 1;
        s1 = s1 < T ? s1 : T - 1;
        fa = perm ? perm[s0] : (unsigned)s0;
        fb = perm ? perm[s1] : (unsigned)s1;
    };
    auto home_of = [&](unsigned fa) {
        int hm = 0;
        if (TWO && home_mul) {
            const unsigned fh = (unsigned)__builtin_amdgcn_readfirstlane((int)fa);
            hm = (home_mul * (int)prev_sym[fh]) >> 5;
            hm = __builtin_amdgcn_readfirstlane(hm < MT ? hm : MT - 1);
        }
        return hm;
    };
    h8 B[NCB][PK::PAIRS];
    float g[NCB];
    unsigned fn[NCB];
    int home_n = 0;
    // WHICH: 1 = the coarse stage's pairs, 2 = the others, 4 = g
    auto load_pairs = [&](h8 (&Bc)[PK::PAIRS], float& gg, unsigned fr, int which) {
        // (the lane half's offset made afresh at every call: as a loop invariant the per-lane base pointer is one more pair of
        // registers alive across stage 1, whose budget has none left)
        int hq = h * 16;
        asm volatile("" : "+v"(hq));
        const unsigned char* p0 = fimg + (size_t)fr * FS + hq;
#pragma unroll
        for (int p = 0; p < PK::PAIRS; ++p)
            if (which & (SweepImg<NC>::coarse_pair(p) ? 1 : 2)) Bc[p] = *(const h8*)(p0 + p * 32);
        if (which & 4) gg = *(const float*)(p0 - hq + PK::PAIRS * 32);
    };
    auto next_frames = [&](long turn) {
#pragma unroll
        for (int hh = 0; hh < NB; ++hh) slot_frames(turn * NB + hh, fn[2 * hh], fn[2 * hh + 1]);
    };
    auto request_B = [&]() {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) load_pairs(B[cb], g[cb], fn[cb], NB == 2 ? 5 : 7);
    };
    if (wave < nturns) {
        next_frames(wave);
        home_n = home_of(fn[0]);
        request_B();
    }
    for (long turn = wave; turn < nturns; turn += nwaves) {
        unsigned f[NCB];
        float gc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) f[cb] = fn[cb], gc[cb] = g[cb];
        const int home = home_n;
        const long tn = turn + nwaves < nturns ? turn + nwaves : turn;  // (the wave's last turn asks for itself again: no load is conditional)
        SW_STAMP(0)  // the turn's B operands (requested behind the previous turn's evaluation) are there
        if constexpr (!TWO) next_frames(tn);  // (two-stage: behind stage 1, whose registers are the kernel's peak)
        unsigned short oldraw = 0;  // (turned into the old cell further down: a use up here would wait for the requests below)
        // FUSE: a block's FP64 rows -> the wave's LDS region, frame by frame (the previous block's reads of it are complete).
        // (RP instructions of 64 pieces of 16 bytes -- an LDS-DMA instruction costs its issue whatever its width: fetched dword
        // by dword the request took 18 k cycles per block --: lane -> piece q = 64 k + lane of the block's padded row-major image
        // = piece q % RP of slot q / RP, whose frame number comes from that slot's lane.  The lane index afresh from the
        // hardware: computed from the kernel's `lane`, every instruction's slot and piece are invariants of the loop -- hoisted,
        // kept in 2 RP registers for the whole kernel, and spilled)
        auto request_rows = [&](unsigned fs) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (fz.incr) oldraw = fz.cells[fs];
            const int lane_q = pre_fresh_lane();
            // (the next instruction's frame numbers are shuffled while this one is issued: 0.70 -> 0.68 ms at M = 1024)
            unsigned fr_next = (unsigned)__shfl((int)fs, (int)((unsigned)lane_q / (unsigned)SL::RP), 64);
#pragma unroll 1
            for (int k = 0; k < (SW_EXP(4) ? 0 : SL::RP); ++k) {
                const unsigned q = (unsigned)(k * 64 + lane_q);
                const unsigned slot = q / (unsigned)SL::RP, pc = q - slot * (unsigned)SL::RP;
                unsigned fr_ = fr_next;
                {
                    const unsigned qn = (unsigned)((k + 1 < SL::RP ? k + 1 : k) * 64 + lane_q);
                    fr_next = (unsigned)__shfl((int)fs, (int)(qn / (unsigned)SL::RP), 64);
                }
                if (SW_EXP(64)) fr_ = fs;
                if (SW_EXP(32)) fr_ &= ~1u;
                // (a padding piece fetches the row's first piece again: any valid address)
                const char* gp = (const char*)fz.aos + ((size_t)fr_ * (size_t)(NC * 8) + (size_t)((pc < (unsigned)SL::RP0 ? pc : 0u) * 16u));
                __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(wlds + k * 1024), 16, 0, 0);
            }
#ifdef E2VQ_SWEEP_STAMP
            {   // (issue side of the requests only: no drain)
                const unsigned long long sw_now = __builtin_amdgcn_s_memtime();
                sw_acc[4] += sw_now - sw_t;
                sw_t = sw_now;
            }
#endif
        };
        // the first block's rows are needed behind stage 2 -- in this wave's in-order queue they only delay the first tile
        // (whose wait, behind a loop of requests, is for everything outstanding: the last row's latency once per turn)
        if constexpr (FUSE) request_rows(h ? f[1] : f[0]);
        float k1[NCB], k2[NCB], k3[NCB];
        int ntl = 0;

        if constexpr (TWO) {
            // ---- stage 1: every tile, weight levels 0 and 1; tiles in cyclic order from the turn's home tile -----------------
            // 2 x 1.27 x E2 of the lane's frames (coarse-key units; header)
            // Two register sets of coarse granules, one tile requested ahead (measured: one, two or three tiles of distance
            // make no difference)
            auto tile_at = [&](int i) {
                i = i < MT ? i : MT - 1;
                return home + i < MT ? home + i : home + i - MT;
            };
            constexpr int ND = 2;  // register sets
            h8 A[ND][NU];
#pragma unroll
            for (int k = 0; k < ND - 1; ++k) sweep_load_tile<NC>(A[k], cimg, tile_at(k), lane, true);
            // (per column block only U lives across the jobs: the threshold is made from it and the frame's g when a job is
            // digested -- four operations against eight registers)
            float U[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) U[cb] = pinf;
            // accumulators of two jobs: each job digests the values of the one before it (the next column block of the same
            // tile, or the last one of the tile before); the "previous" values of the very first job are +inf -- its digest runs
            // like any other (a branch around it would take the job's epilogue out from between its MFMAs), changes nothing
            // and is barred from flagging --, and one more epilogue follows the loop
            f16v acc0[2], acc1[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[0][r] = 0.f, acc1[1][r] = pinf;
            unsigned bits_acc = 0u;  // flags of the column blocks digested so far of the tile whose last one is still to come
            int tile_prev = 0;
            // column block CB of `tile`: flag, U, threshold; with the tile's last column block its list entry
            auto digest = [&](float m, float& Uc, float gcb, int cb, int tile, bool valid) {
                const float Dc = 2.54f * (257.f * (gcb + ymax1) + (129.f * NC + 2.f));
                const float thrc = Uc > 0.f && Uc < pinf ? __builtin_fmaf(Uc, 1.000001f, Dc) : pinf;
                const bool fl = !(m > thrc);  // (negated comparison: a NaN key flags its tile)
                Uc = __builtin_fminf(Uc, m);
                bits_acc |= (valid && __ballot(fl) != 0) ? 1u << cb : 0u;
                if (cb == NCB - 1) {
                    if (bits_acc) {  // (wave-uniform; every lane stores the same word)
                        tlist[ntl] = (unsigned short)((unsigned)tile | bits_acc << 8);
                        ++ntl;
                    }
                    bits_acc = 0u;
                }
            };
            for (int i0 = 0; i0 < MT; i0 += ND) {
#pragma unroll
                for (int k = 0; k < ND; ++k) {
                    const int i = i0 + k;
                    if (!SW_EXP(1)) sweep_load_tile<NC>(A[(k + ND - 1) % ND], cimg, tile_at(i + ND - 1), lane, true);
                    if (i < MT) {  // (wave-uniform)
                        const int tile = tile_at(i);
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb) {
                            const float m = (cb & 1) ? sweep_coarse_job_pinned<NC>(acc1, A[k], B[cb], acc0)
                                                     : sweep_coarse_job_pinned<NC>(acc0, A[k], B[cb], acc1);
                            if (cb == 0)
                                digest(m, U[NCB - 1], gc[NCB - 1], NCB - 1, tile_prev, i > 0);
                            else
                                digest(m, U[cb - 1], gc[cb - 1], cb - 1, tile, true);
                        }
                        tile_prev = tile;
                    }
                }
            }
            {   // the last job's values
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = __builtin_fmaf(acc1[0][r], 512.f, acc1[1][r]);
                float m = __builtin_fminf(v[0], v[1]);
#pragma unroll
                for (int r = 2; r < 16; r += 2) m = __builtin_fminf(m, __builtin_fminf(v[r], v[r + 1]));
                digest(m, U[NCB - 1], gc[NCB - 1], NCB - 1, tile_prev, true);
            }
            njobs += (unsigned long long)NCB * MT;
            if (SW_EXP(2)) {
                ntl = 1;
                tlist[0] = (unsigned short)((unsigned)home | ((1u << NCB) - 1u) << 8);
            }
            SW_STAMP(1)  // stage 1
            next_frames(tn);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) k1[cb] = k2[cb] = k3[cb] = __int_as_float(0x7f7fffff);
        // (operands of the key epilogue that have to be registers; made behind stage 1, whose budget has none to spare)
        int maskv = idxmask;
        asm volatile("" : "+v"(maskv));
        float ninf = -__builtin_inff();
        asm volatile("" : "+v"(ninf));
        if constexpr (!TWO) {
            // ---- one stage: every tile with all its k-steps (frames that are not grouped, or data that flags most tiles) ------
            h8 Acur[NU], Anext[NU];
            sweep_load_tile<NC>(Acur, cimg, 0, lane, false);
            for (int t = 0; t < MT; ++t) {
                sweep_load_tile<NC>(Anext, cimg, t + 1 < MT ? t + 1 : t, lane, false);
                sweep_full_job<NC>(Acur, B[0], t, k1[0], k2[0], k3[0], maskv, ninf);
                sweep_full_job<NC>(Acur, B[1], t, k1[1], k2[1], k3[1], maskv, ninf);
#pragma unroll
                for (int u = 0; u < NU; ++u) Acur[u] = Anext[u];
            }
            njobs += 2ull * MT;
            nflag += 2ull * MT;
            home_n = 0;
        }
#pragma unroll
        for (int hh = 0; hh < NB; ++hh) {
        const long b = turn * NB + hh;
        // (the second block's rows: its turn of the wave's LDS region begins when the first block's sums are done; stage 2 of
        // the block runs behind the request -- a wave's vector-memory operations return in order, so the block's first tile
        // arrives with its rows, and its k-steps cover what is left of the evaluation's wait)
        if (FUSE && hh) request_rows(h ? f[NCB - 1] : f[NCB - 2]);
        if constexpr (TWO) {
            // ---- stage 2: the block's flagged tiles with all their k-steps and the key epilogue -------------------------------
            if constexpr (NB == 2) {
                // (the second block's coarse pairs again: 40 registers that the first block's stage 2 and tail need more)
                load_pairs(B[2 * hh], g[2 * hh], f[2 * hh], hh == 0 ? 2 : 3);
                load_pairs(B[2 * hh + 1], g[2 * hh + 1], f[2 * hh + 1], hh == 0 ? 2 : 3);
            }
            const unsigned sh = 8u + 2u * hh;
            auto next_e = [&](int j) {
                while (j < ntl && ((tlist[j] >> sh) & 3u) == 0u) ++j;
                return j;
            };
            int j = next_e(0);
            if (j < ntl) {
                h8 Acur[NU], Anext[NU];
                unsigned e = tlist[j];
                sweep_load_tile<NC>(Acur, cimg, (int)(e & 0xffu), lane, false);
                while (j < ntl) {
                    const int jn = next_e(j + 1);
                    const unsigned en = tlist[jn < ntl ? jn : j];
                    sweep_load_tile<NC>(Anext, cimg, (int)(en & 0xffu), lane, false);
                    const int tile = (int)(e & 0xffu);
                    const unsigned eb = e >> sh;
                    if (eb & 1u) sweep_full_job<NC>(Acur, B[2 * hh], tile, k1[2 * hh], k2[2 * hh], k3[2 * hh], maskv, ninf);
                    if (eb & 2u) sweep_full_job<NC>(Acur, B[2 * hh + 1], tile, k1[2 * hh + 1], k2[2 * hh + 1], k3[2 * hh + 1], maskv, ninf);
                    nflag += (eb & 1u) + ((eb >> 1) & 1u);
#pragma unroll
                    for (int u = 0; u < NU; ++u) Acur[u] = Anext[u];
                    e = en;
                    j = jn;
                }
            }
            if (hh == 0) home_n = home_of(fn[0]);  // (the next turn's list entries, requested behind stage 1, have arrived)
        }
        SW_STAMP(2)  // stage 2 (or the one-stage loop)
        // the next turn's B operands: their registers are free from here on (FUSE: requested further down)
        if constexpr (!FUSE) request_B();

        const unsigned f0 = NB == 2 && hh ? f[NCB - 2] : f[0], f1 = NB == 2 && hh ? f[NCB - 1] : f[1];
        const float gc0 = NB == 2 && hh ? gc[NCB - 2] : gc[0], gc1 = NB == 2 && hh ? gc[NCB - 1] : gc[1];
        const float ka1 = NB == 2 && hh ? k1[NCB - 2] : k1[0], ka2 = NB == 2 && hh ? k2[NCB - 2] : k2[0],
                    ka3 = NB == 2 && hh ? k3[NCB - 2] : k3[0];
        const float kb1 = NB == 2 && hh ? k1[NCB - 1] : k1[1], kb2 = NB == 2 && hh ? k2[NCB - 1] : k2[1],
                    kb3 = NB == 2 && hh ? k3[NCB - 1] : k3[1];
        // ---- lane = slot b * 64 + lane: merge the two lane halves of its frame's keys, certify the top two ------------------
        const int hb = h << 2;
        float a1, a2, a3, q1, q2, q3;
        {
            const float o1 = __int_as_float(__float_as_int(ka1) | hb), o2 = __int_as_float(__float_as_int(ka2) | hb),
                        o3 = __int_as_float(__float_as_int(ka3) | hb);
            const float r1 = __int_as_float(__float_as_int(kb1) | hb), r2 = __int_as_float(__float_as_int(kb2) | hb),
                        r3 = __int_as_float(__float_as_int(kb3) | hb);
            a1 = h ? r1 : o1, a2 = h ? r2 : o2, a3 = h ? r3 : o3;  // own frame's keys (slot 32 h + col)
            q1 = h ? o1 : r1, q2 = h ? o2 : r2, q3 = h ? o3 : r3;  // the partner's frame's keys
        }
        const float b1 = __shfl_xor(q1, 32, 64), b2 = __shfl_xor(q2, 32, 64), b3 = __shfl_xor(q3, 32, 64);
        const float t3 = med3f(a2, a3, b1), t2 = med3f(a1, a2, b1), t1 = med3f(a1, b1, ninf);
        const float u3 = med3f(t2, t3, b2), u2 = med3f(t1, t2, b2);
        const float w3 = med3f(u2, u3, b3);
        const float gfr = h ? gc1 : gc0;
        const unsigned fr = h ? f1 : f0;
        float tau = 1.27f * (512.f * (gfr + ymax1 + (NC + 4.0f)) + relk * t1);
        if constexpr (FUSE) {  // (round 6: the smallest key with its tile's own tolerance: k_pre_codebook's table, k_pass_pre_lds)
            const int tl1 = (__float_as_int(t1) & ~idxmask) >> 5;
            const float2 tt = ((const float2*)(cimg + (size_t)MT * PK::TILE_E))[tl1 < MT ? tl1 : 0];
            tau = 1.27f * (256.f * __builtin_fmaf(tt.x, gfr, tt.y) + 256.f * (gfr + ymax1 + (NC + 4.0f)) + relk * t1);
        }
        const bool cert = t1 >= 1.0e-30f && t1 < 1.0e37f && w3 > t1 + tau;
        const bool amb = !(u2 > t1 + tau);
        const unsigned c1 = (unsigned)(__float_as_int(t1) & ~idxmask), c2 = (unsigned)(__float_as_int(u2) & ~idxmask);
        const bool live = b * 64 + lane < T;
        if constexpr (!FUSE) {
            if (live) cand[fr] = c1 | c2 << 13 | (amb ? CAND_AMB : 0u) | (cert ? CAND_CERT : 0u);
        } else {
            // ---- exact evaluation: the canonical chain acc = fma(r[n], cq[n], acc), n ascending from +0.0; r from the frame's
            // LDS row (inline asm, eight at a time, the fmas pinned behind them: see k_finish), cq gathered from L2 ----------
            double2 x[NH], y[NH];
            const unsigned la = lds_rows + (unsigned)(lane * SL::ROW_BYTES);
            {
                const double2* r1 = (const double2*)(fz.cbq + (size_t)c1 * NPAD);
                const double2* r2 = (const double2*)(fz.cbq + (size_t)c2 * NPAD);
#pragma unroll
                for (int n2 = 0; n2 < NH; ++n2) x[n2] = SW_EXP(16) ? make_double2(1.0, 1.0) : r1[n2];  // (rows are padded to a multiple of 8 doubles)
                // (the runner-up's row only where it can matter: a gather costs the texture path a request per lane)
#pragma unroll
                for (int n2 = 0; n2 < NH; ++n2) y[n2] = make_double2(0.0, 0.0);
                if (amb && !SW_EXP(16)) {
#pragma unroll
                    for (int n2 = 0; n2 < NH; ++n2) y[n2] = r2[n2];
                }
            }
            // the codeword rows -- and, older, this block's rows (LDS-DMA) -- have arrived
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            asm volatile("" ::: "memory");
            double d1 = 0.0, d2 = 0.0;
#pragma unroll
            for (int b0 = 0; b0 < NC; b0 += 8) {
                double fv[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (b0 + k < NC) asm volatile("ds_read_b64 %0, %1" : "=v"(fv[k]) : "v"(la + (unsigned)((b0 + k) * 8)));
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(fv[0]), "+v"(fv[1]), "+v"(fv[2]), "+v"(fv[3]), "+v"(fv[4]), "+v"(fv[5]), "+v"(fv[6]), "+v"(fv[7]));
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (b0 + k < NC) {
                        d1 = __builtin_fma(fv[k], ((b0 + k) & 1) ? x[(b0 + k) >> 1].y : x[(b0 + k) >> 1].x, d1);
                        d2 = __builtin_fma(fv[k], ((b0 + k) & 1) ? y[(b0 + k) >> 1].y : y[(b0 + k) >> 1].x, d2);
                    }
                asm volatile("" : "+v"(d1), "+v"(d2));
            }
            const bool take_b = amb && (d2 < d1 || (d2 == d1 && c2 < c1));
            const double best = take_b ? d2 : d1;
            int idx = take_b ? (int)c2 : (int)c1;
            // the next turn's B operands: the codeword rows' registers are free now
            if (hh == NB - 1) request_B();
            const bool skip = !cert;
            idx = skip ? 0 : idx;
#ifdef E2VQ_SWEEP_STAMP
            {
                const unsigned long long sw_now = __builtin_amdgcn_s_memtime();
                sw_acc[5] += sw_now - sw_t;
                sw_t = sw_now;
            }
#endif
            // ---- outputs; uncertified frames go to the fallback list ----
            if (live) {
                if (skip) {
                    fz.fb_list[atomicAdd(&ps->fb_count, 1)] = (int)fr;
                } else {
                    if (fz.sym) fz.sym[fr] = (unsigned short)idx;
                    if (fz.dmin) fz.dmin[fr] = best;
                    fz.cells[fr] = (unsigned short)idx;
                    const double e = best - 1.0;
                    int h0, l0, h1, l1;
                    if (fast_d) {  // (kernel-uniform; same limbs as fix2: vq_fixed.h)
                        fix2_mul(e, scale_d, h0, l0);
                        fix2_mul(e * e, scale_d2, h1, l1);
                    } else {
                        fix2(e, sh_d, h0, l0);
                        fix2(e * e, sh_d2, h1, l1);
                    }
                    dacc0 += h0;
                    dacc1 += l0;
                    dacc2 += h1;
                    dacc3 += l1;
                }
            }
#ifdef E2VQ_SWEEP_STAMP
            {
                const unsigned long long sw_now = __builtin_amdgcn_s_memtime();
                sw_acc[6] += sw_now - sw_t;
                sw_t = sw_now;
            }
#endif
            // ---- the block's contributions to the cell sums, reduced in the block ----
            {
                const int incr = fz.incr;
                const int oldc = (incr == 2 ? 2 : 1) * (int)oldraw;
                const bool mov = live && !skip && (!incr || oldc != idx);
                const bool infam = incr == 2 && idx == oldc + 1;
                const int keyN = infam ? fz.M + (oldc >> 1) : idx;  // (rows of the side table: behind the codebook's)
                const bool hasN = mov, hasO = mov && incr != 0 && !infam;
                u64 mN = __ballot(hasN), mO = __ballot(hasO);
                // every contributing lane turns ITS frame's row into limb pairs in place (8 bytes either way): the conversion
                // -- six FP64 operations per value -- is paid once per block with all lanes at work, and adding a frame to
                // a row is then one 4-byte LDS read and one 64-bit add per lane
                if (mov) {
                    double* fr = (double*)(wlds + lane * SL::ROW_BYTES);
                    if (fast_fix) {
#pragma unroll 4
                        for (int n = 0; n < NC; ++n) {
                            int hh, ll;
                            fix2_mul(fr[n], scale_r, hh, ll);
                            *(int2*)&fr[n] = make_int2(hh, ll);
                        }
                    } else {
#pragma unroll 1
                        for (int n = 0; n < NC; ++n) {
                            int hh, ll;
                            fix2(fr[n], sh_r, hh, ll);
                            *(int2*)&fr[n] = make_int2(hh, ll);
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef E2VQ_SWEEP_STAMP
                {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const unsigned long long sw_now = __builtin_amdgcn_s_memtime();
                    sw_conv += sw_now - sw_t;
                    sw_t = sw_now;
                }
#endif
                // lane e < 2 NC: limb e of the row (2 n = hi, 2 n + 1 = lo of coefficient n); rows of more than 32 coefficients:
                // the limbs beyond the 64th ride in a second value of the lanes e < 2 NC - 64
                const int* lrows = (const int*)wlds;
                constexpr int RW = SL::ROW_BYTES / 4;
                while ((mN | mO) != 0) {
                    const int key = mN != 0 ? __builtin_amdgcn_readlane(keyN, (int)__builtin_ctzll(mN))
                                            : __builtin_amdgcn_readlane(oldc, (int)__builtin_ctzll(mO));
                    const u64 sN = __ballot(hasN && keyN == key), sO = __ballot(hasO && oldc == key);
                    i64 a0 = 0, a1 = 0;
                    for (int sgn = 0; sgn < 2; ++sgn) {
                        u64 w = sgn ? sO : sN;
                        while (w != 0) {
                            const int j = (int)__builtin_ctzll(w);
                            w &= w - 1;
                            const int v0 = lane < 2 * NC ? lrows[j * RW + lane] : 0;
                            const int v1 = (2 * NC > 64 && lane < 2 * NC - 64) ? lrows[j * RW + 64 + lane] : 0;
                            a0 += sgn ? -(i64)v0 : (i64)v0;
                            a1 += sgn ? -(i64)v1 : (i64)v1;
                        }
                    }
                    i64* dst = key < fz.M ? fz.rows + (size_t)key * RS : fz.fam + (size_t)(key - fz.M) * RS;
                    if (SW_EXP(8)) a0 = 0, a1 = 0;
                    if (lane < 2 * NC && a0 != 0) atomicAdd((u64*)&dst[lane], (u64)a0);
                    if (2 * NC > 64 && lane < 2 * NC - 64 && a1 != 0) atomicAdd((u64*)&dst[64 + lane], (u64)a1);
                    if (lane == 63 && !SW_EXP(8)) {
                        const i64 cnt = (i64)__builtin_popcountll(sN) - (i64)__builtin_popcountll(sO);
                        if (cnt != 0) atomicAdd((u64*)&dst[2 * NC], (u64)cnt);
                    }
                    mN &= ~sN;
                    mO &= ~sO;
#ifdef E2VQ_SWEEP_STAMP
                    sw_acc[7] += 1;  // (rows added to: a count, not cycles)
#endif
                }
            }
        }
#ifdef E2VQ_SWEEP_STAMP
        {   // (merge, certification, store -- without the drain: the B loads just issued belong to the next block's stamp 0)
            const unsigned long long sw_now = __builtin_amdgcn_s_memtime();
            sw_acc[3] += sw_now - sw_t;
            sw_t = sw_now;
            sw_n += 1;
        }
#endif
    }  // (blocks of the turn)
    }
#ifdef E2VQ_SWEEP_STAMP
    if (lane == 0) {
        for (int k = 0; k < 8; ++k) atomicAdd(&g_sweep_stamps[k], sw_acc[k]);
        atomicAdd(&g_sweep_stamps[8], sw_n);
        atomicAdd(&g_sweep_stamps[9], 1ull);
        atomicAdd(&g_sweep_stamps[10], nflag);
        atomicAdd(&g_sweep_stamps[11], njobs);
        atomicAdd(&g_sweep_stamps[12], sw_conv);
    }
#endif
    if (counters && lane == 0 && njobs) {
        atomicAdd(&counters->flagged, nflag);
        atomicAdd(&counters->jobs, njobs);
    }
    if constexpr (FUSE) {  // the wave's distortion sums -> the distortion columns of one row (any row: totals only)
        i64 v[4] = {dacc0, dacc1, dacc2, dacc3};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            for (int dd = 32; dd >= 1; dd >>= 1) v[k] += __shfl_xor(v[k], dd, 64);
            if (lane == 0 && v[k] != 0) atomicAdd((u64*)&fz.rows[(long)(wave % (32 * MT)) * RS + 2 * NC + 1 + k], (u64)v[k]);
        }
    }
}

// ---- k_finish ---------------------------------------------------------------------------------------------------------------
// Every frame once, in its natural order; bound by reading the FP64 rows (296 B per frame at P = 36) -- if enough of them are
// in flight: twelve waves per CU (three per SIMD), each working on HALF blocks of 32 frames whose rows arrive in
// the wave's LDS region by LDS-DMA (no registers).  Lanes 0..31 = the half block's frames evaluating their first candidate,
// lanes 32..63 = the same frames' runner-up (only where its key was within reach): the canonical chain
// acc = fma(r[n], cq[n], acc), n ascending from +0.0, r from the frame's LDS row, cq gathered from L2.
// A wave's turn: gather the codeword rows (i) -> wait for them and for rows (i), requested a turn ago -> chains (i) -> request
// rows (i + 1), load candidates (i + 1) -> outputs, distortion sums, records (i).  (Its vector-memory operations retire in
// order: waiting for the short gathers behind a second buffer's requests would wait for those too -- the other eleven waves
// of the CU are what keeps rows in flight.)
// Outputs: symbol / distortion (or the fallback list); the distortion sums per lane, reduced once at the end of the kernel;
// the frame's contribution to the cell sums as records -- (frame, cell within its bin, sign) into this workgroup's region of
// the bin, the slot from one LDS atomic per record -- for k_reduce_records: the statements of k_pass_pre_lds<.., 2> behind its
// certification, unchanged in what they compute.
template <int NC>
struct FinLds {
    static constexpr int HALF_BYTES = 32 * NC * 8;
    static constexpr int K16 = HALF_BYTES / 1024, K4 = (HALF_BYTES - K16 * 1024) / 256;
    static_assert(K16 * 1024 + K4 * 256 == HALF_BYTES, "a half block of frames is a whole number of 256-byte pieces");
    static constexpr int WAVE_BYTES = HALF_BYTES + 256;  // + the cells of the previous pass (32 u16, fetched as 64 dwords)
    // twelve waves per workgroup = three per SIMD: the codeword row of a lane is 2 * NH registers, and 128 of them do not
    // hold the chains (the records' plan allows for any wave count up to sixteen: prefilter_records_plan)
    static constexpr int WAVES = 12;
    static_assert(WAVES * WAVE_BYTES + 512 <= E2VQ_LDS_BYTES || !PreLds<NC>::OK, "twelve half-block regions fit");
};

template <int NC>
__global__ __launch_bounds__(FinLds<NC>::WAVES * 64) void k_finish(const double* __restrict__ aos, long T, long nblocks,
                                                   const unsigned* __restrict__ cand, PreScalars* __restrict__ ps,
                                                   const double* __restrict__ cbq, int MT, const DevScalars* __restrict__ sc,
                                                   const u64* __restrict__ l1max_bits, unsigned short* __restrict__ sym,
                                                   double* __restrict__ dmin, i64* __restrict__ rows, int* __restrict__ fb_list,
                                                   unsigned short* __restrict__ prev_sym, int incr, PreRec rec,
                                                   SweepCounters* __restrict__ counters, unsigned long long* host_counters)
{
    typedef FinLds<NC> FL;
    constexpr int W = FL::WAVES;
    constexpr int RS = (2 * NC + 5 + 7) & ~7, NPAD = (NC + 7) & ~7, NH = (NC + 1) / 2;
    // the two-stage sweep in front of this kernel left its counters: to the host (it adapts), and zero for the next sweep
    if (counters && host_counters && blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(host_counters, counters->flagged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_counters + 1, counters->jobs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        counters->flagged = 0;
        counters->jobs = 0;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long wave = (long)blockIdx.x * W + wib;
    const long nwaves = (long)gridDim.x * W;
    const long nhalf = (T + 31) / 32;
    (void)nblocks;
    unsigned char* wbase = smem + (size_t)wib * FL::WAVE_BYTES;
    int* const rcnt = (int*)(smem + (size_t)W * FL::WAVE_BYTES);
    if (threadIdx.x < 64) rcnt[threadIdx.x] = 0;
    __syncthreads();
    const int Ed = dist_exponent(sc->maxabs, __longlong_as_double((i64)*l1max_bits));
    const int sh_d = 30 - Ed, sh_d2 = 30 - 2 * Ed;
    auto pow2 = [](int e) { return __longlong_as_double((long long)(1023 + e) << 52); };  // |e| <= 1000
    const bool fast_d = sh_d >= -1000 && sh_d <= 1000 && sh_d2 >= -1000 && sh_d2 <= 1000;
    const double scale_d = pow2(fast_d ? sh_d : 0), scale_d2 = pow2(fast_d ? sh_d2 : 0);
    const int fl = lane & 31;
    const bool second = lane >= 32;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const unsigned lds_rows = (unsigned)(unsigned long long)(lptr_t)wbase;  // the region's LDS address (for the asm reads)
    // rows of half block hb (the resident copy is padded with zero rows to whole 64-frame blocks) and its cells of the
    // previous pass -> the wave's region
    auto request = [&](long hb) {
        const char* g = (const char*)(aos + hb * (long)(32 * NC));
#pragma unroll
        for (int k = 0; k < FL::K16; ++k)
            __builtin_amdgcn_global_load_lds((gptr_t)(g + k * 1024 + lane * 16), (lptr_t)(wbase + k * 1024), 16, 0, 0);
#pragma unroll
        for (int k = 0; k < FL::K4; ++k)
            __builtin_amdgcn_global_load_lds((gptr_t)(g + FL::K16 * 1024 + k * 256 + lane * 4), (lptr_t)(wbase + FL::K16 * 1024 + k * 256),
                                             4, 0, 0);
        if (incr)  // (64 dwords: the half block's 32 cells and 192 bytes beyond them, which the array is padded for)
            __builtin_amdgcn_global_load_lds((gptr_t)((const char*)(prev_sym + hb * 32) + lane * 4), (lptr_t)(wbase + FL::HALF_BYTES), 4, 0,
                                             0);
    };
    auto load_cand = [&](long hb) { return hb < nhalf && hb * 32 + fl < T ? cand[hb * 32 + fl] : 0u; };
    // this lane's codeword row of the half block whose candidates are `cd` (the runner-up's only where it can matter)
    double2 x[NH];
    auto gather = [&](unsigned cd) {
        const int c = (int)(second ? (cd >> 13) & 0x1fffu : cd & 0x1fffu);
        const double2* r = (const double2*)(cbq + (long)c * NPAD);
        // (the runner-up's row only where its value will be looked at: the gathers -- 64 lanes x 16 bytes out of 64 different
        // rows per instruction -- are what this kernel is bound by, not the rows from HBM)
#pragma unroll
        for (int n2 = 0; n2 < NH; ++n2) x[n2] = make_double2(0.0, 0.0);
        if (!second || (cd & CAND_AMB) != 0) {
#pragma unroll
            for (int n2 = 0; n2 < NH; ++n2) x[n2] = r[n2];  // (rows are padded to a multiple of 8 doubles)
        }
    };
    i64 dacc0 = 0, dacc1 = 0, dacc2 = 0, dacc3 = 0;
    unsigned cd_nxt = 0u;
    if (wave < nhalf) {
        request(wave);
        cd_nxt = load_cand(wave);
    }
    for (long hb = wave; hb < nhalf; hb += nwaves) {
        const long t = hb * 32 + fl;
        const bool live = t < T;
        const unsigned cd = cd_nxt;
        const bool cert = (cd & CAND_CERT) != 0, amb = (cd & CAND_AMB) != 0;
        const int c1 = (int)(cd & 0x1fffu), c2 = (int)((cd >> 13) & 0x1fffu);
        // ---- chains: this lane's codeword row from L2; the half's rows, requested an iteration ago, land meanwhile (the
        // compiler waits for the LDS-DMA in front of the LDS reads itself) ----
        gather(cd);
        // everything requested so far has arrived: the codeword rows, and -- older -- this half's rows (LDS-DMA: the compiler
        // does not order LDS reads behind it by itself)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        asm volatile("" ::: "memory");
        // The frame's coefficients come from its LDS row eight at a time, by inline asm with its own counter wait: left to
        // the compiler all 37 reads are hoisted in front of the chain -- 74 registers beside the 74 of the codeword row.
        double d = 0.0;
        const unsigned la = lds_rows + (unsigned)(fl * NC * 8);
#pragma unroll
        for (int b0 = 0; b0 < NC; b0 += 8) {
            double f[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (b0 + k < NC) asm volatile("ds_read_b64 %0, %1" : "=v"(f[k]) : "v"(la + (unsigned)((b0 + k) * 8)));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (b0 + k < NC) d = __builtin_fma(f[k], ((b0 + k) & 1) ? x[(b0 + k) >> 1].y : x[(b0 + k) >> 1].x, d);
            asm volatile("" : "+v"(d));  // (pins this batch's fmas in front of the next batch's reads: volatile asm keeps its
                                         // order, plain arithmetic does not -- the optimiser sank all 37 behind the last read)
        }
        int old = 0;
        if (incr) {
            unsigned o16;
            asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(o16) : "v"(lds_rows + (unsigned)(FL::HALF_BYTES + 2 * fl)));
            old = (incr == 2 ? 2 : 1) * (int)o16;
        }
        // ---- the next half: its rows (the region's reads are done) and its candidates ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const long hn = hb + nwaves;
        if (hn < nhalf) {  // (wave-uniform)
            request(hn);
            cd_nxt = load_cand(hn);
        }
        const double dB = __shfl(d, fl + 32, 64);  // (lanes 0..31: their frame's runner-up)
        const bool take_b = amb && (dB < d || (dB == d && c2 < c1));
        const double best = take_b ? dB : d;
        int idx = take_b ? c2 : c1;
        const bool skip = !cert;
        idx = skip ? 0 : idx;
        const bool own = !second && live;  // this lane speaks for a frame
        // ---- outputs; uncertified frames go to the fallback list -----------------------------------------------------------
        if (own) {
            if (skip) {
                fb_list[atomicAdd(&ps->fb_count, 1)] = (int)t;
            } else {
                if (sym) sym[t] = (unsigned short)idx;
                if (dmin) dmin[t] = best;
                const double e = best - 1.0;
                int h0, l0, h1, l1;
                if (fast_d) {  // (kernel-uniform; same limbs as fix2: vq_fixed.h)
                    fix2_mul(e, scale_d, h0, l0);
                    fix2_mul(e * e, scale_d2, h1, l1);
                } else {
                    fix2(e, sh_d, h0, l0);
                    fix2(e * e, sh_d2, h1, l1);
                }
                dacc0 += h0;
                dacc1 += l0;
                dacc2 += h1;
                dacc3 += l1;
            }
        }
        if (own && !skip && prev_sym) prev_sym[t] = (unsigned short)idx;
        // ---- records: '+' for the cell the frame is in now -- the side table's row when it landed in the odd child of its
        // family (seeded pass) --, '-' for the cell an incremental mover left ---------------------------------------------------
        const bool mov = own && !skip && (!incr || old != idx);
        if (mov) {
            const bool infam = incr == 2 && idx == old + 1;
            const int vN = infam ? rec.nbins_rows * rec.bin_cells + (old >> 1) : idx;
            const int binN = (int)(((unsigned)vN * rec.magic) >> 22);
            uint2* const region0 = rec.recs + (size_t)blockIdx.x * (size_t)rec.nbins * (size_t)rec.cap;
            const int pN = __hip_atomic_fetch_add(&rcnt[binN], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            region0[(size_t)binN * rec.cap + pN] = make_uint2((unsigned)t, (unsigned)(vN - binN * rec.bin_cells));
            if (incr != 0 && !infam) {
                const int binO = (int)(((unsigned)old * rec.magic) >> 22);
                const int pO = __hip_atomic_fetch_add(&rcnt[binO], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                region0[(size_t)binO * rec.cap + pO] = make_uint2((unsigned)t, (unsigned)(old - binO * rec.bin_cells) | 0x10000u);
            }
        }
    }
    // the wave's distortion sums -> the distortion columns of one row (any row: only their column totals are ever used)
    {
        i64 v[4] = {dacc0, dacc1, dacc2, dacc3};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            for (int dd = 32; dd >= 1; dd >>= 1) v[k] += __shfl_xor(v[k], dd, 64);
            if (lane == 0 && v[k] != 0) atomicAdd((u64*)&rows[(long)(wave % (32 * MT)) * RS + 2 * NC + 1 + k], (u64)v[k]);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < rec.nbins) rec.counts[(size_t)blockIdx.x * rec.nbins + threadIdx.x] = rcnt[threadIdx.x];
}

// ---- launch wrappers ----------------------------------------------------------------------------------------------------------
#ifdef E2VQ_SWEEP_STAMP
}  // namespace e2vq
extern "C" int e2vq_debug_sweep_exp(int mode)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(e2vq::g_sweep_exp), &mode, sizeof(int)) != hipSuccess;
}

extern "C" int e2vq_debug_sweep_stamps(unsigned long long* out16, int reset)
{
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(e2vq::g_sweep_stamps), 16 * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(e2vq::g_sweep_stamps), z, 16 * 8) != hipSuccess) return 1;
    }
    return 0;
}
namespace e2vq {
#endif
static bool sweep_has_nc(int NC)
{
    switch (NC) {
#define X(N) case N:
        E2VQ_PRE_NC_LIST(X)
#undef X
        return true;
        default: return false;
    }
}

bool sweep_supported(int NC, int M)
{
    if (!sweep_has_nc(NC) || !prefilter_supports(NC, M) || M > 8192) return false;
    switch (NC) {
#define X(N) case N: return PreLds<N>::OK;
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return false;
    }
}

// the fused sorted pass (launch_pass_sorted) on top of that: its rows fit the waves' LDS regions, its flagged-tile list the
// codebook's tiles -- asked BEFORE anything of the pass is enqueued (e2vq_pass falls back to the record kernels otherwise)
bool sweep_fused_supported(int NC, int M)
{
    if (!sweep_supported(NC, M) || M / 32 > 256) return false;
    switch (NC) {
#define X(N) case N: return SweepLds<N>::FUSE_OK;
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return false;
    }
}

size_t sweep_frame_image_bytes(int NC, long nblocks64)
{
    switch (NC) {
#define X(N) case N: return (size_t)nblocks64 * 64 * SweepImg<N>::FS;
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return 0;
    }
}

void launch_sweep_frames(const double* aos, long T, long nblocks64, int NC, const int* ea, void* img, hipStream_t s)
{
    const long nframes = nblocks64 * 64;
    const int grid = (int)((2 * nframes + 127) / 128 < 8192 ? (2 * nframes + 127) / 128 : 8192);
    switch (NC) {
#define X(N)                                                                                                           \
    case N:                                                                                                            \
        hipLaunchKernelGGL((k_frames_fm<N>), dim3(grid), dim3(128), 0, s, aos, T, nframes, ea, (unsigned char*)img);   \
        break;
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: break;
    }
}

__global__ void k_sweep_counters_out(SweepCounters* counters, unsigned long long* host)
{
    __hip_atomic_store(host, counters->flagged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(host + 1, counters->jobs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    counters->flagged = 0;
    counters->jobs = 0;
}
void launch_sweep_counters_out(void* counters, void* host_counters, hipStream_t s)
{
    hipLaunchKernelGGL(k_sweep_counters_out, dim3(1), dim3(1), 0, s, (SweepCounters*)counters, (unsigned long long*)host_counters);
}

size_t sort_scratch_bytes() { return (size_t)2 * SORT_MAX_BINS * sizeof(int) + 2 * sizeof(SweepCounters); }

// scratch: sort_scratch_bytes() bytes, zeroed once by the caller when it is allocated (the kernels leave the histogram zeroed)
int launch_sort_by_cell(const unsigned short* key, long T, long nblocks64, int nbins, void* scratch, unsigned* perm, hipStream_t s)
{
    if (nbins < 1 || nbins > SORT_MAX_BINS || T < 1) return 1;
    int* hist = (int*)scratch;
    int* cursor = hist + SORT_MAX_BINS;
    long nwg = (T + 8191) / 8192;
    nwg = nwg > 256 ? 256 : nwg;
    const long chunk = ((T + nwg - 1) / nwg + 63) / 64 * 64;
    const long used = (T + chunk - 1) / chunk;
    hipLaunchKernelGGL(k_sort_hist, dim3((unsigned)used), dim3(SORT_TPB), (size_t)nbins * sizeof(int), s, key, T, chunk, nbins, hist);
    hipLaunchKernelGGL(k_sort_base, dim3(1), dim3(SORT_TPB), 0, s, hist, nbins, cursor);
    hipLaunchKernelGGL(k_sort_scatter, dim3((unsigned)used), dim3(SORT_TPB), (size_t)2 * nbins * sizeof(int), s, key, T, chunk,
                       nbins, cursor, perm, nblocks64 * 64);
    return 0;
}

void* sweep_counters_of(void* sort_scratch) { return (char*)sort_scratch + (size_t)2 * SORT_MAX_BINS * sizeof(int); }
// a second pair of words: the sweeps whose flagged share the host does not fetch (the passes behind a level's first) add here,
// so that the executed k-steps of a timed region can be read back afterwards (e2vq_sweep_executed)
void* sweep_totals_of(void* sort_scratch) { return (char*)sweep_counters_of(sort_scratch) + sizeof(SweepCounters); }

int launch_sweep_candidates(int NC, bool two_stage, const void* fimg, const unsigned* perm, long T, long nblocks, const void* cimg,
                            const void* ps, int M, const unsigned short* prev_sym, int home_mul, unsigned* cand, void* counters,
                            hipStream_t s)
{
    if (!sweep_supported(NC, M)) return 1;
    int bits = 0;
    while ((1 << bits) < M) ++bits;
    const int idxmask = ~((1 << bits) - 1);
    const int MT = M / 32;
    if (MT > 256) return 1;
    long g = (nblocks + 7) / 8;
    const int grid = (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
    const SweepFuse none{};
    switch (NC) {
#define X(N)                                                                                                           \
    case N:                                                                                                            \
        if (two_stage)                                                                                                 \
            hipLaunchKernelGGL((k_sweep_cand<N, true, false>), dim3(grid), dim3(512), 8 * 512, s, (const unsigned char*)fimg, perm, T, \
                               nblocks, (const h8*)cimg, (PreScalars*)ps, MT, idxmask, prev_sym, home_mul, cand,       \
                               (SweepCounters*)counters, none);                                                        \
        else                                                                                                           \
            hipLaunchKernelGGL((k_sweep_cand<N, false, false>), dim3(grid), dim3(512), 8 * 512, s, (const unsigned char*)fimg, perm, T, \
                               nblocks, (const h8*)cimg, (PreScalars*)ps, MT, idxmask, prev_sym, home_mul, cand,       \
                               (SweepCounters*)counters, none);                                                        \
        return 0;
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return 1;
    }
}

// the fused pass over grouped frames (perm from launch_sort_by_cell): sweep, exact evaluation, outputs, cell sums.  `cells`
// holds every frame's cell of the previous pass (incr 1) or of the parents (incr 2) and receives the new ones.
template <int NC>
static int launch_pass_sorted_t(bool two_stage, bool one_block, const void* fimg, const unsigned* perm, long T, long nblocks, const void* cimg, void* ps,
                                int M, const SweepFuse& fz, void* counters, hipStream_t s)
{
    if constexpr (SweepLds<NC>::FUSE_OK) {
        int bits = 0;
        while ((1 << bits) < M) ++bits;
        const int idxmask = ~((1 << bits) - 1);
        const int MT = M / 32;
        constexpr int W = SweepLds<NC>::WAVES_FUSE;
        long g = (nblocks + W - 1) / W;
        const int grid = (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
        const size_t lds = (size_t)W * SweepLds<NC>::WAVE_BYTES_FUSE;
        auto go = [&](auto kernel) {
            (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, E2VQ_LDS_BYTES);
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(W * 64), lds, s, (const unsigned char*)fimg, perm, T, nblocks, (const h8*)cimg,
                               (PreScalars*)ps, MT, idxmask, (const unsigned short*)fz.cells, fz.incr, (unsigned*)nullptr,
                               (SweepCounters*)counters, fz);
        };
        if (two_stage && one_block && SweepImg<NC>::TWO_BLOCKS_FIT)
            go(k_sweep_cand<NC, true, true, true>);
        else if (two_stage)
            go(k_sweep_cand<NC, true, true>);
        else
            go(k_sweep_cand<NC, false, true>);
        return 0;
    }
    return 1;
}

int launch_pass_sorted(int NC, bool two_stage, bool one_block, const void* fimg, const unsigned* perm, long T, long nblocks, const void* cimg, void* ps,
                       const double* cbq, int M, const double* aos, const DevScalars* sc, const unsigned long long* l1max_bits,
                       unsigned short* sym, double* dmin, long long* rows, long long* fam, int* fb_list, unsigned short* cells, int incr,
                       void* counters, hipStream_t s)
{
    if (!sweep_supported(NC, M) || M / 32 > 256 || !perm || !cells || incr == 0 || (incr == 2 && !fam)) return 1;
    SweepFuse fz{};
    fz.aos = aos;
    fz.cbq = cbq;
    fz.sc = sc;
    fz.l1max_bits = (const u64*)l1max_bits;
    fz.sym = sym;
    fz.dmin = dmin;
    fz.rows = (i64*)rows;
    fz.fam = (i64*)fam;
    fz.fb_list = fb_list;
    fz.cells = cells;
    fz.incr = incr;
    fz.M = M;
    switch (NC) {
#define X(N) \
    case N: return launch_pass_sorted_t<N>(two_stage, one_block, fimg, perm, T, nblocks, cimg, ps, M, fz, counters, s);
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return 1;
    }
}

template <int NC>
static int launch_finish_t(const double* aos, long T, long nblocks, const unsigned* cand, void* ps, const double* cbq, int M,
                           const DevScalars* sc, const unsigned long long* l1max_bits, unsigned short* sym, double* dmin,
                           long long* rows, int* fb_list, unsigned short* prev_sym, int incr, const PassRecords* records,
                           void* counters, void* host_counters, hipStream_t s)
{
    if constexpr (PreLds<NC>::OK) {
        constexpr int WAVES = PreLds<NC>::WAVES;
        long g = (nblocks + WAVES - 1) / WAVES;
        const int grid = (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
        if (!records || records->grid != grid || records->nbins > 64) return 1;
        PreRec rec{};
        rec.recs = (uint2*)records->recs;
        rec.counts = records->counts;
        rec.nbins = records->nbins;
        rec.nbins_rows = records->nbins_rows;
        rec.bin_cells = records->bin_cells;
        rec.cap = records->cap;
        rec.magic = records->magic;
        (void)hipFuncSetAttribute((const void*)k_finish<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, E2VQ_LDS_BYTES);
        hipLaunchKernelGGL((k_finish<NC>), dim3(grid), dim3(FinLds<NC>::WAVES * 64), (size_t)FinLds<NC>::WAVES * FinLds<NC>::WAVE_BYTES + 256, s, aos, T,
                           nblocks, cand, (PreScalars*)ps, cbq, M / 32, sc, (const u64*)l1max_bits, sym, dmin, (i64*)rows, fb_list,
                           prev_sym, incr, rec, (SweepCounters*)counters, (unsigned long long*)host_counters);
        return 0;
    }
    return 1;
}

int launch_finish(int NC, const double* aos, long T, long nblocks, const unsigned* cand, void* ps, const double* cbq, int M,
                  const DevScalars* sc, const unsigned long long* l1max_bits, unsigned short* sym, double* dmin, long long* rows,
                  int* fb_list, unsigned short* prev_sym, int incr, const PassRecords* records, void* counters,
                  void* host_counters, hipStream_t s)
{
    switch (NC) {
#define X(N)                                                                                                           \
    case N:                                                                                                            \
        return launch_finish_t<N>(aos, T, nblocks, cand, ps, cbq, M, sc, l1max_bits, sym, dmin, rows, fb_list, prev_sym, incr, \
                                  records, counters, host_counters, s);
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return 1;
    }
}

}  // namespace e2vq
