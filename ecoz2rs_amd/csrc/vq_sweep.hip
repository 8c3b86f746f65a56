// vq_sweep.hip -- round 5: the accumulating prefiltered pass as a chain of kernels, each bound by one thing
//
//   k_sort_*        once per level: the frames' numbers grouped by the cell they had when the level began (a counting sort of
//                   2-byte keys: `perm`, 4 bytes per frame).  Nothing but this list is moved: every kernel below addresses
//                   frames by their number.
//   k_sweep_cand    (the matrix pipe) one wave = 64 *slots* of the sorted list; the B operands -- the f16 limb images of its
//                   frames -- are gathered through `perm` from a frame-major image (256 bytes per frame at P = 36, two cache
//                   lines).  Per frame it emits the two codewords that can be the nearest one, and whether that is certain:
//                   4 bytes.  No FP64 frame, no LDS row, no output besides those 4 bytes.
//   k_finish        (HBM) every frame once, in its natural order: the canonical FP64 chain for the one or two candidates
//                   (lane = frame, rows staged in LDS by LDS-DMA), symbol / distortion out, distortion sums, and the frame's
//                   contribution to the cell sums as 8-byte records -- what the tail of k_pass_pre_lds did inside the sweep.
//   k_reduce_records (vq_prefilter.hip) folds the records into the rows, as in round 4.
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
// 93-99 % when they are not (profiles/r05_skip_feasibility.txt): the sweep issues 0.58 of the limb products and -- more
// to the point, since the key epilogue's VALU operations bound round 4's tile loop -- 1.5 instead of 6 VALU operations
// per value in stage 1.  Nothing is decided by a key: a frame whose top two cannot be certified goes to the FP64
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

struct SweepCounters {
    unsigned long long flagged;  // (tile, column block) jobs that ran stage 2
    unsigned long long jobs;     // (tile, column block) jobs in all
};

template <int NC, bool TWO>
__global__ __launch_bounds__(512, 2) void k_sweep_cand(const unsigned char* __restrict__ fimg, const unsigned* __restrict__ perm,
                                                       long T, long nblocks, const h8* __restrict__ cimg,
                                                       const PreScalars* __restrict__ ps, int MT, int idxmask,
                                                       const unsigned short* __restrict__ prev_sym, int home_mul,
                                                       unsigned* __restrict__ cand, SweepCounters* __restrict__ counters)
{
    typedef PrePack<NC> PK;
    constexpr int NU = PK::NU, FS = SweepImg<NC>::FS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long wave = (long)blockIdx.x * 8 + wib;
    const long nwaves = (long)gridDim.x * 8;
    unsigned* tlist = (unsigned*)smem + wib * 256;  // flagged tiles of the block: tile | column-block bits << 16 (MT <= 256)
    const float ymax1 = __int_as_float(ps->ymax_bits);
    const float relk = __int_as_float((127 + __builtin_popcount(~idxmask) - 21) << 23);  // 2 rho, rho = 2^-(22-idxbits)
    int maskv = idxmask;
    asm volatile("" : "+v"(maskv));
    float ninf = -__builtin_inff();
    asm volatile("" : "+v"(ninf));
    const float pinf = __builtin_inff();
    const int col = lane & 31, h = lane >> 5;
    unsigned long long nflag = 0, njobs = 0;

    for (long b = wave; b < nblocks; b += nwaves) {
        // the block's slots -> frames; lane (h, col) holds lane half h of the granules of slots col and 32 + col
        long s0 = b * 64 + col, s1 = s0 + 32;
        s0 = s0 < T ? s0 : T - 1;
        s1 = s1 < T ? s1 : T - 1;
        const unsigned f0 = perm ? perm[s0] : (unsigned)s0, f1 = perm ? perm[s1] : (unsigned)s1;
        const unsigned char* p0 = fimg + (size_t)f0 * FS;
        const unsigned char* p1 = fimg + (size_t)f1 * FS;
        h8 B[2][PK::PAIRS];
#pragma unroll
        for (int p = 0; p < PK::PAIRS; ++p) {
            B[0][p] = *(const h8*)(p0 + p * 32 + h * 16);
            B[1][p] = *(const h8*)(p1 + p * 32 + h * 16);
        }
        const float g0 = *(const float*)(p0 + PK::PAIRS * 32), g1 = *(const float*)(p1 + PK::PAIRS * 32);
        float k1[2], k2[2], k3[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) k1[cb] = k2[cb] = k3[cb] = __int_as_float(0x7f7fffff);

        if constexpr (TWO) {
            // ---- stage 1: every tile, weight levels 0 and 1; tiles in cyclic order from the block's home tile ---------------
            int home = 0;
            if (home_mul) {
                const unsigned fh = (unsigned)__builtin_amdgcn_readfirstlane((int)f0);
                home = (home_mul * (int)prev_sym[fh]) >> 5;
                home = __builtin_amdgcn_readfirstlane(home < MT ? home : MT - 1);
            }
            // 2 x 1.27 x E2 of the lane's two frames (coarse-key units; header)
            const float D0 = 2.54f * (257.f * (g0 + ymax1) + (129.f * NC + 2.f));
            const float D1 = 2.54f * (257.f * (g1 + ymax1) + (129.f * NC + 2.f));
            float U0 = pinf, U1 = pinf, thr0 = pinf, thr1 = pinf;
            int ntl = 0;
            h8 Acur[NU], Anext[NU];
            sweep_load_tile<NC>(Acur, cimg, home, lane, true);
#pragma unroll 2
            for (int i = 0; i < MT; ++i) {
                const int tile = home + i < MT ? home + i : home + i - MT;
                int tn = tile + 1 < MT ? tile + 1 : 0;
                tn = i + 1 < MT ? tn : tile;  // (the last iteration reloads its own tile: no load is conditional)
                sweep_load_tile<NC>(Anext, cimg, tn, lane, true);
                const float m0 = sweep_coarse_job<NC>(Acur, B[0]);
                const float m1 = sweep_coarse_job<NC>(Acur, B[1]);
                // (negated comparisons: a NaN key flags its tile)
                const bool fl0 = !(m0 > thr0), fl1 = !(m1 > thr1);
                U0 = __builtin_fminf(U0, m0);
                U1 = __builtin_fminf(U1, m1);
                thr0 = U0 > 0.f ? __builtin_fmaf(U0, 1.000001f, D0) : pinf;
                thr1 = U1 > 0.f ? __builtin_fmaf(U1, 1.000001f, D1) : pinf;
                const unsigned bits = (__ballot(fl0) != 0 ? 1u : 0u) | (__ballot(fl1) != 0 ? 2u : 0u);
                if (bits) {  // (wave-uniform; every lane stores the same word)
                    tlist[ntl] = (unsigned)tile | bits << 16;
                    ++ntl;
                }
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (SweepImg<NC>::coarse_unique(u)) Acur[u] = Anext[u];
            }
            njobs += 2ull * MT;
            // ---- stage 2: the flagged tiles with all their k-steps and the key epilogue ---------------------------------------
            if (ntl > 0) {
                unsigned e = tlist[0];
                sweep_load_tile<NC>(Acur, cimg, (int)(e & 0xffffu), lane, false);
                for (int j = 0; j < ntl; ++j) {
                    const unsigned en = tlist[j + 1 < ntl ? j + 1 : j];
                    sweep_load_tile<NC>(Anext, cimg, (int)(en & 0xffffu), lane, false);
                    const int tile = (int)(e & 0xffffu);
                    if (e & 0x10000u) sweep_full_job<NC>(Acur, B[0], tile, k1[0], k2[0], k3[0], maskv, ninf);
                    if (e & 0x20000u) sweep_full_job<NC>(Acur, B[1], tile, k1[1], k2[1], k3[1], maskv, ninf);
                    nflag += ((e >> 16) & 1u) + ((e >> 17) & 1u);
#pragma unroll
                    for (int u = 0; u < NU; ++u) Acur[u] = Anext[u];
                    e = en;
                }
            }
        } else {
            // ---- one stage: every tile with all its k-steps (frames that are not grouped, or data that flags most tiles) ------
            h8 Acur[NU], Anext[NU];
            sweep_load_tile<NC>(Acur, cimg, 0, lane, false);
#pragma unroll 2
            for (int t = 0; t < MT; ++t) {
                sweep_load_tile<NC>(Anext, cimg, t + 1 < MT ? t + 1 : t, lane, false);
                sweep_full_job<NC>(Acur, B[0], t, k1[0], k2[0], k3[0], maskv, ninf);
                sweep_full_job<NC>(Acur, B[1], t, k1[1], k2[1], k3[1], maskv, ninf);
#pragma unroll
                for (int u = 0; u < NU; ++u) Acur[u] = Anext[u];
            }
            njobs += 2ull * MT;
            nflag += 2ull * MT;
        }

        // ---- lane = slot b * 64 + lane: merge the two lane halves of its frame's keys, certify the top two ------------------
        const int hb = h << 2;
        float a1, a2, a3, q1, q2, q3;
        {
            const float o1 = __int_as_float(__float_as_int(k1[0]) | hb), o2 = __int_as_float(__float_as_int(k2[0]) | hb),
                        o3 = __int_as_float(__float_as_int(k3[0]) | hb);
            const float r1 = __int_as_float(__float_as_int(k1[1]) | hb), r2 = __int_as_float(__float_as_int(k2[1]) | hb),
                        r3 = __int_as_float(__float_as_int(k3[1]) | hb);
            a1 = h ? r1 : o1, a2 = h ? r2 : o2, a3 = h ? r3 : o3;  // own frame's keys (slot 32 h + col)
            q1 = h ? o1 : r1, q2 = h ? o2 : r2, q3 = h ? o3 : r3;  // the partner's frame's keys
        }
        const float b1 = __shfl_xor(q1, 32, 64), b2 = __shfl_xor(q2, 32, 64), b3 = __shfl_xor(q3, 32, 64);
        const float t3 = med3f(a2, a3, b1), t2 = med3f(a1, a2, b1), t1 = med3f(a1, b1, ninf);
        const float u3 = med3f(t2, t3, b2), u2 = med3f(t1, t2, b2);
        const float w3 = med3f(u2, u3, b3);
        const float g = h ? g1 : g0;
        const unsigned f = h ? f1 : f0;
        const float tau = 1.27f * (512.f * (g + ymax1 + (NC + 4.0f)) + relk * t1);
        const bool cert = t1 >= 1.0e-30f && t1 < 1.0e37f && w3 > t1 + tau;
        const bool amb = !(u2 > t1 + tau);
        const unsigned c1 = (unsigned)(__float_as_int(t1) & ~idxmask), c2 = (unsigned)(__float_as_int(u2) & ~idxmask);
        if (b * 64 + lane < T) cand[f] = c1 | c2 << 13 | (amb ? CAND_AMB : 0u) | (cert ? CAND_CERT : 0u);
    }
    if (counters && lane == 0 && njobs) {
        atomicAdd(&counters->flagged, nflag);
        atomicAdd(&counters->jobs, njobs);
    }
}

// ---- k_finish ---------------------------------------------------------------------------------------------------------------
// One wave = 64 consecutive frames (their natural order), lane = frame: the block's FP64 rows arrive in the wave's LDS region
// by LDS-DMA (as in k_pass_pre_lds), the candidates' codeword rows are gathered from L2, the canonical chain
// acc = fma(r[n], cq[n], acc), n ascending from +0.0, decides; outputs, distortion sums and the records of the frame's
// contribution follow -- the statements of k_pass_pre_lds<.., 2> behind its certification, unchanged in what they compute.
template <int NC>
__global__ __launch_bounds__(512, 2) void k_finish(const double* __restrict__ aos, long T, long nblocks,
                                                   const unsigned* __restrict__ cand, PreScalars* __restrict__ ps,
                                                   const double* __restrict__ cbq, int MT, const DevScalars* __restrict__ sc,
                                                   const u64* __restrict__ l1max_bits, unsigned short* __restrict__ sym,
                                                   double* __restrict__ dmin, i64* __restrict__ rows, int* __restrict__ fb_list,
                                                   unsigned short* __restrict__ prev_sym, int incr, PreRec rec,
                                                   SweepCounters* __restrict__ counters, unsigned long long* host_counters)
{
    typedef PreLds<NC> PL;
    // the two-stage sweep in front of this kernel left its counters: to the host (it adapts), and zero for the next sweep
    if (counters && host_counters && blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(host_counters, counters->flagged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_counters + 1, counters->jobs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        counters->flagged = 0;
        counters->jobs = 0;
    }
    constexpr int TPBM = PL::WAVES * 64;
    constexpr int RS = (2 * NC + 5 + 7) & ~7, NPAD = (NC + 7) & ~7;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ln = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long wave = (long)blockIdx.x * (TPBM >> 6) + wib;
    const long nwaves = (long)gridDim.x * (TPBM >> 6);
    unsigned char* wbase = smem + (size_t)wib * PL::WAVE_BYTES;
    const double* stage = (const double*)wbase;
    const unsigned short* prevs = (const unsigned short*)(wbase + PL::STAGE_BYTES + 256);
    int* const rcnt = (int*)(smem + (size_t)PL::WAVES * PL::WAVE_BYTES);
    if (threadIdx.x < 64) rcnt[threadIdx.x] = 0;
    __syncthreads();
    const int Ed = dist_exponent(sc->maxabs, __longlong_as_double((i64)*l1max_bits));
    const int sh_d = 30 - Ed, sh_d2 = 30 - 2 * Ed;
    auto pow2 = [](int e) { return __longlong_as_double((long long)(1023 + e) << 52); };  // |e| <= 1000
    const bool fast_d = sh_d >= -1000 && sh_d <= 1000 && sh_d2 >= -1000 && sh_d2 <= 1000;
    const double scale_d = pow2(fast_d ? sh_d : 0), scale_d2 = pow2(fast_d ? sh_d2 : 0);
    unsigned cd_next = 0u;
    if (wave < nblocks) {
        pre_lds_request<NC, false>(aos, nullptr, incr ? prev_sym : nullptr, wave, ln, wbase);
        cd_next = wave * 64 + ln < T ? cand[wave * 64 + ln] : 0u;
    }
    for (long b = wave; b < nblocks; b += nwaves) {
        const long t = b * 64 + ln;
        const bool live = t < T;
        const unsigned cd = cd_next;
        const bool cert = (cd & CAND_CERT) != 0, amb = (cd & CAND_AMB) != 0;
        const int c1 = (int)(cd & 0x1fffu), c2 = (int)((cd >> 13) & 0x1fffu);
        double best;
        int idx;
        {
            constexpr int NH = (NC + 1) / 2;
            const double2* r1 = (const double2*)(cbq + (long)c1 * NPAD);
            const double2* r2 = (const double2*)(cbq + (long)c2 * NPAD);
            double2 x[NH], y[NH];
#pragma unroll
            for (int n2 = 0; n2 < NH; ++n2) x[n2] = r1[n2];  // (rows are padded to a multiple of 8 doubles)
#pragma unroll
            for (int n2 = 0; n2 < NH; ++n2) y[n2] = make_double2(0.0, 0.0);
            if (amb) {
#pragma unroll
                for (int n2 = 0; n2 < NH; ++n2) y[n2] = r2[n2];
            }
            // the block's rows (and its cells of the previous pass) have landed: everything older than the codeword gathers
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            asm volatile("" ::: "memory");
            const double* fr = stage + ln * NC;
            double d1 = 0.0, d2 = 0.0;
#pragma unroll
            for (int n2 = 0; n2 < NH; ++n2) {
                if ((n2 & 3) == 0) asm volatile("" ::: "memory");
                const double f0 = fr[2 * n2];
                d1 = __builtin_fma(f0, x[n2].x, d1);
                d2 = __builtin_fma(f0, y[n2].x, d2);
                if (2 * n2 + 1 < NC) {
                    const double f1 = fr[2 * n2 + 1];
                    d1 = __builtin_fma(f1, x[n2].y, d1);
                    d2 = __builtin_fma(f1, y[n2].y, d2);
                }
            }
            const bool take_b = amb && (d2 < d1 || (d2 == d1 && c2 < c1));
            best = take_b ? d2 : d1;
            idx = take_b ? c2 : c1;
        }
        const int old = incr ? (incr == 2 ? 2 : 1) * (int)prevs[ln] : 0;
        const bool skip = !cert;
        idx = skip ? 0 : idx;
        // the next block's rows: the LDS region is free once this block's reads are done
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (b + nwaves < nblocks) {
            pre_lds_request<NC, false>(aos, nullptr, incr ? prev_sym : nullptr, b + nwaves, ln, wbase);
            cd_next = (b + nwaves) * 64 + ln < T ? cand[(b + nwaves) * 64 + ln] : 0u;
        }
        // ---- outputs; uncertified frames go to the fallback list -----------------------------------------------------------
        if (live) {
            if (skip) {
                fb_list[atomicAdd(&ps->fb_count, 1)] = (int)t;
            } else {
                if (sym) sym[t] = (unsigned short)idx;
                if (dmin) dmin[t] = best;
            }
        }
        i64 dsum;
        {
            int h0 = 0, l0 = 0, h1 = 0, l1 = 0;
            if (live && !skip) {
                const double e = best - 1.0;
                if (fast_d) {  // (kernel-uniform; same limbs as fix2: vq_fixed.h)
                    fix2_mul(e, scale_d, h0, l0);
                    fix2_mul(e * e, scale_d2, h1, l1);
                } else {
                    fix2(e, sh_d, h0, l0);
                    fix2(e * e, sh_d2, h1, l1);
                }
            }
            i64 d0 = h0, d1 = l0, d2 = h1, d3 = l1;
            for (int d = 32; d >= 1; d >>= 1) {
                d0 += __shfl_xor(d0, d, 64);
                d1 += __shfl_xor(d1, d, 64);
                d2 += __shfl_xor(d2, d, 64);
                d3 += __shfl_xor(d3, d, 64);
            }
            dsum = ln == 0 ? d0 : ln == 1 ? d1 : ln == 2 ? d2 : d3;
        }
        // ---- records: (frame, cell within its bin, sign) into this workgroup's region of the bin (k_pass_pre_lds, ACC = 2) ---
        const bool mov = live && !skip && (!incr || old != idx);
        {
            const bool infam = incr == 2 && idx == old + 1;
            const int vN = infam ? rec.nbins_rows * rec.bin_cells + (old >> 1) : idx;
            const bool hasN = mov, hasO = mov && incr != 0 && !infam;
            const int binN = (int)(((unsigned)vN * rec.magic) >> 22), binO = (int)(((unsigned)old * rec.magic) >> 22);
            int rankN = 0, rankO = 0, cntv = 0;
            u64 pn = __ballot(hasN), po = __ballot(hasO);
            while ((pn | po) != 0) {
                const int r = pn != 0 ? __builtin_amdgcn_readlane(binN, (int)__builtin_ctzll(pn))
                                      : __builtin_amdgcn_readlane(binO, (int)__builtin_ctzll(po));
                const bool inN = hasN && binN == r, inO = hasO && binO == r;
                const u64 sn = __ballot(inN), so = __ballot(inO);
                const int cn = __builtin_popcountll(sn);
                if (inN) rankN = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(sn >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)sn, 0u));
                if (inO) rankO = cn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(so >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)so, 0u));
                if (ln == r) cntv = cn + __builtin_popcountll(so);
                pn &= ~sn;
                po &= ~so;
            }
            int basev = 0;
            if (cntv > 0) basev = __hip_atomic_fetch_add(&rcnt[ln], cntv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int baseN = __shfl(basev, binN, 64), baseO = __shfl(basev, binO, 64);
            uint2* const region0 = rec.recs + (size_t)blockIdx.x * (size_t)rec.nbins * (size_t)rec.cap;
            if (hasN)
                region0[(size_t)binN * rec.cap + (baseN + rankN)] = make_uint2((unsigned)t, (unsigned)(vN - binN * rec.bin_cells));
            if (hasO)
                region0[(size_t)binO * rec.cap + (baseO + rankO)] =
                    make_uint2((unsigned)t, (unsigned)(old - binO * rec.bin_cells) | 0x10000u);
        }
        if (ln < 4 && dsum != 0) atomicAdd((u64*)&rows[(long)(b % (32 * MT)) * RS + 2 * NC + 1 + ln], (u64)dsum);
        if (prev_sym && live && !skip) prev_sym[t] = (unsigned short)idx;
    }
    __syncthreads();
    if ((int)threadIdx.x < rec.nbins) rec.counts[(size_t)blockIdx.x * rec.nbins + threadIdx.x] = rcnt[threadIdx.x];
}

// ---- launch wrappers ----------------------------------------------------------------------------------------------------------
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

size_t sort_scratch_bytes() { return (size_t)2 * SORT_MAX_BINS * sizeof(int) + sizeof(SweepCounters); }

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
    switch (NC) {
#define X(N)                                                                                                           \
    case N:                                                                                                            \
        if (two_stage)                                                                                                 \
            hipLaunchKernelGGL((k_sweep_cand<N, true>), dim3(grid), dim3(512), 8 * 256 * 4, s, (const unsigned char*)fimg, perm, T, \
                               nblocks, (const h8*)cimg, (const PreScalars*)ps, MT, idxmask, prev_sym, home_mul, cand,  \
                               (SweepCounters*)counters);                                                              \
        else                                                                                                           \
            hipLaunchKernelGGL((k_sweep_cand<N, false>), dim3(grid), dim3(512), 8 * 256 * 4, s, (const unsigned char*)fimg, perm, T, \
                               nblocks, (const h8*)cimg, (const PreScalars*)ps, MT, idxmask, prev_sym, home_mul, cand,  \
                               (SweepCounters*)counters);                                                              \
        return 0;
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
        hipLaunchKernelGGL((k_finish<NC>), dim3(grid), dim3(WAVES * 64), (size_t)WAVES * PreLds<NC>::WAVE_BYTES + 256, s, aos, T,
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
