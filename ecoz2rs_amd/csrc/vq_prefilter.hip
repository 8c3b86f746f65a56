// vq_prefilter.hip -- the prefiltered sweep: same results as the FP64 sweep (bit for bit), a fraction of its time.
//
// Idea.  argmin_m d(r, c_m) needs the FP64 chain only for the codewords that can win.  A cheap approximation
// d~ with a PROVEN error bound eps finds them: every codeword whose d~ exceeds the smallest d~ by more than
// 2 eps is out.  The approximation runs on the f16 matrix pipe and is made of exact integer arithmetic, so the
// bound needs no assumption about the hardware's summation order:
//
//   * per coefficient n a power of two a_n (vq learn: > max_t |r_t[n]|, one scan per upload; vq quantize: from the
//     codebook, so that no pass over the data is needed), per frame a power of two A_t,
//     per codebook a power of two C:   xi = r / (a A_t),  eta = c a / C,  both in (-1, 1),  d = A_t C sum xi eta
//   * xi -> three integer limbs  X1 = rint(xi 2^9) (|X1| <= 512), X2, X3 (|.| <= 256):
//         xi = X1 2^-9 + X2 2^-18 + X3 2^-27 + rho,  |rho| <= 2^-28;   eta likewise (Y1, Y2, Y3, sigma)
//   * the limb products of equal weight are summed by v_mfma_f32_32x32x16_f16 into separate f32 accumulators
//         W0 = sum X1 Y1,   W1 = sum X1 Y2 + X2 Y1,   W2 = sum X1 Y3 + X2 Y2 + X3 Y1
//     every partial sum is an integer below 2^24 (NC 2^18, 2 NC 2^17, 5 NC 2^16 with NC <= 41), i.e. exactly
//     representable: the MFMA results are exact in any summation order (checked on hardware: tools/probe/pre_sweep.hip)
//   * v = W0 2^18 + W1 2^9 + W2 (two f32 fmas), key = v with its low mantissa bits replaced by the codeword index;
//     a running (min, 2nd, 3rd) of the keys per frame costs three VALU ops per value
//   * |2^36 sum xi eta - key| <= 2^8 (sum|xi| + max_m sum|eta_m| + NC + 4) + |key| 2^-(22 - idxbits)     
//     If the third key is farther from the first than twice that (x1.27), the true argmin is one of the first two:
//     both are evaluated with the canonical FP64 chain (on the FP64 matrix pipe, as the diagonal of a 16x16 tile of
//     gathered codewords -- the same instruction sequence as the full sweep, so bit-identical values) and compared
//     exactly (ties: lower index).  Otherwise -- or if the smallest key is not a positive normal number -- the frame
//     goes to a list that k_pass_mfma<SRC = 2> sweeps in full FP64 right after.  Nothing is ever decided by d~.
//
// Why three limbs: d = sum r[n] cq[n] is the small difference of large terms (sum xi eta ~ 3e-4 against sum |xi eta|
// ~ 1 on LPC data), and the two best codewords of a frame are typically 5 % apart: the approximation must resolve
// ~1e-6 of the term scale.  Two limbs (2^-19, 8 MFMAs per tile) were tried in round 2: bit-identical results, but
// 99.7 % of the frames failed the certification and took the FP64 fallback.
//
// K-slot packing for any NC = P + 1 <= 41 (PrePack): with G = NC / 16 and R = NC % 16, each limb fills G "pairs" of 16
// coefficients (two 8-half granules: lane halves h = 0, 1 of the 32x32x16 operand); the three R-coefficient tails
// are packed back to back into ceil(3R / 16) further pairs.  W0 takes the pairs that hold X1 slots, W1 those with X1 or
// X2 slots, W2 all of them; the codeword image puts the matching limb -- or zero -- in each slot.
// NC = 37: 7 pairs = 224 B per frame, 3 + 5 + 7 = 15 MFMAs per 32x32 tile (222 of 240 slots carry a product).
#include "vq_pre_common.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

namespace e2vq {

// ---- the pass ---------------------------------------------------------------------------------------------

// canonical FP64 chain of frame j of tile ft against codeword `cand` (per lane: the candidate of frame lane&15),
// evaluated as the diagonal of one 16x16 MFMA tile: the instruction sequence of k_pass_mfma, hence its values.
// The codeword row comes from the row-major cbq: the four q lanes of a frame read adjacent 8 bytes, so a gather
// instruction touches 16 rows; the rows are L2-resident (303 KB at M = 1024).
template <int NC>
struct PreCand {
    double a[(NC + 3) / 4];
    double tail;
};
template <int NC>
__device__ __forceinline__ PreCand<NC> pre_gather(int cand, const double* __restrict__ cbq, int q)
{
    constexpr int NS = (NC + 3) / 4, REM = NC - 4 * (NS - 1), NPAD = (NC + 7) & ~7;
    constexpr int NSM = REM == 1 ? NS - 1 : NS;
    PreCand<NC> c;
    const double* row = cbq + (long)cand * NPAD;
#pragma unroll
    for (int st = 0; st < NSM; ++st) c.a[st] = row[4 * st + q];
    c.tail = REM == 1 ? row[NC - 1] : 0.0;
    return c;
}
template <int NC>
__device__ __forceinline__ double pre_exact(const double (&Bft)[2 * ((((NC + 3) / 4) + 1) / 2)], const PreCand<NC>& c, int j)
{
    constexpr int NS = (NC + 3) / 4, REM = NC - 4 * (NS - 1);
    constexpr bool TAILV = REM == 1;
    constexpr int NSM = TAILV ? NS - 1 : NS;
    d4 acc = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a[0], Bft[0], (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
#pragma unroll
    for (int st = 1; st < NSM; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(c.a[st], Bft[st], acc, 0, 0, 0);
    const int rg = j >> 2;
    double v = rg == 0 ? acc[0] : rg == 1 ? acc[1] : rg == 2 ? acc[2] : acc[3];  // D[row 4 rg + q][col j], row = j
    if (TAILV) v = __builtin_fma(Bft[NS - 1], c.tail, v);
    return __shfl(v, 16 * (j & 3) + j, 64);  // the lane with q = j & 3 holds the diagonal element of frame j
}

// ---- quantize, fused (MODE 6 of k_pass_pre): the wave builds the limb images of its 64 frames itself ---------------
// The 64 row-major frames of block b go to the wave's LDS stage (coalesced 8-byte loads; frames >= T read as zero) and
// stay there for the exact evaluation after the sweep, so the sweep kernel reads every frame exactly once and no limb
// image ever travels through HBM (the separate preparation pass wrote and the sweep re-read 228 B per frame).
// Lane (col = lane & 31, h = lane >> 5) owns the B granules of frames col (column block 0) and 32 + col (column block 1)
// for lane half h: of the full pairs those are the coefficients n = 16 g + 8 h + e, e = 0..7, with all three limbs --
// the two halves split the coefficients between them; the R tail coefficients are split by both.  Same scales and the
// same pre_split as k_pre_quant_prep: identical images.  gq[cb] = sum_n |xi_n| of the frame, rounded up.
template <int NC>
__device__ __forceinline__ void pre_build_block(const double* __restrict__ aos, long b, long T, int lane,
                                                double* __restrict__ stage, const int* __restrict__ eas,
                                                h8 (&B)[2][PrePack<NC>::PAIRS], float (&gq)[2])
{
    typedef PrePack<NC> PK;
    constexpr int G = PK::G, R = PK::R, NL = PK::NL;
    const long base = b * 64 * NC, total = T * NC;
    if ((b + 1) * 64 <= T) {  // (wave-uniform) a full block: LDS-DMA, 1 KB per instruction, no registers involved
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        constexpr int BYTES = 64 * NC * 8, K16 = BYTES / 1024, K4 = (BYTES - K16 * 1024) / 256;
        static_assert(K16 * 1024 + K4 * 256 == BYTES, "a block of frames is a whole number of 256-byte pieces");
        const char* g16 = (const char*)(aos + base) + lane * 16;
        const char* g4 = (const char*)(aos + base) + K16 * 1024 + lane * 4;
        char* l = (char*)stage;
#pragma unroll
        for (int k = 0; k < K16; ++k)
            __builtin_amdgcn_global_load_lds((gptr_t)(g16 + k * 1024), (lptr_t)(l + k * 1024), 16, 0, 0);
#pragma unroll
        for (int k = 0; k < K4; ++k)
            __builtin_amdgcn_global_load_lds((gptr_t)(g4 + k * 256), (lptr_t)(l + K16 * 1024 + k * 256), 4, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {  // the last, partial block: element by element, frames >= T as zeros
        for (int i = lane; i < 64 * NC; i += 64) stage[i] = base + i < total ? aos[base + i] : 0.0;
    }
    const int col = lane & 31, h = lane >> 5;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        __builtin_amdgcn_sched_barrier(0);  // one frame after the other: interleaved, the two spill ~80 registers
        const double* row = stage + (32 * cb + col) * NC;
        constexpr int NV = 8 * G + R;  // the lane's values of this frame: its half of the full pairs, then the tail
        double x[NV];
        int sh[NV];                    // -ea[n], then -ea[n] - eA
        int eA = -100000;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int n = i < 8 * G ? 16 * (i >> 3) + 8 * h + (i & 7) : 16 * G + (i - 8 * G);
            x[i] = row[n];
            sh[i] = -eas[n];
            // ilogb(x) + 1 - ea[n] for x != 0 (v_frexp_exp_i32_f64: denormals included)
            const int ex = x[i] != 0.0 ? __builtin_amdgcn_frexp_exp(x[i]) + sh[i] : -100000;
            eA = ex > eA ? ex : eA;
        }
        {
            const int o = __shfl_xor(eA, 32, 64);
            eA = o > eA ? o : eA;
        }
        if (eA == -100000) eA = 0;  // an all-zero frame
        double gsum = 0.0;
        _Float16 lim[NL][NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const double xi = ldexp(x[i], sh[i] - eA);
            if (i < 8 * G || h == 0) gsum += fabs(xi);  // (both halves hold the tail: counted once)
            const double s1 = xi * 512.0, l1 = __builtin_rint(s1);  // (pre_split, kept in FP64 up to the f16 conversion)
            const double s2 = (s1 - l1) * 512.0, l2 = __builtin_rint(s2);
            const double s3 = (s2 - l2) * 512.0, l3 = __builtin_rint(s3);
            lim[0][i] = (_Float16)(float)l1;
            lim[1][i] = (_Float16)(float)l2;
            lim[2][i] = (_Float16)(float)l3;
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int fl = 0; fl < NL; ++fl)
#pragma unroll
                for (int e = 0; e < 8; ++e) B[cb][fl * G + g][e] = lim[fl][8 * g + e];
        if constexpr (R > 0) {
#pragma unroll
            for (int tp = 0; tp < PK::TAILP; ++tp) {
                h8 o0 = {0, 0, 0, 0, 0, 0, 0, 0}, o1 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    int fl, n;
                    PK::slot(NL * G + tp, 0, e, fl, n);
                    if (n >= 0) o0[e] = lim[fl][8 * G + n - 16 * G];
                    PK::slot(NL * G + tp, 1, e, fl, n);
                    if (n >= 0) o1[e] = lim[fl][8 * G + n - 16 * G];
                }
                B[cb][NL * G + tp] = h == 0 ? o0 : o1;
            }
        }
        gsum += __shfl_xor(gsum, 32, 64);
        gq[cb] = (float)gsum * 1.000001f;
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ---- diagnostics (-DE2VQ_PRE_STAMP, tools/probe/pre_stamps.py): where a wave's cycles go, phase by phase ----------
// s_memtime deltas summed per wave in scalar registers, added to a global table at the end; the stamped build also
// drains the vector-memory counter at the phase ends, so that a phase pays for the loads it waits on.  Never defined
// in the product build.
#ifdef E2VQ_PRE_STAMP
__device__ unsigned long long g_pre_stamps[32];
#define E2VQ_STAMP_DECL unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = 0, st_n = 0;
#define E2VQ_STAMP_START st_t = __builtin_amdgcn_s_memtime();
#define E2VQ_STAMP(i)                                                    \
    {                                                                    \
        const unsigned long long st_now = __builtin_amdgcn_s_memtime();  \
        st_acc[i] += st_now - st_t;                                      \
        st_t = st_now;                                                   \
    }
#define E2VQ_STAMP_DRAIN(i)                                              \
    {                                                                    \
        if (E2VQ_PRE_STAMP == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* (2: stamps without drains) */ \
        E2VQ_STAMP(i)                                                    \
    }
#else
#define E2VQ_STAMP_DECL
#define E2VQ_STAMP_START
#define E2VQ_STAMP(i)
#define E2VQ_STAMP_DRAIN(i)
#endif

// One wave = 64 frames, independent of every other wave (no LDS sharing, no barriers): the codeword tile images come
// straight from L2 (9 KB per 32 codewords at NC = 37; 16 B per lane and granule) -- measured as fast as a workgroup-shared
// LDS ring (tools/probe/pre_sweep.hip -DDIRECT_L2), and it lets the two waves of a SIMD drift apart so that one sweeps
// (matrix pipe) while the other is in its latency-bound evaluate / accumulate phase.
// ROT (round 6, fused quantize): a rotating tile loop -- the next tile requested granule by granule behind the current tile's last
// readers (inline asm, hand-placed waits: pre_job) -- in place of a loop that loads a tile and uses it at once (every tile's L2
// latency exposed but for the SIMD's other wave).  Needs at least two tiles.
template <int NC, int MODE, int TPBM, bool ROT = false>
__global__ __launch_bounds__(TPBM, 2) void k_pass_pre(const double* __restrict__ blk, long T, long nblocks,
                                                  const h8* __restrict__ fimg, const float* __restrict__ fg,
                                                  const h8* __restrict__ cimg, PreScalars* __restrict__ ps,
                                                  const double* __restrict__ cbq, int MT, int idxmask,
                                                  const DevScalars* __restrict__ sc, const u64* __restrict__ l1max_bits,
                                                  unsigned short* __restrict__ sym, double* __restrict__ dmin,
                                                  i64* __restrict__ rows, int* __restrict__ fb_list, int stagger,
                                                  unsigned short* __restrict__ prev_sym, int incr,
                                                  const double* __restrict__ aos, const int* __restrict__ ea)
{
    // MODE 0: assignment only (limb image from fimg; FP64 frames from blk, or from the row-major payload aos);
    // MODE 6: assignment only, fused quantize -- limb image AND FP64 frames from the row-major payload aos, through the
    //         wave's LDS stage (pre_build_block).  (Round 2's accumulating modes 2 / 5 of this kernel left in round 5:
    //         training passes run k_pass_pre_lds or the kernels of vq_sweep.hip.)
    typedef PrePack<NC> PK;
    static_assert(MODE == 0 || MODE == 6, "assignment-only modes");
    constexpr bool QF = MODE == 6;
    E2VQ_STAMP_DECL
    constexpr int NS = (NC + 3) / 4, NP = (NS + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // (the wave's index as a scalar: block numbers and the LDS stage stay in SGPRs -- the rotating loop has no VGPR to spare)
    const int lane0 = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long wave = (long)blockIdx.x * (TPBM >> 6) + wib;
    const long nwaves = (long)gridDim.x * (TPBM >> 6);
    double* stage = (double*)smem + wib * (64 * NC);                       // QF: the wave's 64 row-major FP64 frames
    int* eas = (int*)((double*)smem + (TPBM >> 6) * (64 * NC));            // QF: per-coefficient scale exponents
    if constexpr (QF) {
        for (int n = threadIdx.x; n < NC; n += TPBM) eas[n] = ea[n];
        __syncthreads();
    }
    (void)sc;
    (void)l1max_bits;
    (void)rows;
    (void)prev_sym;
    (void)incr;
    int maskv = idxmask;
    asm volatile("" : "+v"(maskv));
    float ninf = -__builtin_inff();
    asm volatile("" : "+v"(ninf));
    const float ymax1 = __int_as_float(ps->ymax_bits);
    // 2 rho, rho = 2^-(22-idxbits): put together from its bit pattern, so that it stays in a scalar register (as a float
    // division it lived in a VGPR across the tile loop)
    const float relk = __int_as_float((127 + __builtin_popcount(~idxmask) - 21) << 23);

    // Waves w and w + 4 of an 8-wave workgroup share a SIMD.  Started together they would reach their evaluate /
    // accumulate phases together; half a block period of delay for waves 4..7 makes the phases alternate.
    if (stagger && TPBM == 512 && nblocks >= 2 * nwaves && wib >= 4)
        for (int i = 0; i < stagger * MT / 8; ++i) __builtin_amdgcn_s_sleep(127);  // ~8k cycles each; a tile ~2k

    (void)lane0;
    for (long b = wave; b < nblocks; b += nwaves) {
        E2VQ_STAMP_START
        // ---- f16 limb images of the wave's 64 frames: B operands, resident for the sweep ----------
        h8 B[2][PK::PAIRS];
        float gq[2] = {0.f, 0.f};
        // (ROT: the lane index afresh in front of and behind the tile loop, so that nothing derived from it lives across it)
        int lane = ROT ? pre_fresh_lane() : lane0;
        if constexpr (QF) {
            pre_build_block<NC>(aos, b, T, lane, stage, eas, B, gq);
        } else {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int p = 0; p < PK::PAIRS; ++p) B[cb][p] = fimg[((b * 2 + cb) * PK::PAIRS + p) * 64 + lane];
        }
        E2VQ_STAMP_DRAIN(0)  // limb images (or the fused build of them) are there
        float k1[2], k2[2], k3[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) k1[cb] = k2[cb] = k3[cb] = __int_as_float(0x7f7fffff);

        f16v acc0[3], acc1[3];
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[l][r] = l == 2 ? 3.0e38f : 0.f;

        if constexpr (ROT) {
            static_assert(MODE == 6, "the rotating loop serves the fused quantize");
            constexpr int NU = PK::NU;
            constexpr unsigned TILE_IMG = (unsigned)PK::TILE_E * 16u;
            const char* cimg_c = (const char*)cimg;
            const unsigned lo_t = (unsigned)pre_fresh_lane() * 16u;
            // ONE register set (two made this kernel spill -- 36 registers, with scratch reloads inside a loop whose waits count
            // the vector-memory operations in flight: not an option): job 1 of tile t is the last reader of the set; behind the
            // last use of each granule it requests the same granule of tile t + 1 (inline asm), and job 0 of tile t + 1 waits
            // for each in front of its first use with s_waitcnt vmcnt(NU - 1 - rank).  Half a tile of MFMAs (and the partner
            // wave's) lies between a request and its use, where the plain loop had none.
            h8 A[NU];
            pre_load_tile_asm<NC>(A, cimg_c, lo_t);  // tile 0: one exposed L2 latency per block
            pre_job<NC, 0, false>(acc0, B[0], A, acc1, 0xffff, k1[1], k2[1], k3[1], maskv, ninf, nullptr, 0u);
            pre_job<NC, 0, true>(acc1, B[1], A, acc0, 0, k1[0], k2[0], k3[0], maskv, ninf, cimg_c + TILE_IMG, lo_t);
            for (int t = 1; t < MT - 1; ++t) {
                pre_job<NC, NU, false>(acc0, B[0], A, acc1, t - 1, k1[1], k2[1], k3[1], maskv, ninf, nullptr, 0u);
                pre_job<NC, 0, true>(acc1, B[1], A, acc0, t, k1[0], k2[0], k3[0], maskv, ninf, cimg_c + (size_t)(t + 1) * TILE_IMG, lo_t);
            }
            pre_job<NC, NU, false>(acc0, B[0], A, acc1, MT - 2, k1[1], k2[1], k3[1], maskv, ninf, nullptr, 0u);
            pre_job<NC, 0, false>(acc1, B[1], A, acc0, MT - 1, k1[0], k2[0], k3[0], maskv, ninf, nullptr, 0u);
        } else {
            for (int t = 0; t < MT; ++t) {
                h8 A[PK::NU];
#pragma unroll
                for (int u = 0; u < PK::NU; ++u) A[u] = cimg[(long)t * PK::TILE_E + u * 64 + lane];
                // (t = 0: the "previous" accumulators hold 3e38)
                pre_job_pinned<NC>(acc0, B[0], A, acc1, (t - 1) & 0xffff, k1[1], k2[1], k3[1], maskv, ninf);
                pre_job_pinned<NC>(acc1, B[1], A, acc0, t, k1[0], k2[0], k3[0], maskv, ninf);
            }
        }
        pre_epilogue<NC>(acc1, MT - 1, k1[1], k2[1], k3[1], maskv, ninf);
        if constexpr (ROT) lane = pre_fresh_lane();
        const int q = lane >> 4, j = lane & 15;

        E2VQ_STAMP(1)  // tile loop
        // ---- per frame: merge the two lane halves (rows 4h..4h+3 of every 8), certify the top two -------------
        int c1[2], c2[2], c3[2];
        bool cert[2], amb[2], amb3[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int hb = (lane >> 5) << 2;
            const float a1 = __int_as_float(__float_as_int(k1[cb]) | hb), a2 = __int_as_float(__float_as_int(k2[cb]) | hb),
                        a3 = __int_as_float(__float_as_int(k3[cb]) | hb);
            const float b1 = __shfl_xor(a1, 32, 64), b2 = __shfl_xor(a2, 32, 64), b3 = __shfl_xor(a3, 32, 64);
            const float t3 = med3f(a2, a3, b1), t2 = med3f(a1, a2, b1), t1 = med3f(a1, b1, ninf);
            const float u3 = med3f(t2, t3, b2), u2 = med3f(t1, t2, b2);
            const float w3 = med3f(u2, u3, b3);
            // t1 <= u2 <= w3: the three smallest keys of frame 32 cb + (lane & 31)
            const long t = b * 64 + 32 * cb + (lane & 31);
            const float g = QF ? gq[cb] : (t < T ? fg[t] : 0.f);
            float tau = 1.27f * (512.f * (g + ymax1 + (NC + 4.0f)) + relk * t1);
            if constexpr (QF) {  // (round 6: the smallest key with its tile's own tolerance -- see k_pass_pre_lds)
                const int tl = (__float_as_int(t1) & ~idxmask) >> 5;
                const float2 tt = ((const float2*)(cimg + (size_t)MT * PK::TILE_E))[tl < MT ? tl : 0];
                tau = 1.27f * (256.f * __builtin_fmaf(tt.x, g, tt.y) + 256.f * (g + ymax1 + (NC + 4.0f)) + relk * t1);
            }
            cert[cb] = t1 >= 1.0e-30f && t1 < 1.0e37f && w3 > t1 + tau;
            amb[cb] = !(u2 > t1 + tau);  // the runner-up is within reach: it needs the exact evaluation too
            c1[cb] = __float_as_int(t1) & ~idxmask;
            c2[cb] = __float_as_int(u2) & ~idxmask;
            // round 6 (fused quantize): where the third key is within reach too, the top THREE are evaluated if every other
            // codeword is out of reach -- pre_fourth_bound: a lower bound of the frame's fourth-smallest key from what the two
            // lane halves kept.  (The same argument as for two: a codeword whose key exceeds t1 + tau cannot be the nearest.)
            c3[cb] = __float_as_int(w3) & ~idxmask;
            amb3[cb] = QF && !cert[cb] && t1 >= 1.0e-30f && t1 < 1.0e37f && pre_fourth_bound(a2, a3, b2, b3) > t1 + tau;
            if (QF) cert[cb] = cert[cb] || amb3[cb];
        }

        if constexpr (QF) {
            // fused quantize (round 3): the frames are in the wave's LDS stage, so each lane evaluates ITS frame
            // (frame lane of the block = column block lane >> 5, column lane & 31; both halves hold the merged keys)
            // against both candidates with the canonical chain -- acc = fma(r[n], cq[n], acc), n ascending from +0.0,
            // the oracle's definition, which the FP64 MFMA reproduces -- instead of the diagonals of 16x16 MFMA tiles
            const int hsel = lane >> 5;
            const bool certl = hsel ? cert[1] : cert[0];
            const int ca = hsel ? c1[1] : c1[0], cb2 = hsel ? c2[1] : c2[0];
            constexpr int NH = (NC + 1) / 2, NPADQ = (NC + 7) & ~7;
            const double2* r1 = (const double2*)(cbq + (long)ca * NPADQ);
            const double2* r2 = (const double2*)(cbq + (long)cb2 * NPADQ);
            double2 x[NH], y[NH];
#pragma unroll
            for (int n2 = 0; n2 < NH; ++n2) {  // (rows are padded to a multiple of 8 doubles)
                x[n2] = r1[n2];
                y[n2] = r2[n2];
            }
            const double* fr = stage + lane * NC;
            double d1 = 0.0, d2 = 0.0;
#pragma unroll
            for (int n2 = 0; n2 < NH; ++n2) {
                if ((n2 & 3) == 0) asm volatile("" ::: "memory");  // (frame reads eight at a time, not all hoisted)
                const double f0 = fr[2 * n2];
                d1 = __builtin_fma(f0, x[n2].x, d1);
                d2 = __builtin_fma(f0, y[n2].x, d2);
                if (2 * n2 + 1 < NC) {
                    const double f1 = fr[2 * n2 + 1];
                    d1 = __builtin_fma(f1, x[n2].y, d1);
                    d2 = __builtin_fma(f1, y[n2].y, d2);
                }
            }
            // (a runner-up whose key is out of reach cannot win: comparing it anyway changes nothing)
            const bool take_b = d2 < d1 || (d2 == d1 && cb2 < ca);
            double bestq = take_b ? d2 : d1;
            int idxq = take_b ? cb2 : ca;
            const bool third = hsel ? amb3[1] : amb3[0];
            if (__ballot(third) != 0) {  // (wave-uniform; rare on data that certifies well)
                const int cc = hsel ? c3[1] : c3[0];
                const double2* r3 = (const double2*)(cbq + (long)cc * NPADQ);
#pragma unroll
                for (int n2 = 0; n2 < NH; ++n2) y[n2] = third ? r3[n2] : make_double2(0.0, 0.0);
                double d3 = 0.0;
#pragma unroll
                for (int n2 = 0; n2 < NH; ++n2) {
                    if ((n2 & 3) == 0) asm volatile("" ::: "memory");
                    d3 = __builtin_fma(fr[2 * n2], y[n2].x, d3);
                    if (2 * n2 + 1 < NC) d3 = __builtin_fma(fr[2 * n2 + 1], y[n2].y, d3);
                }
                const bool take_c = third && (d3 < bestq || (d3 == bestq && cc < idxq));
                bestq = take_c ? d3 : bestq;
                idxq = take_c ? cc : idxq;
            }
            const long t = b * 64 + lane;
            if (t < T) {
                if (!certl) {
                    fb_list[atomicAdd(&ps->fb_count, 1)] = (int)t;
                } else {
                    if (sym) sym[t] = (unsigned short)idxq;
                    if (dmin) dmin[t] = bestq;
                }
            }
            // (the next block's LDS-DMA overwrites the stage: this block's reads are complete first)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            continue;
        }

        E2VQ_STAMP_DRAIN(2)  // certification, cells of the previous pass
        // ---- FP64 frames (MFMA operand layout of k_pass_mfma), exact evaluation of the two candidates -----------
        double Bf[4][2 * NP];
        if constexpr (QF)  // fused quantize: the frames are still in the wave's LDS stage
            load_block_frames_stage<NC>(stage, lane, Bf);
        else if (MODE == 0 && aos)  // quantize: the FP64 frames come straight from the row-major payload
            load_block_frames_rowmajor<NC>(aos, b, T, lane, Bf);
        else
            load_block_frames<NC>(blk, b, lane, Bf);
        E2VQ_STAMP_DRAIN(3)  // FP64 frames
        double best[4];
        int idx[4];
        bool skip[4];
        {
            int ca[4], cbx[4];
            bool two[4];
            PreCand<NC> g[4];
            // the four gathers of a round are issued together: two exposed L2 latencies per block instead of eight
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const int src = 16 * (ft & 1) + j;  // a lane of half 0 that holds frame 16 ft + j of column block ft / 2
                ca[ft] = __shfl(c1[ft >> 1], src, 64);
                cbx[ft] = __shfl(c2[ft >> 1], src, 64);
                skip[ft] = __shfl((int)cert[ft >> 1], src, 64) == 0;
                two[ft] = __ballot(__shfl((int)amb[ft >> 1], src, 64) != 0) != 0;  // wave-uniform
                g[ft] = pre_gather<NC>(ca[ft], cbq, q);
            }
            E2VQ_STAMP_DRAIN(4)  // gathers of the first candidates
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                best[ft] = pre_exact<NC>(Bf[ft], g[ft], j);
                idx[ft] = ca[ft];
            }
            E2VQ_STAMP(5)  // exact chains of the first candidates
            if (two[0] || two[1] || two[2] || two[3]) {
#pragma unroll
                for (int ft = 0; ft < 4; ++ft)
                    if (two[ft]) g[ft] = pre_gather<NC>(cbx[ft], cbq, q);
#pragma unroll
                for (int ft = 0; ft < 4; ++ft)
                    if (two[ft]) {
                        const double db = pre_exact<NC>(Bf[ft], g[ft], j);
                        const bool take_b = db < best[ft] || (db == best[ft] && cbx[ft] < ca[ft]);
                        best[ft] = take_b ? db : best[ft];
                        idx[ft] = take_b ? cbx[ft] : ca[ft];
                    }
            }
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) idx[ft] = skip[ft] ? 0 : idx[ft];
            E2VQ_STAMP(6)  // the runners-up
        }


        // ---- outputs: lane 16q + j owns frame b*64 + lane; uncertified frames go to the fallback list ---------
        {
            const double bs = q == 0 ? best[0] : q == 1 ? best[1] : q == 2 ? best[2] : best[3];
            const int is = q == 0 ? idx[0] : q == 1 ? idx[1] : q == 2 ? idx[2] : idx[3];
            const bool sk = q == 0 ? skip[0] : q == 1 ? skip[1] : q == 2 ? skip[2] : skip[3];
            const long t = b * 64 + lane;
            if (t < T) {
                if (sk) {
                    fb_list[atomicAdd(&ps->fb_count, 1)] = (int)t;
                } else {
                    if (sym) sym[t] = (unsigned short)is;
                    if (dmin) dmin[t] = bs;
                }
            }
        }
        E2VQ_STAMP(7)  // outputs
        E2VQ_STAMP(8)  // accumulate (issue side: the atomics drain later)
#ifdef E2VQ_PRE_STAMP
        st_n += 1;
#endif
    }
#ifdef E2VQ_PRE_STAMP
    {
        E2VQ_STAMP_START
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        E2VQ_STAMP(9)  // drain at the wave's end
        if (lane == 0) {
            for (int k = 0; k < 12; ++k) atomicAdd(&g_pre_stamps[k], st_acc[k]);
            atomicAdd(&g_pre_stamps[12], st_n);
            atomicAdd(&g_pre_stamps[13], 1ull);
        }
    }
#endif
}

// ---- k_pass_pre_lds: the accumulating prefiltered pass with the FP64 frames of a block staged in LDS ------------------
// Same sweep, same keys, same certification, same bits as k_pass_pre.  Everything around the tile loop is built on one
// fact (round 3, stamps of tools/probe/pre_stamps.py): a wave's vector-memory operations retire IN ORDER, so every load
// issued after a block's atomics -- a register reload from scratch, the next codeword tile, the next block's limb images
// -- waits until those atomics have been performed at the memory side (thousands of cycles with every CU adding).
//   * All atomics of a block go out in ONE burst at the very end of the block, behind the loads of the next block's limb
//     images and of its first TWO codeword tiles; what follows them in the queue (the LDS-DMA of the next block's FP64
//     frames, the operands of tile 2) is not needed before two tiles have been swept.
//   * The FP64 frames arrive by LDS-DMA (global_load_lds_dwordx4 from a row-major resident copy: no registers, issued a
//     whole tile loop ahead), with the tolerance terms and the cells of the previous pass behind them.
//   * Exact evaluation: lane = frame.  Each lane runs the canonical chain acc = fma(r[n], cq[n], acc), n ascending from
//     +0.0 -- the oracle's definition itself, which the FP64 MFMA reproduces (tools/probe/mfma64.hip) -- for both of its
//     candidates, r from its LDS row, cq rows gathered from L2.
//   * Accumulate in place: a frame that contributes converts its own LDS row to the (hi, lo) limb pairs where the
//     doubles were (8 bytes either way), then the wave adds the rows of those frames to their cells, four frames per
//     step (four 64-lane adds + one carrying the four row tails and the count).  Only frames that moved are touched.
//   * Round 4 -- the tile loop: codeword tiles are NU unique granules (PrePack: 9 instead of 15 at NC = 37) in TWO
//     register sets (even / odd tiles).  The operands of tile t + 2 are requested granule by granule behind their last
//     readers in job 1 of tile t and waited for one by one in job 0 of tile t + 2: a whole tile of MFMAs (two jobs of the
//     partner wave as well) lies between a request and its use, where round 3's single set left half a tile -- and the
//     burst of a block's atomics has two wait-free tiles to drain under instead of one.
template <int NC, bool ROT, int ACC>
__global__ __launch_bounds__(512, 2) void k_pass_pre_lds(const double* __restrict__ aos, long T, long nblocks,
                                                         const h8* __restrict__ fimg, const float* __restrict__ fg,
                                                         const h8* __restrict__ cimg, PreScalars* __restrict__ ps,
                                                         const double* __restrict__ cbq, int MT, int idxmask,
                                                         const DevScalars* __restrict__ sc,
                                                         const u64* __restrict__ l1max_bits,
                                                         unsigned short* __restrict__ sym, double* __restrict__ dmin,
                                                         i64* __restrict__ rows, int* __restrict__ fb_list, int stagger,
                                                         unsigned short* __restrict__ prev_sym, int incr,
                                                         i64* __restrict__ fam, PreRec rec)
{
    constexpr bool ACCUM = ACC != 0;
    // incr: 0 = full accumulation; 1 = incremental (rows and cells of the previous pass persist: only frames that changed
    // cell are moved); 2 = the seeded first pass of a level (vq_update.hip, k_seed_family): a frame's old cell is the even
    // child 2 * prev_sym of the cell it had at the previous size; one that lands in the odd child 2 * prev_sym + 1 adds its
    // limbs to row prev_sym of the side table `fam` and nothing else
    // ACC = 1: the cell sums as a burst of atomics per block.  ACC = 2: the contributions are RECORDED
    // -- 8 bytes each, into the region of (this workgroup, bin of cells) -- and k_reduce_records adds them up (`fam` is
    // not used: a frame that lands in the odd child of its family is recorded for the side table's bin)
    typedef PrePack<NC> PK;
    typedef PreLds<NC> PL;
    constexpr int TPBM = PL::WAVES * 64;
    static_assert(ACC == 1 || ACC == 2, "burst of atomics or records");
    static_assert(ACC != 1 || PL::BURST_OK, "rows of more than 80 elements are recorded, not added in a burst");
    constexpr int NU = PK::NU;
    constexpr int RS = (2 * NC + 5 + 7) & ~7, NPAD = (NC + 7) & ~7;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // (the wave's index as a scalar: everything derived from it -- block numbers, the LDS region -- stays in SGPRs)
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long wave = (long)blockIdx.x * (TPBM >> 6) + wib;
    const long nwaves = (long)gridDim.x * (TPBM >> 6);
    unsigned char* wbase = smem + (size_t)wib * PL::WAVE_BYTES;
    double* stage = (double*)wbase;
    const int* stagei = (const int*)wbase;
    const float* fgs = (const float*)(wbase + PL::STAGE_BYTES);
    const unsigned short* prevs = (const unsigned short*)(wbase + PL::STAGE_BYTES + 256);
    E2VQ_STAMP_DECL
    // (ACC = 2) records written so far into each of this workgroup's regions: one LDS word per bin, behind the waves' regions
    int* const rcnt = (int*)(smem + (size_t)PL::WAVES * PL::WAVE_BYTES);
    if constexpr (ACC == 2) {
        if (threadIdx.x < 64) rcnt[threadIdx.x] = 0;
        __syncthreads();
    }

    const int sh_r = sc->sh_r;
    const int Ed = dist_exponent(sc->maxabs, __longlong_as_double((i64)*l1max_bits));
    const int sh_d = 30 - Ed, sh_d2 = 30 - 2 * Ed;
    const bool fast_fix = sh_r >= -1000 && sh_r <= 1000;  // fix2_mul applies (always, but for absurdly scaled data)
    // (powers of two put together from their bit patterns with integer arithmetic: uniform values that stay in scalar
    // registers -- ldexp / a float division would leave them in VGPRs for the whole kernel, i.e. spilled across the tile loop
    // and reloaded from scratch in the middle of the phases that must not wait on the vector-memory counter)
    auto pow2 = [](int e) { return __longlong_as_double((long long)(1023 + e) << 52); };  // |e| <= 1000
    const double scale_r = pow2(fast_fix ? sh_r : 0);
    const bool fast_d = sh_d >= -1000 && sh_d <= 1000 && sh_d2 >= -1000 && sh_d2 <= 1000;
    const double scale_d = pow2(fast_d ? sh_d : 0), scale_d2 = pow2(fast_d ? sh_d2 : 0);
    const float ymax1 = __int_as_float(ps->ymax_bits);
    const float relk = __int_as_float((127 + __builtin_popcount(~idxmask) - 21) << 23);  // 2 rho, rho = 2^-(22-idxbits)
    // ROT: the rotating, double-buffered tile loop; it needs at least four tiles and an even count (the register set of a
    // tile is its parity).  Any other count -- a base codebook of unusual size -- runs the instantiation with the simple
    // loop: one set, one wait per tile (a kernel of its own, so that its register needs do not disturb this one's).

    // (partner waves w and w + 4 of a SIMD: half a block period apart, as in k_pass_pre)
    if (stagger && nblocks >= 2 * nwaves && wib >= 4)
        for (int i = 0; i < stagger * MT / 8; ++i) __builtin_amdgcn_s_sleep(127);

    // limb images of the block and the first two codeword tiles: loop-carried, requested for the NEXT block before the
    // current block's atomics go out
    // (addresses as uniform base + 32-bit lane offset: the compiler keeps no 64-bit per-lane pointers alive -- and
    // spilled -- across the phases; a reload from scratch behind the atomics would wait for them)
    h8 B[2][PK::PAIRS];
    h8 A0[NU], A1[NU];
    constexpr unsigned BLOCK_IMG = 2u * PK::PAIRS * 64u * 16u;  // bytes of a block's limb image
    constexpr unsigned TILE_IMG = (unsigned)PK::TILE_E * 16u;   // bytes of a codeword tile's limb image
#define E2VQ_LDS_LOAD_B(BLK, LN)                                                                              \
    {                                                                                                         \
        const char* fb_ = (const char*)fimg + (size_t)(BLK) * BLOCK_IMG;                                      \
        const unsigned lo_ = (unsigned)(LN) * 16u;                                                            \
        _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) _Pragma("unroll") for (int p = 0; p < PK::PAIRS; ++p) \
            B[cb][p] = *(const h8*)(fb_ + (lo_ + (unsigned)((cb * PK::PAIRS + p) * 1024)));                   \
    }
#define E2VQ_LDS_LOAD_A(ASET, TILE, LN)                                                                       \
    {                                                                                                         \
        const char* cb_ = (const char*)cimg + (size_t)(TILE) * TILE_IMG;                                      \
        const unsigned lo_ = (unsigned)(LN) * 16u;                                                            \
        _Pragma("unroll") for (int u = 0; u < NU; ++u) ASET[u] = *(const h8*)(cb_ + (lo_ + (unsigned)(u * 1024))); \
    }
    if (wave < nblocks) {
        E2VQ_LDS_LOAD_B(wave, lane)
        E2VQ_LDS_LOAD_A(A0, 0, lane)
        E2VQ_LDS_LOAD_A(A1, 1, lane)  // (every codebook of this kernel has at least two tiles: M >= 64)
        pre_lds_request<NC>(aos, fg, (ACCUM && incr) ? prev_sym : nullptr, wave, lane, wbase);
    }
    // (as before every block's atomics: both ways into the block loop arrive with no register load pending, so the
    // compiler puts no counter wait in front of tile 0)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

    for (long b = wave; b < nblocks; b += nwaves) {
        E2VQ_STAMP_START
        // per-lane constants of the tile loop are made afresh for every block (from scalars, the lane index from the
        // hardware): kept alive across the blocks they would be spilled during the evaluate phase and reloaded here --
        // behind the previous block's atomics
        int maskv;
        asm volatile("v_mov_b32 %0, %1" : "=v"(maskv) : "s"(idxmask));  // (scalar source: nothing to keep in a VGPR)
        float ninf = -__builtin_inff();
        asm volatile("" : "+v"(ninf));
        const int lane_t = pre_fresh_lane();
        const unsigned lo_t = (unsigned)lane_t * 16u;
        float k1[2], k2[2], k3[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) k1[cb] = k2[cb] = k3[cb] = __int_as_float(0x7f7fffff);
        f16v acc0[3], acc1[3];
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[l][r] = l == 2 ? 3.0e38f : 0.f;
        // Tiles 0 and 1 came with the block's limb images, complete before the previous block's atomics went out: they run
        // without a counter wait while those atomics drain; the loads of tile 2 queue behind them.  The in-loop loads are
        // inline asm with hand-placed waits: any load the compiler can see in this loop makes it insert counter waits
        // that, at t = 0, would wait for the atomics (the counter is in-order and the number of atomics unknown).
        // One tile = job 0 (column block 0, epilogue of the previous tile's job 1) + job 1 (column block 1, epilogue of
        // job 0); the "previous" accumulators of tile 0 hold 3e38.
        const char* cimg_c = (const char*)cimg;
#define E2VQ_TILE(T_, ASET, WAIT_, LOADS_)                                                                              \
    {                                                                                                                   \
        pre_job<NC, WAIT_, false>(acc0, B[0], ASET, acc1, ((T_) - 1) & 0xffff, k1[1], k2[1], k3[1], maskv, ninf, nullptr, 0u); \
        pre_job<NC, 0, LOADS_>(acc1, B[1], ASET, acc0, (T_), k1[0], k2[0], k3[0], maskv, ninf,                           \
                               cimg_c + (size_t)((T_) + 2) * TILE_IMG, lo_t);                                           \
    }
        if constexpr (ROT) {
            E2VQ_TILE(0, A0, 0, true)  // (requests tile 2)
            E2VQ_TILE(1, A1, 0, true)  // (requests tile 3)
            for (int t = 2; t < MT - 2; t += 2) {
                E2VQ_TILE(t, A0, 2 * NU, true)
                E2VQ_TILE(t + 1, A1, 2 * NU, true)
            }
            E2VQ_TILE(MT - 2, A0, 2 * NU, false)  // (tile MT - 1's loads are the NU younger ones; none beyond the last tile
            E2VQ_TILE(MT - 1, A1, NU, false)      //  is ever requested: nothing younger here)
        } else {
            E2VQ_TILE(0, A0, 0, false)
            E2VQ_TILE(1, A1, 0, false)
            for (int t = 2; t < MT; ++t) {
                pre_load_tile_asm<NC>(A0, cimg_c + (size_t)t * TILE_IMG, lo_t);
                E2VQ_TILE(t, A0, 0, false)
            }
            // (two tiles only: no load was waited for in the loop, and the block's LDS-DMA must have landed below)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#undef E2VQ_TILE
        pre_epilogue<NC>(acc1, MT - 1, k1[1], k2[1], k3[1], maskv, ninf);
        E2VQ_STAMP(1)  // tile loop

        // Everything below recomputes its addresses from an opaque copy of the lane index: hoisted out of the block loop
        // they would live across the tile loop, be spilled there, and their reloads would queue behind the atomics.
        const int ln = pre_fresh_lane();
        const long bn = b + nwaves;

        // ---- ln = frame b*64 + ln = (column block ln >> 5, column ln & 31): the ln's half of that frame's
        // keys is its own k*[ln >> 5]; the other half sits in the partner ln (ln ^ 32) as ITS k*[ln >> 5] --
        const int half = ln >> 5;
        const int hb = half << 2;
        float a1, a2, a3, s1, s2, s3;
        {
            const float o1 = __int_as_float(__float_as_int(k1[0]) | hb), o2 = __int_as_float(__float_as_int(k2[0]) | hb),
                        o3 = __int_as_float(__float_as_int(k3[0]) | hb);
            const float p1 = __int_as_float(__float_as_int(k1[1]) | hb), p2 = __int_as_float(__float_as_int(k2[1]) | hb),
                        p3 = __int_as_float(__float_as_int(k3[1]) | hb);
            a1 = half ? p1 : o1, a2 = half ? p2 : o2, a3 = half ? p3 : o3;  // own frame's keys
            s1 = half ? o1 : p1, s2 = half ? o2 : p2, s3 = half ? o3 : p3;  // the partner's frame's keys
        }
        const float b1 = __shfl_xor(s1, 32, 64), b2 = __shfl_xor(s2, 32, 64), b3 = __shfl_xor(s3, 32, 64);
        const float t3 = med3f(a2, a3, b1), t2 = med3f(a1, a2, b1), t1 = med3f(a1, b1, ninf);
        const float u3 = med3f(t2, t3, b2), u2 = med3f(t1, t2, b2);
        const float w3 = med3f(u2, u3, b3);
        // t1 <= u2 <= w3: the three smallest keys of the ln's frame (the merge is symmetric in the two halves)
        // The block's LDS-DMA has landed: it was requested before the tile loop, and the loads of the last codeword tile --
        // younger, completed in order -- have been consumed.  (No counter wait here: it would also wait for the limb
        // images just requested.)  The fence keeps the compiler from reading the LDS rows any earlier.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        asm volatile("" ::: "memory");
        const long t = b * 64 + ln;
        const bool live = t < T;
        const float g = live ? fgs[ln] : 0.f;
        // round 6: the smallest key's own tolerance is its TILE's -- 2^8 (c_t g + d_t) from the table behind the codebook image
        // (k_pre_codebook: the tile's limbs sit 2^-s_t below the global scale) --, every other key's the old bound
        const int tl1 = (__float_as_int(t1) & ~idxmask) >> 5;
        const float2 tt = ((const float2*)(cimg + (size_t)MT * PK::TILE_E))[tl1 < MT ? tl1 : 0];
        const float tau = 1.27f * (256.f * __builtin_fmaf(tt.x, g, tt.y) + 256.f * (g + ymax1 + (NC + 4.0f)) + relk * t1);
        const bool keyok = t1 >= 1.0e-30f && t1 < 1.0e37f;
        const bool cert2 = keyok && w3 > t1 + tau;
        const bool amb = !(u2 > t1 + tau);  // the runner-up is within reach: it needs the exact evaluation too
        // round 6: the third key within reach as well -- the top THREE are evaluated exactly where every other codeword is out
        // of reach (pre_fourth_bound: a lower bound of the frame's fourth-smallest key from what the two lane halves kept; the
        // argument is the one for two candidates).  On data whose distortions are small differences of large terms this
        // certifies about half of the frames two candidates leave to the FP64 fallback sweep.
        const bool amb3 = !cert2 && keyok && pre_fourth_bound(a2, a3, b2, b3) > t1 + tau;
        const bool cert = cert2 || amb3;
        const int c1 = __float_as_int(t1) & ~idxmask, c2 = __float_as_int(u2) & ~idxmask, c3 = __float_as_int(w3) & ~idxmask;
        const int old = (ACCUM && incr) ? (incr == 2 ? 2 : 1) * (int)prevs[ln] : 0;
        E2VQ_STAMP(2)  // merge, certification

        // ---- exact evaluation of both candidates: the canonical chain, one frame per ln ---------------------------
        double best;
        int idx;
        {
            // both codeword rows are requested at once (one exposed L2 latency); the frame's coefficients come from
            // its LDS row eight at a time, fenced, so that the compiler does not hoist all 37 reads above the chains
            constexpr int NH = (NC + 1) / 2;
            const double2* r1 = (const double2*)(cbq + (long)c1 * NPAD);
            const double2* r2 = (const double2*)(cbq + (long)c2 * NPAD);
            double2 x[NH], y[NH];
#pragma unroll
            for (int n2 = 0; n2 < NH; ++n2) x[n2] = r1[n2];  // (rows are padded to a multiple of 8 doubles)
            // (the runner-up's row only where it can matter: a gather costs the texture path a request per lane)
#pragma unroll
            for (int n2 = 0; n2 < NH; ++n2) y[n2] = make_double2(0.0, 0.0);
            if (amb) {
#pragma unroll
                for (int n2 = 0; n2 < NH; ++n2) y[n2] = r2[n2];
            }
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
            if (__ballot(amb3) != 0) {  // (wave-uniform; the third candidate's row takes the runner-up's registers)
                const double2* r3 = (const double2*)(cbq + (long)c3 * NPAD);
#pragma unroll
                for (int n2 = 0; n2 < NH; ++n2) y[n2] = amb3 ? r3[n2] : make_double2(0.0, 0.0);
                double d3 = 0.0;
#pragma unroll
                for (int n2 = 0; n2 < NH; ++n2) {
                    if ((n2 & 3) == 0) asm volatile("" ::: "memory");
                    d3 = __builtin_fma(fr[2 * n2], y[n2].x, d3);
                    if (2 * n2 + 1 < NC) d3 = __builtin_fma(fr[2 * n2 + 1], y[n2].y, d3);
                }
                const bool take_c = amb3 && (d3 < best || (d3 == best && c3 < idx));
                best = take_c ? d3 : best;
                idx = take_c ? c3 : idx;
            }
        }
        const bool skip = !cert;
        idx = skip ? 0 : idx;
        E2VQ_STAMP(3)  // exact evaluation

        // ---- the next block's limb images and first two codeword tiles: requested now -- behind the register-hungry
        // chains, ahead of the outputs and the distortion sums, which cover part of their latency -- and complete before
        // the atomics go out
        {   // (unconditional -- the wave's last block reloads its own images: a conditional load would keep the old B and A
            // alive through the evaluation above)
            const long bl = bn < nblocks ? bn : b;
            E2VQ_LDS_LOAD_B(bl, ln)
            E2VQ_LDS_LOAD_A(A0, 0, ln)
            E2VQ_LDS_LOAD_A(A1, 1, ln)
        }

        // ---- outputs; uncertified frames go to the fallback list -----------------------------------------------------
        if (live) {
            if (skip) {
                fb_list[atomicAdd(&ps->fb_count, 1)] = (int)t;
            } else {
                if (sym) sym[t] = (unsigned short)idx;
                if (dmin) dmin[t] = best;
            }
        }
        // the block's distortion sums (e and e^2 of every certified frame as limb pairs): only their column totals are
        // ever used, so the wave adds them up and one lane per element adds them to the distortion columns of some row
        // with the block's other atomics (no per-lane running sums to keep across the tile loops)
        i64 dsum;
        {
            // (tried: four ds_add_u64 per lane into four words of the wave's LDS region instead of the shuffles below -- 64
            // adds on one address serialise at ~100 cycles each: 27 k cycles per block against 2.7 k)
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
        E2VQ_STAMP(4)  // outputs

        u64 movers = 0;
        bool bulk = false;
        // ---- frames that contribute: all of a full pass, the movers of an incremental one ---------------------------------
        const bool mov = ACCUM && live && !skip && (!incr || old != idx);
        if constexpr (ACC == 1) {
        // Many contributors (a full pass; the first incremental pass of a level): every contributing lane converts its own
        // row to limb pairs in place -- 37 conversions per lane, whatever the number of contributors.  Few: the lanes of
        // each atomic convert just the value they add (below) -- one conversion per lane and contributor.
        movers = __ballot(mov);
        bulk = __builtin_popcountll(movers) > 28;  // wave-uniform
        if (bulk) {
            if (mov) {
                double* fr = stage + ln * NC;
                if (fast_fix) {
#pragma unroll
                    for (int n = 0; n < NC; ++n) {
                        int hi, lo;
                        fix2_mul(fr[n], scale_r, hi, lo);
                        *(int2*)&fr[n] = make_int2(hi, lo);
                    }
                } else {
#pragma unroll 1
                    for (int n = 0; n < NC; ++n) {
                        int hi, lo;
                        fix2(fr[n], sh_r, hi, lo);
                        *(int2*)&fr[n] = make_int2(hi, lo);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        }  // ACC == 1
        E2VQ_STAMP(5)  // limb conversion (bulk)

        // ---- the block's atomics, four frames per step ------------------------------------------------------------------
        // Every load of this block -- and of the next block's limb images and first tiles -- has to be complete before the
        // first atomic: nothing may wait on the vector-memory counter from here to tile 2 of the next block.  The builtin
        // (not inline asm) so that the compiler's own counter bookkeeping sees it and inserts no later wait for those registers.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        if constexpr (ACC == 2) {
            // A contribution = (frame, cell, sign): '+' for the cell the frame is in now -- row `old >> 1` of the side table
            // when it landed in the odd child of its family (seeded pass) --, '-' for the cell an incremental mover left.
            // Records of one bin go to this workgroup's region of that bin, at positions handed out by the LDS counter:
            // per distinct bin among the wave's records one round of ballots gives every record its rank and lane `bin` the
            // count; ONE ds_add_rtn (lane = bin) reserves the space for all bins, two shuffles hand the bases back.
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
        if constexpr (ACC == 1) {
            u64 mm = movers;
            const int tq = ln >> 4, te = ln & 15;
            constexpr int NE = PL::NE;            // limb pairs + count
            constexpr bool TAIL = NE > 64;        // rows longer than a wave: one more add carries four row tails
            constexpr int NT = TAIL ? NE - 64 : 0;  // tail elements, the count last
            // lane e of a 64-lane add handles element e = 2 n + limb of a frame's contribution: it reads coefficient n from
            // the frame's LDS row and converts it itself (pairs of lanes convert the same value and keep one limb each): one
            // conversion per lane and contributing frame, where converting whole rows in place cost every lane 37 of them
            auto limb = [&](double x, int odd) -> int {
                int hi, lo;
                if (fast_fix)
                    fix2_mul(x, scale_r, hi, lo);
                else
                    fix2(x, sh_r, hi, lo);
                return odd ? lo : hi;
            };
            while (mm != 0) {
                int f[4], cell[4], oldc[4], v[4];
                bool on[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    on[k] = mm != 0;  // wave-uniform
                    f[k] = on[k] ? (int)__builtin_ctzll(mm) : 0;
                    mm = on[k] ? (mm & (mm - 1)) : mm;
                    cell[k] = __builtin_amdgcn_readlane(idx, f[k]);
                    oldc[k] = __builtin_amdgcn_readlane(old, f[k]);
                    const int e = ln;  // element handled in the full-width add
                    v[k] = e >= 2 * NC ? 1  // (e == 2 NC, short rows only: the count)
                                       : (bulk ? stagei[f[k] * (2 * NC) + e] : limb(stage[f[k] * NC + (e >> 1)], e & 1));
                }
                int tv = 0, tcell = 0, told = 0;
                bool ton = false;
                if constexpr (TAIL) {
                    const int tf = tq == 0 ? f[0] : tq == 1 ? f[1] : tq == 2 ? f[2] : f[3];
                    ton = tq == 0 ? on[0] : tq == 1 ? on[1] : tq == 2 ? on[2] : on[3];
                    tcell = tq == 0 ? cell[0] : tq == 1 ? cell[1] : tq == 2 ? cell[2] : cell[3];
                    told = tq == 0 ? oldc[0] : tq == 1 ? oldc[1] : tq == 2 ? oldc[2] : oldc[3];
                    const int et = 64 + te;  // (te < NT - 1: a limb; te == NT - 1: the count)
                    tv = te >= NT - 1 ? 1 : (bulk ? stagei[tf * (2 * NC) + et] : limb(stage[tf * NC + (et >> 1)], et & 1));
                    ton = ton && te < NT;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!on[k]) continue;  // wave-uniform
                    if (TAIL || ln < NE) {
                        if (incr == 2 && cell[k] == oldc[k] + 1) {  // (wave-uniform) in-family odd child: one add, to the side table
                            atomicAdd((u64*)&fam[(long)(oldc[k] >> 1) * RS + ln], (u64)(i64)v[k]);
                        } else {
                            atomicAdd((u64*)&rows[(long)cell[k] * RS + ln], (u64)(i64)v[k]);
                            if (incr) atomicAdd((u64*)&rows[(long)oldc[k] * RS + ln], (u64)(-(i64)v[k]));
                        }
                    }
                }
                if (TAIL && ton) {
                    if (incr == 2 && tcell == told + 1) {  // (per 16-lane group)
                        atomicAdd((u64*)&fam[(long)(told >> 1) * RS + 64 + te], (u64)(i64)tv);
                    } else {
                        atomicAdd((u64*)&rows[(long)tcell * RS + 64 + te], (u64)(i64)tv);
                        if (incr) atomicAdd((u64*)&rows[(long)told * RS + 64 + te], (u64)(-(i64)tv));
                    }
                }
            }
        }
        if (ln < 4 && dsum != 0) atomicAdd((u64*)&rows[(long)(b % (32 * MT)) * RS + 2 * NC + 1 + ln], (u64)dsum);
        if (ACCUM && prev_sym && live && !skip) prev_sym[t] = (unsigned short)idx;
        E2VQ_STAMP(6)  // atomics (issue)

        // ---- the next block's frames: the LDS rows are free once this block's reads are done ---------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (bn < nblocks) pre_lds_request<NC>(aos, fg, (ACCUM && incr) ? prev_sym : nullptr, bn, ln, wbase);
        E2VQ_STAMP(7)
#ifdef E2VQ_PRE_STAMP
        st_n += 1;
#endif
    }
#undef E2VQ_LDS_LOAD_A
#undef E2VQ_LDS_LOAD_B
    if constexpr (ACC == 2) {  // what each of the workgroup's regions holds
        __syncthreads();
        if ((int)threadIdx.x < rec.nbins) rec.counts[(size_t)blockIdx.x * rec.nbins + threadIdx.x] = rcnt[threadIdx.x];
    }
#ifdef E2VQ_PRE_STAMP
    {
        E2VQ_STAMP_START
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        E2VQ_STAMP(9)  // drain at the wave's end
        if (lane == 0) {
            for (int k = 0; k < 12; ++k) atomicAdd(&g_pre_stamps[16 + k], st_acc[k]);
            atomicAdd(&g_pre_stamps[12 + 16], st_n);
            atomicAdd(&g_pre_stamps[13 + 16], 1ull);
        }
    }
#endif
}

// ---- k_reduce_records (round 4): folds the records of an ACC = 2 pass into the rows ---------------------------------------
// The sweep wrote one 8-byte record per contribution -- (frame, cell within its bin | sign << 16) -- into the region of
// (sweeping workgroup, bin); the regions' fill counts are in `counts`.  grid = bins x slices: a workgroup owns one bin
// (<= REC_BIN_CELLS cells) and the regions of the sweeping workgroups s, s + nslices, ... of that bin, which it walks as ONE
// list (prefix sums of their counts in LDS), REC_CHUNK records at a time:
//   1. counting sort of the chunk by cell, in LDS: one ds_add_rtn per record gives its rank within its cell, a scan of the
//      counts the cells' offsets, and every record's (frame | sign << 31) goes to its sorted place;
//   2. every wave walks an equal span of the sorted chunk, eight records at a time (the next eight rows are already
//      requested): lane n loads coefficient n of the row (296 B, coalesced), converts it to its two limbs and adds them --
//      negated for a '-' record -- to two 64-bit sums IN REGISTERS;
//   3. where the cell changes (and at the span's end) the sums are added to the bin's table in LDS -- not to global memory:
//      a wave's vector-memory operations retire in order, the next rows would wait behind the atomics --; at the end the
//      table's non-zero words go to the global rows with one atomic each (the family side table for the bins behind the rows').
// A table's ds_add_u64 retires one to two lanes per clock and CU: adding every RECORD to it (75 lane-adds) bounded the first
// version of this kernel (and k_accum_ranges) at ~75 clocks per record and CU, whatever the number of waves.  Here a
// record costs two 32-bit LDS atomics, one row load and ~15 vector instructions, and the table sees one add per (span, cell).
// The work follows the number of contributions, not the number of frames; a workgroup whose regions are empty returns at
// once.  Exact 64-bit integers: any order, same bits.
constexpr int REC_BIN_CELLS = 60;   // x 640 B (NC = 37) = 38 KB of table + 25 KB: two workgroups per CU
constexpr int REC_BIN_CELLS_BIG = 120;  // (codebooks that would need more than 64 bins of 60 cells)
constexpr int REC_TPB = 512;
constexpr int REC_PER_THREAD = 4;
constexpr int REC_CHUNK = REC_TPB * REC_PER_THREAD;
constexpr int REC_BATCH = 8;

// (three workgroups per CU need <= 85 VGPRs: 66 as compiled -- the allocation is touchy, a store of `all` straight from the
// scan cost 104 registers and a third of the kernel's speed; check with tools/isa_regs.py after touching this kernel)
template <int NC>
__global__ __launch_bounds__(REC_TPB, 4) void k_reduce_records(const double* __restrict__ aos, PreRec rec, int nregions,
                                                            const DevScalars* __restrict__ sc,
                                                            i64* __restrict__ rows, i64* __restrict__ fam)
{
    constexpr int RS = (2 * NC + 5 + 7) & ~7;
    static_assert(NC < 64, "lane n = coefficient n, lane NC = the count");
    __shared__ int pre[260];      // pre[k] = the bin's records in front of region k; pre[256] = all
    __shared__ int hist[128];     // records of the chunk per cell
    __shared__ int start[128];    // offset of a cell's records in `sorted`
    __shared__ unsigned sorted[REC_CHUNK];
    __shared__ unsigned char scell[REC_CHUNK];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    i64* const table = (i64*)smem;  // bin_cells x RS
    // ---- which records are this workgroup's: the bins share the grid in proportion to their totals (a seeded pass puts half
    // of all records into the side table's bins), a bin's share divides the bin's list -- its regions one after the other --
    // into equal portions.  Every workgroup derives the same plan from the counts (grid x bins words, out of L2).
    __shared__ int btot[64];   // records per bin
    __shared__ int ubase[66];  // first workgroup of a bin; [64] = workgroups in use, [65] = records of the pass
    const int lane0 = threadIdx.x & 63;
    {
        const int b = (int)threadIdx.x >> 3, part = (int)threadIdx.x & 7;  // eight threads per bin
        int sum = 0;
        if (b < rec.nbins)
            for (int w = part; w < nregions; w += 8) sum += rec.counts[(size_t)w * rec.nbins + b];
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        sum += __shfl_xor(sum, 4, 64);
        if (b < 64 && part == 0) btot[b] = b < rec.nbins ? sum : 0;
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // wave 0: lane = bin
        const int tb = btot[lane0];
        long all = tb;
        for (int d = 32; d >= 1; d >>= 1) all += __shfl_xor(all, d, 64);
        const int nonempty = __builtin_popcountll(__ballot(tb > 0));
        const long spare = (long)gridDim.x - nonempty;  // (the launch has at least one workgroup per bin)
        int u = tb > 0 ? 1 + (int)((long)tb * spare / (all > 0 ? all : 1)) : 0;
        int v = u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(v, d, 64);
            if (lane0 >= d) v += o;
        }
        ubase[lane0] = v - u;
        if (lane0 == 63) ubase[64] = v;
        if (lane0 == 0) ubase[65] = (int)(all > 0x7fffffff ? 0x7fffffff : all);  // (published below)
    }
    __syncthreads();
    // (the host picks the next pass's accumulate by the number of records: few -> the burst of atomics is cheaper than this kernel)
    if (blockIdx.x == 0 && threadIdx.x == 0 && rec.total_out)
        __hip_atomic_store(rec.total_out, (long long)ubase[65], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)blockIdx.x >= ubase[64]) return;  // (workgroup-uniform)
    int r = 0;  // the last non-empty bin that starts at or before this workgroup
    for (int b = 0; b < rec.nbins; ++b)
        if (btot[b] > 0 && ubase[b] <= (int)blockIdx.x) r = b;
    const int nunits = (r == 63 ? ubase[64] : ubase[r + 1]) - ubase[r];
    const int portion = (int)blockIdx.x - ubase[r];
    const int per = (btot[r] + nunits - 1) / nunits;
    const int g_begin = portion * per, g_end = g_begin + per < btot[r] ? g_begin + per : btot[r];
    // prefix sums of the bin's counts over the regions (<= 256): four per lane of wave 0
    if (threadIdx.x < 64) {
        int c4[4], sum4 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int w = lane0 * 4 + k;
            c4[k] = w < nregions ? rec.counts[(size_t)w * rec.nbins + r] : 0;
            sum4 += c4[k];
        }
        int v = sum4;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(v, d, 64);
            if (lane0 >= d) v += o;
        }
        int acc = v - sum4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pre[lane0 * 4 + k] = acc;
            acc += c4[k];
        }
        if (lane0 == 63) pre[256] = acc;
    }
    __syncthreads();
    const int nreg = nregions;
    const int total = g_end - g_begin;
    if (total <= 0) return;  // (workgroup-uniform)
    for (int i = threadIdx.x; i < rec.bin_cells * RS; i += REC_TPB) table[i] = 0;  // (the first barrier of the chunk loop follows)
    const int sh_r = sc->sh_r;
    const bool fast_fix = sh_r >= -1000 && sh_r <= 1000;
    const double scale_r = __longlong_as_double((long long)(1023 + (fast_fix ? sh_r : 0)) << 52);
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // the bins behind the rows' belong to the family side table
    i64* const dst = r < rec.nbins_rows ? rows + (size_t)r * rec.bin_cells * RS : fam + (size_t)(r - rec.nbins_rows) * rec.bin_cells * RS;
    constexpr int WAVES = REC_TPB / 64;
    for (int c0 = 0; c0 < total; c0 += REC_CHUNK) {
        const int n = total - c0 < REC_CHUNK ? total - c0 : REC_CHUNK;
        if (threadIdx.x < 128) hist[threadIdx.x] = 0;
        __syncthreads();
        // ---- 1. the chunk's records: fetched (record g of the list lives in the region with pre[k] <= g < pre[k + 1]), ranked
        {
            uint2 rc[REC_PER_THREAD];
            int rank[REC_PER_THREAD];
#pragma unroll
            for (int k = 0; k < REC_PER_THREAD; ++k) {
                const int i = (int)threadIdx.x + k * REC_TPB;
                rc[k] = make_uint2(0u, 0u);
                rank[k] = 0;
                if (i < n) {
                    const int g = g_begin + c0 + i;
                    int lo = 0, hi = nreg - 1;  // (empty regions never match)
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (pre[mid] <= g) lo = mid; else hi = mid - 1;
                    }
                    rc[k] = rec.recs[((size_t)lo * rec.nbins + r) * (size_t)rec.cap + (size_t)(g - pre[lo])];
                }
            }
#pragma unroll
            for (int k = 0; k < REC_PER_THREAD; ++k)
                if ((int)threadIdx.x + k * REC_TPB < n)
                    rank[k] = __hip_atomic_fetch_add(&hist[rc[k].y & 0x7Fu], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __syncthreads();
            if (wib == 0) {  // exclusive scan of the 128 counts: two per lane
                const int a = hist[2 * lane], b = hist[2 * lane + 1];
                int v = a + b;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(v, d, 64);
                    if (lane >= d) v += o;
                }
                start[2 * lane] = v - a - b;
                start[2 * lane + 1] = v - b;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < REC_PER_THREAD; ++k)
                if ((int)threadIdx.x + k * REC_TPB < n) {
                    const int pos = start[rc[k].y & 0x7Fu] + rank[k];
                    sorted[pos] = rc[k].x | ((rc[k].y >> 16) << 31);
                    scell[pos] = (unsigned char)(rc[k].y & 0x7Fu);
                }
            __syncthreads();
        }
        // ---- 2. + 3. equal spans of the sorted chunk
        const int span = ((n + WAVES * REC_BATCH - 1) / (WAVES * REC_BATCH)) * REC_BATCH;
        const int j0 = wib * span, j1 = j0 + span < n ? j0 + span : n;
        if (j0 < j1) {
            i64 acc_hi = 0, acc_lo = 0;
            int acc_n = 0, cur = -1;
            auto flush = [&]() {
                if (cur >= 0) {
                    i64* row = table + cur * RS;
                    if (lane < NC) {
                        if (acc_hi != 0) atomicAdd((u64*)&row[2 * lane], (u64)acc_hi);
                        if (acc_lo != 0) atomicAdd((u64*)&row[2 * lane + 1], (u64)acc_lo);
                    } else if (lane == NC && acc_n != 0) {
                        atomicAdd((u64*)&row[2 * NC], (u64)(i64)acc_n);
                    }
                }
                acc_hi = 0;
                acc_lo = 0;
                acc_n = 0;
            };
            unsigned wA[REC_BATCH], wB[REC_BATCH];
            int cA[REC_BATCH], cB[REC_BATCH];
            double xA[REC_BATCH], xB[REC_BATCH];
            auto load = [&](int j, unsigned (&w)[REC_BATCH], int (&c)[REC_BATCH], double (&x)[REC_BATCH]) {
                const unsigned wv = sorted[j + (lane & 7)];
                const int cv = scell[j + (lane & 7)];
#pragma unroll
                for (int k = 0; k < REC_BATCH; ++k) {
                    w[k] = (unsigned)__builtin_amdgcn_readlane((int)wv, k);
                    c[k] = __builtin_amdgcn_readlane(cv, k);
                    x[k] = (j + k < j1 && lane < NC) ? aos[(long)(w[k] & 0x7FFFFFFFu) * NC + lane] : 0.0;
                }
            };
            load(j0, wA, cA, xA);
            for (int j = j0; j < j1; j += REC_BATCH) {
                if (j + REC_BATCH < j1) load(j + REC_BATCH, wB, cB, xB);
#pragma unroll
                for (int k = 0; k < REC_BATCH; ++k) {
                    if (j + k >= j1) break;  // wave-uniform
                    if (cA[k] != cur) {      // wave-uniform
                        flush();
                        cur = cA[k];
                    }
                    int h, l;
                    if (fast_fix)
                        fix2_mul(xA[k], scale_r, h, l);
                    else
                        fix2(xA[k], sh_r, h, l);
                    if (wA[k] >> 31) {
                        acc_hi -= (i64)h;
                        acc_lo -= (i64)l;
                        acc_n -= 1;
                    } else {
                        acc_hi += (i64)h;
                        acc_lo += (i64)l;
                        acc_n += 1;
                    }
                }
#pragma unroll
                for (int k = 0; k < REC_BATCH; ++k) {
                    wA[k] = wB[k];
                    cA[k] = cB[k];
                    xA[k] = xB[k];
                }
            }
            flush();
        }
        __syncthreads();  // (hist / sorted are rewritten by the next chunk)
    }
    for (int i = threadIdx.x; i < rec.bin_cells * RS; i += REC_TPB) {
        const i64 v = table[i];
        if (v != 0) atomicAdd((u64*)&dst[i], (u64)v);  // (cells beyond the codebook's last never receive a record: zero)
    }
}

// ---- launch wrappers ---------------------------------------------------------------------------------------
#ifdef E2VQ_PRE_STAMP
}  // namespace e2vq
extern "C" int e2vq_debug_pre_stamps(unsigned long long* out32, int reset)
{
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(e2vq::g_pre_stamps), 32 * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(e2vq::g_pre_stamps), z, 32 * 8) != hipSuccess) return 1;
    }
    return 0;
}
namespace e2vq {
#endif
static inline int pre_grid(long items, int per_block, int cap)
{
    long g = (items + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

static bool pre_has_nc(int NC) { return prefilter_supports(NC, 64); }

// fused quantize keeps the 64 row-major FP64 frames of every wave in LDS: eight waves for P <= 38, seven at P = 40
// (waves per workgroup: eight while their stages fit; P = 40 runs seven)
__host__ __device__ constexpr int fused_quantize_waves(int NC)
{
    return (E2VQ_LDS_BYTES - NC * 4) / (64 * NC * 8) >= 8 ? 8 : (E2VQ_LDS_BYTES - NC * 4) / (64 * NC * 8);
}
__host__ __device__ constexpr bool fused_quantize_fits(int NC) { return fused_quantize_waves(NC) >= 6; }
bool prefilter_fused_quantize(int NC) { return pre_has_nc(NC) && fused_quantize_fits(NC) && !getenv("ECOZ2_VQ_QUANTIZE_UNFUSED"); }

// the accumulating pass with LDS-staged frames (k_pass_pre_lds) needs (64 frames + 512 B) of LDS per wave: eight waves for P <= 38, seven at P = 40;
template <int NC> static constexpr bool lds_stage_ok_t() { return PreLds<NC>::OK; }
static int pre_lds_waves(int NC)
{
    switch (NC) {
#define X(N) case N: return PreLds<N>::WAVES;
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return 8;
    }
}
// the burst of atomics inside the sweep kernel (and the seeded first pass without records) serves rows of at most 80 elements
bool prefilter_burst_supported(int NC)
{
    switch (NC) {
#define X(N) case N: return PreLds<N>::BURST_OK;
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return false;
    }
}
bool prefilter_lds_stage(int NC)
{
    switch (NC) {
#define X(N) case N: return lds_stage_ok_t<N>();
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return false;
    }
}

// accumulate = false: assignment only.  prev_sym (optional): the cell of every frame is recorded there; with
// `incremental` the rows must be those of the previous pass over the same frames (distortion elements zeroed) and
// prev_sym its cells.  Runs after launch_prefilter_codebook of the same pass; afterwards
// *prefilter_fallback_count(ps) frames wait in fb_list for launch_pass_fallback.
template <int NC>
static int launch_pass_prefiltered_t(bool accumulate, const double* blk, long T, long nblocks, const void* fimg,
                                     const float* fg, const void* cimg, void* ps, const double* cbq, int M,
                                     const DevScalars* sc, const unsigned long long* l1max_bits, unsigned short* sym,
                                     double* dmin, long long* rows, int* fb_list, unsigned short* prev_sym, bool incremental,
                                     hipStream_t s, const double* aos, const int* ea_fused,
                                     const double* aos_resident, long long* family_table, const PassRecords* records)
{
    constexpr int IMG = 2 * NC + 5 + IMG_STRIDE_PAD;
    constexpr int TPBM = 512;  // 8 waves = 2 per SIMD, one persistent workgroup per CU
    const size_t lds = (size_t)(TPBM / 64) * 16 * IMG * 4;
    int bits = 0;
    while ((1 << bits) < M) ++bits;
    const int idxmask = ~((1 << bits) - 1);
    const int grid = pre_grid(nblocks, TPBM / 64, 256);
    const int stagger = 1;  // (partner waves of a SIMD half a block period apart)
    if (accumulate) {
        // (an accumulating pass needs the row-major resident copy, LDS room for a block of it, and -- for rows of more than
        // 80 elements -- the recorded accumulate: the caller sends anything else to the plain FP64 sweep)
        if (!(aos_resident && prefilter_lds_stage(NC) && (records || PreLds<NC>::BURST_OK))) return 1;
        // round 3: FP64 frames staged in LDS, lane-per-frame exact evaluation, one burst of atomics per block
        if constexpr (PreLds<NC>::OK) {
            const int MT = M / 32;
            PreRec rec{};
            if (records) {
                if (records->nbins > 64) return 1;
                rec.recs = (uint2*)records->recs;
                rec.counts = records->counts;
                rec.nbins = records->nbins;
                rec.nbins_rows = records->nbins_rows;
                rec.bin_cells = records->bin_cells;
                rec.cap = records->cap;
                rec.magic = records->magic;
            }
            constexpr int WAVES = PreLds<NC>::WAVES;
            const int grid_w = pre_grid(nblocks, WAVES, 256);
            if (records && records->grid != grid_w) return 1;  // (planned for another launch)
            auto go = [&](auto kernel) {
                (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, E2VQ_LDS_BYTES);
                hipLaunchKernelGGL(kernel, dim3(grid_w), dim3(WAVES * 64), (size_t)WAVES * PreLds<NC>::WAVE_BYTES + (records ? 256 : 0), s,
                                   aos_resident, T, nblocks, (const h8*)fimg, fg, (const h8*)cimg, (PreScalars*)ps, cbq, MT,
                                   idxmask, sc, (const u64*)l1max_bits, sym, dmin, rows, fb_list, stagger, prev_sym,
                                   family_table ? 2 : (incremental ? 1 : 0), (i64*)family_table, rec);
            };
            const bool rot = MT >= 4 && (MT & 1) == 0;
            // (two register sets of codeword granules: up to nine granules per tile -- P = 40 has eleven and would spill)
            constexpr bool ROT_OK = PrePack<NC>::NU <= 9;
            if (records) {  // (the cell sums follow in launch_reduce_records)
                if constexpr (ROT_OK) {
                    if (rot) {
                        go(k_pass_pre_lds<NC, true, 2>);
                        return 0;
                    }
                }
                go(k_pass_pre_lds<NC, false, 2>);
            } else if constexpr (PreLds<NC>::BURST_OK) {
                if constexpr (ROT_OK) {
                    if (rot) {
                        go(k_pass_pre_lds<NC, true, 1>);
                        return 0;
                    }
                }
                go(k_pass_pre_lds<NC, false, 1>);
            } else {
                return 1;
            }
        }
    } else if (ea_fused) {  // fused quantize: limb images built in the kernel from the row-major payload
        if constexpr (fused_quantize_fits(NC)) {
            constexpr int QT = fused_quantize_waves(NC) * 64;
            const size_t lds6 = (size_t)(QT / 64) * 64 * NC * 8 + (size_t)NC * sizeof(int);
            auto goq = [&](auto kernel) {
                (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, E2VQ_LDS_BYTES);
                hipLaunchKernelGGL(kernel, dim3(pre_grid(nblocks, QT / 64, 256)), dim3(QT), lds6, s, (const double*)nullptr, T, nblocks,
                                   (const h8*)nullptr, (const float*)nullptr, (const h8*)cimg, (PreScalars*)ps, cbq, M / 32,
                                   idxmask, sc, (const u64*)l1max_bits, sym, dmin, rows, fb_list, stagger, prev_sym, 0, aos,
                                   ea_fused);
            };
            // (round 6: the rotating tile loop from two tiles on, eight waves per workgroup; P = 40's seven-wave instantiation
            // keeps the plain loop)
            if constexpr (QT == 512) {
                if (M / 32 >= 2) {
                    goq(k_pass_pre<NC, 6, QT, true>);
                    return 0;
                }
            }
            goq(k_pass_pre<NC, 6, QT, false>);
        } else {
            return 1;
        }
    } else {
        (void)hipFuncSetAttribute((const void*)k_pass_pre<NC, 0, TPBM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  E2VQ_LDS_BYTES);
        hipLaunchKernelGGL((k_pass_pre<NC, 0, TPBM>), dim3(grid), dim3(TPBM), lds, s, blk, T, nblocks, (const h8*)fimg, fg,
                           (const h8*)cimg, (PreScalars*)ps, cbq, M / 32, idxmask, sc, (const u64*)l1max_bits, sym,
                           dmin, rows, fb_list, stagger, prev_sym, incremental ? 1 : 0, aos, (const int*)nullptr);
    }
    return 0;
}

int launch_pass_prefiltered(int NC, bool accumulate, const double* blk, long T, long nblocks, const void* fimg,
                            const float* fg, const void* cimg, void* ps, const double* cbq, int M,
                            const DevScalars* sc, const unsigned long long* l1max_bits, unsigned short* sym, double* dmin,
                            long long* rows, int* fb_list, unsigned short* prev_sym, bool incremental, hipStream_t s, const double* rowmajor_frames, const int* ea_fused, const double* resident_rowmajor,
                            long long* family_table, const PassRecords* records)
{
    if (records && (!accumulate || !resident_rowmajor || !prefilter_lds_stage(NC) || !prev_sym)) return 1;
    if (family_table && (!accumulate || incremental || !resident_rowmajor || !prefilter_lds_stage(NC))) return 1;
    if (!prefilter_supports(NC, M)) return 1;
    if (ea_fused && (accumulate || !rowmajor_frames || !prefilter_fused_quantize(NC))) return 1;
    switch (NC) {
#define X(N)                                                                                                          \
    case N:                                                                                                           \
        return launch_pass_prefiltered_t<N>(accumulate, blk, T, nblocks, fimg, fg, cimg, ps, cbq, M, sc, l1max_bits, sym, \
                                            dmin, rows, fb_list, prev_sym, incremental, s, rowmajor_frames, \
                                            ea_fused, resident_rowmajor, family_table, records);
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return 1;
    }
}

// ---- the recorded accumulate (k_pass_pre_lds<.., 2> + k_reduce_records) ------------------------------------------------------
bool prefilter_records_plan(int NC, int M, bool family, long nblocks, PassRecords* plan, size_t* recs_bytes)
{
    if (!pre_has_nc(NC) || !prefilter_lds_stage(NC) || NC >= 64 || !prefilter_supports(NC, M)) return false;
    if (nblocks * 64 >= (1L << 31)) return false;  // (frame numbers are 32-bit in a record)
    const int RS = (2 * NC + 5 + 7) & ~7;
    (void)RS;
    const int max_cells = (M + (family ? M / 2 : 0) + 2 * REC_BIN_CELLS) / REC_BIN_CELLS <= 64 ? REC_BIN_CELLS : REC_BIN_CELLS_BIG;
    const int nb_rows = (M + max_cells - 1) / max_cells;
    const int bc = (M + nb_rows - 1) / nb_rows;  // balanced bins
    const int fam_cells = family ? M / 2 : 0;
    const int nb_fam = (fam_cells + bc - 1) / bc;
    if (nb_rows + nb_fam > 64) return false;  // (a bin's count travels in lane `bin` of the recording wave)
    const unsigned magic = (unsigned)(((1u << 22) + bc - 1) / bc);
    const int ncells = (nb_rows + nb_fam) * bc;
    for (int c = 0; c < ncells; ++c)
        if ((int)(((unsigned)c * magic) >> 22) != c / bc) return false;
    const int waves = pre_lds_waves(NC);  // waves per sweeping workgroup
    const int grid = pre_grid(nblocks, waves, 256);
    // every frame of a recording workgroup twice in one bin ('+' and '-'); a workgroup of w <= 16 waves that deals blocks (or
    // half blocks) out wave by wave covers at most nblocks / grid + w blocks
    (void)waves;
    const long cap = 2 * 64 * ((nblocks + grid - 1) / grid + 16);
    if (cap >= (1L << 30)) return false;
    plan->grid = grid;
    plan->nbins = nb_rows + nb_fam;
    plan->nbins_rows = nb_rows;
    plan->bin_cells = bc;
    plan->magic = magic;
    plan->cap = (int)cap;
    if (recs_bytes) *recs_bytes = (size_t)grid * (size_t)plan->nbins * (size_t)cap * 8;
    return true;
}

// few: an incremental pass (a fraction of the frames recorded): one reducing workgroup per CU keeps the flush small;
// else two per CU
int launch_reduce_records(int NC, const double* aos, const PassRecords& plan, bool few, const DevScalars* sc, long long* rows,
                          long long* family_table, hipStream_t s)
{
    if (plan.nbins > plan.nbins_rows && !family_table) return 1;
    const int RS = (2 * NC + 5 + 7) & ~7;
    // (three workgroups per CU by registers and LDS; the kernel shares them out over the bins by their record counts)
    (void)few;
    const int nwg = 768 < plan.nbins ? plan.nbins : 768;
    if (plan.grid > 256 || plan.bin_cells > 128 || plan.nbins > 64) return 1;  // (the prefix array; the histogram; lane = bin)
    const size_t lds = (size_t)plan.bin_cells * RS * 8;
    PreRec rec{};
    rec.recs = (uint2*)plan.recs;
    rec.counts = plan.counts;
    rec.nbins = plan.nbins;
    rec.nbins_rows = plan.nbins_rows;
    rec.bin_cells = plan.bin_cells;
    rec.cap = plan.cap;
    rec.magic = plan.magic;
    rec.total_out = plan.total_out;
    switch (NC) {
#define X(N)                                                                                                          \
    case N:                                                                                                           \
        if constexpr (N < 64) {                                                                                       \
            (void)hipFuncSetAttribute((const void*)k_reduce_records<N>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                      E2VQ_LDS_BYTES - 32768);                                                        \
            hipLaunchKernelGGL((k_reduce_records<N>), dim3((unsigned)nwg), dim3(REC_TPB), lds, s, aos, rec, plan.grid, sc, \
                               (i64*)rows, (i64*)family_table);                                                       \
            return 0;                                                                                                 \
        }                                                                                                             \
        return 1;
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return 1;
    }
}

}  // namespace e2vq
