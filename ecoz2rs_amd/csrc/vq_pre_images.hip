// vq_pre_images.hip -- the operand images of the prefiltered sweep (vq_prefilter.hip, vq_sweep.hip): the data statistic and
// the per-coefficient exponents, the block-major f16 limb image of the frames (k_pre_frames), quantize's preparation pass
// (k_pre_quant_prep) and its scales from the codebook, the codebook's scale (k_pre_cmax) and limb image in unique granules
// (k_pre_codebook), and their launch wrappers.  The limb arithmetic itself (pre_split, PrePack) is vq_pre_common.h.
// (Split from vq_prefilter.hip in round 5: that file keeps the passes.)
#include "vq_pre_common.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

namespace e2vq {

// ---- data statistic: per-coefficient max |r[n]| over the blocked training set -----------------------------
__global__ void k_pre_colmax(const double* __restrict__ blk, long nblocks, int NC, u64* __restrict__ colmax_bits)
{
    __shared__ u64 smax[E2VQ_MAX_P + 1];
    for (int i = threadIdx.x; i < NC; i += blockDim.x) smax[i] = 0;
    __syncthreads();
    const int NS = (NC + 3) >> 2;
    const long total = nblocks * (long)NC * 64;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int w = (int)(o % ((long)NC * 64));
        const int x = w % (NC * 32);
        const int n = x < (NS - 1) * 128 ? 4 * (x >> 7) + ((x & 127) >> 5) : 4 * (NS - 1) + ((x - (NS - 1) * 128) >> 5);
        const u64 bits = (u64)__double_as_longlong(fabs(blk[o]));
        if (bits > smax[n]) atomicMax(&smax[n], bits);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NC; i += blockDim.x)
        if (smax[i]) atomicMax(&colmax_bits[i], smax[i]);
}

// a_n = 2^ea[n] > max |r[n]|   (ea = 0 for an all-zero coefficient)
__global__ void k_pre_exponents(const u64* __restrict__ colmax_bits, int NC, int* __restrict__ ea)
{
    const int n = threadIdx.x;
    if (n < NC) {
        const double m = __longlong_as_double((i64)colmax_bits[n]);
        ea[n] = m > 0.0 ? ilogb(m) + 1 : 0;
    }
}

// ---- frame image: [blk32][pair][h*32 + col][8 halves], fg[t] = sum_n |xi_n| (rounded up) -----------------
template <int NC>
__global__ __launch_bounds__(64) void k_pre_frames(const double* __restrict__ blk, long T, long nblk32,
                                                   const int* __restrict__ ea, h8* __restrict__ fimg,
                                                   float* __restrict__ fg)
{
    typedef PrePack<NC> PK;
    __shared__ short X[3][32][PK::NCX];
    const int col = threadIdx.x & 31, hh = threadIdx.x >> 5;
    for (long b = blockIdx.x; b < nblk32; b += gridDim.x) {
        const long t = b * 32 + col;
        // frame scale: A_t = 2^eA > max_n |r[n]| 2^-ea[n]  (both halves of the workgroup compute it; cheap)
        int eA = -100000;
        if (t < T)
            for (int n = 0; n < NC; ++n) {
                const double v = blk[mfma_blk_offset(NC, t, n)];
                if (v != 0.0) {
                    const int e = ilogb(v) - ea[n] + 1;
                    eA = e > eA ? e : eA;
                }
            }
        if (eA == -100000) eA = 0;
        double g = 0.0;
        for (int n = hh; n < PK::NCX; n += 2) {
            int L[3] = {0, 0, 0};
            if (n < NC && t < T) {
                const double xi = ldexp(blk[mfma_blk_offset(NC, t, n)], -ea[n] - eA);
                g += fabs(xi);
                pre_split(xi, L);
            }
            X[0][col][n] = (short)L[0];
            X[1][col][n] = (short)L[1];
            X[2][col][n] = (short)L[2];
        }
        g += __shfl_xor(g, 32, 64);
        if (hh == 0 && t < T) fg[t] = (float)g * 1.000001f;
        __syncthreads();
        for (int p = 0; p < PK::PAIRS; ++p) {
            h8 out;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int fl, n;
                PK::slot(p, hh, e, fl, n);
                out[e] = n >= 0 ? (_Float16)(int)X[fl][col][n] : (_Float16)0;
            }
            fimg[(b * PK::PAIRS + p) * 64 + threadIdx.x] = out;
        }
        __syncthreads();
    }
}

// ---- quantize: no data statistic is at hand, so the per-coefficient scales come from the codebook ---------
// a_n = 2^ea[n] with ea[n] = -(ilogb(max_m |c[m][n]|) + 1): every eta = c a is in (-1, 1) with C = 1, and the
// frame scale A_t absorbs whatever range r / a has.  (Any powers of two keep the limb arithmetic exact and the
// bound scale-free; the choice only moves how tight the bound is.)
__global__ void k_pre_ea_from_codebook(const double* __restrict__ cbq, int M, int NC, int NPAD, int* __restrict__ ea)
{
    __shared__ u64 smax[E2VQ_MAX_P + 1];
    for (int i = threadIdx.x; i < NC; i += blockDim.x) smax[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < M * NC; i += blockDim.x) {
        const int m = i / NC, n = i - m * NC;
        const u64 bits = (u64)__double_as_longlong(fabs(cbq[(long)m * NPAD + n]));
        if (bits > smax[n]) atomicMax(&smax[n], bits);
    }
    __syncthreads();
    for (int n = threadIdx.x; n < NC; n += blockDim.x) {
        const double mx = __longlong_as_double((i64)smax[n]);
        ea[n] = mx > 0.0 ? -(ilogb(mx) + 1) : 0;
    }
}

// row-major frames [t][NC] -> (a) the blocked FP64 MFMA layout of k_blockify_mfma, (b) the f16 limb image,
// (c) the tolerance terms, one 64-frame block per workgroup, every global access coalesced through LDS.
// Thread = (frame f = tid & 63, coefficient group tid >> 6): the four waves split the coefficients of every frame for
// the exponent and limb passes, and all index arithmetic of the two output layouts is resolved at compile time per
// k-step / granule pair (round 2: 0.57 -> 0.3 ms per 2^21 frames; the kernel was bound by its instruction count).
template <int NC>
__global__ __launch_bounds__(256) void k_pre_quant_prep(const double* __restrict__ aos, long T, long nblocks,
                                                        const int* __restrict__ ea, double* __restrict__ blk,
                                                        h8* __restrict__ fimg, float* __restrict__ fg)
{
    constexpr int NS = (NC + 3) / 4, REM = NC - 4 * (NS - 1);
    typedef PrePack<NC> PK;
    __shared__ double stage[64 * NC];
    __shared__ short X[3][64][PK::NCX];
    __shared__ int eAs[64];
    __shared__ double gp[4][64];
    __shared__ int eas[NC];
    const int tid = threadIdx.x, f = tid & 63, grp = tid >> 6;
    for (int n = tid; n < NC; n += 256) eas[n] = ea[n];
    for (long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        __syncthreads();  // the previous block's readers of stage / X / eAs are done
        const long base = b * 64 * NC, total = T * NC;
        for (int i = tid; i < 64 * NC; i += 256) stage[i] = base + i < total ? aos[base + i] : 0.0;
        if (tid < 64) eAs[tid] = -100000;
        __syncthreads();
        // (a) blocked FP64 layout (k_blockify_mfma): per 32-frame half u and k-step st, 64 lanes x 2 doubles with
        //     value r[u*32 + 16 h + j][4 st + q], lane = 16 q + j.  Threads 0..127 serve u = 0, the others u = 1.
        if (blk) {  // (nullptr: the sweep reads the FP64 frames from the row-major payload itself)
            const int u = tid >> 7, y = tid & 127, l = y >> 1, h = y & 1, j = l & 15, q = l >> 4;
            const double* src = stage + (u * 32 + h * 16 + j) * NC;
            double* dst = blk + base + u * (NC * 32) + y;
#pragma unroll
            for (int st = 0; st < NS - 1; ++st) dst[st * 128] = src[4 * st + q];
            if (y < REM * 32) {  // last k-step: REM coefficients, [q < REM][j][h]
                const int z = y >> 1, jj = z & 15, qq = z >> 4;
                blk[base + u * (NC * 32) + (NS - 1) * 128 + y] = stage[(u * 32 + (y & 1) * 16 + jj) * NC + 4 * (NS - 1) + qq];
            }
        }
        // frame scale A_t = 2^eA: max over the coefficients, each wave its share, combined by an LDS integer max
        {
            int eA = -100000;
            for (int n = grp; n < NC; n += 4) {
                const double x = stage[f * NC + n];
                if (x != 0.0) {
                    const int e = ilogb(x) - eas[n] + 1;
                    eA = e > eA ? e : eA;
                }
            }
            if (eA != -100000) atomicMax(&eAs[f], eA);
        }
        __syncthreads();
        {
            int eA = eAs[f];
            if (eA == -100000) eA = 0;
            double g = 0.0;
            for (int n = grp; n < PK::NCX; n += 4) {
                int L[3] = {0, 0, 0};
                if (n < NC) {
                    const double xi = ldexp(stage[f * NC + n], -eas[n] - eA);
                    g += fabs(xi);
                    pre_split(xi, L);
                }
                X[0][f][n] = (short)L[0];
                X[1][f][n] = (short)L[1];
                X[2][f][n] = (short)L[2];
            }
            gp[grp][f] = g;
        }
        __syncthreads();
        if (tid < 64) {  // tolerance term: sum |xi| (any order: it is rounded up)
            const long t = b * 64 + tid;
            if (t < T) fg[t] = (float)(((gp[0][tid] + gp[1][tid]) + gp[2][tid]) + gp[3][tid]) * 1.000001f;
        }
        // (b) limb image: granule (pair p, lane l = 32 h + col) of column block cb; two pairs per step, p compile-time
        {
            const int half = tid >> 7, cb = (tid >> 6) & 1, l = tid & 63, hh = l >> 5, fr = 32 * cb + (l & 31);
            auto emit = [&](auto pc) {
                constexpr int p = decltype(pc)::value;
                h8 out;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    int fl, n;
                    PK::slot(p, hh, e, fl, n);
                    out[e] = n >= 0 ? (_Float16)(int)X[fl][fr][n] : (_Float16)0;
                }
                fimg[((b * 2 + cb) * PK::PAIRS + p) * 64 + l] = out;
            };
            pre_for_pairs<PK::PAIRS>(half, emit);
        }
    }
}

// ---- codebook scale: eC = max ilogb(c a) + 1 over the codebook -------------------------------------------
__global__ void k_pre_cmax(const double* __restrict__ cbq, int M, int NC, int NPAD, const int* __restrict__ ea,
                           PreScalars* __restrict__ ps)
{
    __shared__ int smax;
    if (threadIdx.x == 0) smax = 0;
    __syncthreads();
    int mx = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M * NC; i += gridDim.x * blockDim.x) {
        const int m = i / NC, n = i - m * NC;
        const double c = cbq[(long)m * NPAD + n];
        if (c != 0.0) {
            const int e = ilogb(c) + ea[n] + 1 + PRE_EBIAS;
            mx = e > mx ? e : mx;
        }
    }
    if (mx) atomicMax(&smax, mx);
    __syncthreads();
    if (threadIdx.x == 0 && smax) atomicMax(&ps->eC_biased, smax);
}

// ---- codebook image: [tile][unique granule][h*32 + row][8 halves], then the per-tile table [tile] (c_t, d_t) --------------
// Round 6: every TILE of 32 codewords takes its limbs at its own scale.  With eC_t = max ilogb(c a) + 1 over the tile's codewords
// and s_t = min(eC - eC_t, 8) >= 0, the limbs are split from eta' = c a 2^-(eC - s_t) -- 2^s_t times finer than the global
// scale allows where the tile's codewords are small -- and STORED multiplied by 2^-s_t: an f16 holds L 2^-s exactly (|L| <= 512,
// s <= 8), the limb products and their f32 sums are the same integers on a 2^-s grid, so every accumulator, and with it the
// key, comes out in the units of the global scale with no change to any sweep kernel -- but with an error of
//     |2^36 sum xi eta - key|  <=  2^-s_t 2^8 (g + y'_t + NC + 4)  +  rho |key|,      y'_t = max_m sum_n |eta'_mn|
// (the proven bound of DESIGN 4.2 applied to eta').  Since 2^-s_t y'_t = sum |eta| <= ymax, the old bound 2^8 (g + ymax + NC
// + 4) holds for every tile as before: rules that do not look at the table (the two-stage sweep's coarse rule, k_finish)
// stay valid unchanged.  The table gives c_t = 2^-s_t and d_t = 2^-s_t (y'_t + NC + 4), both rounded
// up: a kernel that knows which tile a frame's smallest key came from certifies with 2^8 (c_t g + d_t) for that key and the
// old bound for all the others (k_pass_pre_lds, fused quantize, the fused sorted pass) -- on data whose distortions are small differences of large
// terms the typical tile sits 2-3 bits below the global scale (tools/probe/key_precision_model.py).
template <int NC>
__global__ __launch_bounds__(256) void k_pre_codebook(const double* __restrict__ cbq, int M, int NPAD,
                                                      const int* __restrict__ ea, PreScalars* __restrict__ ps,
                                                      h8* __restrict__ cimg)
{
    typedef PrePack<NC> PK;
    __shared__ short Y[3][32][PK::NCX];
    __shared__ int s_et;
    __shared__ float s_y[32];
    const int tile = blockIdx.x;
    const int MT = (M + 31) / 32;
    const int eC = ps->eC_biased ? ps->eC_biased - PRE_EBIAS : 0;
    if (threadIdx.x == 0) s_et = -100000;
    __syncthreads();
    // the tile's own exponent; every thread keeps the codeword elements it will split (one trip to memory, not two)
    constexpr int KPT = (32 * PK::NCX + 255) / 256;
    double cv[KPT];
    int en[KPT];
    {
        int mx = -100000;
#pragma unroll
        for (int k = 0; k < KPT; ++k) {
            const int i = threadIdx.x + 256 * k;
            const int row = i / PK::NCX, n = i - row * PK::NCX;
            const int m = tile * 32 + row;
            const bool live = i < 32 * PK::NCX && n < NC && m < M;
            cv[k] = live ? cbq[(long)m * NPAD + n] : 0.0;
            en[k] = live ? ea[n] : 0;
            if (cv[k] != 0.0) {
                const int e = ilogb(cv[k]) + en[k] + 1;
                mx = e > mx ? e : mx;
            }
        }
        for (int d = 32; d >= 1; d >>= 1) {
            const int o = __shfl_xor(mx, d, 64);
            mx = o > mx ? o : mx;
        }
        if ((threadIdx.x & 63) == 0 && mx > -100000) atomicMax(&s_et, mx);
    }
    __syncthreads();
    int st = s_et > -100000 ? eC - s_et : 8;  // (an all-zero tile: any scale)
    st = st < 0 ? 0 : (st > 8 ? 8 : st);
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < 32 * PK::NCX) {
            const int row = i / PK::NCX, n = i - row * PK::NCX;
            int L[3] = {0, 0, 0};
            if (cv[k] != 0.0) pre_split(ldexp(cv[k], en[k] - eC + st), L);
            Y[0][row][n] = (short)L[0];
            Y[1][row][n] = (short)L[1];
            Y[2][row][n] = (short)L[2];
        }
    }
    if (threadIdx.x < 32) {  // sum_n |eta| of this tile's codewords -> global max (float bits, rounded up); y'_t at the tile's scale
        const int m = tile * 32 + threadIdx.x;
        double g = 0.0;
        if (m < M)
            for (int n = 0; n < NC; ++n) g += fabs(ldexp(cbq[(long)m * NPAD + n], ea[n] - eC));
        atomicMax(&ps->ymax_bits, __float_as_int((float)g * 1.000001f));
        s_y[threadIdx.x] = (float)ldexp(g, st) * 1.000001f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float y = 0.f;
        for (int r = 0; r < 32; ++r) y = s_y[r] > y ? s_y[r] : y;
        const float c = __int_as_float((127 - st) << 23);  // 2^-s_t
        float2* tab = (float2*)(cimg + (size_t)MT * PK::TILE_E);
        tab[tile] = make_float2(c, c * (y + (float)(NC + 4)) * 1.000001f);
    }
    const float scale = __int_as_float((127 - st) << 23);
    for (int i = threadIdx.x; i < PK::TILE_E; i += 256) {
        h8 out = {0, 0, 0, 0, 0, 0, 0, 0};
        const int u = i >> 6, l = i & 63, hh = l >> 5, row = l & 31;  // unique granule u (PrePack::step_unique)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int cl, n;
            PK::unique_slot(u, hh, e, cl, n);
            if (n >= 0) out[e] = (_Float16)((float)(int)Y[cl][row][n] * scale);  // (exact: |L| <= 512, a power of two >= 2^-8)
        }
        cimg[(long)tile * PK::TILE_E + i] = out;
    }
}

// ---- launch wrappers ---------------------------------------------------------------------------------------
static inline int pre_grid(long items, int per_block, int cap)
{
    long g = (items + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

static bool pre_has_nc(int NC)
{
    switch (NC) {
#define X(N) case N:
        E2VQ_PRE_NC_LIST(X)
#undef X
        return true;
        default: return false;
    }
}
template <int NC> static size_t frame_image_bytes_t(long nb) { return (size_t)nb * 2 * PrePack<NC>::PAIRS * 64 * 16; }
// (the tiles' limb images, then the per-tile table of k_pre_codebook: 8 bytes per tile)
template <int NC> static size_t codebook_image_bytes_t(int M) { return (size_t)((M + 31) / 32) * (PrePack<NC>::TILE_E * 16 + 8); }

// (the codeword index shares the f32 key with the value: at M = 8192 nine mantissa bits are left for the value and 6 % of the
// frames of the bench data go to the FP64 fallback sweep -- still 2.6 x the plain sweep's rate, profiles/r04_big_codebooks.txt;
// one bit less would hand over most frames)
bool prefilter_supports(int NC, int M) { return pre_has_nc(NC) && M >= 64 && M % 32 == 0 && M <= 8192; }
size_t prefilter_frame_image_bytes(int NC, long nblocks64)
{
    switch (NC) {
#define X(N) case N: return frame_image_bytes_t<N>(nblocks64);
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return 0;
    }
}
size_t prefilter_codebook_image_bytes(int NC, int M)
{
    switch (NC) {
#define X(N) case N: return codebook_image_bytes_t<N>(M);
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: return 0;
    }
}
size_t prefilter_scalars_bytes() { return sizeof(PreScalars); }

void launch_prefilter_frames(const double* blk, long T, long nblocks64, int NC, unsigned long long* colmax_bits, int* ea,
                             void* fimg, float* fg, hipStream_t s)
{
    (void)hipMemsetAsync(colmax_bits, 0, (size_t)NC * 8, s);
    hipLaunchKernelGGL(k_pre_colmax, dim3(pre_grid(nblocks64 * NC * 64, 256 * 8, 2048)), dim3(256), 0, s, blk, nblocks64,
                       NC, (u64*)colmax_bits);
    hipLaunchKernelGGL(k_pre_exponents, dim3(1), dim3(256), 0, s, (const u64*)colmax_bits, NC, ea);
    switch (NC) {
#define X(N)                                                                                                        \
    case N:                                                                                                         \
        hipLaunchKernelGGL((k_pre_frames<N>), dim3(pre_grid(nblocks64 * 2, 1, 16384)), dim3(64), 0, s, blk, T,      \
                           nblocks64 * 2, (const int*)ea, (h8*)fimg, fg);                                           \
        break;
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: break;
    }
}

void launch_prefilter_quantize_prep(const double* aos, long T, long nblocks64, int NC, const int* ea, double* blk,
                                    void* fimg, float* fg, hipStream_t s)
{
    switch (NC) {
#define X(N)                                                                                                        \
    case N:                                                                                                         \
        hipLaunchKernelGGL((k_pre_quant_prep<N>), dim3(pre_grid(nblocks64, 1, 4096)), dim3(256), 0, s, aos, T,      \
                           nblocks64, ea, blk, (h8*)fimg, fg);                                          \
        break;
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: break;
    }
}

// the per-coefficient scales of quantize alone (fused quantize needs no preparation pass over the frames)
void launch_prefilter_quantize_scales(const double* cbq, int M, int NC, int* ea, hipStream_t s)
{
    hipLaunchKernelGGL(k_pre_ea_from_codebook, dim3(1), dim3(1024), 0, s, cbq, M, NC, (NC + 7) & ~7, ea);
}

const int* prefilter_fallback_count(const void* ps) { return &((const PreScalars*)ps)->fb_count; }
int* prefilter_codebook_scale(void* ps) { return &((PreScalars*)ps)->eC_biased; }

// zeroes the per-pass scalars (fallback count included) and builds the limb image of the current codebook
void launch_prefilter_codebook(const double* cbq, int M, int NC, const int* ea, void* ps, void* cimg, hipStream_t s,
                               bool scale_ready)
{
    const int NPAD = (NC + 7) & ~7;
    if (!scale_ready) {
        (void)hipMemsetAsync(ps, 0, sizeof(PreScalars), s);
        hipLaunchKernelGGL(k_pre_cmax, dim3(pre_grid((long)M * NC, 1024, 32)), dim3(256), 0, s, cbq, M, NC, NPAD, ea,
                           (PreScalars*)ps);
    }
    switch (NC) {
#define X(N)                                                                                                        \
    case N:                                                                                                         \
        hipLaunchKernelGGL((k_pre_codebook<N>), dim3((M + 31) / 32), dim3(256), 0, s, cbq, M, NPAD, ea,             \
                           (PreScalars*)ps, (h8*)cimg);                                                             \
        break;
        E2VQ_PRE_NC_LIST(X)
#undef X
        default: break;
    }
}

}  // namespace e2vq
