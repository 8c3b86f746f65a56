// pre_sweep.hip -- probe: exact three-limb fixed-point prefilter sweep on the f16 matrix pipe.
//
// Each coefficient is split into three signed integer limbs (|X1| <= 512, |X2|,|X3| <= 256) held as f16; the
// three weight classes W0 = sum X1*Y1, W1 = sum X1*Y2 + X2*Y1, W2 = sum X1*Y3 + X2*Y2 + X3*Y1 accumulate in
// separate f32 accumulators and stay below 2^24, so every partial sum is an exactly representable integer and
// the MFMA result is exact whatever the hardware's summation order.  v = W0*2^18 + W1*2^9 + W2 approximates
// the distortion to ~2^-27 of the operand scales; the top three keys (value | codeword index) per frame come out.
//
// Question: cycles per 32x32 tile (15 MFMAs + 96 VALU ops per 32 frames) and the clock held on random data.
//   hipcc --offload-arch=gfx950 -O3 pre_sweep.hip -o pre_sweep && ./pre_sweep [log2 frames] [M]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));          \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

constexpr int NC = 37;
constexpr int NSTEP = 15;  // 3 + 5 + 7 MFMA k-steps of 16
#ifndef CT_
#define CT_ 2
#endif
#ifndef WPB
#define WPB 8
#endif
constexpr int CT = CT_;    // codeword tiles per LDS chunk
constexpr int TPB = 64 * WPB;

// step -> (level, frame granule pair)
__host__ __device__ constexpr int step_level(int s) { return s < 3 ? 0 : (s < 8 ? 1 : 2); }
__host__ __device__ constexpr int step_pair(int s)
{
    // w0: p0 p1 p6 | w1: p0 p1 p2 p3 p6 | w2: p0 p1 p2 p3 p4 p5 p6
    return s < 3 ? (s == 2 ? 6 : s) : (s < 8 ? (s == 7 ? 6 : s - 3) : (s - 8));
}

__device__ __forceinline__ int med3i(int a, int b, int c)
{
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ int mini(int a, int b)
{
    int d;
    asm("v_min_i32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ __forceinline__ float med3f(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
// epilogue ablations (EPI != 0 gives wrong results; they price the instructions beside the MFMA stream)
#ifndef EPI
#define EPI 0
#endif
#if EPI == 4
#define EPI_V(PREV) const float v = r == 0 ? PREV[0][r] + PREV[1][r] + PREV[2][r] : PREV[r % 3][r]; (void)sidx;
#else
#define EPI_V(PREV) const float v = __builtin_fmaf(PREV[0][r], 262144.f, __builtin_fmaf(PREV[1][r], 512.f, PREV[2][r])); \
                    const float key = __int_as_float((__float_as_int(v) & maskv) | sidx);
#endif
#if EPI == 0   // product: three v_med3_f32
#define EPI_UPDATE(PCB) k3[PCB] = med3f(k2[PCB], k3[PCB], key); k2[PCB] = med3f(k1[PCB], k2[PCB], key); k1[PCB] = med3f(k1[PCB], key, ninf);
#elif EPI == 1 // min only
#define EPI_UPDATE(PCB) k1[PCB] = __builtin_fminf(k1[PCB], key);
#elif EPI == 3 // VOP2 min/max insertion network (5 ops)
#define EPI_UPDATE(PCB) { const float u_ = __builtin_fmaxf(k1[PCB], key); k1[PCB] = __builtin_fminf(k1[PCB], key); const float w_ = __builtin_fmaxf(k2[PCB], u_); k2[PCB] = __builtin_fminf(k2[PCB], u_); k3[PCB] = __builtin_fminf(k3[PCB], w_); }
#elif EPI == 4 // (with EPI_SKIP) nearly nothing: one min per register
#define EPI_UPDATE(PCB) k1[PCB] = __builtin_fminf(k1[PCB], v);
#elif EPI == 5 // top-2 only (2 med3)
#define EPI_UPDATE(PCB) k2[PCB] = med3f(k1[PCB], k2[PCB], key); k1[PCB] = med3f(k1[PCB], key, ninf);
#endif

// keys are compared as floats (v_min/v_med3_f32): positive keys order like their bit patterns
__global__ __launch_bounds__(TPB) void k_pre_sweep(const h8* __restrict__ fimg, long nblk32, const h8* __restrict__ cimg,
                                                   int MT, int idxmask, int4* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    h8* lds = (h8*)smem;  // [4 slots][NSTEP][64]
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const long b64 = (long)blockIdx.x * WPB + wib;

    h8 B[2][7];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const long blk = b64 * 2 + cb;
#pragma unroll
        for (int p = 0; p < 7; ++p) {
            h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            B[cb][p] = blk < nblk32 ? fimg[(blk * 7 + p) * 64 + lane] : z;
        }
    }
    int maskv = idxmask;
    asm volatile("" : "+v"(maskv));
    float ninf = -__builtin_inff();
    asm volatile("" : "+v"(ninf));
    float k1[2], k2[2], k3[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) k1[cb] = k2[cb] = k3[cb] = __int_as_float(0x7f7fffff);

    constexpr int TILE_E = 1024;                        // h8 elements per codeword tile (15 KB padded to 16 KB)
    constexpr int PER_T = TILE_E / TPB;
    h8 pre[PER_T];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int e = i * TPB + threadIdx.x;
            pre[i] = cimg[(long)t * TILE_E + e];
        }
    };
    auto lstore = [&](int slot) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int e = i * TPB + threadIdx.x;
            lds[slot * TILE_E + e] = pre[i];
        }
    };
    gload(0);
    lstore(0);
    gload(1);
    lstore(1);
    __syncthreads();

    f16v acc0[3], acc1[3];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[l][r] = l == 2 ? 3.0e38f : 0.f;

    // one job = the 15 MFMAs of (tile, column block) interleaved with the key epilogue of the previous job
#define JOB(ACC, BC, PREV, PTILE, PCB)                                                                              \
    {                                                                                                               \
        const f16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                                          \
        _Pragma("unroll") for (int s = 0; s < NSTEP; ++s)                                                           \
        {                                                                                                           \
            const int lv = step_level(s), pr = step_pair(s);                                                        \
            const bool first = s == 0 || s == 3 || s == 8;                                                          \
            ACC[lv] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s], BC[pr], first ? zero : ACC[lv], 0, 0, 0);        \
        }                                                                                                           \
        _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                              \
        {                                                                                                           \
            const int sidx = __builtin_amdgcn_readfirstlane((PTILE) * 32 + 8 * (r >> 2) + (r & 3));                 \
            EPI_V(PREV)                                                                                             \
            EPI_UPDATE(PCB)                                                                                         \
        }                                                                                                           \
        _Pragma("unroll") for (int s = 0; s < NSTEP; ++s)                                                           \
        {                                                                                                           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                      \
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                                                      \
        }                                                                                                           \
    }

    for (int t = 0; t < MT; ++t) {
#ifndef NO_REFILL
        gload(t + 2 < MT ? t + 2 : MT - 1);
#endif
        const h8* cur = lds + (t & 3) * TILE_E;
        h8 A[NSTEP];
#ifdef A_RESIDENT  // ablation: no operand traffic at all (wrong results), the bare MFMA + epilogue loop
        (void)cur;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            A[s] = B[0][s % 7];
            asm volatile("" : "+v"(A[s]));
        }
#elif defined(DIRECT_L2)
        (void)cur;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) A[s] = cimg[(long)t * TILE_E + s * 64 + lane];
#else
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) A[s] = cur[s * 64 + lane];
#endif
        JOB(acc0, B[0], acc1, (t - 1) & 0xffff, 1)
        JOB(acc1, B[1], acc0, t, 0)
#ifndef NO_REFILL
        lstore((t + 2) & 3);
        __syncthreads();
#endif
    }
    {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int sidx = __builtin_amdgcn_readfirstlane((MT - 1) * 32 + 8 * (r >> 2) + (r & 3));
            const float v = __builtin_fmaf(acc1[0][r], 262144.f, __builtin_fmaf(acc1[1][r], 512.f, acc1[2][r]));
            const float key = __int_as_float((__float_as_int(v) & maskv) | sidx);
            k3[1] = med3f(k2[1], k3[1], key);
            k2[1] = med3f(k1[1], k2[1], key);
            k1[1] = med3f(k1[1], key, ninf);
        }
    }

    // merge the two lane halves (rows 4h..4h+3 of every 8) of a column; the half bit goes into the index
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int hb = (lane >> 5) << 2;
        const float a1 = __int_as_float(__float_as_int(k1[cb]) | hb), a2 = __int_as_float(__float_as_int(k2[cb]) | hb),
                    a3 = __int_as_float(__float_as_int(k3[cb]) | hb);
        const float b1 = __shfl_xor(a1, 32, 64), b2 = __shfl_xor(a2, 32, 64), b3 = __shfl_xor(a3, 32, 64);
        // top three of {a1<=a2<=a3} U {b1<=b2<=b3}
        const float t3 = med3f(a2, a3, b1), t2 = med3f(a1, a2, b1), t1 = med3f(a1, b1, ninf);
        const float u3 = med3f(t2, t3, b2), u2 = med3f(t1, t2, b2);
        const float w3 = med3f(u2, u3, b3);
        if ((lane >> 5) == cb) {
            const long t = (b64 * 2 + cb) * 32 + (lane & 31);
            if (b64 * 2 + cb < nblk32) out[t] = make_int4(__float_as_int(t1), __float_as_int(u2), __float_as_int(w3), 0);
        }
    }
}

// Variant for occupancy: one 32-frame column block per wave (28 VGPRs of limbs), single-buffered accumulators, A
// operands loaded just in time from L2: ~110 VGPRs, 4 waves per SIMD; the epilogue of a job is not overlapped inside
// the wave but by the other waves.  Costs twice the L2 traffic of the 64-frame-per-wave kernel.
__global__ __launch_bounds__(256, 4) void k_pre_sweep_occ(const h8* __restrict__ fimg, long nblk32,
                                                          const h8* __restrict__ cimg, int MT, int idxmask,
                                                          int4* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    int maskv = idxmask;
    asm volatile("" : "+v"(maskv));
    float ninf = -__builtin_inff();
    asm volatile("" : "+v"(ninf));
    for (long blk = wave; blk < nblk32; blk += nwaves) {
        h8 B[7];
#pragma unroll
        for (int p = 0; p < 7; ++p) B[p] = fimg[(blk * 7 + p) * 64 + lane];
        float k1 = __int_as_float(0x7f7fffff), k2 = k1, k3 = k1;
        for (int t = 0; t < MT; ++t) {
            f16v acc[3];
            const f16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const h8 a = cimg[(long)t * 1024 + s * 64 + lane];
                const int lv = step_level(s), pr = step_pair(s);
                const bool first = s == 0 || s == 3 || s == 8;
                acc[lv] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B[pr], first ? zero : acc[lv], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int sidx = __builtin_amdgcn_readfirstlane(t * 32 + 8 * (r >> 2) + (r & 3));
                const float v = __builtin_fmaf(acc[0][r], 262144.f, __builtin_fmaf(acc[1][r], 512.f, acc[2][r]));
                const float key = __int_as_float((__float_as_int(v) & maskv) | sidx);
                k3 = med3f(k2, k3, key);
                k2 = med3f(k1, k2, key);
                k1 = med3f(k1, key, ninf);
            }
        }
        const int hb = (lane >> 5) << 2;
        const float a1 = __int_as_float(__float_as_int(k1) | hb), a2 = __int_as_float(__float_as_int(k2) | hb),
                    a3 = __int_as_float(__float_as_int(k3) | hb);
        const float b1 = __shfl_xor(a1, 32, 64), b2 = __shfl_xor(a2, 32, 64), b3 = __shfl_xor(a3, 32, 64);
        const float t3 = med3f(a2, a3, b1), t2 = med3f(a1, a2, b1), t1 = med3f(a1, b1, ninf);
        const float u3 = med3f(t2, t3, b2), u2 = med3f(t1, t2, b2);
        const float w3 = med3f(u2, u3, b3);
        if (lane < 32) out[blk * 32 + lane] = make_int4(__float_as_int(t1), __float_as_int(u2), __float_as_int(w3), 0);
    }
}

// ---------------------------------------------------------------------------------------------
static inline unsigned long long sm64(unsigned long long& s)
{
    unsigned long long z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 21;
    const int M = argc > 2 ? atoi(argv[2]) : 1024;
    const long T = 1L << lg;
    const long nblk32 = T / 32;
    const int MT = M / 32;
    unsigned long long seed = 1234;
    // limbs: [3][T][NC], [3][M][NC]
    std::vector<short> X((size_t)3 * T * NC), Y((size_t)3 * M * NC);
    auto fill = [&](std::vector<short>& v, size_t n) {
        for (int l = 0; l < 3; ++l)
            for (size_t i = 0; i < n; ++i) {
                const int lim = l == 0 ? 512 : 256;
#ifndef DATA
#define DATA 0
#endif
                // DATA (round 4: how much of the power-limited clock is the operands' bit activity?): 0 = the product's limbs
                // (X1, Y1 >= 0: keys positive; the lower limbs signed), 1 = every limb unsigned, 2 = zeros, 3 = small unsigned
                // values (0..15), 4 = signed, lower limbs sign-magnitude-sorted (no effect on the encoding: control)
                if (DATA == 2) { v[l * n + i] = 0; (void)sm64(seed); continue; }
                if (DATA == 3) { v[l * n + i] = (short)(sm64(seed) % 16); continue; }
                if (DATA == 1) { v[l * n + i] = (short)(sm64(seed) % (lim + 1)); continue; }
                v[l * n + i] = l == 0 ? (short)(sm64(seed) % (lim + 1)) : (short)((long)(sm64(seed) % (2 * lim + 1)) - lim);  // X1, Y1 >= 0: keys positive
            }
    };
    fill(X, (size_t)T * NC);
    fill(Y, (size_t)M * NC);
    auto XL = [&](int l, long t, int n) -> _Float16 { return n < NC ? (_Float16)X[(size_t)l * T * NC + t * NC + n] : (_Float16)0; };
    auto YL = [&](int l, int m, int n) -> _Float16 { return n < NC ? (_Float16)Y[(size_t)l * M * NC + (size_t)m * NC + n] : (_Float16)0; };

    // frame image: [blk32][p<7][h*32+col][8]
    std::vector<_Float16> fimg((size_t)nblk32 * 7 * 64 * 8);
    for (long b = 0; b < nblk32; ++b)
        for (int p = 0; p < 7; ++p)
            for (int h = 0; h < 2; ++h)
                for (int col = 0; col < 32; ++col) {
                    const long t = b * 32 + col;
                    _Float16* g = &fimg[(((size_t)b * 7 + p) * 64 + h * 32 + col) * 8];
                    for (int e = 0; e < 8; ++e) {
                        if (p < 6)
                            g[e] = XL(p >> 1, t, 16 * (p & 1) + 8 * h + e);
                        else if (h == 0)
                            g[e] = e < 5 ? XL(0, t, 32 + e) : XL(1, t, 32 + e - 5);
                        else
                            g[e] = e < 2 ? XL(1, t, 35 + e) : (e < 7 ? XL(2, t, 32 + e - 2) : (_Float16)0);
                    }
                }
    // codeword image: [tile][s<15][h*32+row][8]; per step the codeword limb that meets frame pair p's limb
    std::vector<_Float16> cimg((size_t)MT * 1024 * 8, (_Float16)0);
    for (int tile = 0; tile < MT; ++tile)
        for (int s = 0; s < NSTEP; ++s) {
            const int lv = step_level(s), pr = step_pair(s);
            for (int h = 0; h < 2; ++h)
                for (int row = 0; row < 32; ++row) {
                    const int m = tile * 32 + row;
                    _Float16* g = &cimg[((size_t)tile * 1024 + (size_t)s * 64 + h * 32 + row) * 8];
                    for (int e = 0; e < 8; ++e) {
                        _Float16 v = 0;
                        if (pr < 6) {
                            const int fl = pr >> 1;           // frame limb of this pair
                            const int cl = lv - fl;           // codeword limb with fl + cl = level
                            if (cl >= 0 && cl <= 2) v = YL(cl, m, 16 * (pr & 1) + 8 * h + e);
                        } else {
                            // tail pair: Ta = [X1[32..36], X2[32..34]], Tb = [X2[35..36], X3[32..36], 0]
                            int fl, n;
                            if (h == 0) { fl = e < 5 ? 0 : 1; n = e < 5 ? 32 + e : 32 + e - 5; }
                            else { fl = e < 2 ? 1 : 2; n = e < 2 ? 35 + e : 32 + e - 2; if (e == 7) n = NC; }
                            const int cl = lv - fl;
                            if (cl >= 0 && cl <= 2 && n < NC) v = YL(cl, m, n);
                        }
                        g[e] = v;
                    }
                }
        }
    h8 *d_f, *d_c;
    int4* d_o;
    CK(hipMalloc(&d_f, fimg.size() * 2));
    CK(hipMalloc(&d_c, cimg.size() * 2));
    CK(hipMalloc(&d_o, (size_t)T * 16));
    CK(hipMemcpy(d_f, fimg.data(), fimg.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_c, cimg.data(), cimg.size() * 2, hipMemcpyHostToDevice));
    int bits = 0;
    while ((1 << bits) < M) ++bits;
    const int idxmask = ~((1 << bits) - 1);
#ifdef LDS_KB  // occupancy knob: a workgroup that asks for > 80 KB is alone on its CU (WPB=4: one wave per SIMD)
    const size_t lds = (size_t)LDS_KB * 1024;
#else
    const size_t lds = (size_t)4 * 1024 * 16;
#endif
    CK(hipFuncSetAttribute((const void*)k_pre_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = (int)((nblk32 / 2 + WPB - 1) / WPB);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        const int R = rep == 0 ? 1 : 10;
#ifdef OCC
        for (int i = 0; i < R; ++i) k_pre_sweep_occ<<<OCC, 256>>>(d_f, nblk32, d_c, MT, idxmask, d_o);
#else
        for (int i = 0; i < R; ++i) k_pre_sweep<<<grid, TPB, lds>>>(d_f, nblk32, d_c, MT, idxmask, d_o);
#endif
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= R;
        const double tiles = (double)nblk32 * MT;
        printf("T=2^%d M=%d: %.3f ms/launch  %.1f M frames/s  %.0f cyc/tile@2.4GHz/SIMD  f16 %.2f PFLOP/s\n", lg, M, ms,
               T / ms / 1e3, ms * 1e-3 * 2.4e9 * 1024 / tiles, tiles * NSTEP * 32768.0 / (ms * 1e-3) / 1e15);
    }
    // verify some frames
    std::vector<int> o((size_t)T * 4);
    CK(hipMemcpy(o.data(), d_o, (size_t)T * 16, hipMemcpyDeviceToHost));
    long bad = 0;
    const long step = T / 257;
    for (long t = 0; t < T; t += step) {
        int best[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
        for (int m = 0; m < M; ++m) {
            long W[3] = {0, 0, 0};
            for (int n = 0; n < NC; ++n)
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; a + b < 3; ++b)
                        W[a + b] += (long)X[(size_t)a * T * NC + t * NC + n] * Y[(size_t)b * M * NC + (size_t)m * NC + n];
            const float v = fmaf((float)W[0], 262144.f, fmaf((float)W[1], 512.f, (float)W[2]));
            int key;
            memcpy(&key, &v, 4);
            key = (key & idxmask) | m;
            for (int k = 0; k < 3; ++k)
                if (key < best[k]) { const int tmp = best[k]; best[k] = key; key = tmp; }
        }
        for (int k = 0; k < 3; ++k)
            if (best[k] != o[t * 4 + k]) {
                if (bad < 5) printf("mismatch t=%ld k=%d: cpu %08x gpu %08x\n", t, k, best[k], o[t * 4 + k]);
                ++bad;
            }
    }
    printf("verify: %ld mismatches over %ld sampled frames (random limbs -> keys may be negative floats; int order)\n", bad,
           (T + step - 1) / step);
    return bad != 0;
}
