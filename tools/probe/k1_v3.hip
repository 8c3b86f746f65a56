// K1 v3 probe: codebook tile in LDS, lane-distributed codeword registers, v_fmac_f64_dpp row_newbcast.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
constexpr int NC = 37, NPAD = 40;

template<int N> __device__ __forceinline__ void fmac_bc(double& acc, double c, double r) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(r), "i"(N));
}

template<int G> struct CW { double2 a[G]; double b[G]; };

// one chain step n for all G codewords
template<int n, int G> __device__ __forceinline__ void step(double (&d)[G], const CW<G>& c, double r) {
    #pragma unroll
    for (int g = 0; g < G; ++g) {
        if constexpr (n < 32) {
            if constexpr ((n & 1) == 0) fmac_bc<(n >> 1)>(d[g], c.a[g].x, r);
            else fmac_bc<(n >> 1)>(d[g], c.a[g].y, r);
        } else {
            fmac_bc<n - 32>(d[g], c.b[g], r);
        }
    }
}
template<int n, int G> struct Chain {
    static __device__ __forceinline__ void run(double (&d)[G], const CW<G>& c, const double (&r)[NC]) {
        step<n, G>(d, c, r[n]);
        if constexpr (n + 1 < NC) Chain<n + 1, G>::run(d, c, r);
    }
};

template<int G, int TPB, int WPS, int MT>
__global__ __launch_bounds__(TPB, WPS)
void k1(const double* __restrict__ frames, const double* __restrict__ cb, int M, long nblocks,
        unsigned short* __restrict__ sym, double* __restrict__ dmin)
{
    __shared__ __attribute__((aligned(16))) double tile[MT * NPAD + 16];
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const long wave = (long)blockIdx.x * (TPB >> 6) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (TPB >> 6);
    const long niter = (nblocks + nwaves - 1) / nwaves;
    for (long it = 0; it < niter; ++it) {
        long b = wave + it * nwaves;
        const bool active = b < nblocks;
        if (!active) b = nblocks - 1;
        double r[NC];
        const double* fb = frames + b * (long)(NC * 64);
        #pragma unroll
        for (int n = 0; n < NC; ++n) r[n] = fb[n * 64 + lane];
        double best = __builtin_inf(); int bi = 0;
        for (int m0 = 0; m0 < M; m0 += MT) {
            __syncthreads();
            for (int i = threadIdx.x * 2; i < MT * NPAD; i += TPB * 2)
                *(double2*)&tile[i] = *(const double2*)&cb[(long)m0 * NPAD + i];
            __syncthreads();
            for (int m = 0; m < MT; m += G) {
                CW<G> c;
                #pragma unroll
                for (int g = 0; g < G; ++g) {
                    c.a[g] = *(const double2*)&tile[(m + g) * NPAD + 2 * j];
                    c.b[g] = tile[(m + g) * NPAD + 32 + j];
                }
                double d[G];
                #pragma unroll
                for (int g = 0; g < G; ++g) d[g] = -0.0;
                Chain<0, G>::run(d, c, r);
                #pragma unroll
                for (int g = 0; g < G; ++g) { bool lt = d[g] < best; best = lt ? d[g] : best; bi = lt ? (m0 + m + g) : bi; }
            }
        }
        if (active) { sym[b * 64 + lane] = (unsigned short)bi; dmin[b * 64 + lane] = best; }
    }
}

static std::vector<double> g_frames_h, g_cb_h;
template<int G, int TPB, int WPS, int MT> void run(const double* d_frames, const double* d_cb, int M, long T, unsigned short* d_sym, double* d_dmin, int wg, bool check) {
    long nblocks = T / 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best_ms = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k1<G,TPB,WPS,MT>), dim3(wg), dim3(TPB), 0, 0, d_frames, d_cb, M, nblocks, d_sym, d_dmin);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_ms) best_ms = ms;
    }
    CK(hipGetLastError());
    double flops = 2.0 * 37 * M * (double)(nblocks * 64);
    printf("G=%d grid=%dx%d wps=%d MT=%d : %.3f ms  %.2f TFLOP/s", G, wg, TPB, WPS, MT, best_ms, flops / best_ms * 1e-9);
    if (check) {
        int nchk = 192; std::vector<unsigned short> s(nchk); std::vector<double> dm(nchk);
        CK(hipMemcpy(s.data(), d_sym, nchk * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(dm.data(), d_dmin, nchk * 8, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < nchk; ++i) {
            long b = i / 64; int within = i % 64;
            double bestv = INFINITY; int bidx = 0;
            for (int m = 0; m < M; ++m) {
                double acc = 0;
                for (int n = 0; n < NC; ++n) {
                    double rv = g_frames_h[(size_t)b * NC * 64 + (size_t)n * 64 + within];
                    double c = g_cb_h[(size_t)m * NPAD + n];
                    acc = (n == 0) ? rv * c : fma(rv, c, acc);
                }
                if (acc < bestv) { bestv = acc; bidx = m; }
            }
            if (bidx != s[i] || bestv != dm[i]) ++bad;
        }
        printf("  check: %d/%d mismatches", bad, nchk);
    }
    printf("\n"); fflush(stdout);
}
int main() {
    const long T = 1L << 21; const int M = 1024;
    g_frames_h.resize((size_t)T * 37); g_cb_h.assign((size_t)M * NPAD + 64, 0.0);
    srand(1);
    for (auto& x : g_frames_h) x = (rand() / (double)RAND_MAX) * 2 - 1;
    for (int m = 0; m < M; ++m) for (int n = 0; n < 37; ++n) g_cb_h[(size_t)m * NPAD + n] = (rand() / (double)RAND_MAX) * 2 - 1;
    double *d_frames, *d_cb, *d_dmin; unsigned short* d_sym;
    CK(hipMalloc(&d_frames, g_frames_h.size() * 8)); CK(hipMalloc(&d_cb, g_cb_h.size() * 8)); CK(hipMalloc(&d_dmin, T * 8)); CK(hipMalloc(&d_sym, T * 2));
    CK(hipMemcpy(d_frames, g_frames_h.data(), g_frames_h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cb, g_cb_h.data(), g_cb_h.size() * 8, hipMemcpyHostToDevice));
    run<4,512,4,128>(d_frames, d_cb, M, T, d_sym, d_dmin, 512, true);    // 4 waves/SIMD: 2 WG/CU x 8 waves
    run<4,512,4,128>(d_frames, d_cb, M, T, d_sym, d_dmin, 1024, false);
    run<4,512,4,128>(d_frames, d_cb, M, T, d_sym, d_dmin, 2048, false);
    run<2,512,4,128>(d_frames, d_cb, M, T, d_sym, d_dmin, 512, true);
    run<2,512,4,128>(d_frames, d_cb, M, T, d_sym, d_dmin, 2048, false);
    run<4,256,4,128>(d_frames, d_cb, M, T, d_sym, d_dmin, 1024, true);   // 4 WG/CU x 4 waves
    run<4,256,4,128>(d_frames, d_cb, M, T, d_sym, d_dmin, 4096, false);
    run<4,1024,4,128>(d_frames, d_cb, M, T, d_sym, d_dmin, 256, true);   // 1 WG/CU x 16 waves
    run<4,1024,4,128>(d_frames, d_cb, M, T, d_sym, d_dmin, 512, false);
    run<8,512,2,128>(d_frames, d_cb, M, T, d_sym, d_dmin, 512, true);
    run<8,512,2,128>(d_frames, d_cb, M, T, d_sym, d_dmin, 2048, false);
    run<4,512,4,256>(d_frames, d_cb, M, T, d_sym, d_dmin, 512, false);
    run<4,512,4,64>(d_frames, d_cb, M, T, d_sym, d_dmin, 512, false);
    return 0;
}
