// K1 v2 probe: F frames/lane x G codewords interleaved (independent chains), chunk-blocked codebook.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef const double __attribute__((address_space(4))) cdouble_k;
constexpr int NC = 37, NQ = 5;  // 5 chunks of 8 coefficients (40 padded)

// codebook layout: cb[mg][q][g][8], mg = m / G, g = m % G
template<int F, int G, int TPB, int WPS>
__global__ __launch_bounds__(TPB, WPS)
void k1(const double* __restrict__ frames, const double* cb_, int M, long nblocks,
        unsigned short* __restrict__ sym, double* __restrict__ dmin)
{
    cdouble_k* cb = (cdouble_k*)cb_;
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (TPB >> 6) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (TPB >> 6);
    for (long b = wave; b < nblocks; b += nwaves) {
        double r[F][NC];
        const double* fb = frames + b * (long)(NC * 64 * F);
        #pragma unroll
        for (int n = 0; n < NC; ++n)
            #pragma unroll
            for (int f = 0; f < F; ++f)
                r[f][n] = fb[(long)n * 64 * F + lane * F + f];
        double best[F]; int bi[F];
        #pragma unroll
        for (int f = 0; f < F; ++f) { best[f] = __builtin_inf(); bi[f] = 0; }
        for (int mg = 0; mg < M / G; ++mg) {
            cdouble_k* cg = cb + (long)mg * (NQ * G * 8);
            double d[F][G];
            #pragma unroll
            for (int q = 0; q < NQ; ++q) {
                #pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int n = q * 8 + j;
                    if (n < NC) {
                        #pragma unroll
                        for (int g = 0; g < G; ++g) {
                            const double c = cg[(q * G + g) * 8 + j];
                            #pragma unroll
                            for (int f = 0; f < F; ++f)
                                d[f][g] = (n == 0) ? r[f][0] * c : __builtin_fma(r[f][n], c, d[f][g]);
                        }
                    }
                }
            }
            #pragma unroll
            for (int g = 0; g < G; ++g)
                #pragma unroll
                for (int f = 0; f < F; ++f) { bool lt = d[f][g] < best[f]; best[f] = lt ? d[f][g] : best[f]; bi[f] = lt ? (mg * G + g) : bi[f]; }
        }
        #pragma unroll
        for (int f = 0; f < F; ++f) {
            sym[(b * 64 + lane) * F + f] = (unsigned short)bi[f];
            dmin[(b * 64 + lane) * F + f] = best[f];
        }
    }
}

static std::vector<double> g_frames_h, g_cb_h;
template<int F, int G, int TPB, int WPS> void run(const double* d_frames, int M, long T, unsigned short* d_sym, double* d_dmin, int wg, bool check) {
    long nblocks = T / (64 * F);
    // build blocked codebook
    std::vector<double> cbb((size_t)(M / G) * NQ * G * 8, 0.0);
    for (int m = 0; m < M; ++m) for (int n = 0; n < NC; ++n)
        cbb[((size_t)(m / G) * NQ + n / 8) * G * 8 + (m % G) * 8 + (n % 8)] = g_cb_h[(size_t)m * 40 + n];
    double* d_cb; CK(hipMalloc(&d_cb, cbb.size() * 8)); CK(hipMemcpy(d_cb, cbb.data(), cbb.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best_ms = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k1<F,G,TPB,WPS>), dim3(wg), dim3(TPB), 0, 0, d_frames, d_cb, M, nblocks, d_sym, d_dmin);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_ms) best_ms = ms;
    }
    double flops = 2.0 * 37 * M * (double)(nblocks * 64 * F);
    printf("F=%d G=%d grid=%dx%d wps=%d : %.3f ms  %.2f TFLOP/s", F, G, wg, TPB, WPS, best_ms, flops / best_ms * 1e-9);
    if (check) {
        // verify first 64*F*2 frames against host chain
        int nchk = 128 * F; std::vector<unsigned short> s(nchk); std::vector<double> dm(nchk);
        CK(hipMemcpy(s.data(), d_sym, nchk * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(dm.data(), d_dmin, nchk * 8, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < nchk; ++i) {
            long b = i / (64 * F); int within = i % (64 * F);
            double bestv = INFINITY; int bidx = 0;
            for (int m = 0; m < M; ++m) {
                double acc = 0;
                for (int n = 0; n < NC; ++n) {
                    double rv = g_frames_h[(size_t)b * NC * 64 * F + (size_t)n * 64 * F + within];
                    double c = g_cb_h[(size_t)m * 40 + n];
                    acc = (n == 0) ? rv * c : fma(rv, c, acc);
                }
                if (acc < bestv) { bestv = acc; bidx = m; }
            }
            if (bidx != s[i] || bestv != dm[i]) ++bad;
        }
        printf("  check: %d/%d mismatches", bad, nchk);
    }
    printf("\n"); fflush(stdout);
    CK(hipFree(d_cb));
}
int main() {
    const long T = 1L << 21; const int M = 1024;
    g_frames_h.resize((size_t)T * 37); g_cb_h.assign((size_t)M * 40, 0.0);
    srand(1);
    for (auto& x : g_frames_h) x = (rand() / (double)RAND_MAX) * 2 - 1;
    for (int m = 0; m < M; ++m) for (int n = 0; n < 37; ++n) g_cb_h[(size_t)m * 40 + n] = (rand() / (double)RAND_MAX) * 2 - 1;
    double *d_frames, *d_dmin; unsigned short* d_sym;
    CK(hipMalloc(&d_frames, g_frames_h.size() * 8)); CK(hipMalloc(&d_dmin, T * 8)); CK(hipMalloc(&d_sym, T * 2));
    CK(hipMemcpy(d_frames, g_frames_h.data(), g_frames_h.size() * 8, hipMemcpyHostToDevice));
    run<1,1,256,5>(d_frames, M, T, d_sym, d_dmin, 2048, true);
    run<1,2,256,5>(d_frames, M, T, d_sym, d_dmin, 2048, true);
    run<1,4,256,5>(d_frames, M, T, d_sym, d_dmin, 2048, true);
    run<1,4,256,4>(d_frames, M, T, d_sym, d_dmin, 2048, false);
    run<1,4,256,4>(d_frames, M, T, d_sym, d_dmin, 1024, false);
    run<1,4,256,5>(d_frames, M, T, d_sym, d_dmin, 1280, false);
    run<1,4,512,4>(d_frames, M, T, d_sym, d_dmin, 512, false);
    run<1,4,1024,4>(d_frames, M, T, d_sym, d_dmin, 256, false);
    run<1,8,256,4>(d_frames, M, T, d_sym, d_dmin, 2048, true);
    run<1,8,256,4>(d_frames, M, T, d_sym, d_dmin, 1024, false);
    run<2,2,256,3>(d_frames, M, T, d_sym, d_dmin, 2048, true);
    run<2,2,256,3>(d_frames, M, T, d_sym, d_dmin, 768, false);
    run<2,4,256,2>(d_frames, M, T, d_sym, d_dmin, 2048, true);
    run<2,4,256,2>(d_frames, M, T, d_sym, d_dmin, 512, false);
    run<2,4,256,3>(d_frames, M, T, d_sym, d_dmin, 768, false);
    return 0;
}
