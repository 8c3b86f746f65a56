#include <hip/hip_runtime.h>
#include <stdint.h>
typedef const double __attribute__((address_space(4))) cdouble_k;

template<int NC, int NPAD, int F, int TPB>
__global__ __launch_bounds__(TPB)
void k1(const double* __restrict__ frames, const double* cb_, int M, long nblocks,
        unsigned short* __restrict__ sym, double* __restrict__ dmin)
{
    cdouble_k* cb = (cdouble_k*)cb_;
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
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
        for (int m = 0; m < M; ++m) {
            cdouble_k* c = cb + (long)m * NPAD;
            double d[F];
            #pragma unroll
            for (int f = 0; f < F; ++f) d[f] = r[f][0] * c[0];
            #pragma unroll
            for (int n = 1; n < NC; ++n)
                #pragma unroll
                for (int f = 0; f < F; ++f) d[f] = __builtin_fma(r[f][n], c[n], d[f]);
            #pragma unroll
            for (int f = 0; f < F; ++f) { bool lt = d[f] < best[f]; best[f] = lt ? d[f] : best[f]; bi[f] = lt ? m : bi[f]; }
        }
        #pragma unroll
        for (int f = 0; f < F; ++f) {
            sym[(b * 64 + lane) * F + f] = (unsigned short)bi[f];
            dmin[(b * 64 + lane) * F + f] = best[f];
        }
    }
}

#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template<int F, int TPB> void run(const double* d_frames, const double* d_cb, int M, long T, unsigned short* d_sym, double* d_dmin, int wg) { const int threads = TPB;
    long nblocks = T / (64 * F);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k1<37,40,F,TPB>), dim3(wg), dim3(TPB), 0, 0, d_frames, d_cb, M, nblocks, d_sym, d_dmin);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double flops = 2.0 * 37 * M * (double)T;
        printf("F=%d M=%d T=%ld grid=%dx%d : %.3f ms  %.2f Gframes/s  %.2f TFLOP/s (fp64)\n", F, M, T, wg, threads, ms, T / ms * 1e-6, flops / ms * 1e-9);
    }
}
int main() {
    const long T = 1L << 21; const int M = 1024;
    std::vector<double> h((size_t)T * 37), cb((size_t)M * 40, 0.0);
    srand(1);
    for (auto& x : h) x = (rand() / (double)RAND_MAX) * 2 - 1;
    for (int m = 0; m < M; ++m) for (int n = 0; n < 37; ++n) cb[(size_t)m * 40 + n] = (rand() / (double)RAND_MAX) * 2 - 1;
    double *d_frames, *d_cb, *d_dmin; unsigned short* d_sym;
    CK(hipMalloc(&d_frames, h.size() * 8)); CK(hipMalloc(&d_cb, cb.size() * 8)); CK(hipMalloc(&d_dmin, T * 8)); CK(hipMalloc(&d_sym, T * 2));
    CK(hipMemcpy(d_frames, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cb, cb.data(), cb.size() * 8, hipMemcpyHostToDevice));
    for (int wg : {256, 512, 1024, 2048}) {
        run<1,256>(d_frames, d_cb, M, T, d_sym, d_dmin, wg);
        run<1,512>(d_frames, d_cb, M, T, d_sym, d_dmin, wg);
        run<2,256>(d_frames, d_cb, M, T, d_sym, d_dmin, wg);
        run<2,512>(d_frames, d_cb, M, T, d_sym, d_dmin, wg);
        run<3,256>(d_frames, d_cb, M, T, d_sym, d_dmin, wg);
    }
    run<1,1024>(d_frames, d_cb, M, T, d_sym, d_dmin, 256);
    run<1,1024>(d_frames, d_cb, M, T, d_sym, d_dmin, 512);
    for (int m : {2, 16, 64, 256}) { run<1,256>(d_frames, d_cb, m, T, d_sym, d_dmin, 2048); run<2,256>(d_frames, d_cb, m, T, d_sym, d_dmin, 2048); }
    return 0;
}
