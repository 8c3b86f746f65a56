// K1 ablation probe (diagnostic only): which part of the sweep limits FP64 throughput?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef const double __attribute__((address_space(4))) cdouble_k;
constexpr int NC = 37, NPAD = 40;

// MODE 0: baseline  1: no argmin (min only)  2: same codeword every iter (K$-hot)  3: barrier every 8 codewords
// MODE 4: codeword via LDS broadcast reads
template<int F, int TPB, int MODE>
__global__ __launch_bounds__(TPB)
void k1(const double* __restrict__ frames, const double* cb_, int M, long nblocks,
        unsigned short* __restrict__ sym, double* __restrict__ dmin)
{
    cdouble_k* cb = (cdouble_k*)cb_;
    __shared__ double lds_cb[MODE == 4 ? 256 * NPAD : 1];
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (TPB >> 6) + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * (TPB >> 6);
    const long niter = (nblocks + nwaves - 1) / nwaves;
    for (long it = 0; it < niter; ++it) {
        long b = wave + it * nwaves;
        const bool active = b < nblocks;
        if (!active) b = nblocks - 1;
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
        if constexpr (MODE == 4) {
            for (int m0 = 0; m0 < M; m0 += 256) {
                __syncthreads();
                for (int i = threadIdx.x; i < 256 * NPAD; i += TPB) lds_cb[i] = cb_[(long)m0 * NPAD + i];
                __syncthreads();
                for (int m = 0; m < 256; ++m) {
                    const double* c = lds_cb + m * NPAD;
                    double d[F];
                    #pragma unroll
                    for (int f = 0; f < F; ++f) d[f] = r[f][0] * c[0];
                    #pragma unroll
                    for (int n = 1; n < NC; ++n)
                        #pragma unroll
                        for (int f = 0; f < F; ++f) d[f] = __builtin_fma(r[f][n], c[n], d[f]);
                    #pragma unroll
                    for (int f = 0; f < F; ++f) { bool lt = d[f] < best[f]; best[f] = lt ? d[f] : best[f]; bi[f] = lt ? (m0 + m) : bi[f]; }
                }
            }
        } else {
            double cs[NC];
            if constexpr (MODE == 5) {
                #pragma unroll
                for (int n = 0; n < NC; ++n) cs[n] = cb[n];
            }
            for (int m = 0; m < M; ++m) {
                if constexpr (MODE == 5) {
                    #pragma unroll
                    for (int n = 0; n < NC; ++n) asm volatile("" : "+s"(cs[n]));
                }
                cdouble_k* c0_ = cb + (MODE == 2 ? (long)(m & 1) : (MODE == 5 ? 0L : (long)m)) * NPAD;
                if constexpr (MODE == 3) { if ((m & 7) == 0) __builtin_amdgcn_s_barrier(); }
                double d[F];
                #pragma unroll
                for (int f = 0; f < F; ++f) d[f] = r[f][0] * (MODE == 5 ? cs[0] : c0_[0]);
                #pragma unroll
                for (int n = 1; n < NC; ++n)
                    #pragma unroll
                    for (int f = 0; f < F; ++f) d[f] = __builtin_fma(r[f][n], (MODE == 5 ? cs[n] : c0_[n]), d[f]);
                #pragma unroll
                for (int f = 0; f < F; ++f) {
                    if constexpr (MODE == 1) { best[f] = __builtin_fmin(best[f], d[f]); }
                    else { bool lt = d[f] < best[f]; best[f] = lt ? d[f] : best[f]; bi[f] = lt ? m : bi[f]; }
                }
            }
        }
        if (active) {
            #pragma unroll
            for (int f = 0; f < F; ++f) {
                sym[(b * 64 + lane) * F + f] = (unsigned short)bi[f];
                dmin[(b * 64 + lane) * F + f] = best[f];
            }
        }
    }
}

template<int F, int TPB, int MODE> void run(const double* d_frames, const double* d_cb, int M, long T, unsigned short* d_sym, double* d_dmin, int wg) {
    long nblocks = T / (64 * F);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best_ms = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k1<F,TPB,MODE>), dim3(wg), dim3(TPB), 0, 0, d_frames, d_cb, M, nblocks, d_sym, d_dmin);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best_ms) best_ms = ms;
    }
    double flops = 2.0 * 37 * M * (double)(nblocks * 64 * F);
    printf("MODE=%d F=%d grid=%dx%d : %.3f ms  %.2f TFLOP/s\n", MODE, F, wg, TPB, best_ms, flops / best_ms * 1e-9);
    fflush(stdout);
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
#define ALLM(F,TPB,WG) run<F,TPB,5>(d_frames,d_cb,M,T,d_sym,d_dmin,WG); run<F,TPB,0>(d_frames,d_cb,M,T,d_sym,d_dmin,WG); run<F,TPB,1>(d_frames,d_cb,M,T,d_sym,d_dmin,WG); run<F,TPB,2>(d_frames,d_cb,M,T,d_sym,d_dmin,WG); run<F,TPB,3>(d_frames,d_cb,M,T,d_sym,d_dmin,WG); run<F,TPB,4>(d_frames,d_cb,M,T,d_sym,d_dmin,WG);
    ALLM(1,256,1024)   // 4 waves/SIMD balanced
    ALLM(2,256,512)    // 2 waves/SIMD balanced
    ALLM(2,256,768)
    return 0;
}
