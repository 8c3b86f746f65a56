// What does the argmin (v_cmp_f64 + v_cndmask) cost next to a 37-FMA chain?  (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
constexpr int NR = 37;
// MODE 0: chain only (sum into best)   1: + v_cmp + 3 cndmask (standard)   2: deferred: compare previous d while chaining current
// MODE 3: v_min_f64 only                4: cmp + cndmask idx + v_min value     5: standard but 2 codewords per iteration
template<int MODE>
__global__ __launch_bounds__(256) void k(double* out, const double* in, int iters) {
    double r[NR];
    #pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = in[threadIdx.x + 64 * i];
    double s[NR];
    #pragma unroll
    for (int i = 0; i < NR; ++i) s[i] = in[i + 7];          // uniform -> SGPRs
    double best = 1e300; int bi = 0; double dprev = 1e300;
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int n = 0; n < NR; ++n) asm volatile("" : "+s"(s[n]));
        double d = r[0] * s[0];
        #pragma unroll
        for (int n = 1; n < NR; ++n) d = __builtin_fma(r[n], s[n], d);
        if constexpr (MODE == 0) { best += d; }
        else if constexpr (MODE == 1) { bool lt = d < best; best = lt ? d : best; bi = lt ? it : bi; }
        else if constexpr (MODE == 2) { bool lt = dprev < best; best = lt ? dprev : best; bi = lt ? it : bi; dprev = d; }
        else if constexpr (MODE == 3) { best = __builtin_fmin(best, d); }
        else if constexpr (MODE == 4) { bool lt = d < best; bi = lt ? it : bi; best = __builtin_fmin(best, d); }
        else if constexpr (MODE == 5) {
            #pragma unroll
            for (int n = 0; n < NR; ++n) asm volatile("" : "+s"(s[n]));
            double d2 = r[0] * s[0];
            #pragma unroll
            for (int n = 1; n < NR; ++n) d2 = __builtin_fma(r[n], s[n], d2);
            bool lt = d < best; best = lt ? d : best; bi = lt ? it : bi;
            bool lt2 = d2 < best; best = lt2 ? d2 : best; bi = lt2 ? it + 1 : bi;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = best + bi + dprev;
}
template<int MODE> void run(int wg, double* d_out, double* d_in) {
    const int iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE>), dim3(wg), dim3(256), 0, 0, d_out, d_in, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double flops = 2.0 * NR * (MODE == 5 ? 2 : 1) * (double)iters * wg * 256;
    printf("MODE=%d wg=%d : %.3f ms %.2f TFLOP/s\n", MODE, wg, best, flops / best * 1e-9); fflush(stdout);
}
int main() {
    double *d_out, *d_in; CK(hipMalloc(&d_out, 4096 * 256 * 8)); CK(hipMalloc(&d_in, 65536 * 8)); { double* h = (double*)malloc(65536*8); srand(3); for (int i = 0; i < 65536; ++i) h[i] = rand() / (double)RAND_MAX * 2 - 1; CK(hipMemcpy(d_in, h, 65536*8, hipMemcpyHostToDevice)); }
    for (int wg : {1024, 2048}) { run<0>(wg, d_out, d_in); run<1>(wg, d_out, d_in); run<3>(wg, d_out, d_in); }
    return 0;
}
