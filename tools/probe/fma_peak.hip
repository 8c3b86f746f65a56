// Practical FP64 VALU ceiling + in-kernel clock on MI355X (diagnostic, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template<int CH>
__global__ __launch_bounds__(256) void fma_loop(double* out, double s0, double s1, int iters, unsigned long long* stamps) {
    double acc[CH];
    #pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int k = 0; k < 8; ++k)
            #pragma unroll
            for (int i = 0; i < CH; ++i) acc[i] = __builtin_fma(acc[i], s0, s1);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    #pragma unroll
    for (int i = 0; i < CH; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
    }
}

template<int CH> void run(int wg, int iters, double* d_out, unsigned long long* d_st) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int nw = wg * 4;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((fma_loop<CH>), dim3(wg), dim3(256), 0, 0, d_out, 0.999999, 1e-7, iters, d_st);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> st(2 * nw);
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> clk(nw);
        for (int w = 0; w < nw; ++w) clk[w] = (double)st[2 * w] / (double)st[2 * w + 1] * 100.0; // MHz
        std::sort(clk.begin(), clk.end());
        double flops = 2.0 * CH * 8.0 * iters * (double)wg * 256;
        printf("CH=%d wg=%d iters=%d: %.3f ms  %.2f TFLOP/s  clock median %.0f MHz (min %.0f max %.0f)\n",
               CH, wg, iters, ms, flops / ms * 1e-9, clk[nw / 2], clk[0], clk[nw - 1]);
    }
}
int main() {
    double* d_out; unsigned long long* d_st;
    CK(hipMalloc(&d_out, 8192 * 256 * 8)); CK(hipMalloc(&d_st, 8192 * 4 * 16));
    for (int wg : {256, 512, 1024, 2048}) {
        run<2>(wg, 20000, d_out, d_st);
        run<4>(wg, 10000, d_out, d_st);
        run<8>(wg, 5000, d_out, d_st);
    }
    // long run to see sustained clocks (~1 s)
    run<8>(2048, 100000, d_out, d_st);
    return 0;
}
