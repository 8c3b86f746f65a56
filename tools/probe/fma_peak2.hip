// FP64 FMA issue-rate microbench: does operand diversity (many distinct r VGPRs) or DPP slow v_fmac_f64?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

template<int N> __device__ __forceinline__ void fmac_bc(double& acc, double c, double r) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(r), "i"(N));
}
__device__ __forceinline__ void fmac_v(double& acc, double c, double r) {
    asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(c), "v"(r));
}
__device__ __forceinline__ void fmac_s(double& acc, double c, double r) {
    asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "s"(c), "v"(r));
}

// MODE 0: VGPR c, NR distinct r registers; 1: DPP c; 2: SGPR c
template<int MODE, int CH, int NR>
__global__ __launch_bounds__(256) void k(double* out, const double* in, double s0, int iters) {
    double r[NR], acc[CH], c[4];
    #pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = in[threadIdx.x + 64 * i];
    #pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = in[threadIdx.x + 7 * i] * 1e-3;
    #pragma unroll
    for (int i = 0; i < CH; ++i) acc[i] = i;
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int n = 0; n < NR; ++n)
            #pragma unroll
            for (int i = 0; i < CH; ++i) {
                if constexpr (MODE == 0) fmac_v(acc[i], c[i & 3], r[n]);
                else if constexpr (MODE == 1) fmac_bc<3>(acc[i], c[i & 3], r[n]);
                else fmac_s(acc[i], s0, r[n]);
            }
    }
    double s = 0;
    #pragma unroll
    for (int i = 0; i < CH; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template<int MODE, int CH, int NR> void run(int wg, double* d_out, double* d_in) {
    const int iters = 4000 / NR * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE,CH,NR>), dim3(wg), dim3(256), 0, 0, d_out, d_in, 0.5, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double flops = 2.0 * CH * NR * (double)iters * wg * 256;
    printf("MODE=%d CH=%d NR=%d wg=%d : %.3f ms %.2f TFLOP/s\n", MODE, CH, NR, wg, best, flops / best * 1e-9); fflush(stdout);
}
int main() {
    double *d_out, *d_in; CK(hipMalloc(&d_out, 4096 * 256 * 8)); CK(hipMalloc(&d_in, 65536 * 8)); CK(hipMemset(d_in, 0, 65536 * 8));
    for (int wg : {512, 1024, 2048}) {
        run<0,4,1>(wg, d_out, d_in); run<0,4,8>(wg, d_out, d_in); run<0,4,37>(wg, d_out, d_in);
        run<1,4,1>(wg, d_out, d_in); run<1,4,8>(wg, d_out, d_in); run<1,4,37>(wg, d_out, d_in);
        run<2,4,1>(wg, d_out, d_in); run<2,4,37>(wg, d_out, d_in);
        run<0,1,37>(wg, d_out, d_in); run<0,2,37>(wg, d_out, d_in); run<0,8,37>(wg, d_out, d_in);
    }
    return 0;
}
