// Do FP64 VALU ops slow a v_mfma_f64 stream (shared DP datapath?) while 32-bit VALU ops overlap?  (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef double d4 __attribute__((ext_vector_type(4)));
// MODE 0: MFMA only; 1: + KV v_min_f64 per 9 MFMAs... per group of 36 MFMAs add NV ops: 1 = f64 min, 2 = f64 cmp+cndmask, 3 = i32 ops, 4 = f64 fma
template<int MODE, int NV>
__global__ __launch_bounds__(256, 2) void k(const double* in, double* out, int iters) {
    const int l = threadIdx.x & 63;
    double a[9], b[4][9];
    for (int i = 0; i < 9; ++i) { a[i] = in[l + 64 * i]; for (int f = 0; f < 4; ++f) b[f][i] = in[l + 64 * (i + 9 + f)]; }
    double x[8]; int y[8];
    for (int i = 0; i < 8; ++i) { x[i] = in[l + i]; y[i] = l + i; }
    double sink = 0;
    for (int it = 0; it < iters; ++it) {
        d4 acc[4];
        #pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[f][0], (d4){0,0,0,0}, 0, 0, 0);
        #pragma unroll
        for (int s = 1; s < 9; ++s)
            #pragma unroll
            for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[f][s], acc[f], 0, 0, 0);
        #pragma unroll
        for (int v = 0; v < NV; ++v) {
            const double d = acc[v & 3][(v >> 2) & 3];
            if constexpr (MODE == 1) x[v & 7] = __builtin_fmin(x[v & 7], d);
            else if constexpr (MODE == 2) { bool lt = d < x[v & 7]; y[v & 7] = lt ? it : y[v & 7]; }
            else if constexpr (MODE == 3) { int hi = __double2hiint(d); y[v & 7] = (hi < y[v & 7]) ? hi ^ it : y[v & 7] + 1; }
            else if constexpr (MODE == 4) x[v & 7] = __builtin_fma(d, 1e-9, x[v & 7]);
        }
        if (MODE == 0 || NV == 0) sink += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    }
    double s = sink;
    for (int i = 0; i < 8; ++i) s += x[i] + y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template<int MODE, int NV> void run(const double* d_in, double* d_out) {
    const int iters = 4000, wg = 512;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE,NV>), dim3(wg), dim3(256), 0, 0, d_in, d_out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    double flops = 2.0 * 1024 * 36.0 * iters * (double)wg * 4;
    printf("MODE=%d NV=%2d : %.3f ms  %.2f TFLOP/s (MFMA flops only)\n", MODE, NV, best, flops / best * 1e-9); fflush(stdout);
}
int main() {
    std::vector<double> h(64 * 32); srand(2); for (auto& x : h) x = rand() / (double)RAND_MAX * 2 - 1;
    double *d_in, *d_out; CK(hipMalloc(&d_in, h.size() * 8)); CK(hipMalloc(&d_out, 2048 * 256 * 8));
    CK(hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    run<0,0>(d_in, d_out);
    run<1,16>(d_in, d_out); run<1,48>(d_in, d_out);
    run<2,16>(d_in, d_out); run<2,48>(d_in, d_out);
    run<3,16>(d_in, d_out); run<3,48>(d_in, d_out); run<3,96>(d_in, d_out);
    run<4,16>(d_in, d_out); run<4,48>(d_in, d_out);
    return 0;
}
