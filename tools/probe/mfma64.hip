// v_mfma_f64_16x16x4_f64: (1) is D = k-ordered single-rounded fma chain starting from C? (2) sustained rate on random data
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef double d4 __attribute__((ext_vector_type(4)));

// one wave: D[16x16] = A[16x4] * B[4x16] + C ; lane l: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15,row=(l>>4)+4*reg
__global__ void k_check(const double* A, const double* B, const double* C, double* D) {
    const int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
    d4 c;
    for (int rg = 0; rg < 4; ++rg) c[rg] = C[((l >> 4) + 4 * rg) * 16 + (l & 15)];
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int rg = 0; rg < 4; ++rg) D[((l >> 4) + 4 * rg) * 16 + (l & 15)] = d[rg];
}

template<int NACC>
__global__ __launch_bounds__(256) void k_rate(const double* in, double* out, int iters, unsigned long long* stamps) {
    const int l = threadIdx.x & 63;
    double a[10], b[10];
    for (int i = 0; i < 10; ++i) { a[i] = in[l + 64 * i]; b[i] = in[l + 64 * (i + 10)]; }
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int kk = 0; kk < 10; ++kk)
            #pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[(kk + i) % 10], acc[i], 0, 0, 0);
        #pragma unroll
        for (int i = 0; i < NACC; ++i) { acc[i][0] *= 1e-3; acc[i][1] *= 1e-3; acc[i][2] *= 1e-3; acc[i][3] *= 1e-3; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (l == 0) { int w = blockIdx.x * 4 + (threadIdx.x >> 6); stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0; }
}
template<int NACC> void rate(int wg, const double* d_in, double* d_out, unsigned long long* d_st) {
    const int iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_rate<NACC>), dim3(wg), dim3(256), 0, 0, d_in, d_out, iters, d_st);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    std::vector<unsigned long long> st(2 * wg * 4); CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    double clk = (double)st[2 * (wg * 2)] / (double)st[2 * (wg * 2) + 1] * 100.0;
    double flops = 2.0 * 16 * 16 * 4 * 10.0 * NACC * iters * (double)wg * 4;
    printf("mfma_f64 NACC=%d wg=%d : %.3f ms %.2f TFLOP/s  clock %.0f MHz\n", NACC, wg, best, flops / best * 1e-9, clk); fflush(stdout);
}
int main() {
    // (1) rounding-order check
    std::vector<double> A(64), B(64), C(256), D(256);
    srand(5);
    int bad_fwd = 0, bad_rev = 0, bad_unfused = 0;
    double *dA, *dB, *dC, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 2048)); CK(hipMalloc(&dD, 2048));
    for (int trial = 0; trial < 200; ++trial) {
        for (auto& x : A) x = (rand() / (double)RAND_MAX * 2 - 1) * pow(2.0, rand() % 40 - 20);
        for (auto& x : B) x = (rand() / (double)RAND_MAX * 2 - 1) * pow(2.0, rand() % 40 - 20);
        for (auto& x : C) x = (trial & 1) ? 0.0 : (rand() / (double)RAND_MAX * 2 - 1);
        CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double f = C[i * 16 + j], r = C[i * 16 + j], u = C[i * 16 + j];
            for (int k = 0; k < 4; ++k) f = fma(A[i * 4 + k], B[k * 16 + j], f);
            for (int k = 3; k >= 0; --k) r = fma(A[i * 4 + k], B[k * 16 + j], r);
            for (int k = 0; k < 4; ++k) { volatile double p = A[i * 4 + k] * B[k * 16 + j]; u = u + p; }
            if (D[i * 16 + j] != f) ++bad_fwd;
            if (D[i * 16 + j] != r) ++bad_rev;
            if (D[i * 16 + j] != u) ++bad_unfused;
        }
    }
    printf("mfma_f64_16x16x4 vs host: k-ascending fma chain mismatches=%d, k-descending=%d, unfused=%d (of %d)\n", bad_fwd, bad_rev, bad_unfused, 200 * 256);
    // (2) rate on random data
    std::vector<double> h(64 * 20); for (auto& x : h) x = rand() / (double)RAND_MAX * 2 - 1;
    double *d_in, *d_out; unsigned long long* d_st; CK(hipMalloc(&d_in, h.size() * 8)); CK(hipMalloc(&d_out, 4096 * 256 * 8)); CK(hipMalloc(&d_st, 4096 * 4 * 16));
    CK(hipMemcpy(d_in, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    for (int wg : {256, 512, 1024}) { rate<1>(wg, d_in, d_out, d_st); rate<2>(wg, d_in, d_out, d_st); rate<4>(wg, d_in, d_out, d_st); rate<8>(wg, d_in, d_out, d_st); }
    return 0;
}
