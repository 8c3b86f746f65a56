// LDS atomic throughput (tools/probe): how many cycles does a 64-lane ds_add_u64 / ds_add_u32 / plain ds_write_b64 take
// per wave-instruction with 8 waves per CU hammering a workgroup table (lane-distinct addresses, rows picked per step)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
template <int KIND>
__global__ __launch_bounds__(512) void k(int iters, int rows, u64* out)
{
    extern __shared__ u64 tbl[];
    unsigned* t32 = (unsigned*)tbl;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < rows * 80; i += 512) tbl[i] = 0;
    __syncthreads();
    unsigned r = w * 7 + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            r = r * 1664525u + 1013904223u;
            const int row = (int)((r >> 8) % (unsigned)rows);  // wave-uniform
            if (KIND == 0) atomicAdd(&tbl[row * 80 + lane], (u64)(it + lane));
            if (KIND == 1) {
                atomicAdd(&t32[(row * 80 + lane) * 2], (unsigned)(it + lane));
                atomicAdd(&t32[(row * 80 + lane) * 2 + 1], (unsigned)it);
            }
            if (KIND == 2) atomicAdd(&t32[row * 160 + lane], (unsigned)(it + lane));       // 32-bit, consecutive dwords
            if (KIND == 3) tbl[row * 80 + lane] = (u64)(it + lane);                         // plain 64-bit store
        }
    }
    __syncthreads();
    u64 s = 0;
    for (int i = threadIdx.x; i < rows * 80; i += 512) s += tbl[i];
    if (s == 12345) out[0] = s;
}
template <int KIND>
static void run(const char* name, int rows)
{
    u64* out;
    hipMalloc(&out, 8);
    const int iters = 2000, grid = 256;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(512), (size_t)rows * 80 * 8, 0, 10, rows, out);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(512), (size_t)rows * 80 * 8, 0, iters, rows, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)iters * 8 * 8 * (KIND == 1 ? 2 : 1);  // wave-instructions per CU
    printf("%-28s rows=%3d: %.3f ms -> %.1f ns per wave-instruction per CU (%.1f cycles at 2.4 GHz)\n", name, rows, ms,
           ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
    hipFree(out);
}
int main()
{
    for (int rows : {2, 16, 128}) {
        run<0>("ds_add_u64", rows);
        run<1>("2 x ds_add_u32 (same 8 bytes)", rows);
        run<2>("ds_add_u32 (consecutive)", rows);
        run<3>("ds_write_b64", rows);
    }
    return 0;
}
