// Are workgroup-scope int64 atomics into a per-XCD table copy (a) exact, (b) faster than agent-scope atomics
// into one shared table?  (diagnostic)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
typedef unsigned long long u64;
constexpr int RS = 80;

__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// MODE 0: agent scope, one table.  MODE 1: workgroup scope, table copy = XCC id.  MODE 2: agent scope, per-XCC copy
template<int MODE>
__global__ __launch_bounds__(512) void k(u64* tables, int M, int iters, unsigned seed, int* xcc_seen) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int xcc = xcc_id();
    if (lane == 0) atomicOr(&xcc_seen[xcc], 1);
    u64* tab = tables + (MODE == 0 ? 0 : (size_t)xcc * M * RS);
    unsigned s = seed ^ (wave * 2654435761u);
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const int cell = (s >> 8) % M;
        u64* row = tab + (size_t)cell * RS;
        const u64 v = (u64)(lane + 1);
        if (MODE == 1) {
            __hip_atomic_fetch_add(&row[lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane < 15) __hip_atomic_fetch_add(&row[64 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add(&row[lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane < 15) __hip_atomic_fetch_add(&row[64 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__global__ void k_combine(const u64* tables, int n, int copies, u64* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { u64 s = 0; for (int c = 0; c < copies; ++c) s += tables[(size_t)c * n + i]; out[i] = s; }
}
template<int MODE> void run(int M, int iters) {
    const int copies = 16, wg = 256;
    u64 *d_tab, *d_out; int* d_seen;
    const size_t n = (size_t)M * RS;
    CK(hipMalloc(&d_tab, copies * n * 8)); CK(hipMalloc(&d_out, n * 8)); CK(hipMalloc(&d_seen, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(d_tab, 0, copies * n * 8)); CK(hipMemset(d_seen, 0, 64));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE>), dim3(wg), dim3(512), 0, 0, d_tab, M, iters, 1234u, d_seen);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    hipLaunchKernelGGL(k_combine, dim3((n + 255) / 256), dim3(256), 0, 0, d_tab, (int)n, copies, d_out);
    std::vector<u64> h(n); CK(hipMemcpy(h.data(), d_out, n * 8, hipMemcpyDeviceToHost));
    std::vector<int> seen(16); CK(hipMemcpy(seen.data(), d_seen, 64, hipMemcpyDeviceToHost));
    // expected: every wave adds (lane+1) to element lane (<79) of a pseudo-random row, iters times
    std::vector<u64> exp(n, 0);
    const int nwaves = wg * 8;
    for (int w = 0; w < nwaves; ++w) { unsigned s = 1234u ^ (w * 2654435761u); for (int it = 0; it < iters; ++it) { s = s * 1664525u + 1013904223u; int cell = (s >> 8) % M; for (int l = 0; l < 64; ++l) exp[(size_t)cell * RS + l] += l + 1; for (int l = 0; l < 15; ++l) exp[(size_t)cell * RS + 64 + l] += l + 1; } }
    size_t bad = 0; for (size_t i = 0; i < n; ++i) if (h[i] != exp[i]) ++bad;
    int nx = 0; for (int x : seen) nx += x;
    double bytes = (double)nwaves * iters * 79 * 8;
    printf("MODE=%d M=%4d : %.3f ms  %.2f TB/s of atomic payload  xcc ids seen=%d  mismatching elements=%zu\n", MODE, M, best, bytes / best * 1e-9, nx, bad);
    fflush(stdout);
    CK(hipFree(d_tab)); CK(hipFree(d_out)); CK(hipFree(d_seen));
}
int main() {
    for (int M : {256, 1024, 4096}) { run<0>(M, 1024); run<1>(M, 1024); run<2>(M, 1024); }
    return 0;
}
