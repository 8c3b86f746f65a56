// what do the allocations of an upload cost? (hipMalloc of GBs, pinned staging, frees) -- tools/probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipFree(0);
    for (int rep = 0; rep < 2; ++rep) {
        void *d = nullptr, *h[2] = {nullptr, nullptr}, *d2 = nullptr;
        double t0 = now();
        hipMalloc(&d, (size_t)2960 << 20);
        double t1 = now();
        hipHostMalloc(&h[0], (size_t)78 << 20, hipHostMallocDefault);
        hipHostMalloc(&h[1], (size_t)78 << 20, hipHostMallocDefault);
        double t2 = now();
        hipMalloc(&d2, (size_t)2200 << 20);
        double t3 = now();
        hipMemset(d, 0, (size_t)2960 << 20);
        hipDeviceSynchronize();
        double t4 = now();
        hipFree(d);
        double t5 = now();
        hipHostFree(h[0]);
        hipHostFree(h[1]);
        double t6 = now();
        hipFree(d2);
        double t7 = now();
        printf("rep %d: hipMalloc 2.96 GB %.1f ms | 2 x hipHostMalloc 78 MB %.1f ms | hipMalloc 2.2 GB %.1f ms | memset 2.96 GB %.1f ms | hipFree 2.96 GB %.1f ms | 2 x hipHostFree %.1f ms | hipFree 2.2 GB %.1f ms\n",
               rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3, (t7 - t6) * 1e3);
    }
    return 0;
}
