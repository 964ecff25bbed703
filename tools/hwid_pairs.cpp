// Which workgroups of a 512-workgroup launch (256 threads, ~75 KB of LDS: two per CU) share a CU?  (tools/, not part of the product)
// hipcc --offload-arch=gfx950 -O2 -o build/hwid_pairs tools/hwid_pairs.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ void k(unsigned* out)
{
    extern __shared__ float sm[];
    unsigned id, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = id; out[2 * blockIdx.x + 1] = xcc; sm[0] = 1.f; }
    long t0 = clock64();
    while (clock64() - t0 < 2000000) {}
}
int main()
{
    const int wgs = 512;
    unsigned* d;
    hipMalloc(&d, wgs * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 75 * 1024, 0, d);
    std::vector<unsigned> h(wgs * 2);
    hipMemcpy(h.data(), d, wgs * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int w = 0; w < wgs; ++w) {
        const unsigned id = h[2 * w], xcc = h[2 * w + 1] & 15;
        const unsigned key = (xcc << 16) | (((id >> 13) & 7) << 8) | ((id >> 8) & 15);
        cu[key].push_back(w);
    }
    std::map<int, int> diff, cnt;
    int shown = 0;
    for (auto& kv : cu) {
        cnt[(int)kv.second.size()]++;
        if (kv.second.size() == 2) diff[kv.second[1] - kv.second[0]]++;
        if (shown++ < 12) { printf("cu %06x:", kv.first); for (int w : kv.second) printf(" %d", w); printf("\n"); }
    }
    printf("CUs seen: %zu; workgroups per CU:", cu.size());
    for (auto& c : cnt) printf(" %d:%d", c.first, c.second);
    printf("\nblock-index difference of the pairs:");
    for (auto& c : diff) printf(" %d:%d", c.first, c.second);
    printf("\n");
    return 0;
}
