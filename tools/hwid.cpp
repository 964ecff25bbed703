// Where do the waves of a 256-thread workgroup land?  (tools/, not part of the product)  hipcc --offload-arch=gfx950 -O2 -o build/hwid tools/hwid.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
    // stay resident for a while so that workgroups do not reuse slots
    long t0 = clock64();
    while (clock64() - t0 < 200000) {}
}
int main()
{
    const int wgs = 200, waves = 4;
    unsigned* d;
    hipMalloc(&d, wgs * waves * 4);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(waves * 64), 0, 0, d);
    std::vector<unsigned> h(wgs * waves);
    hipMemcpy(h.data(), d, wgs * waves * 4, hipMemcpyDeviceToHost);
    int hist[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < wgs; ++w) {
        unsigned mask = 0;
        for (int i = 0; i < waves; ++i) mask |= 1u << ((h[w * waves + i] >> 4) & 3);
        hist[__builtin_popcount(mask)]++;
        if (w < 6) {
            printf("wg %d:", w);
            for (int i = 0; i < waves; ++i) printf(" [simd %u cu %u se %u raw %08x]", (h[w * waves + i] >> 4) & 3, (h[w * waves + i] >> 8) & 15, (h[w * waves + i] >> 13) & 7, h[w * waves + i]);
            printf("\n");
        }
    }
    printf("distinct SIMDs per workgroup: 1:%d 2:%d 3:%d 4:%d\n", hist[1], hist[2], hist[3], hist[4]);
    return 0;
}
