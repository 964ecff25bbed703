// Practical HBM ceilings on the box for the STFT's traffic shape: a 210 MB write, a 66 MB read,
// and both together (tools/, not part of the product).  hipcc --offload-arch=gfx950 -O3 -o build/bench_hbm tools/bench_hbm.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void fill_kernel(float4* __restrict__ dst, long n4)
{
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) dst[i] = v;
}
__global__ void fill_nt_kernel(float4* __restrict__ dst, long n4)
{
    typedef float vf4 __attribute__((ext_vector_type(4)));
    const vf4 v = {1.f, 2.f, 3.f, 4.f};
    vf4* d = reinterpret_cast<vf4*>(dst);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(v, &d[i]);
}
__global__ void read_kernel(const float4* __restrict__ src, long n4, float* __restrict__ out)
{
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
// the STFT's shape: read r4 float4, write w4 float4 per thread-iteration ratio ~ 1:3.2
__global__ void mix_kernel(const float4* __restrict__ src, long nr4, float4* __restrict__ dst, long nw4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nw4; i += (long)gridDim.x * blockDim.x) {
        float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
        if ((i & 3) == 0 && (i >> 2) < nr4) v = src[i >> 2];
        dst[i] = v;
    }
}
template <typename Fn> static float timeit(Fn fn, int iters = 20)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) fn();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) fn();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}
int main()
{
    const long wbytes = 204800L * 257 * 4, rbytes = 1024L * 16000 * 4;
    float4 *w, *r;
    float* o;
    hipMalloc(&w, wbytes + 64);
    hipMalloc(&r, rbytes + 64);
    hipMalloc(&o, 64);
    hipMemset(r, 0, rbytes);
    const long nw4 = wbytes / 16, nr4 = rbytes / 16;
    for (int grid : {1024, 2048, 4096, 8192, 16384}) {
        for (int bs : {64, 256}) {
            float t1 = timeit([&] { hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(bs), 0, 0, w, nw4); });
            float t1n = timeit([&] { hipLaunchKernelGGL(fill_nt_kernel, dim3(grid), dim3(bs), 0, 0, w, nw4); });
            float t2 = timeit([&] { hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(bs), 0, 0, r, nr4, o); });
            float t2b = timeit([&] { hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(bs), 0, 0, (const float4*)w, nw4, o); });
            float t3 = timeit([&] { hipLaunchKernelGGL(mix_kernel, dim3(grid), dim3(bs), 0, 0, r, nr4, w, nw4); });
            printf("grid %5d x %3d: fill 210MB %.1f us (%.2f TB/s) | nt %.1f us (%.2f TB/s) | read 66MB %.1f us (%.2f TB/s) | read 210MB %.1f us (%.2f TB/s) | mix %.1f us (%.2f TB/s)\n",
                   grid, bs, t1 * 1e3, wbytes / t1 * 1e-9, t1n * 1e3, wbytes / t1n * 1e-9, t2 * 1e3, rbytes / t2 * 1e-9,
                   t2b * 1e3, wbytes / t2b * 1e-9, t3 * 1e3, (wbytes + rbytes) / t3 * 1e-9);
        }
    }
    float tm = timeit([&] { hipMemsetAsync(w, 0, wbytes, 0); });
    printf("hipMemsetAsync 210MB %.1f us (%.2f TB/s)\n", tm * 1e3, wbytes / tm * 1e-9);
    return 0;
}
