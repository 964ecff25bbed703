// A stand-in for a collective's kernel on a one-GPU box (tools/ab_reserve_cus.py): `wgs` workgroups of 256 threads with 32 KB of LDS each
// that keep their CUs busy for `ns` nanoseconds (wall clock), like RCCL's all-gather waiting on xGMI transfers.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/occupy.cpp -o build/liboccupy.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void occupy_kernel(long ns, unsigned* sink)
{
    __shared__ unsigned pad[8192];
    pad[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();   // 100 MHz
    unsigned acc = 0;
    while ((long)(wall_clock64() - t0) * 10 < ns) {
        acc += pad[(threadIdx.x + acc) & 8191];
        __builtin_amdgcn_s_sleep(32);
    }
    if (acc == 0xffffffffu) *sink = acc;
}

__global__ __launch_bounds__(256) void occupy_sleep_kernel(long ns)   // no LDS, no memory: sleeps only
{
    const unsigned long long t0 = wall_clock64();
    while ((long)(wall_clock64() - t0) * 10 < ns) __builtin_amdgcn_s_sleep(127);
}

extern "C" int occupy_sleep(int wgs, long ns, void* stream)
{
    hipLaunchKernelGGL(occupy_sleep_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, ns);
    return (int)hipGetLastError();
}

extern "C" int occupy(int wgs, long ns, void* sink, void* stream)
{
    hipLaunchKernelGGL(occupy_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, ns, (unsigned*)sink);
    return (int)hipGetLastError();
}
