// Calibration microbenchmark (dev tool): what one tick of __builtin_readcyclecounter() is worth on this GPU, in vector
// instructions and in LDS instructions, with 1 / 4 waves per SIMD -- the unit the stamps of tools/bench_stft_pk.cpp are in.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_clock.cpp -o build/bench_clock
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void valu_kernel(float* out, unsigned long long* ticks, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                     "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

__global__ void valu_pk_kernel(float* out, unsigned long long* ticks, int iters)
{
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                     "v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int BYTES>
__global__ void lds_kernel(float* out, unsigned long long* ticks, int iters)
{
    __shared__ __attribute__((aligned(16))) float buf[64 * 4 * 16];
    float* mine = buf + (threadIdx.x >> 6) * 256 + (threadIdx.x & 63) * (BYTES / 4);
    for (int i = threadIdx.x; i < 64 * 4 * 16; i += blockDim.x) buf[i] = i;
    __syncthreads();
    float acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (BYTES == 4) {
                float v;
                asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)mine));
                asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                acc += 0;
                (void)v;
            } else if (BYTES == 8) {
                typedef float v2 __attribute__((ext_vector_type(2)));
                v2 v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)mine));
            } else {
                typedef float v4 __attribute__((ext_vector_type(4)));
                v4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)mine));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

int main()
{
    float* out;
    unsigned long long* ticks;
    hipMalloc(&out, 256 * 1024 * 4 * 4);
    hipMalloc(&ticks, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves_per_cu : {4, 8, 12, 16}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(valu_kernel, dim3(256), dim3(64 * waves_per_cu), 0, 0, out, ticks, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t;
            hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            printf("VALU  %2d waves/CU: %.3f ms, %llu ticks -> %.2f GHz tick rate, %.2f ticks per v_fma_f32 per wave (x %d waves per SIMD); %.1f TFLOP/s\n", waves_per_cu, ms, t,
                   t / (ms * 1e6), (double)t / (8.0 * iters), waves_per_cu / 4, 256.0 * waves_per_cu * 8.0 * iters * 64 * 2 / (ms * 1e-3) / 1e12);
        }
    }
    for (int waves_per_cu : {4, 8, 16}) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(valu_pk_kernel, dim3(256), dim3(64 * waves_per_cu), 0, 0, out, ticks, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t;
        hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        printf("VALU pk %2d waves/CU: %.3f ms, %llu ticks -> %.2f GHz, %.2f ticks per v_pk_fma_f32 per wave; %.1f TFLOP/s\n", waves_per_cu, ms, t, t / (ms * 1e6),
               (double)t / (8.0 * iters), 256.0 * waves_per_cu * 8.0 * iters * 64 * 4 / (ms * 1e-3) / 1e12);
    }
#define LDS_RUN(B)                                                                                                             \
    for (int waves_per_cu : {4, 16}) {                                                                                         \
        hipEventRecord(e0);                                                                                                    \
        hipLaunchKernelGGL(lds_kernel<B>, dim3(256), dim3(64 * waves_per_cu), 0, 0, out, ticks, iters / 10);                   \
        hipEventRecord(e1);                                                                                                    \
        hipEventSynchronize(e1);                                                                                               \
        float ms;                                                                                                              \
        hipEventElapsedTime(&ms, e0, e1);                                                                                      \
        unsigned long long t;                                                                                                  \
        hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);                                                                        \
        printf("LDS b%-3d %2d waves/CU: %.3f ms, %llu ticks, %.2f ticks per CU per instruction (%.1f B per tick per CU)\n", B * 8, waves_per_cu, ms, t, \
               (double)t / (8.0 * (iters / 10) * waves_per_cu), 64.0 * B * 8.0 * (iters / 10) * waves_per_cu / t);           \
    }
    LDS_RUN(4)
    LDS_RUN(8)
    LDS_RUN(16)
    return 0;
}
