// One wave per SIMD issuing v_mfma_f32_16x16x4_f32: what does an instruction cost with operands from registers, and with the
// operands of the next step read from LDS while a step multiplies (the inner loop of csrc/rows_gemm.hip)?  (tools/, not part of
// the product)   hipcc --offload-arch=gfx950 -O3 -o build/bench_mfma_f32 tools/bench_mfma_f32.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE, int NT>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters)
{
    __shared__ float bs[2][32 * 130];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    for (int i = tid; i < 2 * 32 * 130; i += 256) (&bs[0][0])[i] = 1.0f / (1 + (i & 7));
    __syncthreads();
    f4 acc[NT];
    for (int t = 0; t < NT; ++t) acc[t] = f4{0, 0, 0, 0};
    float cv[8];
    for (int t = 0; t < 8; ++t) cv[t] = 0.001f * (lane + t);
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const float* bl = &bs[it & 1][(8 * g) * 130 + n];
        if (MODE == 0) {   // operands from registers
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[t], cv[(t + ct) & 7], acc[ct], 0, 0, 0);
        } else {           // rows_gemm's inner loop
            float bv[2][NT];
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) bv[0][ct] = bl[16 * ct];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t + 1 < 8) {
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) bv[(t + 1) & 1][ct] = bl[(t + 1) * 130 + 16 * ct];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[t], bv[t & 1][ct], acc[ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 2) __syncthreads();
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int NT>
static void run(const char* name, int wgs, int threads)
{
    float* o;
    long long* c;
    hipMalloc(&o, wgs * 256 * 4);
    hipMalloc(&c, wgs * 8);
    const int iters = 400;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, NT>), dim3(wgs), dim3(threads), 0, 0, o, c, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, NT>), dim3(wgs), dim3(threads), 0, 0, o, c, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    long long h;
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 8 * NT;
    printf("%-46s wgs %4d x %3d threads: %7.1f us, %6.1f ns per matrix instruction and wave, %6.1f counter ticks each\n", name, wgs, threads, ms * 1e3,
           ms * 1e6 / nm, (double)h / nm);
    hipFree(o);
    hipFree(c);
}
int main()
{
    run<0, 7>("registers, 7 accumulators", 200, 256);
    run<1, 7>("LDS operands a step ahead, 7 accumulators", 200, 256);
    run<2, 7>("same + barrier per 56", 200, 256);
    run<0, 4>("registers, 4 accumulators", 200, 256);
    run<1, 4>("LDS operands a step ahead, 4 accumulators", 200, 256);
    run<0, 7>("registers, 7 acc, 2 workgroups per CU", 512, 256);
    run<1, 7>("LDS a step ahead, 7 acc, 2 workgroups per CU", 512, 256);
    run<1, 7>("LDS a step ahead, 7 acc, 4 workgroups per CU", 1024, 256);
    return 0;
}
