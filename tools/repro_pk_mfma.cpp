// Stand-alone reproducer attempt for the interaction of DESIGN.md 3.25: packed float32 vector instructions (v_pk_add_f32 ...) in one
// wave while the OTHER wave of the same SIMD issues float32 matrix instructions (v_mfma_f32_4x4x1_16b_f32).
//
//   hipcc --offload-arch=gfx950 -O3 tools/repro_pk_mfma.cpp -o build/repro_pk_mfma && build/repro_pk_mfma
//
// One workgroup of 8 waves per CU: waves 0..3 ("victims", one per SIMD) run a long butterfly recurrence twice -- once on packed
// instructions, once on scalar ones, same roundings -- and count the lanes where the two disagree; waves 4..7 ("aggressors", the
// second wave of each SIMD) run the instruction kind selected by `mode`: 0 nothing | 1 v_mfma_f32_4x4x1 | 2 v_mfma_f32_16x16x4_f32 |
// 3 v_mfma_f32_16x16x32_f16 | 4 v_fma_f32.  Every disagreement is a wrong result of a packed (or scalar) instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ v2f pk_add(v2f a, v2f b) { v2f r; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b) { v2f r; asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f pk_negi(v2f a, v2f b) { v2f r; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f pk_posi(v2f a, v2f b) { v2f r; asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ v2f pk_mul(v2f a, v2f b) { v2f r; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_add(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_sub(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float s_mul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// 4-point DFT butterfly, packed and scalar (same roundings)
__device__ __forceinline__ void dft4_pk(v2f& a0, v2f& a1, v2f& a2, v2f& a3)
{
    const v2f s02 = pk_add(a0, a2), d02 = pk_sub(a0, a2), s13 = pk_add(a1, a3), d13 = pk_sub(a1, a3);
    a0 = pk_add(s02, s13); a2 = pk_sub(s02, s13); a1 = pk_negi(d02, d13); a3 = pk_posi(d02, d13);
}
__device__ __forceinline__ void dft4_sc(v2f& a0, v2f& a1, v2f& a2, v2f& a3)
{
    const v2f s02 = {s_add(a0.x, a2.x), s_add(a0.y, a2.y)}, d02 = {s_sub(a0.x, a2.x), s_sub(a0.y, a2.y)};
    const v2f s13 = {s_add(a1.x, a3.x), s_add(a1.y, a3.y)}, d13 = {s_sub(a1.x, a3.x), s_sub(a1.y, a3.y)};
    a0 = v2f{s_add(s02.x, s13.x), s_add(s02.y, s13.y)};
    a2 = v2f{s_sub(s02.x, s13.x), s_sub(s02.y, s13.y)};
    a1 = v2f{s_add(d02.x, d13.y), s_sub(d02.y, d13.x)};
    a3 = v2f{s_sub(d02.x, d13.y), s_add(d02.y, d13.x)};
}

__global__ __launch_bounds__(512) void repro_kernel(int mode, int iters, unsigned* __restrict__ errors, float* __restrict__ sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 256 registers per wave, as the fused kernel's waves hold: two waves then fill the SIMD's register file
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    __shared__ v2f tile[4][2][16 * 17 * 4];
    if (wave < 4) {   // victims
        unsigned bad = 0;
        for (int rep = 0; rep < iters; ++rep) {
            v2f p[8], s[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = s[i] = v2f{(float)((lane * 7 + i * 3 + rep) % 97) * 0.03125f, (float)((lane * 5 + i + 2 * rep) % 89) * 0.0625f};
            const v2f q = {0.25f, 0.25f};
#pragma unroll
            for (int it = 0; it < 24; ++it) {
                dft4_pk(p[0], p[1], p[2], p[3]); dft4_pk(p[4], p[5], p[6], p[7]);
                dft4_pk(p[0], p[4], p[2], p[6]); dft4_pk(p[1], p[5], p[3], p[7]);
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = pk_mul(p[i], q);
                dft4_sc(s[0], s[1], s[2], s[3]); dft4_sc(s[4], s[5], s[6], s[7]);
                dft4_sc(s[0], s[4], s[2], s[6]); dft4_sc(s[1], s[5], s[3], s[7]);
#pragma unroll
                for (int i = 0; i < 8; ++i) s[i] = v2f{s_mul(s[i].x, 0.25f), s_mul(s[i].y, 0.25f)};
                // a transposition through LDS, as the FFT's (stride 17), for both copies
                {
                    const int j = lane & 15, fl = lane >> 4;
                    v2f* zp = tile[wave][0] + fl * 272;
                    v2f* zs = tile[wave][1] + fl * 272;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { zp[i * 17 + j] = p[i]; zs[i * 17 + j] = s[i]; }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int i = 0; i < 8; ++i) { p[i] = zp[(j & 7) * 17 + ((i + j) & 15)]; s[i] = zs[(j & 7) * 17 + ((i + j) & 15)]; }
                    __builtin_amdgcn_wave_barrier();
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) bad += (p[i].x != s[i].x) + (p[i].y != s[i].y);
        }
        if (bad) atomicAdd(errors + wave, bad);
        if (bad) atomicAdd(errors + 4, 1u);   // lanes with at least one wrong value
    } else {          // aggressors
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float a = 1.0f + lane * 1e-3f, b = 0.5f - lane * 1e-3f;
        f16x8 ha, hb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * (lane + i)); hb[i] = (_Float16)(0.02f * (i + 1)); }
        const int n = iters * (mode == 1 || mode == 5 ? 55 : (mode == 2 ? 55 : (mode == 3 ? 55 : 20)));   // about as long as the victims run
        for (int rep = 0; rep < n; ++rep) {
            if (mode == 1) {
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i & 7], 0, 0, 0);
            } else if (mode == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 7], 0, 0, 0);
            } else if (mode == 3) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i & 7], 0, 0, 0);
            } else if (mode == 5) {   // the elimination's mix: a reciprocal, a quad broadcast, scalings, then a run of 4 x 4 x 1 products
                const float piv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc[0][0] + 1.f), 0x55, 0xf, 0xf, true));
                const float ninv = -__builtin_amdgcn_rcpf(piv);
                float m[7];
#pragma unroll
                for (int c = 0; c < 7; ++c) m[c] = acc[c][1] * ninv + a;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 28; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(m[i % 7], b, acc[i & 7], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if (mode == 4) {
#pragma unroll
                for (int i = 0; i < 64; ++i) acc[i & 7][i & 3] = __builtin_fmaf(a, b, acc[i & 7][i & 3]);
            }
        }
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (t == 123.456f) sink[threadIdx.x] = t;
    }
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned* d_err;
    float* d_sink;
    hipMalloc(&d_err, 8 * sizeof(unsigned));
    hipMalloc(&d_sink, 512 * sizeof(float));
    const char* names[] = {"no aggressor", "v_mfma_f32_4x4x1_16b_f32", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x32_f16", "v_fma_f32", "rcp + dpp + 4x4x1 mix"};
    for (int round = 0; round < 2; ++round)
        for (int mode = 0; mode < 6; ++mode) {
            hipMemset(d_err, 0, 8 * sizeof(unsigned));
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(repro_kernel, dim3(256), dim3(512), 0, 0, mode, iters, d_err, d_sink);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned h[8];
            hipMemcpy(h, d_err, sizeof(h), hipMemcpyDeviceToHost);
            // packed instructions executed by the victims: 256 CUs x 4 waves x iters x 24 x (4 x 8 + 8)
            const double npk = 256.0 * 4 * iters * 24 * 40;
            printf("round %d aggressor %-26s: wrong values %u (lanes affected %u) of %.3g packed wave-instructions, %.2f ms\n", round, names[mode],
                   h[0] + h[1] + h[2] + h[3], h[4], npk, ms);
        }
    return 0;
}
