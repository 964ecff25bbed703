// Stand-alone reproducer attempt (round 5) for the stale-result effect of DESIGN.md 3.25, now that the reduction from the failing
// kernel (tools/hazard/) has named the instruction: v_pk_add_f32 with the halves of src1 crossed (op_sel:[0,1] op_sel_hi:[1,0], the
// +-i rotation of the radix-4 butterfly) delivers, in lanes 48..63, a low half computed from a stale high half of src1 -- only while
// ANOTHER wave with a different instruction stream shares the SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDSA_PK_CROSSED=1 -DDSA_PK_DBG=6 tools/hazard/repro_rot.cpp -o build/repro_rot && build/repro_rot
// (round 6: the product's helpers no longer emit the crossed forms by default -- DSA_PK_CROSSED=1 restores them for this tool; aggressor
//  kinds 10 .. 17 added: the instruction classes of the Newton phase that round 5's list left out)
//
// Victims (waves 0..3 of a workgroup, one per SIMD): the product's own 16-point transform with ONLY the rotations packed
// (DSA_PK_DBG=6: plain adds and the twiddle products on scalar instructions -- the variant with the highest rate inside the kernel,
// ~150 bad frames per 204 800) against its scalar twin on the same inputs, every lane-transform compared.  Aggressors (waves 4..7,
// the second wave of each SIMD): `mode` picks their instruction stream.  256 registers per wave as in the fused kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../diffsptk_amd/csrc/pk_math.h"
using namespace dsa;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 2) void repro(int mode, int iters, unsigned* __restrict__ err, const float* __restrict__ gsrc, float* __restrict__ sink)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];   // 96 KB: a table the aggressors stream + the victims' tiles
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    for (int i = threadIdx.x; i < 24576; i += 512) lds[i] = 1e-3f * (float)(i % 977);
    __syncthreads();
    if (wave < 4) {
        unsigned bad = 0, bad_hi_rows = 0;
        v2f* zf = reinterpret_cast<v2f*>(lds + 20480) + wave * 272;   // (shared by the four lane groups on purpose: values are not checked through it)
        for (int rep = 0; rep < iters; ++rep) {
            v2f v[16], s[16];
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                const v2f_u4 g = *reinterpret_cast<const v2f_u4*>(gsrc + ((rep * 131 + lane * 26 + 2 * i) & 65535));   // samples from memory, as the prologue
                v[i] = s[i] = v2f{g.x + (float)i, g.y - (float)lane};
            }
#pragma unroll
            for (int i = 13; i < 16; ++i) v[i] = s[i] = v2f{0.f, 0.f};
            pk_fft16<true>(v);
            sc_fft16<true>(s);
            unsigned b = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) b += (__builtin_bit_cast(unsigned, v[i].x) != __builtin_bit_cast(unsigned, s[i].x)) + (__builtin_bit_cast(unsigned, v[i].y) != __builtin_bit_cast(unsigned, s[i].y));
            bad += b;
            bad_hi_rows += (b && lane >= 48) ? 1 : 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) zf[(i * 17 + (lane & 15)) & 255] = v[i];   // the transposition's stores
            __builtin_amdgcn_wave_barrier();
        }
        if (bad) { atomicAdd(err, bad); atomicAdd(err + 1, 1u); }
        if (bad_hi_rows) atomicAdd(err + 2, bad_hi_rows);
    } else {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float a = 1.0f + lane * 1e-3f, b = 0.5f - lane * 1e-3f, t = 0.f;
        f16x8 ha, hb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * (lane + i)); hb[i] = (_Float16)(0.02f * (i + 1)); }
        const f16x8* img = reinterpret_cast<const f16x8*>(lds) + lane;
        const int n = iters * 4;
        for (int rep = 0; rep < n; ++rep) {
            const int m = mode == 9 ? 1 + (rep % 8) : (mode == 18 ? 1 + (rep % 17) : mode);
            if (m == 19) {   // the dependent binary16 chain WITHOUT the s_nop
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[0], 0, 0, 0);
            } else if (m == 20) {   // s_nop 7 alone
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("s_nop 7");
            } else if (m == 21) {   // a dependent chain of 4 x 4 x 1 products + s_nop
#pragma unroll
                for (int i = 0; i < 8; ++i) { acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[0], 0, 0, 0); asm volatile("s_nop 7"); }
            } else if (m == 22) {   // INDEPENDENT binary16 products (eight accumulators) + s_nop
#pragma unroll
                for (int i = 0; i < 8; ++i) { acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0); asm volatile("s_nop 7"); }
            } else if (m == 23) {   // the dependent binary16 chain + s_nop 0
#pragma unroll
                for (int i = 0; i < 8; ++i) { acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[0], 0, 0, 0); asm volatile("s_nop 0"); }
            }
            if (m == 1) {   // binary16 products fed from LDS, as the chains
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(img[(i * 64 + rep * 8) & 4095], hb, acc[i & 7], 0, 0, 0);
            } else if (m == 2) {   // transcendentals
#pragma unroll
                for (int i = 0; i < 16; ++i) a = __builtin_amdgcn_exp2f(a * 0.5f) + __builtin_amdgcn_rcpf(b + (float)i);
            } else if (m == 3) {   // 4 x 4 x 1 products
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i & 7], 0, 0, 0);
            } else if (m == 4) {   // the compiler's packed multiply-adds
                typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int i = 0; i < 16; ++i) { f2 x = {acc[i & 7][0], acc[i & 7][1]}; x = x * f2{a, a} + f2{b, b}; acc[i & 7][0] = x[0]; acc[i & 7][1] = x[1]; }
            } else if (m == 5) {   // LDS traffic
#pragma unroll
                for (int i = 0; i < 8; ++i) { const f32x4 r = *reinterpret_cast<const f32x4*>(lds + ((lane * 4 + i * 256 + rep * 64) & 16383)); acc[i] += r; }
            } else if (m == 6) {   // DPP + conversions
#pragma unroll
                for (int i = 0; i < 16; ++i) { a += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(b), 0x4e, 0xf, 0xf, true)); b = __builtin_ldexpf(a, -1) + (float)(_Float16)a; }
            } else if (m == 7) {   // memory loads
#pragma unroll
                for (int i = 0; i < 4; ++i) t += gsrc[(rep * 577 + lane + i * 8191) & 65535];
            } else if (m == 8) {   // plain multiply-adds
#pragma unroll
                for (int i = 0; i < 64; ++i) acc[i & 7][i & 3] = __builtin_fmaf(a, b, acc[i & 7][i & 3]);
            } else if (m == 10) {   // cross-row swaps (the reductions of the chains: v_permlane16_swap / v_permlane32_swap)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
                    a = __uint_as_float(p[0]) + 1e-3f; b = __uint_as_float(p[1]) * 0.999f;
                    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
                    a = __uint_as_float(q[0]); b = __uint_as_float(q[1]);
                }
            } else if (m == 11) {   // the binary16 split: v_cvt_pk_f16_f32 + v_fma_mixlo / mixhi_f16
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    h2 hi = {(_Float16)a, (_Float16)b};
                    unsigned lo;
                    asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\ts_nop 0\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0"
                                 : "=&v"(lo) : "v"(hi), "v"(a), "v"(b));
                    a += 1e-3f * (float)(lo & 3); b -= 1e-3f * (float)hi[0];
                }
            } else if (m == 12) {   // (moves between the two register files: not built -- with no accumulation registers allocated to
                                    //  the kernel the instruction faults; the fused kernel's own v_accvgpr moves come with its allocation)
            } else if (m == 13) {   // LDS writes + wave barriers (the windows of the solve)
#pragma unroll
                for (int i = 0; i < 8; ++i) { lds[16384 + ((wave * 64 + lane + 68 * i) & 4095)] = a + (float)i; __builtin_amdgcn_wave_barrier(); a += lds[16384 + ((wave * 64 + lane + 68 * i + 4) & 4095)]; }
            } else if (m == 14) {   // global stores (the history rows)
#pragma unroll
                for (int i = 0; i < 4; ++i) sink[4096 + ((blockIdx.x * 512 + threadIdx.x + 1024 * i) & 65535)] = a + (float)i;
            } else if (m == 15) {   // float32 16 x 16 x 4 products (the Nyquist preload of the second chain)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 7], 0, 0, 0);
            } else if (m == 16) {   // v_log_f32, v_ldexp_f32, v_frexp_exp_i32_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) { a = __builtin_amdgcn_logf(__builtin_fabsf(a) + 1.5f) + __builtin_ldexpf(b, -(__builtin_amdgcn_frexp_expf(a + 3.f) & 3)); }
            } else if (m == 17) {   // a dependent chain of binary16 products into ONE accumulator (waits out the matrix pipe) + s_nop
#pragma unroll
                for (int i = 0; i < 8; ++i) { acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[0], 0, 0, 0); asm volatile("s_nop 7"); }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (t + a + b == 123.456f) sink[threadIdx.x] = t;
    }
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned* d_err; float *d_sink, *d_src;
    hipMalloc(&d_err, 16); hipMalloc(&d_sink, (65536 + 8192) * 4); hipMalloc(&d_src, 65536 * 4 + 64);
    float* h = (float*)malloc(65536 * 4 + 64);
    for (int i = 0; i < 65536 + 16; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(d_src, h, 65536 * 4 + 64, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)repro, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    const char* names[] = {"none", "binary16 products from LDS", "v_exp / v_rcp", "v_mfma_f32_4x4x1", "compiler v_pk_fma_f32", "LDS b128 reads", "DPP + conversions",
                           "memory loads", "v_fma_f32", "1 .. 8 in turn", "v_permlane16/32_swap", "binary16 split (mixlo / mixhi)", "(skipped)",
                           "LDS writes + wave barriers", "global stores", "v_mfma_f32_16x16x4_f32", "v_log / ldexp / frexp", "dependent binary16 chain + s_nop",
                           "1 .. 17 in turn", "dependent binary16 chain, no s_nop", "s_nop 7 alone", "dependent 4x4x1 chain + s_nop 7",
                           "independent binary16 products + s_nop 7", "dependent binary16 chain + s_nop 0"};
    const int first = argc > 2 ? atoi(argv[2]) : 0;
    for (int mode = first; mode < 24; ++mode) {
        hipMemset(d_err, 0, 16);
        hipLaunchKernelGGL(repro, dim3(256), dim3(512), 98304, 0, mode, iters, d_err, d_src, d_sink);
        hipDeviceSynchronize();
        unsigned e[4];
        hipMemcpy(e, d_err, 16, hipMemcpyDeviceToHost);
        printf("aggressor %-28s: wrong values %u in %u lanes (lane-transforms in lanes 48..63: %u) of %.3g lane-transforms\n", names[mode], e[0], e[1], e[2],
               256.0 * 4 * 64 * iters);
    }
    return 0;
}
