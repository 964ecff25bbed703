// Minimal stand-alone reproducer (round 6) of the stale-result effect of DESIGN.md 4 on gfx950 (MI355X):
//
//   a packed float32 instruction whose LOW result half reads a HIGH source half (a SET op_sel bit) returns wrong values while ANOTHER
//   wave of its SIMD issues v_mfma_f32_16x16x32_f16 back to back from registers.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hazard/repro_min.cpp -o build/repro_min && build/repro_min
//
// A workgroup of 8 waves, two per SIMD (256 registers asked for, as the fused STFT -> mel-cepstrum kernel).  Waves 0..3 are VICTIMS: a
// loop of ONE packed instruction form on changing data, every result compared with the same arithmetic on one-component instructions.
// Waves 4..7 are AGGRESSORS: a loop of one instruction kind.  Everything is in registers: no LDS, no memory traffic in either loop.
// Output: wrong results per (victim form, aggressor kind).  What round 6 measured with it: profiles/r06_hazard_standalone_repro.txt.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// victim forms: r = f(a, b); `ref` is the same arithmetic on scalar instructions
template <int FORM>
__device__ __forceinline__ void victim_op(v2f a, v2f b, v2f& r, v2f& ref)
{
    if (FORM == 0) {          // crossed add: (a.lo + b.hi, a.hi - b.lo) -- the +-i rotation of the radix-4 butterfly
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        ref = v2f{a.x + b.y, a.y - b.x};
    } else if (FORM == 1) {   // high-half broadcast multiply: (a.lo * b.hi, a.hi * b.hi)
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
        ref = v2f{a.x * b.y, a.y * b.y};
    } else if (FORM == 2) {   // the complex product's second instruction: (-a.hi * b.hi + a.lo, a.lo * b.hi + a.hi)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(a));
        ref = v2f{__builtin_fmaf(-a.y, b.y, a.x), __builtin_fmaf(a.x, b.y, a.y)};
    } else if (FORM == 3) {   // PLAIN packed add (no modifier)
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        ref = v2f{a.x + b.x, a.y + b.y};
    } else if (FORM == 4) {   // low-half broadcast (op_sel_hi only: the HIGH result reads a LOW half): (a.lo * b.lo, a.hi * b.lo)
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
        ref = v2f{a.x * b.x, a.y * b.x};
    } else if (FORM == 5) {   // negations only: (a.lo - b.lo, a.hi + b.hi)
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        ref = v2f{a.x - b.x, a.y + b.y};
    } else if (FORM == 6) {   // a register-pair move with swapped halves (not arithmetic, same operand routing)
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(b), "v"(b));
        ref = v2f{b.y, b.y};
    }
}

template <int FORM>
__global__ __launch_bounds__(512, 2) void repro(int aggressor, int iters, unsigned* __restrict__ err, float* __restrict__ sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 4) {
        unsigned bad = 0, bad_low = 0, bad_rows3 = 0;
        v2f a = v2f{1.0f + 1e-3f * lane, 0.5f - 2e-3f * lane}, b = v2f{-0.25f + 3e-3f * lane, 2.0f + 1e-3f * lane};
        for (int rep = 0; rep < iters; ++rep) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                v2f r, ref;
                victim_op<FORM>(a, b, r, ref);
                const bool lo = __builtin_bit_cast(unsigned, r.x) != __builtin_bit_cast(unsigned, ref.x);
                const bool hi = __builtin_bit_cast(unsigned, r.y) != __builtin_bit_cast(unsigned, ref.y);
                bad += lo + hi;
                bad_low += lo;
                bad_rows3 += (lo || hi) && lane >= 48;
                // new data for the next instruction (bounded, never NaN): from the REFERENCE values so that an error does not propagate
                a = v2f{ref.y * 0.5f + 0.37f, ref.x * 0.25f - 0.11f};
                b = v2f{b.y * 0.75f + 0.01f * (float)i, b.x * 0.5f + 1.0f};
            }
        }
        if (bad) { atomicAdd(err, bad); atomicAdd(err + 1, bad_low); atomicAdd(err + 2, bad_rows3); }
        if (a.x + b.x == 123.456f) sink[threadIdx.x] = a.y;
    } else {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        f16x8 ha, hb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * (lane + i)); hb[i] = (_Float16)(0.02f * (i + 1)); }
        float x = 1.0f + lane * 1e-3f, y = 0.5f;
        for (int rep = 0; rep < iters; ++rep) {
            if (aggressor == 1) {          // binary16 products, eight independent accumulators
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i & 7], 0, 0, 0);
            } else if (aggressor == 2) {   // binary16 products into ONE accumulator
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[0], 0, 0, 0);
            } else if (aggressor == 3) {   // float32 4 x 4 x 1 products
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, acc[i & 7], 0, 0, 0);
            } else if (aggressor == 4) {   // float32 16 x 16 x 4 products
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i & 7], 0, 0, 0);
            } else if (aggressor == 5) {   // plain vector multiply-adds
#pragma unroll
                for (int i = 0; i < 64; ++i) acc[i & 7][i & 3] = __builtin_fmaf(x, y, acc[i & 7][i & 3]);
            } else if (aggressor == 6) {   // bfloat16 products (the other 16-bit input type of the same pipe)
                typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                bf16x8 ba, bb;
#pragma unroll
                for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(float)ha[i]; bb[i] = (__bf16)(float)hb[i]; }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[i & 7], 0, 0, 0);
            }
        }
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (t + x == 123.456f) sink[threadIdx.x] = t;
    }
}

template <int FORM>
static void run_form(const char* name, int iters, unsigned* d_err, float* d_sink)
{
    const char* agg[] = {"no second wave's work", "v_mfma_f32_16x16x32_f16 x 8 acc", "v_mfma_f32_16x16x32_f16, 1 acc", "v_mfma_f32_4x4x1_16b_f32",
                         "v_mfma_f32_16x16x4_f32", "v_fma_f32", "v_mfma_f32_16x16x32_bf16"};
    printf("victim: %s\n", name);
    for (int a = 0; a < 7; ++a) {
        hipMemset(d_err, 0, 16);
        hipLaunchKernelGGL(repro<FORM>, dim3(256), dim3(512), 0, 0, a, iters, d_err, d_sink);
        hipDeviceSynchronize();
        unsigned e[4];
        hipMemcpy(e, d_err, 16, hipMemcpyDeviceToHost);
        printf("    aggressor %-34s wrong halves %10u (low halves %10u; results in lanes 48..63 %10u) of %.3g results\n", agg[a], e[0], e[1], e[2],
               256.0 * 4 * 64 * 16 * iters);
    }
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    unsigned* d_err;
    float* d_sink;
    hipMalloc(&d_err, 16);
    hipMalloc(&d_sink, 4096);
    run_form<0>("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]   (crossed: the rotation)", iters, d_err, d_sink);
    run_form<1>("v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,1]                  (high-half broadcast)", iters, d_err, d_sink);
    run_form<2>("v_pk_fma_f32 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] (complex product)", iters, d_err, d_sink);
    run_form<3>("v_pk_add_f32                                               (plain)", iters, d_err, d_sink);
    run_form<4>("v_pk_mul_f32 op_sel_hi:[1,0]                               (low-half broadcast: op_sel_hi only)", iters, d_err, d_sink);
    run_form<5>("v_pk_add_f32 neg_lo:[0,1]                                  (negation only)", iters, d_err, d_sink);
    run_form<6>("v_pk_mov_b32 op_sel:[1,0]                                  (half-swapping move)", iters, d_err, d_sink);
    return 0;
}
