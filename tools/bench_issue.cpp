// Issue-rate calibration (dev tool, round 3; replaces the round-2 tool whose 8-instruction loop body measured the branch
// bubble and whose single recorded wave measured wave 0's share of the arbiter, not the clock).
//
// What it measures: cycles per wave64 instruction, by instruction KIND, with 1 / 2 / 4 / 8 waves resident per SIMD on every
// CU of the chip, separated from the clock.
//   * loop body = 256 instructions (16 independent accumulators x 16 repeats) + 3 scalar instructions of loop control;
//   * EVERY wave stores its own s_memtime start / end; the host reports the mean / min / max of the per-wave tick counts and
//     the span (last end - first start) next to the wall time of the launch (HIP events);
//   * operands are random finite values chosen so that accumulators neither overflow nor denormalise (the data-dependent
//     power effect of accumulators running to inf made v_fmac look slower than v_fma in the old tool);
//   * the number of resident waves is fixed by the dynamic LDS size (160 KB / waves-per-SIMD per 256-thread workgroup) and
//     grid = 256 CUs x waves-per-SIMD, so every wave of the launch is resident from start to end;
//   * run it under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES` (tools/issue_pmc.sh) to get
//     instructions / GRBM_GUI_ACTIVE: the clock-independent issue rate; tools/issue_pmc_summary.py joins the two.
// It also answers the design questions of the mel-cepstral solve: what a v_mfma_f32_4x4x1_16b_f32 costs, whether it issues
// beside vector instructions of the same wave / of another wave of the SIMD, what its dependent latency is, and its lane layout.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_issue.cpp -o build/bench_issue
//   build/bench_issue [kind-substring] [iters]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));              \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

// 16 accumulators a0..a15 are asm operands %0..%15; %16, %17 are the two sources
#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)
#define OPS16                                                                                                          \
    "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), \
        "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
#define BODY256(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S)

__device__ __forceinline__ unsigned rnd(unsigned x)
{
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    return x;
}
// random in [lo, hi)
__device__ __forceinline__ float rndf(unsigned& s, float lo, float hi)
{
    s = rnd(s);
    return lo + (hi - lo) * (float)(s >> 8) * (1.f / 16777216.f);
}

struct Stamp {
    unsigned long long t0, t1;
};
__device__ __forceinline__ void stamp(Stamp* st, unsigned long long t0, unsigned long long t1)
{
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        st[w].t0 = t0;
        st[w].t1 = t1;
    }
}

// T = float / v2 / v4 accumulators.  b in [0.4, 0.6), c in [0.5, 1.5): acc <- acc * b + c converges to c / (1 - b)
#define KERNEL(NAME, T, INSTR, INIT)                                                                   \
    __global__ __launch_bounds__(256) void NAME(float* out, Stamp* st, int iters)                      \
    {                                                                                                  \
        extern __shared__ float dyn_lds[];                                                             \
        unsigned s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);                               \
        T a[16];                                                                                       \
        for (int i = 0; i < 16; ++i) a[i] = INIT;                                                      \
        T b = INIT * 0.f + rndf(s, 0.4f, 0.6f), c = INIT * 0.f + rndf(s, 0.5f, 1.5f);                  \
        if (iters < 0) dyn_lds[threadIdx.x] = 0.f;                                                     \
        const unsigned long long t0 = __builtin_readcyclecounter();                                    \
        for (int i = 0; i < iters; ++i) asm volatile(BODY256(INSTR) : OPS16 : "v"(b), "v"(c));         \
        const unsigned long long t1 = __builtin_readcyclecounter();                                    \
        T r = a[0];                                                                                    \
        for (int i = 1; i < 16; ++i) r += a[i];                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = ((float*)&r)[0];                                         \
        stamp(st, t0, t1);                                                                             \
    }

#define F1 (rndf(s, 0.5f, 1.5f))
#define F2 (v2{rndf(s, 0.5f, 1.5f), rndf(s, 0.5f, 1.5f)})
#define F4 (v4{rndf(s, 0.5f, 1.5f), rndf(s, 0.5f, 1.5f), rndf(s, 0.5f, 1.5f), rndf(s, 0.5f, 1.5f)})

#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %16, %17\n"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %16, %17\n"   /* acc += b c: grows by < 1 per instruction, 1e6 instructions: finite */
#define I_FMAC_DPP(n) "v_fmac_f32_dpp %" #n ", %16, %17 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
#define I_MOV_DPP(n) "v_mov_b32_dpp %" #n ", %16 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_ADD_DPP(n) "v_add_f32_dpp %" #n ", %16, %17 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %16\n"
#define I_ADD(n) "v_add_f32 %" #n ", %16, %17\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %16, %17\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %16, %17, vcc\n"
#define I_EXP(n) "v_exp_f32 %" #n ", %16\n"
#define I_LOG(n) "v_log_f32 %" #n ", %17\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %17\n"
#define I_LDEXP(n) "v_ldexp_f32 %" #n ", %17, 3\n"
#define I_MAX3(n) "v_max3_f32 %" #n ", %" #n ", %16, %17\n"
#define I_CVT_PK(n) "v_cvt_pk_f16_f32 %" #n ", %16, %17\n"
#define I_FMA_MIX(n) "v_fma_mix_f32 %" #n ", %16, -1.0, %17 op_sel_hi:[1,0,0]\n"
#define I_AND(n) "v_and_b32 %" #n ", %16, %17\n"
#define I_ADD_U32(n) "v_add_u32 %" #n ", %16, %17\n"
#define I_PK_FMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %16, %17\n"
#define I_PK_MUL(n) "v_pk_mul_f32 %" #n ", %16, %17\n"
#define I_PK_ADD(n) "v_pk_add_f32 %" #n ", %16, %17\n"
#define I_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %16, %17\n"

KERNEL(k_fma, float, I_FMA, F1)
KERNEL(k_fmac, float, I_FMAC, F1)
KERNEL(k_fmac_dpp, float, I_FMAC_DPP, F1)
KERNEL(k_mov_dpp, float, I_MOV_DPP, F1)
KERNEL(k_add_dpp, float, I_ADD_DPP, F1)
KERNEL(k_mov, float, I_MOV, F1)
KERNEL(k_add, float, I_ADD, F1)
KERNEL(k_mul, float, I_MUL, F1)
KERNEL(k_cndmask, float, I_CNDMASK, F1)
KERNEL(k_exp, float, I_EXP, F1)
KERNEL(k_log, float, I_LOG, F1)
KERNEL(k_rcp, float, I_RCP, F1)
KERNEL(k_ldexp, float, I_LDEXP, F1)
KERNEL(k_max3, float, I_MAX3, F1)
KERNEL(k_cvt_pk, float, I_CVT_PK, F1)
KERNEL(k_fma_mix, float, I_FMA_MIX, F1)
KERNEL(k_and, float, I_AND, F1)
KERNEL(k_add_u32, float, I_ADD_U32, F1)
KERNEL(k_pk_fma, v2, I_PK_FMA, F2)
KERNEL(k_pk_mul, v2, I_PK_MUL, F2)
KERNEL(k_pk_add, v2, I_PK_ADD, F2)

// dependent chains: ILP 1 / 2 / 4 (the same 256 instructions per iteration, fewer accumulators)
#define I_FMA_D1(n) "v_fma_f32 %0, %0, %16, %17\n"
#define I_FMA_D2(n) "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n"
#define I_FMA_D4(n) "v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REP4(S) S(0) S(1) S(2) S(3)
#define REP16_D1(S) REP16(S)
#define REP16_D2(S) REP8(S)
#define REP16_D4(S) REP4(S)
#define BODY_DEP(R, S) R(S) R(S) R(S) R(S) R(S) R(S) R(S) R(S) R(S) R(S) R(S) R(S) R(S) R(S) R(S) R(S)
#define KERNEL_DEP(NAME, R, INSTR)                                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, Stamp* st, int iters)                      \
    {                                                                                                  \
        extern __shared__ float dyn_lds[];                                                             \
        unsigned s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);                               \
        float a[16];                                                                                   \
        for (int i = 0; i < 16; ++i) a[i] = F1;                                                        \
        float b = rndf(s, 0.4f, 0.6f), c = rndf(s, 0.5f, 1.5f);                                        \
        if (iters < 0) dyn_lds[threadIdx.x] = 0.f;                                                     \
        const unsigned long long t0 = __builtin_readcyclecounter();                                    \
        for (int i = 0; i < iters; ++i) asm volatile(BODY_DEP(R, INSTR) : OPS16 : "v"(b), "v"(c));     \
        const unsigned long long t1 = __builtin_readcyclecounter();                                    \
        float r = a[0];                                                                                \
        for (int i = 1; i < 16; ++i) r += a[i];                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                       \
        stamp(st, t0, t1);                                                                             \
    }
KERNEL_DEP(k_fma_dep1, REP16_D1, I_FMA_D1)
KERNEL_DEP(k_fma_dep2, REP16_D2, I_FMA_D2)
KERNEL_DEP(k_fma_dep4, REP16_D4, I_FMA_D4)
// a quad-broadcast multiply-add whose DPP source was written by the PREVIOUS instruction's neighbour (2 wait states needed:
// the chain of the elimination: scale the pivot row, s_nop 1, then DPP readers)
#define I_DPP_CHAIN(n) "v_mul_f32 %1, %16, %17\n s_nop 1\n v_fmac_f32_dpp %0, %1, %17 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
KERNEL_DEP(k_dpp_after_write, REP16_D1, I_DPP_CHAIN)

// ---- matrix pipe ----
#define I_MFMA4(n) "v_mfma_f32_4x4x1_16b_f32 %" #n ", %16, %17, %" #n "\n"
#define I_MFMA16F32(n) "v_mfma_f32_16x16x4_f32 %" #n ", %16, %17, %" #n "\n"
// acc += a b with a, b ~ 0.5, 1: grows by < 1.5 per instruction
#define KERNEL_MFMA(NAME, INSTR, NINSTR_NOTE)                                                          \
    __global__ __launch_bounds__(256) void NAME(float* out, Stamp* st, int iters)                      \
    {                                                                                                  \
        extern __shared__ float dyn_lds[];                                                             \
        unsigned s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);                               \
        v4 a[16];                                                                                      \
        for (int i = 0; i < 16; ++i) a[i] = F4;                                                        \
        float b = rndf(s, 0.4f, 0.6f), c = rndf(s, 0.5f, 1.5f);                                        \
        if (iters < 0) dyn_lds[threadIdx.x] = 0.f;                                                     \
        const unsigned long long t0 = __builtin_readcyclecounter();                                    \
        for (int i = 0; i < iters; ++i) asm volatile(INSTR : OPS16 : "v"(b), "v"(c));                  \
        const unsigned long long t1 = __builtin_readcyclecounter();                                    \
        v4 r = a[0];                                                                                   \
        for (int i = 1; i < 16; ++i) r += a[i];                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3];                               \
        stamp(st, t0, t1);                                                                             \
    }
KERNEL_MFMA(k_mfma_4x4x1, BODY256(I_MFMA4), 256)
KERNEL_MFMA(k_mfma_16x16x4_f32, BODY256(I_MFMA16F32), 256)
// dependent accumulate chain on ONE accumulator (srcC = vdst of the previous one: the hardware interlocks this case)
#define I_MFMA4_D1(n) "v_mfma_f32_4x4x1_16b_f32 %0, %16, %17, %0\n"
#define I_MFMA4_D2(n) "v_mfma_f32_4x4x1_16b_f32 %0, %16, %17, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %16, %17, %1\n"
KERNEL_MFMA(k_mfma_4x4x1_dep1, BODY_DEP(REP16_D1, I_MFMA4_D1), 256)
KERNEL_MFMA(k_mfma_4x4x1_dep2, BODY_DEP(REP16_D2, I_MFMA4_D2), 256)
// one 4x4x1 product and k independent vector multiply-adds per slot, same wave (does the vector instruction issue in the
// shadow of the product?): 128 products + 128 k vector instructions per iteration
#define BODY128(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S) REP16(S)
#define KERNEL_MIX(NAME, INSTR)                                                                        \
    __global__ __launch_bounds__(256) void NAME(float* out, Stamp* st, int iters)                      \
    {                                                                                                  \
        extern __shared__ float dyn_lds[];                                                             \
        unsigned s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);                               \
        v4 a[16];                                                                                      \
        for (int i = 0; i < 16; ++i) a[i] = F4;                                                        \
        float b = rndf(s, 0.4f, 0.6f), c = rndf(s, 0.5f, 1.5f);                                        \
        float f0 = F1, f1 = F1, f2 = F1, f3 = F1;                                                      \
        if (iters < 0) dyn_lds[threadIdx.x] = 0.f;                                                     \
        const unsigned long long t0 = __builtin_readcyclecounter();                                    \
        for (int i = 0; i < iters; ++i)                                                                \
            asm volatile(BODY128(INSTR) : OPS16, "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b), "v"(c)); \
        const unsigned long long t1 = __builtin_readcyclecounter();                                    \
        v4 r = a[0];                                                                                   \
        for (int i = 1; i < 16; ++i) r += a[i];                                                        \
        out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + f0 + f1 + f2 + f3;           \
        stamp(st, t0, t1);                                                                             \
    }
// operand numbering inside KERNEL_MIX: %0..%15 accumulators, %16..%19 f0..f3, %20 b, %21 c
#define J_MIX0(n) "v_mfma_f32_4x4x1_16b_f32 %" #n ", %20, %21, %" #n "\n"
#define J_MIX1(n) J_MIX0(n) "v_fma_f32 %16, %16, %20, %21\n"
#define J_MIX2(n) J_MIX1(n) "v_fma_f32 %17, %17, %20, %21\n"
#define J_MIX4(n) J_MIX2(n) "v_fma_f32 %18, %18, %20, %21\n v_fma_f32 %19, %19, %20, %21\n"
KERNEL_MIX(k_mix_mfma4_only128, J_MIX0)
KERNEL_MIX(k_mix_mfma4_valu1, J_MIX1)
KERNEL_MIX(k_mix_mfma4_valu2, J_MIX2)
KERNEL_MIX(k_mix_mfma4_valu4, J_MIX4)
#define J_VALU4(n) "v_fma_f32 %16, %16, %20, %21\n v_fma_f32 %17, %17, %20, %21\n v_fma_f32 %18, %18, %20, %21\n v_fma_f32 %19, %19, %20, %21\n"
KERNEL_MIX(k_mix_valu4_only, J_VALU4)

// binary16 products (the chains of the mel-cepstral kernel) alone, and alternating with 4x4x1 float32 products: one pipe or two?
__global__ __launch_bounds__(256) void k_mfma_f16(float* out, Stamp* st, int iters)
{
    extern __shared__ float dyn_lds[];
    unsigned s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);
    v4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = F4;
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)rndf(s, 0.4f, 0.6f);
        b[i] = (_Float16)rndf(s, 0.01f, 0.02f);
    }
    if (iters < 0) dyn_lds[threadIdx.x] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            c[r & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[r & 7], 0, 0, 0);
            asm volatile("" : "+v"(c[r & 7]));
        }
    const unsigned long long t1 = __builtin_readcyclecounter();
    v4 r = c[0];
    for (int i = 1; i < 8; ++i) r += c[i];
    out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    stamp(st, t0, t1);
}
__global__ __launch_bounds__(256) void k_mfma_f16_plus_4x4(float* out, Stamp* st, int iters)
{
    extern __shared__ float dyn_lds[];
    unsigned s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);
    v4 c[8], d[8];
    for (int i = 0; i < 8; ++i) c[i] = F4, d[i] = F4;
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)rndf(s, 0.4f, 0.6f);
        b[i] = (_Float16)rndf(s, 0.01f, 0.02f);
    }
    const float fa = rndf(s, 0.4f, 0.6f), fb = rndf(s, 0.5f, 1.5f);
    if (iters < 0) dyn_lds[threadIdx.x] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int r = 0; r < 32; ++r) {   // 32 binary16 + 32 float32 4x4x1 products per iteration
            c[r & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[r & 7], 0, 0, 0);
            d[r & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(fa, fb, d[r & 7], 0, 0, 0);
            asm volatile("" : "+v"(c[r & 7]), "+v"(d[r & 7]));
        }
    const unsigned long long t1 = __builtin_readcyclecounter();
    v4 r = c[0] + d[0];
    for (int i = 1; i < 8; ++i) r += c[i] + d[i];
    out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    stamp(st, t0, t1);
}

// two KINDS of wave on one SIMD: 512-thread workgroups, waves 0-3 run 4x4x1 products, waves 4-7 vector multiply-adds (waves w
// and w + 4 share a SIMD).  mode 0: both; 1: only the product waves work; 2: only the vector waves work.
__global__ __launch_bounds__(512) void k_pair_mfma4_valu(float* out, Stamp* st, int iters, int mode)
{
    extern __shared__ float dyn_lds[];
    unsigned s = 0x9E3779B9u * (blockIdx.x * 512 + threadIdx.x + 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float b = rndf(s, 0.4f, 0.6f), c = rndf(s, 0.5f, 1.5f);
    if (iters < 0) dyn_lds[threadIdx.x] = 0.f;
    float res = 0.f;
    unsigned long long t0, t1;
    if (wave < 4) {
        v4 a[16];
        for (int i = 0; i < 16; ++i) a[i] = F4;
        const int n = mode == 2 ? 0 : iters;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n; ++i) asm volatile(BODY256(I_MFMA4) : OPS16 : "v"(b), "v"(c));
        t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 16; ++i) res += a[i][0] + a[i][3];
    } else {
        float a[16];
        for (int i = 0; i < 16; ++i) a[i] = F1;
        const int n = mode == 1 ? 0 : iters;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < n; ++i) asm volatile(BODY256(I_FMA) : OPS16 : "v"(b), "v"(c));
        t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 16; ++i) res += a[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
    stamp(st, t0, t1);
}

// lane layout of v_mfma_f32_4x4x1_16b_f32: A = lane + 1, B = 100 (lane + 1), C = 0
__global__ void k_layout(float* out)
{
    const float a = threadIdx.x + 1.f, b = 100.f * (threadIdx.x + 1.f);
    v4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = c[i];
}

struct Result {
    double wall_ms, ticks_mean, ticks_min, ticks_max, span;
};

template <typename L>
static Result measure(L launch, int nwaves, Stamp* st_d)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<Stamp> st(nwaves);
    Result best{1e30, 0, 0, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(st_d, 0, nwaves * sizeof(Stamp)));
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipGetLastError());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0) continue;   // clock ramp
        CHECK(hipMemcpy(st.data(), st_d, nwaves * sizeof(Stamp), hipMemcpyDeviceToHost));
        double sum = 0, mn = 1e30, mx = 0;
        unsigned long long first = ~0ull, last = 0;
        int cnt = 0;
        for (auto& q : st) {
            const double d = (double)(q.t1 - q.t0);
            if (d < 1000) continue;   // idle waves of the pair kernel
            sum += d; mn = std::min(mn, d); mx = std::max(mx, d);
            first = std::min(first, q.t0); last = std::max(last, q.t1);
            ++cnt;
        }
        if (ms < best.wall_ms) best = Result{ms, sum / std::max(cnt, 1), mn, mx, (double)(last - first)};
    }
    return best;
}

int main(int argc, char** argv)
{
    const char* filter = argc > 1 ? argv[1] : "";
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    float* out;
    Stamp* st;
    CHECK(hipMalloc(&out, 256 * 8 * 512 * sizeof(float)));
    CHECK(hipMalloc(&st, 256 * 8 * 8 * sizeof(Stamp)));

    {   // lane layout
        CHECK(hipMemset(out, 0, 256 * sizeof(float)));
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out);
        float h[256];
        CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const int blk = l >> 2, j = l & 3;
                const float want = (4 * blk + i + 1.f) * 100.f * (4 * blk + j + 1.f);   // D[blk][i][j] = A[blk][i] B[blk][j] in register i of lane 4 blk + j
                if (h[l * 4 + i] != want) ++bad;
            }
        printf("layout v_mfma_f32_4x4x1_16b_f32: D[blk][i][j] in register i of lane 4 blk + j, A[blk][i] on lane 4 blk + i, B[blk][j] on lane 4 blk + j: %s (%d mismatches); lane 5 regs = %g %g %g %g\n",
               bad ? "NO" : "yes", bad, h[20], h[21], h[22], h[23]);
    }

    struct K {
        const char* name;
        void (*fn)(float*, Stamp*, int);
        int vec_per_iter;     // vector-ALU instructions per loop iteration
        int mat_per_iter;     // matrix instructions per loop iteration
    };
    const K kinds[] = {
        {"v_fma_f32", k_fma, 256, 0}, {"v_fmac_f32", k_fmac, 256, 0}, {"v_fmac_f32_dpp(quad_perm)", k_fmac_dpp, 256, 0},
        {"v_mov_b32_dpp", k_mov_dpp, 256, 0}, {"v_add_f32_dpp", k_add_dpp, 256, 0}, {"v_mov_b32", k_mov, 256, 0},
        {"v_add_f32", k_add, 256, 0}, {"v_mul_f32", k_mul, 256, 0}, {"v_cndmask_b32(vcc)", k_cndmask, 256, 0},
        {"v_exp_f32", k_exp, 256, 0}, {"v_log_f32", k_log, 256, 0}, {"v_rcp_f32", k_rcp, 256, 0}, {"v_ldexp_f32", k_ldexp, 256, 0},
        {"v_max3_f32", k_max3, 256, 0}, {"v_cvt_pk_f16_f32", k_cvt_pk, 256, 0}, {"v_fma_mix_f32", k_fma_mix, 256, 0},
        {"v_and_b32", k_and, 256, 0}, {"v_add_u32", k_add_u32, 256, 0}, {"v_pk_fma_f32", k_pk_fma, 256, 0},
        {"v_pk_mul_f32", k_pk_mul, 256, 0}, {"v_pk_add_f32", k_pk_add, 256, 0},
        {"v_fma_f32 dependent (ILP 1)", k_fma_dep1, 256, 0}, {"v_fma_f32 ILP 2", k_fma_dep2, 256, 0}, {"v_fma_f32 ILP 4", k_fma_dep4, 256, 0},
        {"v_mul; s_nop 1; v_fmac_dpp of it", k_dpp_after_write, 512, 0},
        {"mfma_f32_4x4x1_16b (16 accumulators)", k_mfma_4x4x1, 0, 256}, {"mfma_f32_4x4x1 dependent (1 acc)", k_mfma_4x4x1_dep1, 0, 256},
        {"mfma_f32_4x4x1 2 accumulators", k_mfma_4x4x1_dep2, 0, 256}, {"mfma_f32_16x16x4_f32", k_mfma_16x16x4_f32, 0, 256},
        {"mfma_f32_16x16x32_f16 (8 acc)", k_mfma_f16, 0, 32}, {"f16 product + 4x4x1 alternating", k_mfma_f16_plus_4x4, 0, 64},
        {"mix: 128 x 4x4x1 alone", k_mix_mfma4_only128, 0, 128}, {"mix: 128 x (4x4x1 + 1 v_fma)", k_mix_mfma4_valu1, 128, 128},
        {"mix: 128 x (4x4x1 + 2 v_fma)", k_mix_mfma4_valu2, 256, 128}, {"mix: 128 x (4x4x1 + 4 v_fma)", k_mix_mfma4_valu4, 512, 128},
        {"mix: 128 x 4 v_fma alone (4 acc)", k_mix_valu4_only, 512, 0},
    };
    printf("iters %d; per cell: wall ms | cycles per instruction and WAVE (mean of every wave's own s_memtime ticks; min..max) | per SIMD = per wave / waves | span ticks / wall = counter GHz\n", iters);
    for (const K& k : kinds) {
        if (*filter && !strstr(k.name, filter)) continue;
        printf("%-38s", k.name);
        for (int wps : {1, 2, 4, 8}) {
            const int lds = (160 * 1024 / wps) & ~255;
            CHECK(hipFuncSetAttribute((const void*)k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            const int grid = 256 * wps;
            const Result r = measure([&] { hipLaunchKernelGGL(k.fn, dim3(grid), dim3(256), lds, 0, out, st, iters); }, grid * 4, st);
            const double n = (double)(k.vec_per_iter + k.mat_per_iter) * iters;
            printf(" | %dw %7.3f ms %6.2f (%5.2f..%5.2f) simd %5.2f  %4.2f GHz", wps, r.wall_ms, r.ticks_mean / n, r.ticks_min / n,
                   r.ticks_max / n, r.ticks_mean / n / wps, r.span / (r.wall_ms * 1e6));
        }
        printf("\n");
    }
    if (!*filter || strstr("pair", filter)) {
        const int lds = 160 * 1024;
        CHECK(hipFuncSetAttribute((const void*)k_pair_mfma4_valu, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int mode = 0; mode < 3; ++mode) {
            const Result r = measure([&] { hipLaunchKernelGGL(k_pair_mfma4_valu, dim3(256), dim3(512), lds, 0, out, st, iters, mode); }, 256 * 8, st);
            printf("pair (one 4x4x1 wave + one v_fma wave per SIMD), %s: wall %7.3f ms, working waves: %6.2f cycles per instruction (min %5.2f max %5.2f)\n",
                   mode == 0 ? "both working" : (mode == 1 ? "product waves only" : "vector waves only"), r.wall_ms,
                   r.ticks_mean / (256.0 * iters), r.ticks_min / (256.0 * iters), r.ticks_max / (256.0 * iters));
        }
    }
    return 0;
}
