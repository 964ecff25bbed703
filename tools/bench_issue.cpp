// Calibration microbenchmark (dev tool): cycles one wave needs per instruction, by instruction KIND, with 1 / 2 / 4 waves per
// SIMD resident -- the unit in which the mel-cepstral kernels (2 and 1 waves per SIMD, bound by vector issue) pay for their
// instruction mix.  Eight independent accumulators per wave, 20 000 iterations of 8 instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_issue.cpp -o build/bench_issue
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define OPS8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

#define KERNEL32(NAME, INSTR)                                                                                       \
    __global__ void NAME(float* out, unsigned long long* ticks, int iters)                                         \
    {                                                                                                               \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = a0 * 0.5f;                                                                                        \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                 \
        for (int i = 0; i < iters; ++i) asm volatile(REP8(INSTR) : OPS8 : "v"(b), "s"(-1L));                        \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                         \
        if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;                                                \
    }
#define KERNEL64(NAME, INSTR)                                                                                       \
    __global__ void NAME(float* out, unsigned long long* ticks, int iters)                                         \
    {                                                                                                               \
        v2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        v2 b = a0 * 0.5f;                                                                                           \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                 \
        for (int i = 0; i < iters; ++i) asm volatile(REP8(INSTR) : OPS8 : "v"(b), "s"(-1L));                        \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.x + a2.x + a3.x + a4.y + a5.y + a6.y + a7.y;         \
        if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;                                                \
    }

#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %" #n "\n"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %8, %8\n"
#define I_FMAC_DPP(n) "v_fmac_f32_dpp %" #n ", %8, %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
#define I_MOV_DPP(n) "v_mov_b32_dpp %" #n ", %8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_ADD_DPP_ROW(n) "v_add_f32_dpp %" #n ", %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_ADD_DPP_BCAST(n) "v_add_f32_dpp %" #n ", %8, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_ADD(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CNDMASK_S(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, %9\n"
#define I_EXP(n) "v_exp_f32 %" #n ", %8\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %8\n"
#define I_LDEXP(n) "v_ldexp_f32 %" #n ", %" #n ", %8\n"
#define I_MAX3(n) "v_max3_f32 %" #n ", %" #n ", %8, %8\n"
#define I_CVT_PK(n) "v_cvt_pk_f16_f32 %" #n ", %8, %8\n"
#define I_FMA_MIX(n) "v_fma_mix_f32 %" #n ", %8, -1.0, %" #n " op_sel_hi:[1,0,0]\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_ADD_U32(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_PK_FMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %" #n "\n"
#define I_PK_MUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define I_PK_ADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define I_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %8, %" #n "\n"
#define I_READLANE(n) "v_readlane_b32 s20, %" #n ", 3\n"

KERNEL32(k_fma, I_FMA)
KERNEL32(k_fmac, I_FMAC)
KERNEL32(k_fmac_dpp, I_FMAC_DPP)
KERNEL32(k_mov_dpp, I_MOV_DPP)
KERNEL32(k_add_dpp_row, I_ADD_DPP_ROW)
KERNEL32(k_add_dpp_bcast, I_ADD_DPP_BCAST)
KERNEL32(k_mov, I_MOV)
KERNEL32(k_add, I_ADD)
KERNEL32(k_cndmask, I_CNDMASK)
KERNEL32(k_cndmask_s, I_CNDMASK_S)
KERNEL32(k_exp, I_EXP)
KERNEL32(k_rcp, I_RCP)
KERNEL32(k_ldexp, I_LDEXP)
KERNEL32(k_max3, I_MAX3)
KERNEL32(k_cvt_pk, I_CVT_PK)
KERNEL32(k_fma_mix, I_FMA_MIX)
KERNEL32(k_and, I_AND)
KERNEL32(k_add_u32, I_ADD_U32)
KERNEL64(k_pk_fma, I_PK_FMA)
KERNEL64(k_pk_mul, I_PK_MUL)
KERNEL64(k_pk_add, I_PK_ADD)
KERNEL64(k_fma64, I_FMA64)

// matrix pipe: eight independent 16x16x32 binary16 products per iteration
__global__ void k_mfma(float* out, unsigned long long* ticks, int iters)
{
    v4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    h8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)(threadIdx.x * 0.01f + i);
        b[i] = (_Float16)(0.5f - i * 0.1f);
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c7, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

// a mix as in the elimination: 7 quad-broadcast FMAs per s_nop-free group versus 1 broadcast + 3 packed + 1 plain
__global__ void k_mix_dpp7(float* out, unsigned long long* ticks, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = a0 * 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i)
        asm volatile(I_FMAC_DPP(0) I_FMAC_DPP(1) I_FMAC_DPP(2) I_FMAC_DPP(3) I_FMAC_DPP(4) I_FMAC_DPP(5) I_FMAC_DPP(6) : OPS8 : "v"(b));
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}
__global__ void k_mix_pk5(float* out, unsigned long long* ticks, int iters)
{
    v2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f;
    v2 m = a0 * 0.25f, b = a0 * 0.5f;
    float s = threadIdx.x, mf = s * 0.125f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i)
        asm volatile("v_mov_b32_dpp %6, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                     "v_pk_fma_f32 %0, %3, %5, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %3, %5, %1 op_sel_hi:[0,1,1]\n"
                     "v_pk_fma_f32 %2, %3, %5, %2 op_sel_hi:[0,1,1]\n v_fmac_f32 %4, %6, %6\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(m), "+v"(s) : "v"(b), "v"(mf));
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.y + a2.x + m.x + s;
    if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <typename K>
static void run(const char* name, K kern, int per_iter, float* out, unsigned long long* ticks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    printf("%-34s", name);
    for (int wps : {1, 2, 4}) {
        float ms = 0;
        unsigned long long t = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(256 * wps), 0, 0, out, ticks, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        }
        // ticks of the 100 MHz-class counter are converted through the event time: cycles at the clock the run sustained
        printf(" | %dw: %6.2f ns/instr/wave %5.2f GHz", wps, ms * 1e6 / ((double)per_iter * iters), t / (ms * 1e6));
    }
    printf("\n");
}

int main()
{
    float* out;
    unsigned long long* ticks;
    hipMalloc(&out, 256 * 1024 * 4 * 4);
    hipMalloc(&ticks, 8);
    printf("ns per instruction per wave (every wave of the SIMD runs the same stream; 1 / 2 / 4 waves per SIMD); tick rate of s_memtime in GHz\n");
#define RUN(k) run(#k, k, 8, out, ticks)
    RUN(k_fma); RUN(k_fmac); RUN(k_fmac_dpp); RUN(k_mov_dpp); RUN(k_add_dpp_row); RUN(k_add_dpp_bcast); RUN(k_mov); RUN(k_add);
    RUN(k_cndmask); RUN(k_cndmask_s); RUN(k_exp); RUN(k_rcp); RUN(k_ldexp); RUN(k_max3); RUN(k_cvt_pk); RUN(k_fma_mix); RUN(k_and);
    RUN(k_add_u32); RUN(k_pk_fma); RUN(k_pk_mul); RUN(k_pk_add); RUN(k_fma64); RUN(k_mfma);
    run("k_mix_dpp7 (7 slots of work)", k_mix_dpp7, 7, out, ticks);
    run("k_mix_pk5 (7 slots of work)", k_mix_pk5, 7, out, ticks);
    return 0;
}
