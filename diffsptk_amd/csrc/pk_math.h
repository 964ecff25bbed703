// Packed-float32 complex arithmetic on register pairs (inline assembly with VOP3P operand modifiers) and the 16-point
// register FFT built from it: shared by the packed STFT kernels (stft_pk.h, stft_bwd_pk.h) and the STFT prologue of the fused
// STFT -> mel-cepstrum kernel (mcep_mfma_f16.h).  Modifier semantics are checked by tools/test_pk_asm.cpp.
#pragma once

#ifndef FFT16_OUT
#define FFT16_OUT(k) (4 * ((k)&3) + ((k) >> 2))   // register that holds X[k] after fft16 / pk_fft16
#endif

namespace dsa {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f_u4 __attribute__((ext_vector_type(2), aligned(4)));   // a pair of floats at a 4-byte aligned address

// ---- packed complex helpers: (lo, hi) = (re, im).  Modifier semantics checked by tools/test_pk_asm.cpp. ----
__device__ __forceinline__ v2f pk_add(v2f a, v2f b)
{
    v2f r;
#if defined(DSA_PK_DBG) && (DSA_PK_DBG & 4)
    asm("v_add_f32 %0, %1, %2" : "=v"(r.x) : "v"(a.x), "v"(b.x));
    asm("v_add_f32 %0, %1, %2" : "=v"(r.y) : "v"(a.y), "v"(b.y));
#else
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
#endif
    return r;
}
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b)
{
    v2f r;
#if defined(DSA_PK_DBG) && (DSA_PK_DBG & 4)
    asm("v_sub_f32 %0, %1, %2" : "=v"(r.x) : "v"(a.x), "v"(b.x));
    asm("v_sub_f32 %0, %1, %2" : "=v"(r.y) : "v"(a.y), "v"(b.y));
#else
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
#endif
    return r;
}
// DSA_PK_DBG (reduction builds of tools/hazard/, never the product): 1 the +-i rotations on scalar instructions, 2 the scalar-register
// twiddle products on scalar instructions, 4 plain packed add / sub on scalar instructions, 8 / 16 two wait states before / after
// every +-i rotation, 32 the rotations as a plain packed add of a half-swapped copy (v_pk_mov-free: two v_mov), 64 volatile
#ifndef DSA_PK_DBG
#define DSA_PK_DBG 0
#endif
// DSA_PK_CROSSED (A/B and tools/hazard/ builds only, 0 in the product): 1 restores the one-instruction forms whose LOW result half
// reads a HIGH source half (a set op_sel bit: the +-i rotation, the second instruction of a complex product, the split's sums).
// That is the instruction class of every transient wrong result DESIGN.md 4 recorded, and its trigger is not understood, so NO
// shipped kernel executes it: the same roundings come from two one-component instructions (an addition / subtraction costs 2
// datapath cycles against 4 for the packed pair: no loss; a multiply-add 4 against 4 for the pair: the complex product pays 4
// cycles).  tests/test_host_cpu.py::test_no_crossed_packed_float32 disassembles the built library and enforces it.
#ifndef DSA_PK_CROSSED
#define DSA_PK_CROSSED 0
#endif
#if DSA_PK_DBG & 8
#define DSA_PK_ROT_PRE "s_nop 1\n\t"
#else
#define DSA_PK_ROT_PRE ""
#endif
#if DSA_PK_DBG & 16
#define DSA_PK_ROT_POST "\n\ts_nop 1"
#else
#define DSA_PK_ROT_POST ""
#endif
__device__ __forceinline__ v2f pk_add_negi(v2f a, v2f b)   // a - i b = (a.re + b.im, a.im - b.re)
{
    v2f r;
#if (DSA_PK_DBG & 1) || !DSA_PK_CROSSED
    float rx, ry;
    asm("v_add_f32 %0, %1, %2" : "=v"(rx) : "v"(a.x), "v"(b.y));
    asm("v_sub_f32 %0, %1, %2" : "=v"(ry) : "v"(a.y), "v"(b.x));
    r = v2f{rx, ry};
#elif DSA_PK_DBG & 32
    v2f bs;
    asm("v_mov_b32 %0, %1" : "=v"(bs.x) : "v"(b.y));
    asm("v_mov_b32 %0, %1" : "=v"(bs.y) : "v"(b.x));
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(bs));
#else
    asm(DSA_PK_ROT_PRE "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" DSA_PK_ROT_POST : "=v"(r) : "v"(a), "v"(b));
#endif
    return r;
}
__device__ __forceinline__ v2f pk_add_posi(v2f a, v2f b)   // a + i b = (a.re - b.im, a.im + b.re)
{
    v2f r;
#if (DSA_PK_DBG & 1) || !DSA_PK_CROSSED
    float rx, ry;
    asm("v_sub_f32 %0, %1, %2" : "=v"(rx) : "v"(a.x), "v"(b.y));
    asm("v_add_f32 %0, %1, %2" : "=v"(ry) : "v"(a.y), "v"(b.x));
    r = v2f{rx, ry};
#elif DSA_PK_DBG & 32
    v2f bs;
    asm("v_mov_b32 %0, %1" : "=v"(bs.x) : "v"(b.y));
    asm("v_mov_b32 %0, %1" : "=v"(bs.y) : "v"(b.x));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(bs));
#else
    asm(DSA_PK_ROT_PRE "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" DSA_PK_ROT_POST : "=v"(r) : "v"(a), "v"(b));
#endif
    return r;
}
__device__ __forceinline__ v2f pk_add_conj(v2f a, v2f b)   // a + conj(b) = (a.re + b.re, a.im - b.im)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_sub_conj(v2f a, v2f b)   // a - conj(b) = (a.re - b.re, a.im + b.im)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a.lo + b.hi, a.lo - b.hi): the real-FFT split's sums (a, b may be the same pair)
__device__ __forceinline__ v2f pk_lo_pm_hi(v2f a, v2f b)
{
    v2f r;
#if DSA_PK_CROSSED
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
#else
    asm("v_add_f32 %0, %1, %2" : "=v"(r.x) : "v"(a.x), "v"(b.y));
    asm("v_sub_f32 %0, %1, %2" : "=v"(r.y) : "v"(a.x), "v"(b.y));
#endif
    return r;
}
// (a.hi - b.lo, -a.hi - b.lo)
__device__ __forceinline__ v2f pk_hi_mp_lo(v2f a, v2f b)
{
    v2f r;
#if DSA_PK_CROSSED
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(r) : "v"(a), "v"(b));
#else
    asm("v_sub_f32 %0, %1, %2" : "=v"(r.x) : "v"(a.y), "v"(b.x));
    asm("v_sub_f32 %0, -%1, %2" : "=v"(r.y) : "v"(a.y), "v"(b.x));
#endif
    return r;
}
__device__ __forceinline__ v2f pk_mul(v2f a, v2f b)
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ v2f pk_fma_sc(v2f a, v2f b, v2f c)   // c uniform, in a scalar register pair
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
__device__ __forceinline__ v2f pk_mul_s(v2f a, v2f b)   // b uniform, in a scalar register pair
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
}
// complex product a * t, t = (c, s) in VECTOR registers: (a.re c - a.im s, a.im c + a.re s)
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f t)
{
    v2f t1, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t1) : "v"(a), "v"(t));
#if DSA_PK_CROSSED
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(t), "v"(t1));
#else
    asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(r.x) : "v"(a.y), "v"(t.y), "v"(t1.x));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r.y) : "v"(a.x), "v"(t.y), "v"(t1.y));
#endif
    return r;
}
// the same with the constant t in a SCALAR register pair (the radix-16 twiddles: uniform, 10 scalar registers)
__device__ __forceinline__ v2f pk_cmul_s(v2f a, v2f t)
{
    v2f t1, r;
#if DSA_PK_DBG & 2
    asm("v_mul_f32 %0, %2, %1" : "=v"(t1.x) : "v"(a.x), "s"(t.x));
    asm("v_mul_f32 %0, %2, %1" : "=v"(t1.y) : "v"(a.y), "s"(t.x));
    asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(r.x) : "v"(a.y), "s"(t.y), "v"(t1.x));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r.y) : "v"(a.x), "s"(t.y), "v"(t1.y));
    return r;
#endif
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t1) : "v"(a), "s"(t));
#if DSA_PK_CROSSED
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "s"(t), "v"(t1));
#else
    asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(r.x) : "v"(a.y), "s"(t.y), "v"(t1.x));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r.y) : "v"(a.x), "s"(t.y), "v"(t1.y));
#endif
    return r;
}

// ---- options of ShortTimeFourierTransform beyond the plain configuration (round 6), shared by the packed STFT kernel and the
// prologue of the fused STFT -> mel-cepstrum kernel so that both round alike ----
// sum over the 16 lanes of a frame group, the SAME bits in all 16: butterflies whose two partners add the same pair of values
// (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror) -- a rotation scheme would associate differently per lane
__device__ __forceinline__ float row16_sum(float s)
{
    s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0xB1, 0xf, 0xf, true));    // i ^ 1
    s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0x4E, 0xf, 0xf, true));    // i ^ 2
    s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0x141, 0xf, 0xf, true));   // row_half_mirror: i <-> 7 - i
    s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s), 0x140, 0xf, 0xf, true));   // row_mirror: i <-> 15 - i
    return s;
}
// frame.py:139-140 (zmean): the mean of the frame's L samples, 16 lanes x NR pairs each (lane j holds samples 2 j + 32 m1 (+ 1));
// elements past the frame were selected to zero by the caller and are not touched (`in0` / `in1`: the element belongs to the frame).
// Hand-placed instructions throughout: left to the compiler's pairing, this code came out with a crossed packed form
// (tests/test_host_cpu.py::test_no_crossed_packed_float32 caught it).
template <int NR>
__device__ __forceinline__ void pk_zero_mean(v2f (&r)[NR], const bool (&in0)[NR], const bool (&in1)[NR], int L)
{
    float s = 0.f;
#pragma unroll
    for (int m1 = 0; m1 < NR; ++m1) {   // (zeros outside the frame)
        float t;
        asm("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(r[m1].x), "v"(r[m1].y));
        asm("v_add_f32 %0, %1, %2" : "=v"(s) : "v"(s), "v"(t));
    }
    const float mean = row16_sum(s) / (float)L;
    v2f mm;
    asm("v_mov_b32 %0, %1" : "=v"(mm.x) : "v"(mean));
    asm("v_mov_b32 %0, %1" : "=v"(mm.y) : "v"(mean));
#pragma unroll
    for (int m1 = 0; m1 < NR; ++m1) {
        const v2f d = pk_sub(r[m1], mm);
        r[m1] = v2f{in0[m1] ? d.x : 0.f, in1[m1] ? d.y : 0.f};
    }
}
// maximum over the wave, the same bits in all 64 lanes (spec.py:174-176: the relative floor's per-frame reference)
__device__ __forceinline__ float wave64_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}

// 4-point forward DFT in place (W4 = -i): 8 packed instructions
__device__ __forceinline__ void pk_dft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3)
{
    const v2f s02 = pk_add(a0, a2), d02 = pk_sub(a0, a2), s13 = pk_add(a1, a3), d13 = pk_sub(a1, a3);
    a0 = pk_add(s02, s13);
    a2 = pk_sub(s02, s13);
    a1 = pk_add_negi(d02, d13);
    a3 = pk_add_posi(d02, d13);
}
// the same with a3 == 0 on input (zero padding past the frame: known at compile time): 6 instructions
__device__ __forceinline__ void pk_dft4_z3(v2f& a0, v2f& a1, v2f& a2, v2f& a3)
{
    const v2f s02 = pk_add(a0, a2), d02 = pk_sub(a0, a2), a1in = a1;
    a0 = pk_add(s02, a1in);
    a2 = pk_sub(s02, a1in);
    a1 = pk_add_negi(d02, a1in);
    a3 = pk_add_posi(d02, a1in);
}
// the same with a2 standing for -i a2 (the W16^4 twiddle of the second pass folded into the butterfly)
__device__ __forceinline__ void pk_dft4_negi2(v2f& a0, v2f& a1, v2f& a2, v2f& a3)
{
    const v2f s02 = pk_add_negi(a0, a2), d02 = pk_add_posi(a0, a2), s13 = pk_add(a1, a3), d13 = pk_sub(a1, a3);
    a0 = pk_add(s02, s13);
    a2 = pk_sub(s02, s13);
    a1 = pk_add_negi(d02, d13);
    a3 = pk_add_posi(d02, d13);
}

// 16-point forward DFT in registers (radix 4 x 4), output order as fft16: X[k] in v[FFT16_OUT(k)].
// ZTAIL: v[13], v[14], v[15] are zero on input (never read).  80 packed instructions (74 with ZTAIL).
template <bool ZTAIL>
__device__ __forceinline__ void pk_fft16(v2f (&v)[16])
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    pk_dft4(v[0], v[4], v[8], v[12]);
    if (ZTAIL) {
        pk_dft4_z3(v[1], v[5], v[9], v[13]);
        pk_dft4_z3(v[2], v[6], v[10], v[14]);
        pk_dft4_z3(v[3], v[7], v[11], v[15]);
    } else {
        pk_dft4(v[1], v[5], v[9], v[13]);
        pk_dft4(v[2], v[6], v[10], v[14]);
        pk_dft4(v[3], v[7], v[11], v[15]);
    }
    // after the first pass v[n0 + 4q] = B[n0][q]; twiddle by W16^(n0 q) = (cos, -sin)(2 pi n0 q / 16)
    v[5] = pk_cmul_s(v[5], v2f{C1, -S1});     // e = 1
    v[9] = pk_cmul_s(v[9], v2f{R2, -R2});     // e = 2
    v[13] = pk_cmul_s(v[13], v2f{S1, -C1});   // e = 3
    v[6] = pk_cmul_s(v[6], v2f{R2, -R2});     // e = 2
    //   v[10]: e = 4, a factor -i, folded into the q = 2 butterfly below
    v[14] = pk_cmul_s(v[14], v2f{-R2, -R2});  // e = 6
    v[7] = pk_cmul_s(v[7], v2f{S1, -C1});     // e = 3
    v[11] = pk_cmul_s(v[11], v2f{-R2, -R2});  // e = 6
    v[15] = pk_cmul_s(v[15], v2f{-C1, S1});   // e = 9
    pk_dft4(v[0], v[1], v[2], v[3]);
    pk_dft4(v[4], v[5], v[6], v[7]);
    pk_dft4_negi2(v[8], v[9], v[10], v[11]);
    pk_dft4(v[12], v[13], v[14], v[15]);
}

// ---------------------------------------------------------------------------------------------
// Scalar twins (one v_add / v_mul / v_fma per component, hand-placed like the packed forms so that the compiler's packed-float32
// selection cannot re-pair them), same roundings as the packed helpers -- the results are bit-identical.
// Why they exist (round 4, tools/dbg_fused*.py): inside the fused STFT -> mel-cepstrum kernel a wave runs this FFT while the
// other wave of its SIMD runs the v_mfma_f32_4x4x1 products of the elimination.  With v_pk_*_f32 in the FFT, about one
// instruction in 2e5 delivered a stale result in one 16-lane group (58 +- 10 wrong frames per 204 800, non-deterministic, only
// while another wave was in its Newton phase; none with the elimination on v_fmac_f32_dpp, none with n_iter = 0, none without
// packed instructions; wait states, LDS fences and the compiler's own packed code changed nothing; NOT reproduced in isolation by
// tools/repro_pk_mfma.cpp -- it is this kernel's conditions, not a general rule).  The packed instruction and
// the float32 matrix instruction share the SIMD's float32 datapath; scalar vector instructions next to the matrix instruction are
// exact in every run.  Additions and multiplications cost 2 datapath cycles each against 4 for a packed pair, so only the
// multiply-adds (a fifth of the transform) pay for the split.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sc_add1(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sc_sub1(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sc_nsub1(float a, float b) { float r; asm("v_sub_f32 %0, -%1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }   // -a - b
__device__ __forceinline__ float sc_mul1(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float sc_mul1s(float a, float b) { float r; asm("v_mul_f32 %0, %2, %1" : "=v"(r) : "v"(a), "s"(b)); return r; }   // b uniform
__device__ __forceinline__ float sc_fma1(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float sc_fma1sc(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c)); return r; }   // c uniform
__device__ __forceinline__ float sc_fma1s(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c)); return r; }    // b uniform
__device__ __forceinline__ float sc_nfma1(float a, float b, float c) { float r; asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }  // -a b + c
__device__ __forceinline__ float sc_nfma1s(float a, float b, float c) { float r; asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c)); return r; }

__device__ __forceinline__ v2f sc_add(v2f a, v2f b) { return v2f{sc_add1(a.x, b.x), sc_add1(a.y, b.y)}; }
__device__ __forceinline__ v2f sc_sub(v2f a, v2f b) { return v2f{sc_sub1(a.x, b.x), sc_sub1(a.y, b.y)}; }
__device__ __forceinline__ v2f sc_add_negi(v2f a, v2f b) { return v2f{sc_add1(a.x, b.y), sc_sub1(a.y, b.x)}; }   // a - i b
__device__ __forceinline__ v2f sc_add_posi(v2f a, v2f b) { return v2f{sc_sub1(a.x, b.y), sc_add1(a.y, b.x)}; }   // a + i b
__device__ __forceinline__ v2f sc_add_conj(v2f a, v2f b) { return v2f{sc_add1(a.x, b.x), sc_sub1(a.y, b.y)}; }   // a + conj(b)
__device__ __forceinline__ v2f sc_sub_conj(v2f a, v2f b) { return v2f{sc_sub1(a.x, b.x), sc_add1(a.y, b.y)}; }   // a - conj(b)
__device__ __forceinline__ v2f sc_mul(v2f a, v2f b) { return v2f{sc_mul1(a.x, b.x), sc_mul1(a.y, b.y)}; }
// complex product a * t, rounded as pk_cmul: (fma(-a.im, s, a.re c), fma(a.re, s, a.im c))
__device__ __forceinline__ v2f sc_cmul(v2f a, v2f t)
{
    const float t1x = sc_mul1(a.x, t.x), t1y = sc_mul1(a.y, t.x);
    return v2f{sc_nfma1(a.y, t.y, t1x), sc_fma1(a.x, t.y, t1y)};
}
__device__ __forceinline__ v2f sc_cmul_s(v2f a, v2f t)   // t uniform (scalar registers)
{
    const float t1x = sc_mul1s(a.x, t.x), t1y = sc_mul1s(a.y, t.x);
    return v2f{sc_nfma1s(a.y, t.y, t1x), sc_fma1s(a.x, t.y, t1y)};
}
__device__ __forceinline__ void sc_dft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3)
{
    const v2f s02 = sc_add(a0, a2), d02 = sc_sub(a0, a2), s13 = sc_add(a1, a3), d13 = sc_sub(a1, a3);
    a0 = sc_add(s02, s13);
    a2 = sc_sub(s02, s13);
    a1 = sc_add_negi(d02, d13);
    a3 = sc_add_posi(d02, d13);
}
__device__ __forceinline__ void sc_dft4_z3(v2f& a0, v2f& a1, v2f& a2, v2f& a3)
{
    const v2f s02 = sc_add(a0, a2), d02 = sc_sub(a0, a2), a1in = a1;
    a0 = sc_add(s02, a1in);
    a2 = sc_sub(s02, a1in);
    a1 = sc_add_negi(d02, a1in);
    a3 = sc_add_posi(d02, a1in);
}
__device__ __forceinline__ void sc_dft4_negi2(v2f& a0, v2f& a1, v2f& a2, v2f& a3)
{
    const v2f s02 = sc_add_negi(a0, a2), d02 = sc_add_posi(a0, a2), s13 = sc_add(a1, a3), d13 = sc_sub(a1, a3);
    a0 = sc_add(s02, s13);
    a2 = sc_sub(s02, s13);
    a1 = sc_add_negi(d02, d13);
    a3 = sc_add_posi(d02, d13);
}
// pk_fft16 on scalar instructions: 16-point forward DFT in registers, X[k] in v[FFT16_OUT(k)]
template <bool ZTAIL>
__device__ __forceinline__ void sc_fft16(v2f (&v)[16])
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    sc_dft4(v[0], v[4], v[8], v[12]);
    if (ZTAIL) {
        sc_dft4_z3(v[1], v[5], v[9], v[13]);
        sc_dft4_z3(v[2], v[6], v[10], v[14]);
        sc_dft4_z3(v[3], v[7], v[11], v[15]);
    } else {
        sc_dft4(v[1], v[5], v[9], v[13]);
        sc_dft4(v[2], v[6], v[10], v[14]);
        sc_dft4(v[3], v[7], v[11], v[15]);
    }
    v[5] = sc_cmul_s(v[5], v2f{C1, -S1});
    v[9] = sc_cmul_s(v[9], v2f{R2, -R2});
    v[13] = sc_cmul_s(v[13], v2f{S1, -C1});
    v[6] = sc_cmul_s(v[6], v2f{R2, -R2});
    v[14] = sc_cmul_s(v[14], v2f{-R2, -R2});
    v[7] = sc_cmul_s(v[7], v2f{S1, -C1});
    v[11] = sc_cmul_s(v[11], v2f{-R2, -R2});
    v[15] = sc_cmul_s(v[15], v2f{-C1, S1});
    sc_dft4(v[0], v[1], v[2], v[3]);
    sc_dft4(v[4], v[5], v[6], v[7]);
    sc_dft4_negi2(v[8], v[9], v[10], v[11]);
    sc_dft4(v[12], v[13], v[14], v[15]);
}

}  // namespace dsa
