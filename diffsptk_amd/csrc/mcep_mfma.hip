// Tuned mel-cepstral analysis for gfx950: float32, fft_length 512, cep_order 24.
// (MelCepstralAnalysis._forward, diffsptk/modules/mcep.py:189-224, in the composed-matrix form
//  described in mcep.hip.)  tools/proto_mcep_mfma.py is a lane-level numpy model of the data flow.
//
// This file holds what the forward (mcep_mfma_f16.h) and backward (mcep_mfma_bwd_f16.h) kernels share -- the
// LDS carve-up constants, the quad-layout column-cyclic elimination of the 25 x 25 system -- and the host side:
// operand-image preparation and the launches.  The library keeps NO device memory of its own: the binary16
// operand images of (G, D, E) are written once per configuration into a caller-owned buffer
// (dsa_mcep_images_bytes / dsa_mcep_prepare) and the tile-queue counters of a launch live in a caller-owned
// scratch (DSA_SCRATCH_BYTES), so launches on any number of streams / devices never share mutable state.
// (Earlier kernel generations -- float32-MFMA chains, row-cyclic elimination, role-split waves -- are in the
// git history; DESIGN.md section 3.2 keeps their measurements.)
//
// Mapping.  One wave64 owns 16 frames for the whole Newton iteration; lane l = (n = l & 15: frame, g = l >> 4:
// lane group).  The frames are the N (column) dimension of the MFMAs, so products come out TRANSPOSED:
//     d^T (256 x 16) = D^T mc^T,   rt^T (48 x 16) = E^T e^T.
// In the C/D layout lane (n, g) register r of tile mt holds bin 16 mt + 4 g + r of frame n -- exactly a B operand
// if the k-steps of the second product are enumerated in that order (the E^T image is laid out for it), so
// e = exp(log X - 2 d) feeds the second chain straight from the accumulator registers: no transpose, no LDS
// round trip, log X stays in 64 VGPRs for all iterations.  Bin 256 (Nyquist) and output rt[48] are one extra
// k-step / one VALU dot product instead of padded tiles.
#include "common.h"
#include "th_solve_reg.h"

#include <stdlib.h>
#include <utility>

namespace dsa {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4_u4 __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte access at dword alignment

namespace mm {
constexpr int H = 256, K = 257;   // nfft = 512
constexpr int M1 = 25, M2 = 49;   // cep_order 24
constexpr int KS = 7;             // k-steps of the first product (28 >= M1 coefficients)
constexpr int NR = 7;             // local rows per lane group (ceil(M1 / 4))
constexpr int RS = 68;            // per-frame stride of the rt / rr windows in LDS (floats; 68 % 32 = 4:
                                  // the quad-layout reads of 8 frames x 4 lanes hit 32 distinct banks)
// LDS carve-up (floats)
constexpr int DT_OFF = 0;                       // [16 mt][2 half][64 lane][4]
constexpr int ET_OFF = DT_OFF + 16 * 2 * 64 * 4;  // [3 it][16 mt][64 lane][4 r]
constexpr int E48_OFF = ET_OFF + 3 * 16 * 64 * 4; // [16 mt][4 g][4 r]
constexpr int E256_OFF = E48_OFF + 256;         // [48] + [1] = E[256][0..48]
constexpr int D256_OFF = E256_OFF + 52;         // [28]
constexpr int AV_OFF = D256_OFF + 28;           // [28]
constexpr int WAVE_OFF = AV_OFF + 28;           // per wave: rt [16][RS], rr [16][RS]
constexpr int WAVE_FLOATS = 2 * 16 * RS;
constexpr int LDS_FLOATS = WAVE_OFF + 4 * WAVE_FLOATS;
}  // namespace mm

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

#ifdef DSA_MCEP_TIMING
// Phase stamps of a Newton step (tools/bench_mcep.cpp): the cycle counter goes into SCALAR registers, without a branch (a
// conditional store at every stamp split the step's basic block and with it the instruction schedule being measured); the
// stamps of one (tile, step) are written out by DSA_STAMPS_FLUSH at the end of the step.
__device__ unsigned long long g_mcep_stamps[64];
__device__ unsigned long long g_mcep_slotlog[2048 * 3];   // per wave slot: entry, exit (100 MHz ticks), tiles run
#define DSA_STAMPS_DECL unsigned dsa_st_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define DSA_STAMP(i) dsa_st_[i] = (unsigned)__builtin_readcyclecounter()
#define DSA_STAMPS_FLUSH                                                                \
    do {                                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0 && tile == 0 && iter == 1)             \
            for (int i_ = 0; i_ < 12; ++i_) g_mcep_stamps[i_] = dsa_st_[i_];            \
    } while (0)
#else
#define DSA_STAMPS_DECL
#define DSA_STAMP(i)
#define DSA_STAMPS_FLUSH
#endif

// Branch-free per-lane selection by lane group: gm[i] is all-ones where g == i (hipcc turns
// nested ?: on lane-dependent conditions into exec-mask control flow; the bit form stays VALU).
struct GroupMask {
    unsigned m[4];
    unsigned gt[4];  // gt[i]: all-ones where g > i
};
__device__ __forceinline__ GroupMask make_group_mask(int g)
{
    GroupMask q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        q.m[i] = g == i ? 0xffffffffu : 0u;
        q.gt[i] = g > i ? 0xffffffffu : 0u;
    }
    return q;
}
__device__ __forceinline__ float sel4(const GroupMask& q, float c0, float c1, float c2, float c3)
{
    unsigned r = (__float_as_uint(c0) & q.m[0]) | (__float_as_uint(c1) & q.m[1]) | (__float_as_uint(c2) & q.m[2]) |
                 (__float_as_uint(c3) & q.m[3]);
    return __uint_as_float(r);
}
__device__ __forceinline__ float keep_if(unsigned mask, float v) { return __uint_as_float(__float_as_uint(v) & mask); }
// 1/x: v_rcp_f32 (1 ulp) + one Newton step
__device__ __forceinline__ float rcp_nr(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    return r * __builtin_fmaf(-x, r, 2.f);
}

// Reductions over the four lane groups g = lane >> 4 of a frame (lanes n, n + 16, n + 32, n + 48) without the LDS crossbar:
// v_permlane16_swap / v_permlane32_swap (gfx950) exchange the odd 16-lane rows of one register with the even rows of another
// (resp. the upper half of one with the lower half of the other); with both operands holding x the two results are
// (x of the even row | x of the odd row) on both rows of a pair, so one add / max of the pair is x (op) x[lane ^ 16] -- three
// vector instructions and no memory latency, where __shfl_xor is a ds_bpermute_b32 round trip (~100+ cycles on the wave's
// in-order critical path, six of them per Newton step).
__device__ __forceinline__ float rows_sum4(float x)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ float rows_max4(float x)
{
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __builtin_fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __builtin_fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// value of lane Q of this lane's quad (DPP quad_perm broadcast: a plain VALU move, no LDS)
template <int Q>
__device__ __forceinline__ float quad_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), Q * 0x55, 0xf, 0xf, true));
}

// ---------------------------------------------------------------------------------------------
// Column-cyclic symmetric elimination in the quad layout.  Lane gs of a quad owns COLUMNS
// j = gs + 4c (c = 0..6) of every row; row i keeps its entries c >= i >> 2 (upper triangle plus at
// most three harmless sub-diagonal ones): 109 registers.  At step k the pivot row's own-column
// entries are already local; only the 24 - k multipliers a[k][i] / a[k][k] (by symmetry elements of
// the pivot ROW) are broadcast, each from the lane that owns column i -- no lane-dependent
// selection anywhere.  Column 25 (lane 1, slot c = 6) carries the right-hand side and column 26
// (lane 2, slot c = 6) an optional second one, so they ride along the row updates for free.
// ---------------------------------------------------------------------------------------------
namespace colm {
using namespace mm;
constexpr int row_len(int i) { return 7 - (i >> 2); }
constexpr int row_off(int i)
{
    int o = 0;
    for (int q = 0; q < i; ++q) o += row_len(q);
    return o;
}
constexpr int TOTAL = row_off(M1);  // 109
}  // namespace colm
#define COL_AT(a, i, c) (a)[colm::row_off(i) + (c) - ((i) >> 2)]

// rt0 / rr0: un-shifted windows of this lane's frame; rhs2: second right-hand side (or nullptr)
template <int i>
__device__ __forceinline__ void col_build_rows(float (&a)[colm::TOTAL], const float* rt0, const float* rr0,
                                               const float* avs, const float* rhs2, int gs, const GroupMask& gq)
{
    using namespace mm;
    if constexpr (i < M1) {
        const float* rt_g = rt0 + gs;
        const float* rr_g = rr0 + 27 - gs;
        // entry (i, j = gs + 4c): R[i][j] + Q[i][j] = rr[27 + i - j] + rt[i + j]  (mcep.py:219-221)
#pragma unroll
        for (int c = i >> 2; c < 6; ++c) COL_AT(a, i, c) = rt_g[i + 4 * c] + rr_g[i - 4 * c];
        const float v24 = rt_g[i + 24] + rr_g[i - 24];   // column 24 exists on lane 0 only
        const float r1 = rt0[i] - avs[i];                // mcep.py:216-217
        const float r2 = rhs2 ? rhs2[i] : 0.f;
        COL_AT(a, i, 6) = sel4(gq, v24, r1, r2, 0.f);
        col_build_rows<i + 1>(a, rt0, rr0, avs, rhs2, gs, gq);
    }
}

// The same rows with slot c = 6 read through two per-lane pointers instead of computed three ways and selected:
// lane 0 of a quad: rt[24 + i] + rr[3 + i] (column 24), lane 1: rt[i] + (-alpha_vec[i]) (the right-hand side), lanes 2, 3:
// 0 + 0 (pa6 / pb6 point at this frame's windows, at the negated alpha table, or at a row of zeros).  Bit-identical to
// col_build_rows with rhs2 = nullptr; 6 instructions fewer per row (the forward kernel is bound by vector issue).
template <int i>
__device__ __forceinline__ void col_build_rows_p(float (&a)[colm::TOTAL], const float* rt0, const float* rr0,
                                                 const float* pa6, const float* pb6, int gs)
{
    using namespace mm;
    if constexpr (i < M1) {
        const float* rt_g = rt0 + gs;
        const float* rr_g = rr0 + 27 - gs;
#pragma unroll
        for (int c = i >> 2; c < 6; ++c) COL_AT(a, i, c) = rt_g[i + 4 * c] + rr_g[i - 4 * c];
        COL_AT(a, i, 6) = pa6[i] + pb6[i];
        col_build_rows_p<i + 1>(a, rt0, rr0, pa6, pb6, gs);
    }
}

// acc += quad_bcast<Q>(s0) * s1 in ONE instruction: the DPP quad_perm broadcast is the src0 modifier
// of v_fmac_f32 (hipcc CSEs builtin DPP moves into separate v_mov_b32_dpp instead of fusing them)
template <int Q>
__device__ __forceinline__ void fmac_quad_bcast(float& acc, float s0, float s1)
{
    if constexpr (Q == 0)
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s0), "v"(s1));
    else if constexpr (Q == 1)
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s0), "v"(s1));
    else if constexpr (Q == 2)
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s0), "v"(s1));
    else
        asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s0), "v"(s1));
}

template <int k, int i>
__device__ __forceinline__ void col_update_rows(float (&a)[colm::TOTAL], const float (&pneg)[7])
{
    using namespace mm;
    if constexpr (i < M1) {
        // row_i -= (a[k][i] / a[k][k]) row_k.  The multiplier is element i of the pre-scaled pivot row
        // (pneg = -row_k / a[k][k]), which lives on lane i & 3 of the quad
#pragma unroll
        for (int c = i >> 2; c < 7; ++c) fmac_quad_bcast<(i & 3)>(COL_AT(a, i, c), pneg[(i >> 2) - (k >> 2)], COL_AT(a, k, c));
        col_update_rows<k, i + 1>(a, pneg);
    }
}
template <int k>
__device__ __forceinline__ void col_elim_step(float (&a)[colm::TOTAL])
{
    // v_rcp_f32 (1 ulp) without a Newton step: the error of the pivot reciprocal enters the solution like one more rounding
    // of the elimination, and the outer Newton iteration only ever uses the solve for an UPDATE
    const float ninv = -__builtin_amdgcn_rcpf(quad_bcast<(k & 3)>(COL_AT(a, k, k >> 2)));
    float pneg[7];
    constexpr int NP = 7 - (k >> 2);   // entries of the pivot row this lane owns
#pragma unroll
    for (int c = 0; c < 7; ++c) pneg[c] = c < NP ? COL_AT(a, k, (c < NP ? c : 0) + (k >> 2)) * ninv : 0.f;
    // VALU write -> DPP read of pneg needs 2 wait states, which hipcc cannot see inside inline asm; the
    // dummy in/out operands pin the nop after the multiplies.  Only the NP live entries are operands: naming all
    // seven made the compiler materialise the unused tail (a v_mov 0 each: 66 instructions per 25 x 25 system).
    if constexpr (NP == 7)
        asm volatile("s_nop 1" : "+v"(pneg[0]), "+v"(pneg[1]), "+v"(pneg[2]), "+v"(pneg[3]), "+v"(pneg[4]), "+v"(pneg[5]), "+v"(pneg[6]));
    else if constexpr (NP == 6)
        asm volatile("s_nop 1" : "+v"(pneg[0]), "+v"(pneg[1]), "+v"(pneg[2]), "+v"(pneg[3]), "+v"(pneg[4]), "+v"(pneg[5]));
    else if constexpr (NP == 5)
        asm volatile("s_nop 1" : "+v"(pneg[0]), "+v"(pneg[1]), "+v"(pneg[2]), "+v"(pneg[3]), "+v"(pneg[4]));
    else if constexpr (NP == 4)
        asm volatile("s_nop 1" : "+v"(pneg[0]), "+v"(pneg[1]), "+v"(pneg[2]), "+v"(pneg[3]));
    else if constexpr (NP == 3)
        asm volatile("s_nop 1" : "+v"(pneg[0]), "+v"(pneg[1]), "+v"(pneg[2]));
    else if constexpr (NP == 2)
        asm volatile("s_nop 1" : "+v"(pneg[0]), "+v"(pneg[1]));
    else
        asm volatile("s_nop 1" : "+v"(pneg[0]));
    col_update_rows<k, k + 1>(a, pneg);
    // the pivot row stays behind NORMALISED and negated (-row_k / a_kk: a register renaming, no instruction): the back
    // substitution then needs neither the reciprocal nor its broadcast again
#pragma unroll
    for (int c = 0; c < NP; ++c) COL_AT(a, k, c + (k >> 2)) = pneg[c];
}
template <int... Ks>
__device__ __forceinline__ void col_elim_all(float (&a)[colm::TOTAL], std::integer_sequence<int, Ks...>)
{
    (col_elim_step<Ks>(a), ...);
}

// back substitution of right-hand side RHS (1 or 2): xq[c] accumulates x[gs + 4c]; the slot of the
// right-hand-side column is preset to -1 on its owner lane, so  sum_j U[k][j] x_j - b_k  is one dot product
// (the rows arrive as -U[k][.] / U[k][k] from col_elim_step: the dot product is x_k itself)
template <int k>
__device__ __forceinline__ float col_backsub_step(const float (&a)[colm::TOTAL], float (&xq)[mm::KS],
                                                  const GroupMask& gq)
{
    float sl = 0.f;
#pragma unroll
    for (int c = k >> 2; c < 7; ++c) sl = __builtin_fmaf(COL_AT(a, k, c), xq[c], sl);
    sl += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sl), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
    sl += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sl), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
    // rows are normalised and negated by the elimination: sum_j (-U_kj / U_kk) x_j with x_rhs = -1 IS x_k
    const float xk = sl;
    // the owner lane of column k takes xk, the others keep their slot: one v_cndmask_b32 (the slot is still 0 on the owner)
    xq[k >> 2] = gq.m[k & 3] ? xk : xq[k >> 2];
    return xk;
}
template <int... Ks>
__device__ __forceinline__ void col_backsub_all(const float (&a)[colm::TOTAL], float (&xq)[mm::KS],
                                                const GroupMask& gq, std::integer_sequence<int, Ks...>)
{
    ((void)col_backsub_step<mm::M1 - 1 - Ks>(a, xq, gq), ...);
}


// ---------------------------------------------------------------------------------------------
// The same elimination with its rank-1 updates on v_mfma_f32_4x4x1_16b_f32 (round 3).
// One instruction is 16 independent 4 x 4 outer products D_b += A_b (x) B_b, block b = the lanes 4b .. 4b + 3: exactly the
// quad layout -- 16 systems per wave, lane gs of a quad owning the columns gs + 4c.  The matrix is kept as 4 x 4 BLOCKS,
// block (rg, cg) = rows 4 rg .. 4 rg + 3 x columns 4 cg .. 4 cg + 3 as one register quadruple (register i = row 4 rg + i, the
// lane = the column within the group): the C / D operand.  With the pivot row scaled to m = -row_k / a_kk, the update of
// block (rg, cg) is ONE instruction with A = slot rg of m (lane i: the multiplier of row 4 rg + i -- by symmetry that is
// where it already is) and B = slot cg of the unscaled pivot row (lane j: column 4 cg + j): no broadcast, no DPP, no
// wait states, 305 matrix instructions per 25 x 25 system (+ right-hand sides in column group 6) instead of 986
// v_fmac_f32_dpp.  tools/bench_issue.cpp (profiles/r03_issue_calibration*.txt) has the prices: a v_fmac_f32(_dpp) occupies a
// SIMD for 4 cycles (64 multiply-adds), the 4 x 4 x 1 product for 8 (256): twice the multiply-adds per cycle, on the same
// datapath (the float32 products and the vector ALU do NOT overlap, neither within a wave nor across the waves of a SIMD).
// The elimination's arithmetic is the same fmaf(multiplier, pivot-row entry, entry) per element as in the column-cyclic code
// above (rows <= k of the pivot's own row group are kept by zeroing their lanes of A); the back substitution divides by the
// pivot at the end of each row instead of reading pre-scaled rows.
// ---------------------------------------------------------------------------------------------
namespace blk {
constexpr int NG = 7;                                     // row / column groups (28 >= 25 rows, + right-hand sides)
constexpr int NBLK = NG * (NG + 1) / 2;                   // upper-triangular blocks: 28 quadruples = 112 registers
constexpr int at(int rg, int cg) { return rg * NG - rg * (rg - 1) / 2 + (cg - rg); }
}  // namespace blk

__device__ __forceinline__ f32x4 mfma441(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// rows of R + Q in block layout; slot 6 through the two per-lane pointers of col_build_rows_p (column 24 | right-hand side
// | second right-hand side or zero | zero).  Rows 25 .. 27 are padding: never pivots, never read.
template <int rg>
__device__ __forceinline__ void blk_build_rows(f32x4 (&a)[blk::NBLK], const float* rt0, const float* rr0, const float* pa6,
                                               const float* pb6, int gs)
{
    using namespace mm;
    if constexpr (rg < blk::NG) {
        const float* rt_g = rt0 + gs;
        const float* rr_g = rr0 + 27 - gs;
#ifndef DSA_BLK_BUILD_SCALAR   // the four rows of a block as ONE 16-byte read per window (dword-aligned) and two packed additions
        // (round 5: half the LDS instructions of the build, the same sums; the forward kernels' last 28 bytes of scratch go with it)
        if constexpr (4 * rg + 3 < M1) {
#pragma unroll
            for (int c = rg; c < 6; ++c)
                a[blk::at(rg, c)] = *reinterpret_cast<const f32x4_u4*>(rt_g + 4 * rg + 4 * c) + *reinterpret_cast<const f32x4_u4*>(rr_g + 4 * rg - 4 * c);
            a[blk::at(rg, 6)] = *reinterpret_cast<const f32x4_u4*>(pa6 + 4 * rg) + *reinterpret_cast<const f32x4_u4*>(pb6 + 4 * rg);
        } else
#endif
        {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * rg + i;
                if (row < M1) {
#pragma unroll
                    for (int c = rg; c < 6; ++c) a[blk::at(rg, c)][i] = rt_g[row + 4 * c] + rr_g[row - 4 * c];
                    a[blk::at(rg, 6)][i] = pa6[row] + pb6[row];
                } else {
                    a[blk::at(rg, 6)][i] = 0.f;
                }
            }
        }
        blk_build_rows<rg + 1>(a, rt0, rr0, pa6, pb6, gs);
    }
}

template <int k, int rg>
__device__ __forceinline__ void blk_update_groups(f32x4 (&a)[blk::NBLK], const float (&m)[blk::NG])
{
    if constexpr (rg < blk::NG) {
        // A = the multipliers of rows 4 rg .. 4 rg + 3 (slot rg of the scaled pivot row), B = the pivot row where it stands
#pragma unroll
        for (int c = rg; c < blk::NG; ++c) a[blk::at(rg, c)] = mfma441(m[rg], a[blk::at(k >> 2, c)][k & 3], a[blk::at(rg, c)]);
        blk_update_groups<k, rg + 1>(a, m);
    }
}

// One step.  The pivot row stays in place UNSCALED (writing single elements of the register quadruples makes the compiler
// copy whole quadruples); its scaled copy m = -row_k / a_kk lives for this step only, and the back substitution divides by
// the pivot again (one v_rcp_f32_dpp per row).
template <int k>
__device__ __forceinline__ void blk_elim_step(f32x4 (&a)[blk::NBLK], const GroupMask& gq, float (&ninvs)[mm::M1])
{
    using namespace blk;
    constexpr int c0 = k >> 2, q = k & 3;
    if constexpr (k == mm::M1 - 1) ninvs[k] = -__builtin_amdgcn_rcpf(quad_bcast<q>(a[at(c0, c0)][q]));
    if constexpr (k < mm::M1 - 1) {
        // The vector instructions of a step in ONE run ahead of its matrix instructions: a vector instruction between two
        // 4 x 4 x 1 products costs ~9 cycles on top of its own issue (tools/bench_issue.cpp, "mix" rows), and left alone the
        // scheduler sprinkles the multiplies between the products.
        const float ninv = -__builtin_amdgcn_rcpf(quad_bcast<q>(a[at(c0, c0)][q]));
        ninvs[k] = ninv;
        float m[NG];
#pragma unroll
        for (int c = 0; c < NG; ++c) m[c] = c >= c0 ? a[at(c0, c >= c0 ? c : c0)][q] * ninv : 0.f;
        // the pivot's own row group: rows <= k keep their values (their lanes of A are zero)
        const float m0 = keep_if(gq.gt[q], m[c0]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (q < 3) {
#pragma unroll
            for (int c = c0; c < NG; ++c) a[at(c0, c)] = mfma441(m0, a[at(c0, c)][q], a[at(c0, c)]);
        }
        blk_update_groups<k, c0 + 1>(a, m);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int... Ks>
__device__ __forceinline__ void blk_elim_all(f32x4 (&a)[blk::NBLK], const GroupMask& gq, float (&ninvs)[mm::M1],
                                             std::integer_sequence<int, Ks...>)
{
    (blk_elim_step<Ks>(a, gq, ninvs), ...);
}

// back substitution over the unscaled rows: x_k = -(sum_{j > k} U_kj x_j - b_k) / U_kk with the right-hand-side slot of xq
// preset to -1 on its owner lane (the diagonal and the sub-diagonal lanes of slot k >> 2 still hold 0 in xq).
// Row group by row group: the sums over the FINISHED column groups (c > rg) of the group's four rows are packed two rows to
// an instruction (a register pair of the quadruple times the broadcast x slot); only the row's own column group, the quad
// reduction and the division are serial.
template <int rg>
__device__ __forceinline__ void blk_backsub_group(const f32x4 (&a)[blk::NBLK], float (&xq)[mm::KS], const GroupMask& gq,
                                                  const float (&ninvs)[mm::M1])
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f};
#pragma unroll
    for (int c = rg + 1; c < blk::NG; ++c) {
        const f32x4 v = a[blk::at(rg, c)];
        const f2 x = {xq[c], xq[c]};
        p01 = __builtin_shufflevector(v, v, 0, 1) * x + p01;
        if constexpr (4 * rg + 2 < mm::M1) p23 = __builtin_shufflevector(v, v, 2, 3) * x + p23;
    }
    const float part[4] = {p01[0], p01[1], p23[0], p23[1]};
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        const int k = 4 * rg + i;
        if (k < mm::M1) {
            float sl = __builtin_fmaf(a[blk::at(rg, rg)][i], xq[rg], part[i]);
            sl += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sl), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
            sl += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sl), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
            const float xk = sl * ninvs[k];   // the negated pivot reciprocals of the elimination (quad-uniform)
            xq[rg] = gq.m[i] ? xk : xq[rg];
        }
    }
}
template <int... Gs>
__device__ __forceinline__ void blk_backsub_all(const f32x4 (&a)[blk::NBLK], float (&xq)[mm::KS], const GroupMask& gq,
                                                const float (&ninvs)[mm::M1], std::integer_sequence<int, Gs...>)
{
    (blk_backsub_group<blk::NG - 1 - Gs>(a, xq, gq, ninvs), ...);
}

// The same back substitution taking the pivots' reciprocals again instead of reading the 25 kept by the elimination (the
// two-wave backward: 25 registers fewer across the elimination are worth 25 v_rcp_f32 per right-hand side there).
template <int rg>
__device__ __forceinline__ void blk_backsub_group_r(const f32x4 (&a)[blk::NBLK], float (&xq)[mm::KS], const GroupMask& gq)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p01 = {0.f, 0.f}, p23 = {0.f, 0.f};
#pragma unroll
    for (int c = rg + 1; c < blk::NG; ++c) {
        const f32x4 v = a[blk::at(rg, c)];
        const f2 x = {xq[c], xq[c]};
        p01 = __builtin_shufflevector(v, v, 0, 1) * x + p01;
        if constexpr (4 * rg + 2 < mm::M1) p23 = __builtin_shufflevector(v, v, 2, 3) * x + p23;
    }
    const float part[4] = {p01[0], p01[1], p23[0], p23[1]};
    const f32x4 d = a[blk::at(rg, rg)];
    const float ninv[4] = {-__builtin_amdgcn_rcpf(quad_bcast<0>(d[0])), -__builtin_amdgcn_rcpf(quad_bcast<1>(d[1])),
                           -__builtin_amdgcn_rcpf(quad_bcast<2>(d[2])), -__builtin_amdgcn_rcpf(quad_bcast<3>(d[3]))};
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        const int k = 4 * rg + i;
        if (k < mm::M1) {
            float sl = __builtin_fmaf(d[i], xq[rg], part[i]);
            sl += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sl), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
            sl += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sl), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
            const float xk = sl * ninv[i];
            xq[rg] = gq.m[i] ? xk : xq[rg];
        }
    }
}
template <int... Gs>
__device__ __forceinline__ void blk_backsub_all_r(const f32x4 (&a)[blk::NBLK], float (&xq)[mm::KS], const GroupMask& gq,
                                                  std::integer_sequence<int, Gs...>)
{
    (blk_backsub_group_r<blk::NG - 1 - Gs>(a, xq, gq), ...);
}

// ---------------------------------------------------------------------------------------------
// The same solve for the Toeplitz-plus-Hankel systems of the mel-generalized cepstral analysis (mgcep.py:226-229:
// solve(symmetric_toeplitz(p) + hankel(q), r), n = 24 = the cepstral order): 16 systems per wave in the quad layout, no
// pivoting -- the matrix is the Hessian of a criterion that is convex for -1 <= gamma <= 0 (SPTK's theq() solves it
// without pivoting for the same reason).  The 24 x 24 system rides in the 25 x 25 machinery with row / column 24 as an
// identity pair: row 24 stores slot 6 only (diagonal 1 on lane 0, right-hand side 0 on lane 1) and column 24 of the other
// rows comes through the slot-6 pointers as zeros, so it never couples.  One system per wave with a pivot search
// (th_solve_reg, csrc/mgc.hip) took 0.2 ms per 51 200 systems; this takes 0.02 ms.
// ---------------------------------------------------------------------------------------------
constexpr int kTq = 132;   // floats per system in LDS: q window [0, 52) | mirrored p window [52, 104) | r [104, 132); 132 % 32 = 4: the four lanes of a
                           // system read four consecutive banks and the eight systems of a 32-lane half cover the 32 banks once (136: two systems per bank range)
// `r` rows are r_stride floats apart and start r_off floats in (the Newton step of mgcep hands over its (F, 25) vector with
// the right-hand side in columns 1 .. 24); `add` (or NULL): g = add + solution (the step's update b <- b + solve(..)).
__global__ __launch_bounds__(256) DSA_PK_TARGET void thsolve_quad24_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                             const float* __restrict__ r, long F, float* g,
                                                             int r_stride, int r_off, const float* add)   // (g may be add)
{
    using namespace mm;
    __shared__ __attribute__((aligned(16))) float lds[4 * 16 * kTq + 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* wl = lds + wave * 16 * kTq;
    float* cst = lds + 4 * 16 * kTq;           // [0, 28): zeros | [32, 57): e_24
    if (threadIdx.x < 64) cst[threadIdx.x] = threadIdx.x == 32 + 24 ? 1.f : 0.f;
    __syncthreads();
    const int nq = lane >> 2, gs = lane & 3;
    const GroupMask gq = make_group_mask(gs);
    const long ntiles = (F + 15) / 16;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        __builtin_amdgcn_wave_barrier();
        // stage the 16 records (missing systems: the identity, right-hand side 0).  The tile's q, p and r rows are three
        // CONTIGUOUS runs of memory: each is copied by an unrolled loop of independent loads (as one loop over the record
        // layout with a guarded load per element, every element waited out its own round trip: 43 of the kernel's 59 us)
        const long fbase = tile * 16;
        const long nvalid = (F - fbase) < 16 ? (F - fbase) : 16;
        for (int e = lane; e < 16 * kTq; e += 64) wl[e] = 0.f;
        float vq[12], vp[6], vr[6];
#pragma unroll
        for (int it = 0; it < 12; ++it) {   // 16 x 47 = 752 values of q
            const int idx = lane + 64 * it;
            const bool ok = idx < 752 && idx < nvalid * 47;
            vq[it] = ok ? q[fbase * 47 + (ok ? idx : 0)] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < 6; ++it) {    // 16 x 24 = 384 values of p and of r
            const int idx = lane + 64 * it;
            const bool ok = idx < nvalid * 24;
            vp[it] = ok ? p[fbase * 24 + (ok ? idx : 0)] : ((idx % 24) == 0 ? 1.f : 0.f);   // missing system: p = e_0
            vr[it] = ok ? r[(fbase + (ok ? idx / 24 : 0)) * r_stride + r_off + (ok ? idx % 24 : 0)] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 12; ++it) {
            const int idx = lane + 64 * it;
            if (idx < 752) wl[(idx / 47) * kTq + (idx % 47)] = vq[it];
        }
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int idx = lane + 64 * it, fr = idx / 24, k = idx - fr * 24;
            wl[fr * kTq + 52 + 27 + k] = vp[it];     // mirrored Toeplitz window: p[|d|] at 27 + d
            wl[fr * kTq + 52 + 27 - k] = vp[it];
            wl[fr * kTq + 104 + k] = vr[it];
        }
        __builtin_amdgcn_wave_barrier();
        const float* rt_q = wl + nq * kTq;
        const float* rr_q = rt_q + 52;
        float xq[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, keep_if(gq.m[1], -1.f)};
        bool bad = false;
        {
            f32x4 a[blk::NBLK];
            float ninvs[M1];
            int gsv = gs;
            asm volatile("" : "+v"(gsv));
            const float* zr = cst;
            const float* pa6 = gsv == 0 ? cst + 32 : (gsv == 1 ? rt_q + 104 : zr);   // column 24: e_24 | right-hand side | 0
            blk_build_rows<0>(a, rt_q, rr_q, pa6, zr, gs);
            blk_elim_all(a, gq, ninvs, std::make_integer_sequence<int, M1>{});
            blk_backsub_all(a, xq, gq, ninvs, std::make_integer_sequence<int, blk::NG>{});
            // No pivoting here: that is sound for the positive definite systems of the analysis (gamma in [-1, 0]), and nothing
            // guarantees it for an arbitrary caller.  A pivot that is not positive (ninv = -1 / pivot not negative, or not finite)
            // marks the system: it is solved again below, with row pivoting, by the whole wave (th_solve_reg.h) -- the answer the
            // reference's LAPACK call gives.  (Round 3 wrote NaN and re-solved in a second launch that every call paid for.)
#pragma unroll
            for (int k = 0; k < M1 - 1; ++k) bad |= !(ninvs[k] < 0.f && ninvs[k] > -3.0e38f);
        }
        const long f = tile * 16 + nq;
        if (f < F && !bad) {
#pragma unroll
            for (int c = 0; c < 6; ++c) g[f * 24 + gs + 4 * c] = add ? add[f * 24 + gs + 4 * c] + xq[c] : xq[c];
        }
        unsigned long long marked = __ballot(bad && gs == 0 && f < F);
        while (marked) {   // uniform; normally empty
            const int bl = __builtin_ctzll(marked);
            marked &= marked - 1;
            const int sy = bl >> 2;
            const float* qs2 = wl + sy * kTq;               // q window
            const float* ps2 = qs2 + 52 + 27;               // p[d] at the centre of the mirrored window
            const float rhs = lane < 24 ? qs2[104 + lane] : 0.f;
            int col;
            float sol;
            th_solve_reg<float, 24>(ps2, qs2, rhs, 24, lane, col, sol);
            const long fs = tile * 16 + sy;
            if (lane < 24) g[fs * 24 + col] = add ? add[fs * 24 + col] + sol : sol;
        }
    }
}

int thsolve_quad24_fwd(const void* p, const void* q, const void* r, int64_t F, void* g, hipStream_t st, int r_stride, int r_off,
                       const void* add)
{
    long blocks = ((F + 15) / 16 + 3) / 4;
    if (blocks > 256L * 4) blocks = 256L * 4;
    hipLaunchKernelGGL(thsolve_quad24_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)p, (const float*)q,
                       (const float*)r, (long)F, (float*)g, r_stride, r_off, (const float*)add);
    return check_launch("th_solve_quad_fwd");
}

}  // namespace dsa
#include "mcep_mfma_f16.h"
#include "mcep_mfma_bwd_f16.h"
#include "mcep_mfma_bwd2_f16.h"
#include "mgcep_step_f16.h"
#include "mcep_resid_f16.h"
#include "mcep_resid_bwd_f16.h"
#include "mcep_glogx_f16.h"
#include "mcep_big_f16.h"
#include "mcep_big4_f16.h"
#ifdef DSA_MCEP_BWD_PAIR_EXPERIMENT   // round 5: built, measured, not adopted (tools/experiments/mcep_mfma_bwd_pair.h)
#include "../../tools/experiments/mcep_mfma_bwd_pair.h"
#endif
namespace dsa {

int mcep_mfma_supported(int nfft, int M, int dtype) { return dtype == DSA_F32 && nfft == 512 && M == 24; }

// bytes of the prepared operand images (forward + backward) of one configuration
int64_t mcep_mfma_images_bytes() { return ((int64_t)mhb::IMG_B_BYTES + 255) & ~(int64_t)255; }

// Splits (G, D, E) into the binary16 hi / lo operand images, in MFMA lane order, that both kernels consume
// (two tiny launches; done once per configuration by the caller, not per analysis call).
int mcep_mfma_prepare(const void* G, const void* D, const void* E, void* images, hipStream_t st)
{
    hipLaunchKernelGGL(mcep_h_prep_kernel, dim3((mh::IMG_D + mh::IMG_E + mh::IMG_G + 255) / 256), dim3(256), 0, st,
                       (const float*)G, (const float*)D, (const float*)E, (_Float16*)images);
    hipLaunchKernelGGL(mcep_hb_prep_kernel, dim3((mhb::IMG_EB + mhb::IMG_DB + mhb::IMG_GB + 255) / 256), dim3(256), 0, st,
                       (const float*)G, (const float*)D, (const float*)E, (_Float16*)images);
    return check_launch("mcep_prepare");
}

// `scratch`: DSA_SCRATCH_BYTES of caller-owned device memory; the first two words are this launch's tile counters.
static unsigned int* reset_queue(void* scratch, hipStream_t st, int words = 2)
{
    if (hipMemsetAsync(scratch, 0, words * sizeof(unsigned int), st) != hipSuccess) return nullptr;
    return (unsigned int*)scratch;
}

// `sti`: NULL (spectra in X) or the waveform side of the fused STFT -> mel-cepstrum launch (X is then unused)
int mcep_mfma_fwd(const void* X, int64_t F, int n_iter, const void* G, const void* D, const void* E, const void* av,
                  const void* images, void* scratch, void* mc, void* hist, hipStream_t st, bool scratch_clean = false,
                  const StftIn* sti = nullptr, bool hist_has_rt = false, bool overlapped = false, int reserve_cus = 0)
{
    // DSA_ALGO_HIST_HAS_RT: the caller's history buffer continues behind the (n_iter + 1, F, 25) iterates with (n_iter, F, 49) rows of rt
    float* hist_rt = (hist && hist_has_rt) ? (float*)hist + (size_t)(n_iter + 1) * (size_t)F * mm::M1 : nullptr;
    constexpr int WAVES = 8;
    const int lds_bytes = (sti ? mh::h_lds_floats_fused(WAVES) : mh::h_lds_floats(WAVES)) * 4;
    static std::atomic<uint64_t> attr_devices{0}, attr_devices_fused{0}, attr_devices_rt{0}, attr_devices_fused_rt{0};
    const bool rt = hist_rt != nullptr;
    // reflect / replicate / circular padding, zmean, relative floor: instantiations of their own
    const bool padm = sti && (sti->pad_mode != (int)DSA_PAD_CONSTANT || sti->zmean || sti->floor_lin >= 0.f);
    static std::atomic<uint64_t> attr_devices_padm{0}, attr_devices_padm_rt{0};
    const void* kern = padm ? (rt ? (const void*)mcep_mfma_fwd_kernel_h<WAVES, true, true, true> : (const void*)mcep_mfma_fwd_kernel_h<WAVES, true, false, true>)
                     : sti  ? (rt ? (const void*)mcep_mfma_fwd_kernel_h<WAVES, true, true> : (const void*)mcep_mfma_fwd_kernel_h<WAVES, true, false>)
                            : (rt ? (const void*)mcep_mfma_fwd_kernel_h<WAVES, false, true> : (const void*)mcep_mfma_fwd_kernel_h<WAVES, false, false>);
    if (!ensure_dynamic_lds(kern, lds_bytes, padm ? (rt ? attr_devices_padm_rt : attr_devices_padm)
                                             : sti ? (rt ? attr_devices_fused_rt : attr_devices_fused) : (rt ? attr_devices_rt : attr_devices)))
        return fail(DSA_ERR_LAUNCH, "mcep_mfma: cannot reserve the LDS operand images%s");
    long ntiles16 = (long)((F + 15) / 16);
    long blocks = (ntiles16 + WAVES - 1) / WAVES;
    long grid = blocks < 256 ? blocks : 256;  // one persistent workgroup per CU
    // DSA_ALGO_RESERVE_CUS(n): n CUs stay free for a kernel of ANOTHER stream that has to run beside this launch (a collective's: a
    // persistent workgroup fills its CU's LDS and registers, so nothing else starts on a CU before the launch's tail)
    if (reserve_cus > 0 && grid == 256 && !overlapped) {
        // at least reserve_cus, and up to twice as many where they cost no further round of tiles: a launch is rounds of (8 grid) tiles, a
        // short last round (at most half full) goes to one wave per SIMD pair and takes 0.8 of a round (below).  A side kernel whose
        // workgroup count EQUALS the free CUs was measured to start late now and then (profiles/r06_reserve_cus_ab.txt): slack is cheap
        auto rounds = [&](long g) {
            const long s = g * WAVES, full = ntiles16 / s, rest = ntiles16 - full * s;
            return (double)full + (rest == 0 ? 0.0 : (full > 0 && rest <= s / 2) ? 0.8 : 1.0);
        };
        const long g_hi = 256 - (reserve_cus < 63 ? reserve_cus : 63), g_lo = 256 - 2 * (reserve_cus < 63 ? reserve_cus : 63) < 128 ? 128 : 256 - 2 * (reserve_cus < 63 ? reserve_cus : 63);
        grid = g_hi;
        for (long g = g_hi - 1; g >= g_lo; --g)
            if (rounds(g) <= rounds(g_hi) + 1e-9) grid = g;
    }
    // the kernel zeroes the counters again when its last wave retires: a caller that vouches for a clean scratch
    // (DSA_ALGO_SCRATCH_IS_CLEAN) saves the fill launch
    unsigned int* queue = scratch_clean ? (unsigned int*)scratch : reset_queue(scratch, st, 3);
    if (!queue) return fail(DSA_ERR_LAUNCH, "mcep_mfma: cannot reset the tile queue%s");
    // see the ticket comment in the kernel: a short last round goes to one wave per SIMD pair
    const long slots = grid * WAVES, full = ntiles16 / slots * slots, rest = ntiles16 - full;
    long tiles_shared = (full > 0 && rest > 0 && rest <= slots / 2) ? full : ntiles16;
    // DSA_ALGO_OVERLAPPED_LAUNCHES: the short round packed onto the first tail_wgs workgroups at two waves per SIMD; everyone else
    // exits and leaves its CU to the next launch on the caller's other stream (include/diffsptk_amd.h)
    int tail_wgs = 0;
    if (overlapped && full > 0 && rest > 0 && grid == 256) {
        tiles_shared = full;
        tail_wgs = (int)((rest + WAVES - 1) / WAVES);
    }
#define DSA_MCEP_FWD_LAUNCH(FU, RT, XPTR, STI)                                                                                       \
    hipLaunchKernelGGL((mcep_mfma_fwd_kernel_h<WAVES, FU, RT>), dim3((unsigned)grid), dim3(WAVES * 64), lds_bytes, st, (const float*)(XPTR), \
                       (long)F, n_iter, (const float*)G, (const float*)D, (const float*)E, (const float*)av, (float*)mc, (float*)hist,  \
                       ntiles16, tiles_shared, queue, (const _Float16*)images, STI, hist_rt, tail_wgs)
    if (padm) {
        if (rt) hipLaunchKernelGGL((mcep_mfma_fwd_kernel_h<WAVES, true, true, true>), dim3((unsigned)grid), dim3(WAVES * 64), lds_bytes, st, (const float*)nullptr,
                                   (long)F, n_iter, (const float*)G, (const float*)D, (const float*)E, (const float*)av, (float*)mc, (float*)hist,
                                   ntiles16, tiles_shared, queue, (const _Float16*)images, *sti, hist_rt, tail_wgs);
        else hipLaunchKernelGGL((mcep_mfma_fwd_kernel_h<WAVES, true, false, true>), dim3((unsigned)grid), dim3(WAVES * 64), lds_bytes, st, (const float*)nullptr,
                                (long)F, n_iter, (const float*)G, (const float*)D, (const float*)E, (const float*)av, (float*)mc, (float*)hist,
                                ntiles16, tiles_shared, queue, (const _Float16*)images, *sti, hist_rt, tail_wgs);
        return check_launch("stft512_mcep_fused_fwd");
    }
    if (sti) {
        if (rt) DSA_MCEP_FWD_LAUNCH(true, true, nullptr, *sti);
        else DSA_MCEP_FWD_LAUNCH(true, false, nullptr, *sti);
        return check_launch("stft512_mcep_fused_fwd");
    }
    if (rt) DSA_MCEP_FWD_LAUNCH(false, true, X, StftIn{});
    else DSA_MCEP_FWD_LAUNCH(false, false, X, StftIn{});
#undef DSA_MCEP_FWD_LAUNCH
    return check_launch("mcep_mfma_fwd");
}

// STFT (frame length 400, fft_length 512, power format, constant padding) -> MelCepstralAnalysis (cep_order 24) in one launch
int stft_mcep_fused_fwd(const void* x, int64_t B, int64_t T, int P, int center, const void* window, const void* twiddle, double eps,
                        int n_iter, const void* G, const void* D, const void* E, const void* av, const void* images, void* scratch,
                        void* mc, void* hist, void* X_out, hipStream_t st, bool scratch_clean, bool hist_has_rt, bool overlapped, int pad_mode,
                        int zmean, float floor_lin, int reserve_cus)
{
    const int64_t N = T <= 0 ? 0 : (T - 1) / P + 1;
    StftIn sti;
    sti.x = (const float*)x;
    sti.Tlen = (long)T;
    sti.N = (long)N;
    sti.P = P;
    sti.left = center ? mh::FU_LC / 2 : 0;
    sti.w = (const float*)window;
    sti.twiddle = (const float*)twiddle;
    sti.eps = (float)eps;
    sti.X_out = (float*)X_out;
    sti.pad_mode = pad_mode;
    sti.zmean = zmean;
    sti.floor_lin = floor_lin;
    return mcep_mfma_fwd(nullptr, B * N, n_iter, G, D, E, av, images, scratch, mc, hist, st, scratch_clean, &sti, hist_has_rt, overlapped, reserve_cus);
}

// mcep.py:208-222, all n_iter steps in one persistent launch, for orders 32 .. 54 (the 48 kHz set-ups fft_length 2048 / order 49 and
// 1024 / order 34 among them); DSA_ERR_UNSUPPORTED (no error text) otherwise --
// the caller then runs the step as two launches.  `images`: dsa_mcep_resid_prepare's.
int mcep_big_newton(const void* logx, int64_t F, int K, const void* mc_in, int M1, const void* images, const void* av, int n_iter,
                    void* mc_out, hipStream_t st)
{
    const int ks1 = (M1 + 31) / 32, nt = (2 * M1 - 1 + 15) / 16;
    // the orders of the octet-layout solver, 35 .. 54, and -- quad layout -- 32 .. 34 (the 48 kHz set-up fft_length 1024 / order 34)
    if (!(ks1 == 2 && M1 >= 33 && M1 <= 55 && K >= 4)) return DSA_ERR_UNSUPPORTED;
    static const bool off = [] { const char* e = getenv("DSA_MCEP_BIG"); return e && e[0] == '0'; }();   // A/B: the two-launch step
    if (off) return DSA_ERR_UNSUPPORTED;
    // The kernel's INSTANTIATION is chosen by the order alone; its tile shape by the batch.  A persistent launch is rounds of 256 tiles
    // (one eight-wave workgroup per CU) and a round takes a tile's time however few tiles it has.  Narrow tiles (64 frames, two
    // waves per 16-frame group: the shortest step) for up to one round; beyond it a PLAN: W rounds of wide tiles (128 frames, a wave per
    // group: 1.6 x a narrow round's time for twice the frames) over the first W x 32 768 frames, narrow rounds over the rest -- the W
    // that minimises the sum, two launches on the caller's stream over disjoint frames.  Both shapes -- and the two launches per
    // step -- give the same bits (mcep_resid_f16.h sums the even and the odd stages separately, as the two waves of a narrow group /
    // the one wave of a wide group do), so a frame's result does not depend on the plan (tests/test_gpu_configs.py).
    // Measured, us per analysis, fft_length 2048 / order 49, 10 steps (profiles/r06_mcep_big_newton.txt, two launches per step ->
    // narrow): 200 frames 696 -> 593, 3 200 724 -> 630, 12 800 920 -> 672, 20 000 1 250 -> 1 300, 40 000 2 027 -> 2 008, 102 400
    // 4 813 -> 4 720; with the plan: profiles/r06_mcep_big_wide.txt.
    const char* wide_e = getenv("DSA_MCEP_BIG_WIDE");   // A/B and tests: 0 never, 1 always (read per call: the tests switch it in-process)
    const int wide_env = wide_e ? atoi(wide_e) : -1;
    const bool quad = M1 <= 35;
    // Quad-layout orders: the rounds of 32 768 frames run as TWIN workgroups (mcep_big4_f16.h: two four-wave workgroups per CU, the
    // second half a step late: 1024 / 34 at 122 880 frames 2.15 -> 1.99 ms, profiles/r06_mcep_big_twin.txt) instead of eight-wave wide
    // tiles (DSA_MCEP_BIG_TWIN=0, A/B); the same bits either way.  At the octet-layout orders the twin shape measured no faster than
    // the plan (each workgroup streams the images itself: twice the bytes from L2 per frame) and is not instantiated.
    const char* twin_e = getenv("DSA_MCEP_BIG_TWIN");
    const bool twin = quad && !(twin_e && twin_e[0] == '0');
    const long FN = 256L * 64, FW = 256L * 128;          // frames a full round of narrow / wide (or twin) tiles takes
    long wide_rounds = 0;
    if (wide_env > 0 || (wide_env < 0 && twin && F > FN)) {
        // (twin workgroups take 64-frame tiles like the narrow shape: beyond one narrow round every frame goes to them -- measured
        //  against twin rounds + narrow rounds for the rest, 1024 / 34: 79 200 frames 1.28 / 1.34 ms, 245 760 3.62 / 3.88)
        wide_rounds = (long)((F + FW - 1) / FW);
    } else if (wide_env < 0 && F > FN) {
        // a wide round's time in units of a narrow round's (measured: 1.09 / 0.67 ms at 2048 / 49, 0.48 / 0.34 ms at 1024 / 34)
        const double tw = quad ? 1.42 : 1.63;
        double best = (double)((F + FN - 1) / FN);
        for (long w = 1; w <= (long)((F + FW - 1) / FW); ++w) {
            const long rest = F - w * FW;
            const double c = w * tw + (rest > 0 ? (double)((rest + FN - 1) / FN) : 0.0);
            if (c < best - 1e-9) {
                best = c;
                wide_rounds = w;
            }
        }
    }
    const long F_wide = wide_rounds * FW < F ? wide_rounds * FW : (wide_rounds > 0 ? (long)F : 0);
    const char* stag_e = getenv("DSA_MCEP_BIG_STAGGER");
    const int stagger = stag_e ? atoi(stag_e) : 5;                // x ~8 k cycles (measured 4 .. 6 best: profiles/r06_mcep_big_twin.txt)
#define DSA_BIG_NEWTON_1(NTV, NGV, NMINV, QUADV, WIDEV, F0, FC)                                                                         \
    do {                                                                                                                                \
        constexpr int lds_b = mbg::lds_floats<2, NTV, NGV, QUADV, WIDEV>() * 4;                                                         \
        static_assert(lds_b <= 160 * 1024, "mcep_big_newton: LDS");                                                                     \
        static std::atomic<uint64_t> attr{0};                                                                                           \
        if (!ensure_dynamic_lds((const void*)mcep_big_newton_kernel<2, NTV, NGV, NMINV, QUADV, WIDEV>, lds_b, attr))                    \
            return fail(DSA_ERR_LAUNCH, "mcep_big_newton: cannot reserve LDS%s");                                                       \
        const long tiles_ = ((FC) + (WIDEV ? 127 : 63)) / (WIDEV ? 128 : 64);                                                           \
        const long grid_ = tiles_ < 256 ? tiles_ : 256;   /* persistent: one eight-wave workgroup per CU (105 / 148 KB of LDS) */        \
        hipLaunchKernelGGL((mcep_big_newton_kernel<2, NTV, NGV, NMINV, QUADV, WIDEV>), dim3((unsigned)grid_), dim3(512), lds_b, st,     \
                           (const float*)logx + (F0) * (long)K, (long)(FC), K, (const float*)mc_in + (F0) * (long)M1, M1,               \
                           (const _Float16*)images, (const float*)av, n_iter, (float*)mc_out + (F0) * (long)M1);                        \
    } while (0)
#define DSA_BIG_NEWTON4_1(NTV, NGV, NMINV, QUADV, F0, FC)                                                                               \
    do {                                                                                                                                \
        constexpr int lds_b = mbg4::lds_floats<2, NTV, NGV, QUADV>() * 4;                                                               \
        static_assert(lds_b <= 80 * 1024, "mcep_big_newton: two workgroups per CU");                                                    \
        static std::atomic<uint64_t> attr{0};                                                                                           \
        if (!ensure_dynamic_lds((const void*)mcep_big_newton4_kernel<2, NTV, NGV, NMINV, QUADV>, lds_b, attr))                          \
            return fail(DSA_ERR_LAUNCH, "mcep_big_newton: cannot reserve LDS%s");                                                       \
        const long tiles_ = ((FC) + 63) / 64;                                                                                           \
        const long grid_ = tiles_ < 512 ? tiles_ : 512;   /* persistent: two four-wave workgroups per CU */                             \
        hipLaunchKernelGGL((mcep_big_newton4_kernel<2, NTV, NGV, NMINV, QUADV>), dim3((unsigned)grid_), dim3(256), lds_b, st,           \
                           (const float*)logx + (F0) * (long)K, (long)(FC), K, (const float*)mc_in + (F0) * (long)M1, M1,               \
                           (const _Float16*)images, (const float*)av, n_iter, (float*)mc_out + (F0) * (long)M1, stagger);               \
    } while (0)
#define DSA_BIG_NEWTON_Q(NTV, NGV, NMINV)                                                                                               \
    do {                                                                                                                                \
        if (F_wide > 0 && twin) DSA_BIG_NEWTON4_1(NTV, NGV, NMINV, true, 0L, F_wide);                                                   \
        else if (F_wide > 0) DSA_BIG_NEWTON_1(NTV, NGV, NMINV, true, true, 0L, F_wide);                                                 \
        if (F_wide < (long)F) DSA_BIG_NEWTON_1(NTV, NGV, NMINV, true, false, F_wide, (long)F - F_wide);                                 \
    } while (0)
#define DSA_BIG_NEWTON(NTV, NGV, NMINV, QUADV)                                                                                          \
    do {                                                                                                                                \
        if (F_wide > 0) DSA_BIG_NEWTON_1(NTV, NGV, NMINV, QUADV, true, 0L, F_wide);                                                     \
        if (F_wide < (long)F) DSA_BIG_NEWTON_1(NTV, NGV, NMINV, QUADV, false, F_wide, (long)F - F_wide);                                \
    } while (0)
    // (M1 = order + 1; the solver's instantiations as thsolve_quadn_fwd picks them: quad <9,28> up to 35, <11,36> up to 43, <13,44> up to
    //  51, <14,52>)
    if (quad) {
        DSA_BIG_NEWTON_Q(5, 9, 28);
    } else if (M1 <= 43) {
        if (nt == 5) DSA_BIG_NEWTON(5, 11, 36, false);
        else DSA_BIG_NEWTON(6, 11, 36, false);
    } else if (M1 <= 51) {
        if (nt == 6) DSA_BIG_NEWTON(6, 13, 44, false);
        else DSA_BIG_NEWTON(7, 13, 44, false);
    } else {
        DSA_BIG_NEWTON(7, 14, 52, false);
    }
#undef DSA_BIG_NEWTON
#undef DSA_BIG_NEWTON_Q
#undef DSA_BIG_NEWTON_1
#undef DSA_BIG_NEWTON4_1
    return check_launch("mcep_big_newton");
}

int mcep_mfma_bwd(const void* gmc, const void* X, const void* hist, int64_t F, int n_iter, const void* av,
                  const void* images, void* scratch, void* gX, hipStream_t st, bool has_workspace = false, bool hist_has_rt = false)
{
    const float* hist_rt = hist_has_rt ? (const float*)hist + (size_t)(n_iter + 1) * (size_t)F * mm::M1 : nullptr;
#ifdef DSA_MCEP_BWD_PAIR_EXPERIMENT
    // round-5 experiment: pairs of waves that split the bins of a tile, two waves per SIMD (2.9 ms against 1.57: not adopted)
    static const bool pair_on = [] { const char* e = getenv("DSA_MCEP_BWD_PAIR"); return !(e && e[0] == '0'); }();
    if (pair_on) {
        const int lds_p = mhp::P_LDS_FLOATS * 4;
        static std::atomic<uint64_t> attr_p{0};
        if (!ensure_dynamic_lds((const void*)mcep_mfma_bwd_pair_kernel, lds_p, attr_p))
            return fail(DSA_ERR_LAUNCH, "mcep_mfma_bwd: cannot reserve the LDS operand images%s");
        const long nt = (long)((F + 15) / 16);
        long blocks_p = (nt + mhp::PAIRS - 1) / mhp::PAIRS;
        if (blocks_p > 256) blocks_p = 256;
        hipLaunchKernelGGL(mcep_mfma_bwd_pair_kernel, dim3((unsigned)blocks_p), dim3(512), lds_p, st, (const float*)gmc, (const float*)X,
                           (const float*)hist, (long)F, n_iter, (const float*)av, (float*)gX, nt, (const _Float16*)images);
        return check_launch("mcep_mfma_bwd_pair");
    }
#endif
    // With the forward's rt rows at hand the sweep runs on the two-waves-per-SIMD kernel (mcep_mfma_bwd2_f16.h); DSA_MCEP_BWD2=0: A/B
    const char* e2 = getenv("DSA_MCEP_BWD2");   // (read per call: the tests switch it)
    const bool bwd2_on = !(e2 && e2[0] == '0');
    const bool two = hist_rt && bwd2_on;
    const int waves = two ? mh2::WAVES_2 : mhb::WAVES_B;
    const int lds_bytes = (two ? mh2::C_LDS_FLOATS : mhb::B_LDS_FLOATS) * 4;
    static std::atomic<uint64_t> attr_devices{0};
    if (!two && !ensure_dynamic_lds((const void*)mcep_mfma_bwd_kernel_h<false>, lds_bytes, attr_devices))
        return fail(DSA_ERR_LAUNCH, "mcep_mfma_bwd: cannot reserve the LDS operand images%s");
    unsigned int* queue = reset_queue(scratch, st, 13);
    if (!queue) return fail(DSA_ERR_LAUNCH, "mcep_mfma_bwd: cannot reset the tile queue%s");
    long ntiles16 = (long)((F + 15) / 16);
    long blocks = (ntiles16 + waves - 1) / waves;
    long grid = blocks < 256 ? blocks : 256;
    // A last round that fills at most half of the wave slots is cut into pieces of Newton steps (see the kernel) when the caller's
    // scratch carries the hand-over workspace behind the counters (DSA_ALGO_SCRATCH_HAS_WORKSPACE); DSA_MCEP_SPLIT=0: A/B
    static const bool split_on = [] { const char* e = getenv("DSA_MCEP_SPLIT"); return !(e && e[0] == '0'); }();
    const long slots = grid * waves, rounds = ntiles16 / slots, rest = ntiles16 - rounds * slots;
    int split_tiles = 0, split_pieces = 0;
    if (!two && has_workspace && split_on && rounds >= 1 && rest > 0 && rest <= slots / 2 && rest <= 512 && n_iter >= 2) {
        long pieces = slots / rest;
        if (pieces > n_iter) pieces = n_iter;
        if (pieces > rounds + 1) pieces = rounds + 1;
        if (pieces > 9) pieces = 9;   // one counter word of the scratch per piece level
        if (const char* ep = getenv("DSA_MCEP_SPLIT_PIECES")) { const long cap = atol(ep); if (cap >= 2 && pieces > cap) pieces = cap; }   // (A/B)
        split_tiles = (int)rest;
        split_pieces = (int)pieces;
    }
    if (two) {
        static std::atomic<uint64_t> attr_two{0};
        if (!ensure_dynamic_lds((const void*)mcep_mfma_bwd2_kernel_h, lds_bytes, attr_two))
            return fail(DSA_ERR_LAUNCH, "mcep_mfma_bwd: cannot reserve the LDS operand images%s");
        hipLaunchKernelGGL(mcep_mfma_bwd2_kernel_h, dim3((unsigned)grid), dim3(waves * 64), lds_bytes, st, (const float*)gmc,
                           (const float*)X, (const float*)hist, (long)F, n_iter, (const float*)av, (float*)gX, ntiles16, queue,
                           (const _Float16*)images, hist_rt);
        return check_launch("mcep_mfma_bwd2");
    }
    if (hist_rt) {
        static std::atomic<uint64_t> attr_rt{0};
        if (!ensure_dynamic_lds((const void*)mcep_mfma_bwd_kernel_h<true>, lds_bytes, attr_rt))
            return fail(DSA_ERR_LAUNCH, "mcep_mfma_bwd: cannot reserve the LDS operand images%s");
        hipLaunchKernelGGL(mcep_mfma_bwd_kernel_h<true>, dim3((unsigned)grid), dim3(256), lds_bytes, st, (const float*)gmc,
                           (const float*)X, (const float*)hist, (long)F, n_iter, (const float*)av, (float*)gX, ntiles16, queue,
                           (const _Float16*)images, split_tiles, split_pieces,
                           reinterpret_cast<float*>(static_cast<char*>(scratch) + DSA_SCRATCH_BYTES), hist_rt);
        return check_launch("mcep_mfma_bwd");
    }
    hipLaunchKernelGGL(mcep_mfma_bwd_kernel_h<false>, dim3((unsigned)grid), dim3(256), lds_bytes, st, (const float*)gmc,
                       (const float*)X, (const float*)hist, (long)F, n_iter, (const float*)av, (float*)gX, ntiles16, queue,
                       (const _Float16*)images, split_tiles, split_pieces,
                       reinterpret_cast<float*>(static_cast<char*>(scratch) + DSA_SCRATCH_BYTES), (const float*)nullptr);
    return check_launch("mcep_mfma_bwd");
}

}  // namespace dsa
