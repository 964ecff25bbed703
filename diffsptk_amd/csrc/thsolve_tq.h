// Device side of the batched Toeplitz-plus-Hankel solve (csrc/thsolve_quad.hip: the scheme, the launchers): the quad- and octet-layout
// elimination / back-substitution templates and the two kernels.  A header since round 6: the persistent Newton kernel of the 48 kHz
// set-ups (csrc/mcep_big_f16.h, included by mcep_mfma.hip) runs the octet-layout solve inside its own step loop.
#pragma once

#include <utility>

#include "common.h"
#include "th_solve_reg.h"

namespace dsa {
namespace tq {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma441(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float keep_if(bool c, float v) { return c ? v : 0.f; }
template <int Q>
__device__ __forceinline__ float quad_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), Q * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_sum(float v)
{
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    return v;
}

template <int NG>
struct Blk {
    static constexpr int N = NG * (NG + 1) / 2;
    static constexpr int at(int rg, int cg) { return rg * NG - rg * (rg - 1) / 2 + (cg - rg); }
};

// One elimination step on pivot K (compile time)
template <int NG, int K>
__device__ __forceinline__ void elim_step(f32x4 (&a)[Blk<NG>::N], int gs, bool& bad)
{
    using B = Blk<NG>;
    constexpr int c0 = K >> 2, q = K & 3;
    const float piv = quad_bcast<q>(a[B::at(c0, c0)][q]);
    bad |= !(piv > 0.f && piv < 3.0e38f);
    const float ninv = -__builtin_amdgcn_rcpf(piv);
    float m[NG];
#pragma unroll
    for (int c = c0; c < NG; ++c) m[c] = a[B::at(c0, c)][q] * ninv;
    const float m0 = keep_if(gs > q, m[c0]);   // rows <= k of the pivot's own group keep their values
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (q < 3) {
#pragma unroll
        for (int c = c0; c < NG; ++c) a[B::at(c0, c)] = mfma441(m0, a[B::at(c0, c)][q], a[B::at(c0, c)]);
    }
#pragma unroll
    for (int rg = c0 + 1; rg < NG; ++rg) {
#pragma unroll
        for (int c = rg; c < NG; ++c) a[B::at(rg, c)] = mfma441(m[rg], a[B::at(c0, c)][q], a[B::at(rg, c)]);
    }
    __builtin_amdgcn_sched_barrier(0);
}
// All 4 NG - 1 steps, unconditionally: the right-hand side rides in the LAST column (4 NG - 1) whatever the order, and the rows /
// columns between the order and that column are the identity (pivot 1, multipliers 0) -- no step depends on n at run time (a step
// under its own `if` makes every register quadruple a phi at every branch: 1.1 KB of scratch per lane at NG = 13).
template <int NG, int... Ks>
__device__ __forceinline__ void elim_all(f32x4 (&a)[Blk<NG>::N], int gs, bool& bad, std::integer_sequence<int, Ks...>)
{
    (elim_step<NG, Ks>(a, gs, bad), ...);
}

// x_k = -(sum_{j > k} U_kj x_j - b_k) / U_kk, the right-hand-side slot of xq preset to -1 on its owner lane (the diagonal and
// sub-diagonal lanes of the row's own slot still hold 0 in xq when the row is solved)
template <int NG, int RG, int I>
__device__ __forceinline__ void backsub_row(const f32x4 (&a)[Blk<NG>::N], float (&xq)[NG], const float (&part)[4], int gs)
{
    using B = Blk<NG>;
    if constexpr (4 * RG + I < 4 * NG - 1) {
        const float sl = quad_sum(__builtin_fmaf(a[B::at(RG, RG)][I], xq[RG], part[I]));
        const float diag = quad_bcast<I>(a[B::at(RG, RG)][I]);   // the diagonal element sits on lane I of the quad
        const float xk = -sl * __builtin_amdgcn_rcpf(diag);
        xq[RG] = gs == I ? xk : xq[RG];
    }
}
template <int NG, int RG>
__device__ __forceinline__ void backsub_group(const f32x4 (&a)[Blk<NG>::N], float (&xq)[NG], int gs)
{
    using B = Blk<NG>;
    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = RG + 1; c < NG; ++c) {
        const f32x4 v = a[B::at(RG, c)];
#pragma unroll
        for (int i = 0; i < 4; ++i) part[i] = __builtin_fmaf(v[i], xq[c], part[i]);
    }
    backsub_row<NG, RG, 3>(a, xq, part, gs);
    backsub_row<NG, RG, 2>(a, xq, part, gs);
    backsub_row<NG, RG, 1>(a, xq, part, gs);
    backsub_row<NG, RG, 0>(a, xq, part, gs);
}
template <int NG, int... Gs>
__device__ __forceinline__ void backsub_all(const f32x4 (&a)[Blk<NG>::N], float (&xq)[NG], int gs, std::integer_sequence<int, Gs...>)
{
    (backsub_group<NG, NG - 1 - Gs>(a, xq, gs), ...);
}

// LDS record of a system (floats), REC = 16 NG + 3 (odd: the 16 records of a wave start on different banks):
//   q window  [0, QW), QW = 8 NG - 1: q[k] at k, zeros from 2 n - 1 on           (the Hankel term of element (i, j) is entry i + j <= 2 CN)
//   p window  [QW, QW + 4 NG + 3): p[|d|] at QW + 3 + d, d in [-3, 4 NG), zero for |d| >= n   (upper triangle: j - i >= -3 inside a block)
//   rhs       [QW + 4 NG + 3, REC): r[k] - sub[k], zeros from n on
// NMIN: the smallest order this instantiation is launched for -- rows and columns below it need no mask.
template <int NG, int NMIN>
__global__ __launch_bounds__(256, 1) void thsolve_quadn_kernel(const float* __restrict__ p, int ldp, const float* __restrict__ q, int ldq,
                                                               const float* __restrict__ r, int ldr, const float* __restrict__ sub,
                                                               const float* add, long F, int n, float* g)   // (g may be add: dsa_mcep_newton_update in place)
{
    using B = Blk<NG>;
    constexpr int CN = 4 * NG - 1;              // the right-hand side's column = the largest order
    constexpr int QW = 8 * NG - 1;
    constexpr int PO = QW + 3;                  // p[0]
    constexpr int RO = QW + 4 * NG + 3;         // rhs[0]
    constexpr int REC = ((16 * NG + 3 - 4 + 31) / 32) * 32 + 4;   // >= 16 NG + 3 and = 4 (mod 32): see the octet kernel (rounds 3-4: 16 NG + 3, odd)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* wl = lds + wave * 16 * REC;
    const long ntiles = (F + 15) / 16;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        __builtin_amdgcn_wave_barrier();
        // lane-derived values are derived again per tile from an opaque copy (see thsolve_octn_kernel: hoisted, they are spilled
        // across the elimination and every reload serialises the staging loads)
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int nq = lane >> 2, gs = lane & 3;
        const float subv = (sub && lane < n) ? sub[lane] : 0.f;
#ifdef TQ_STAMPS   // measurement builds only (tools/stamps_thsolve.py): phase durations leave in the output rows
        const long long ts0 = __builtin_readcyclecounter();
#endif
        const long fbase = tile * 16;
        const int nvalid = (int)((F - fbase) < 16 ? (F - fbase) : 16);
        // stage the 16 records, one system per round of the wave (orders up to 55: one load for p and r, two for q); a missing system
        // is the identity with a zero right-hand side.  All 64 loads are issued before the first is used (one round trip to memory
        // instead of four: a tile is a wave's whole life at 12 800 systems, nothing else hides it), zeros go in as 16-byte stores.
        // (Flat index loops with a division per element were a fifth of the kernel.)
        {
            typedef float zf4 __attribute__((ext_vector_type(4)));
            zf4* wz = reinterpret_cast<zf4*>(wl);
            for (int e = lane; e < 4 * REC; e += 64) wz[e] = zf4{0.f, 0.f, 0.f, 0.f};
        }
        float q0[16], q1[16], p0[16], r0[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const long f = fbase + (s < nvalid ? s : 0);
            q0[s] = q[f * (long)ldq + (lane < 2 * n - 1 ? lane : 0)];
            q1[s] = q[f * (long)ldq + (lane + 64 < 2 * n - 1 ? lane + 64 : 0)];
            p0[s] = p[f * (long)ldp + (lane < n ? lane : 0)];
            r0[s] = r[f * (long)ldr + (lane < n ? lane : 0)];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bool ok = s < nvalid;   // uniform
            float* rec = wl + s * REC;
            if (lane < 2 * n - 1) rec[lane] = ok ? q0[s] : 0.f;
            if (lane + 64 < 2 * n - 1) rec[lane + 64] = ok ? q1[s] : 0.f;
            if (lane < n) {
                const float pv = ok ? p0[s] : (lane == 0 ? 1.f : 0.f);
                rec[PO + lane] = pv;
                if (lane >= 1 && lane <= 3) rec[PO - lane] = pv;
                rec[RO + lane] = ok ? r0[s] - subv : 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef TQ_STAMPS
        const long long ts1 = __builtin_readcyclecounter();
#endif
        const float* qs = wl + nq * REC + gs;        // this lane's views: column offset gs folded in
        const float* pw = qs + PO;
        const float* rs = wl + nq * REC + RO;
        f32x4 a[B::N];
        // element (row, col = 4 cg + gs) = p[|col - row|] + q[row + col] -- compile-time offsets from the lane's views -- masked to
        // the order: columns n .. CN - 1 are zero, rows n .. CN - 1 the identity, column CN the right-hand side
#pragma unroll
        for (int rg = 0; rg < NG; ++rg) {
#pragma unroll
            for (int cg = rg; cg < NG; ++cg) {
                const bool cin = 4 * cg + 3 < NMIN || 4 * cg + gs < n;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 4 * rg + i;
                    float v = pw[4 * (cg - rg) - i] + qs[4 * (rg + cg) + i];
                    if (4 * cg + 3 >= NMIN) v = cin ? v : 0.f;
                    if (cg == rg && row >= NMIN && row < CN) v = (gs == i && row >= n) ? 1.f : v;
                    if (cg == NG - 1) v = gs == 3 ? rs[row] : v;
                    a[B::at(rg, cg)][i] = v;
                }
            }
        }
        bool bad = false;
#ifdef TQ_STAMPS
        asm volatile("" ::: "memory");
        const long long ts2 = __builtin_readcyclecounter();
#endif
        elim_all<NG>(a, gs, bad, std::make_integer_sequence<int, CN>{});
#ifdef TQ_STAMPS
        const long long ts3 = __builtin_readcyclecounter();
#endif
        float xq[NG];
#pragma unroll
        for (int c = 0; c < NG; ++c) xq[c] = (4 * c + gs == CN) ? -1.f : 0.f;
        backsub_all<NG>(a, xq, gs, std::make_integer_sequence<int, NG>{});
        // every lane of the quad saw the same pivots.  A system whose elimination met a non-positive or non-finite pivot is solved
        // again HERE, with row pivoting, by the whole wave (one row per lane: th_solve_reg) -- the answer the reference's LAPACK call
        // gives for an arbitrary symmetric system.  (Rounds 3-4 marked such rows with NaN and re-solved them in a second launch that
        // every call paid for.)
        const long f = fbase + nq;
        if (f < F) {
#pragma unroll
            for (int c = 0; c < NG; ++c) {
                const int col = 4 * c + gs;
                if (col < n && !bad) g[f * (long)n + col] = add ? add[f * (long)n + col] + xq[c] : xq[c];
            }
        }
#ifdef TQ_STAMPS
        {
            const long long ts4 = __builtin_readcyclecounter();
            if (f < F && gs == 0) {
                g[f * (long)n + 0] = (float)(ts1 - ts0);   // staging
                g[f * (long)n + 4] = (float)(ts2 - ts1);   // construction
                g[f * (long)n + 8] = (float)(ts3 - ts2);   // elimination
                g[f * (long)n + 12] = (float)(ts4 - ts3);  // back substitution + stores
            }
        }
#endif
        unsigned long long marked = __ballot(bad && gs == 0 && f < F);
        while (marked) {   // uniform; normally empty
            const int bl = __builtin_ctzll(marked);
            marked &= marked - 1;
            const int sy = bl >> 2;
            // opaque copies: everything this cold path derives from (n, lane) is otherwise hoisted out of the tile loop and parked
            // in scratch for the whole kernel
            int nn = n, ln = lane;
            asm volatile("" : "+s"(nn), "+v"(ln));
            const float* qs2 = wl + sy * REC;
            const float* ps2 = qs2 + PO;                    // ps2[d] = p[d]
            const float rhs = ln < nn ? qs2[RO + ln] : 0.f;
            int col;
            float sol;
            th_solve_reg<float, CN <= 32 ? 32 : (CN <= 48 ? 48 : 64)>(ps2, qs2, rhs, nn, ln, col, sol);
            const long fs = fbase + sy;
            if (ln < nn) g[fs * (long)nn + col] = add ? add[fs * (long)nn + col] + sol : sol;
        }
    }
}


// =====================================================================================================================
// Eight lanes per system (orders 36 .. 55): the same elimination with the block COLUMNS dealt to the two quads of an octet
// -- block column cg lives in quad (half) cg & 1, slot (rg, cp) of a lane is block (rg, 2 cp + h) -- so a lane holds 55 / 56
// register quadruples instead of 91 / 105, a wave 8 systems instead of 16, and twice as many waves share the work.
// Why: at 91 quadruples (364 registers) the compiler, splitting them over the two register files, spilled 0.9 KB per lane and
// put 2 340 moves between the files into the elimination; phase stamps of that kernel (tools/stamps_thsolve.py, 12 800 systems
// of order 50, ticks per tile): staging 24 k, construction 18 k, elimination 57 k, back substitution 28 k -- against
// 9 / 5 / 10 / 5 k at order 35, where everything fits: four times the time for three times the work.
// What crosses the halves: the pivot and the multipliers (the scaled pivot row is, by symmetry, the column of multipliers: the
// multiplier of row block rg is born in half rg & 1 and needed by both) -- one DPP row shift by four lanes, restricted to
// the receiving quads by the bank mask, per value; the pivot-row element a block update needs is already in the block's own
// half.  A matrix instruction updates slot (rg, cp) of both halves at once.  For odd rg the slot (rg, rg >> 1) of half 0 is
// block (rg, rg - 1), below the diagonal: never read as a pivot-row element or a multiplier, excluded from the back
// substitution, it only receives updates.
// =====================================================================================================================
template <int H0>
__device__ __forceinline__ float from_half(float v)   // every lane of the octet gets the value its lane gs holds in half H0
{
    const int i = __float_as_int(v);
    // row_shr:4 into the odd quads (bank mask 0b1010) / row_shl:4 into the even quads (0b0101); the other quads keep their own
    return __int_as_float(H0 == 0 ? __builtin_amdgcn_update_dpp(i, i, 0x114, 0xf, 0xa, false)
                                  : __builtin_amdgcn_update_dpp(i, i, 0x104, 0xf, 0x5, false));
}
__device__ __forceinline__ float other_half(float v)   // the partner quad's value
{
    const int i = __float_as_int(v);
    const int t = __builtin_amdgcn_update_dpp(0, i, 0x114, 0xf, 0xa, true);
    return __int_as_float(__builtin_amdgcn_update_dpp(t, i, 0x104, 0xf, 0x5, false));
}

template <int NG>
struct Oct {
    static constexpr int NCP = (NG + 1) / 2;                       // column pairs
    // slots before row block rg: sum over r < rg of NCP - (r >> 1), in closed form (a recursive definition is not folded after
    // unrolling, and a run-time index puts the whole matrix in scratch)
    static constexpr int row_off(int rg) { return rg * NCP - (rg >> 1) * ((rg - 1) >> 1); }
    static constexpr int N = row_off(NG);
    static constexpr int at(int rg, int cp) { return row_off(rg) + (cp - (rg >> 1)); }
};

// the multiplier of row block RG in step K: the scaled pivot-row element of column block RG, born in half RG & 1, sent to both
template <int NG, int K, int RG>
__device__ __forceinline__ float oct_mult(const f32x4 (&a)[Oct<NG>::N], float ninv)
{
    using O = Oct<NG>;
    constexpr int c0 = K >> 2, q = K & 3;
    const float own = a[O::at(c0, RG >> 1)][q] * ninv;
    return (RG & 1) ? from_half<1>(own) : from_half<0>(own);
}
// rows below the pivot's block row, one row block after the other; the multiplier of the next row block is formed while this one's
// matrix instructions issue -- one ahead, not all NG at once: the matrix leaves ~30 registers for everything else at two waves per SIMD
template <int NG, int K, int RG>
__device__ __forceinline__ void oct_rows(f32x4 (&a)[Oct<NG>::N], float m, float ninv)
{
    using O = Oct<NG>;
    constexpr int c0 = K >> 2, q = K & 3;
    if constexpr (RG < NG) {
        float mn = 0.f;
        if constexpr (RG + 1 < NG) mn = oct_mult<NG, K, RG + 1>(a, ninv);
#pragma unroll
        for (int cp = RG >> 1; cp < O::NCP; ++cp) a[O::at(RG, cp)] = mfma441(m, a[O::at(c0, cp)][q], a[O::at(RG, cp)]);
        oct_rows<NG, K, RG + 1>(a, mn, ninv);
    }
}
template <int NG, int K>
__device__ __forceinline__ void oct_elim_step(f32x4 (&a)[Oct<NG>::N], int gs, bool& bad)
{
    using O = Oct<NG>;
    constexpr int c0 = K >> 2, q = K & 3, h0 = c0 & 1, cp0 = c0 >> 1;
    const float piv = from_half<h0>(quad_bcast<q>(a[O::at(c0, cp0)][q]));
    bad |= !(piv > 0.f && piv < 3.0e38f);
    const float ninv = -__builtin_amdgcn_rcpf(piv);
    float mn = 0.f;
    if constexpr (c0 + 1 < NG) mn = oct_mult<NG, K, c0 + 1>(a, ninv);
    if constexpr (q < 3) {
        const float m0 = keep_if(gs > q, oct_mult<NG, K, c0>(a, ninv));   // rows <= k of the pivot's own group keep their values
#pragma unroll
        for (int cp = cp0; cp < O::NCP; ++cp) a[O::at(c0, cp)] = mfma441(m0, a[O::at(c0, cp)][q], a[O::at(c0, cp)]);
    }
    oct_rows<NG, K, c0 + 1>(a, mn, ninv);
    __builtin_amdgcn_sched_barrier(0);
}
template <int NG, int... Ks>
__device__ __forceinline__ void oct_elim_all(f32x4 (&a)[Oct<NG>::N], int gs, bool& bad, std::integer_sequence<int, Ks...>)
{
    (oct_elim_step<NG, Ks>(a, gs, bad), ...);
}

template <int NG, int RG, int I>
__device__ __forceinline__ void oct_backsub_row(const f32x4 (&a)[Oct<NG>::N], float (&xq)[Oct<NG>::NCP], const float (&part)[4], int gs, int h)
{
    using O = Oct<NG>;
    if constexpr (4 * RG + I < 4 * NG - 1) {
        constexpr int hd = RG & 1, cpd = RG >> 1;
        const float dterm = keep_if(h == hd, a[O::at(RG, cpd)][I] * xq[cpd]);   // the diagonal block's own row, solved entries only
        float sl = quad_sum(part[I] + dterm);
        sl += other_half(sl);
        const float diag = from_half<hd>(quad_bcast<I>(a[O::at(RG, cpd)][I]));
        const float xk = -sl * __builtin_amdgcn_rcpf(diag);
        xq[cpd] = (h == hd && gs == I) ? xk : xq[cpd];
    }
}
template <int NG, int RG>
__device__ __forceinline__ void oct_backsub_group(const f32x4 (&a)[Oct<NG>::N], float (&xq)[Oct<NG>::NCP], int gs, int h)
{
    using O = Oct<NG>;
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    // blocks right of the diagonal block: column block 2 cp + h > RG
#pragma unroll
    for (int cp = RG >> 1; cp < O::NCP; ++cp) {
        const f32x4 v = a[O::at(RG, cp)];
        if (2 * cp > RG) {                  // both halves are right of the diagonal
#pragma unroll
            for (int i = 0; i < 4; ++i) part[i] = __builtin_fmaf(v[i], xq[cp], part[i]);
        } else if (2 * cp + 1 > RG) {       // 2 cp == RG: half 1 only (half 0 is the diagonal block, handled per row)
            const float xs = keep_if(h == 1, xq[cp]);
#pragma unroll
            for (int i = 0; i < 4; ++i) part[i] = __builtin_fmaf(keep_if(h == 1, v[i]), xs, part[i]);
        }                                   // 2 cp + 1 == RG: half 1 is the diagonal block, half 0 lies below the diagonal
    }
    oct_backsub_row<NG, RG, 3>(a, xq, part, gs, h);
    oct_backsub_row<NG, RG, 2>(a, xq, part, gs, h);
    oct_backsub_row<NG, RG, 1>(a, xq, part, gs, h);
    oct_backsub_row<NG, RG, 0>(a, xq, part, gs, h);
}
template <int NG, int... Gs>
__device__ __forceinline__ void oct_backsub_all(const f32x4 (&a)[Oct<NG>::N], float (&xq)[Oct<NG>::NCP], int gs, int h,
                                                std::integer_sequence<int, Gs...>)
{
    (oct_backsub_group<NG, NG - 1 - Gs>(a, xq, gs, h), ...);
}

// LDS record of a system (floats): q window [0, QW), QW = 4 NG + 8 NCP - 1 (entry row + col, zeros from 2 n - 1 on) | p window
// p[|d|] at PO + d, d in [-7, 8 NCP), PO = QW + 7 (slots below the diagonal are not built, but a block's views reach back 3) |
// rhs at RO = PO + 8 NCP, 4 NG entries.  REC odd.
#ifndef TQ_OCT_OCC
#define TQ_OCT_OCC 2   // waves per SIMD of the octet kernel (A/B: 1 = 512 registers, no spills, half the residency)
#endif
template <int NG, int NMIN>
__global__ __launch_bounds__(256, TQ_OCT_OCC) void thsolve_octn_kernel(const float* __restrict__ p, int ldp, const float* __restrict__ q, int ldq,
                                                              const float* __restrict__ r, int ldr, const float* __restrict__ sub,
                                                              const float* add, long F, int n, float* g)   // (g may be add: dsa_mcep_newton_update in place)
{
    using O = Oct<NG>;
    constexpr int NCP = O::NCP;
    constexpr int CN = 4 * NG - 1;              // the right-hand side's column = the largest order
    constexpr int QW = 4 * NG + 8 * NCP - 1;
    constexpr int PO = QW + 7;
    constexpr int RO = PO + 8 * NCP;
#ifdef TQ_OCT_REC_ODD   // (A/B: rounds 3-4)
    constexpr int REC = (RO + 4 * NG) | 1;
#else
    // record stride = 8 (mod 32): the eight lanes of a system read eight consecutive banks (column offset 4 h + gs), and the four
    // systems of a 32-lane half then cover the 32 banks exactly once (an odd stride overlapped the systems' bank ranges: PMC
    // lds_conflict_frac 0.67)
    constexpr int REC = ((RO + 4 * NG - 8 + 31) / 32) * 32 + 8;
#endif
    constexpr int CPN = (NG - 1) >> 1, HN = (NG - 1) & 1;   // where column CN lives
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane0 = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* wl = lds + wave * 8 * REC;
    const long ntiles = (F + 7) / 8;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        __builtin_amdgcn_wave_barrier();
        // everything derived from the lane index is derived again per tile, from an opaque copy: hoisted out of the tile loop these
        // values live across the elimination, go to scratch there, and every reload in the staging code below is a
        // s_waitcnt vmcnt(0) that also waits for the staged loads issued so far
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int sy = lane >> 3, h = (lane >> 2) & 1, gs = lane & 3;
        const float subv = (sub && lane < n) ? sub[lane] : 0.f;
#ifdef TQ_STAMPS
        const long long ts0 = __builtin_readcyclecounter();
#endif
        const long fbase = tile * 8;
        const int nvalid = (int)((F - fbase) < 8 ? (F - fbase) : 8);
        for (int e = lane; e < 8 * REC; e += 64) wl[e] = 0.f;
        float q0[8], q1[8], p0[8], r0[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const long f = fbase + (s < nvalid ? s : 0);
            q0[s] = q[f * (long)ldq + (lane < 2 * n - 1 ? lane : 0)];
            q1[s] = q[f * (long)ldq + (lane + 64 < 2 * n - 1 ? lane + 64 : 0)];
            p0[s] = p[f * (long)ldp + (lane < n ? lane : 0)];
            r0[s] = r[f * (long)ldr + (lane < n ? lane : 0)];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const bool ok = s < nvalid;   // uniform
            float* rec = wl + s * REC;
            if (lane < 2 * n - 1) rec[lane] = ok ? q0[s] : 0.f;
            if (lane + 64 < 2 * n - 1) rec[lane + 64] = ok ? q1[s] : 0.f;
            if (lane < n) {
                const float pv = ok ? p0[s] : (lane == 0 ? 1.f : 0.f);
                rec[PO + lane] = pv;
                if (lane >= 1 && lane <= 7) rec[PO - lane] = pv;
                rec[RO + lane] = ok ? r0[s] - subv : 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef TQ_STAMPS
        const long long ts1 = __builtin_readcyclecounter();
#endif
        const int view = sy * REC + 4 * h + gs;         // this lane's views: column offset 4 h + gs folded in
        const float* rs = wl + sy * REC + RO;
        f32x4 a[O::N];
#pragma unroll
        for (int rg = 0; rg < NG; ++rg) {
            const float* qs = wl + view;
            const float* pw = qs + PO;
#pragma unroll
            for (int cp = rg >> 1; cp < NCP; ++cp) {
                const bool below = 2 * cp < rg;                             // half 0 of this slot lies below the diagonal
                const bool cin = 8 * cp + 7 < NMIN || 8 * cp + 4 * h + gs < n;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 4 * rg + i;
                    float v = pw[4 * (2 * cp - rg) - i] + qs[4 * (rg + 2 * cp) + i];
                    if (8 * cp + 7 >= NMIN) v = cin ? v : 0.f;
                    if (below) v = h == 1 ? v : 0.f;
                    if (2 * cp == rg || 2 * cp + 1 == rg) {                 // the diagonal block: identity rows between the order and CN
                        if (row >= NMIN && row < CN) v = (h == (rg & 1) && gs == i && row >= n) ? 1.f : v;
                    }
                    if (cp == CPN) v = (h == HN && gs == 3) ? rs[row] : v;
                    a[O::at(rg, cp)][i] = v;
                }
            }
        }
        bool bad = false;
#ifdef TQ_STAMPS
        asm volatile("" ::: "memory");
        const long long ts2 = __builtin_readcyclecounter();
#endif
        oct_elim_all<NG>(a, gs, bad, std::make_integer_sequence<int, CN>{});
#ifdef TQ_STAMPS
        const long long ts3 = __builtin_readcyclecounter();
#endif
        float xq[NCP];
#pragma unroll
        for (int c = 0; c < NCP; ++c) xq[c] = (c == CPN && h == HN && gs == 3) ? -1.f : 0.f;
        oct_backsub_all<NG>(a, xq, gs, h, std::make_integer_sequence<int, NG>{});
        const long f = fbase + sy;
        if (f < F) {
#pragma unroll
            for (int c = 0; c < NCP; ++c) {
                const int col = 8 * c + 4 * h + gs;
                if (col < n && !bad) g[f * (long)n + col] = add ? add[f * (long)n + col] + xq[c] : xq[c];
            }
        }
#ifdef TQ_STAMPS
        {
            const long long ts4 = __builtin_readcyclecounter();
            if (f < F && (lane & 7) == 0) {
                g[f * (long)n + 0] = (float)(ts1 - ts0);
                g[f * (long)n + 4] = (float)(ts2 - ts1);
                g[f * (long)n + 8] = (float)(ts3 - ts2);
                g[f * (long)n + 12] = (float)(ts4 - ts3);
            }
        }
#endif
        unsigned long long marked = __ballot(bad && (lane & 7) == 0 && f < F);
        while (marked) {   // uniform; normally empty: the whole wave re-solves the system with row pivoting (th_solve_reg.h)
            const int bl = __builtin_ctzll(marked);
            marked &= marked - 1;
            const int sb = bl >> 3;
            int nn = n, ln = lane;
            asm volatile("" : "+s"(nn), "+v"(ln));   // keeps this cold path's address arithmetic out of the tile loop
            const float* qs2 = wl + sb * REC;
            const float* ps2 = qs2 + PO;                    // ps2[d] = p[d]
            const float rhs = ln < nn ? qs2[RO + ln] : 0.f;
            int col;
            float sol;
            th_solve_reg<float, CN <= 48 ? 48 : 64>(ps2, qs2, rhs, nn, ln, col, sol);
            const long fs = fbase + sb;
            if (ln < nn) g[fs * (long)nn + col] = add ? add[fs * (long)nn + col] + sol : sol;
        }
    }
}

}  // namespace tq

}  // namespace dsa
