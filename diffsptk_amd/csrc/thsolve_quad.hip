// Batched Toeplitz-plus-Hankel solve for orders up to 55 on the float32 matrix instruction -- the order-24 scheme of
// csrc/mcep_mfma.hip (thsolve_quad24_kernel / the elimination inside the tuned mel-cepstral kernels) as a template over the size.
//
//   g:(F,n) = solve(T(p) + H(q), r - sub)  (+ add)          mcep.py:216-222, mgcep.py:226-229 (torch.linalg.solve there)
//
// What it replaces: th_solve_fwd_kernel (csrc/mgc.hip) -- one wave per system, one row per lane, row pivoting, the pivot row broadcast
// lane by lane through scalar registers -- took 174 us per 12 800 systems of order 50 (52 % of a Newton step of the mel-cepstral
// analysis at the 48 kHz set-ups, profiles/r04_48k_kernel_trace_v1.txt).
//
// Mapping (gfx950): the four lanes of a QUAD share a system, a wave solves 16 systems at once.  The augmented matrix
// [A | rhs] lives in registers as 4 x 4 blocks -- block (rg, cg), cg >= rg: register i = row 4 rg + i, lane gs = column 4 cg + gs
// -- upper triangle only (A is symmetric).  Step k: the pivot row scaled by -1 / a_kk is, by symmetry, the column of multipliers,
// already on the lanes that own those ROWS; one v_mfma_f32_4x4x1_16b_f32 (sixteen independent 4 x 4 outer products, one per quad)
// updates block (rg, cg) of all 16 systems: A = slot rg of the multipliers, B = slot cg of the unscaled pivot row.  No broadcast, no
// LDS, no pivot search.  The right-hand side rides as column n.  Back substitution: row sums per lane, two quad rotations, one
// multiply by the pivot's reciprocal.
// No pivoting: sound for the positive definite systems of the analysis; a system whose elimination meets a non-positive or
// non-finite pivot is solved again inside the launch by the whole wave with row pivoting (th_solve_reg.h).
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
                                                               const float* __restrict__ add, long F, int n, float* __restrict__ g)
{
    using B = Blk<NG>;
    constexpr int CN = 4 * NG - 1;              // the right-hand side's column = the largest order
    constexpr int QW = 8 * NG - 1;
    constexpr int PO = QW + 3;                  // p[0]
    constexpr int RO = QW + 4 * NG + 3;         // rhs[0]
    constexpr int REC = 16 * NG + 3;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* wl = lds + wave * 16 * REC;
    const int nq = lane >> 2, gs = lane & 3;
    const long ntiles = (F + 15) / 16;
    const float subv = (sub && lane < n) ? sub[lane] : 0.f;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        __builtin_amdgcn_wave_barrier();
        const long fbase = tile * 16;
        const int nvalid = (int)((F - fbase) < 16 ? (F - fbase) : 16);
        // stage the 16 records, one system per round of the wave (orders up to 55: one load for p and r, two for q); a missing system
        // is the identity with a zero right-hand side.  (Flat index loops with a division per element were a fifth of the kernel.)
        for (int e = lane; e < 16 * REC; e += 64) wl[e] = 0.f;
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const bool ok = s < nvalid;   // uniform
            const long f = fbase + (ok ? s : 0);
            float* rec = wl + s * REC;
            const float q0 = q[f * (long)ldq + (lane < 2 * n - 1 ? lane : 0)];
            const float q1 = q[f * (long)ldq + (lane + 64 < 2 * n - 1 ? lane + 64 : 0)];
            const float p0 = p[f * (long)ldp + (lane < n ? lane : 0)];
            const float r0 = r[f * (long)ldr + (lane < n ? lane : 0)];
            if (lane < 2 * n - 1) rec[lane] = ok ? q0 : 0.f;
            if (lane + 64 < 2 * n - 1) rec[lane + 64] = ok ? q1 : 0.f;
            if (lane < n) {
                const float pv = ok ? p0 : (lane == 0 ? 1.f : 0.f);
                rec[PO + lane] = pv;
                if (lane >= 1 && lane <= 3) rec[PO - lane] = pv;
                rec[RO + lane] = ok ? r0 - subv : 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();
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
        elim_all<NG>(a, gs, bad, std::make_integer_sequence<int, CN>{});
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

}  // namespace tq

template <int NG, int NMIN>
static int thsolve_quadn_launch(const void* p, int ldp, const void* q, int ldq, const void* r, int ldr, const void* sub, const void* add,
                                int64_t F, int n, void* g, hipStream_t st)
{
    constexpr int REC = 16 * NG + 3;
    const int lds_bytes = 4 * 16 * REC * (int)sizeof(float);
    static std::atomic<uint64_t> attr{0};
    if (lds_bytes > 48 * 1024 && !ensure_dynamic_lds((const void*)tq::thsolve_quadn_kernel<NG, NMIN>, lds_bytes, attr))
        return fail(DSA_ERR_LAUNCH, "thsolve_quad: cannot reserve LDS%s");
    long blocks = ((F + 15) / 16 + 3) / 4;
    if (blocks > 256) blocks = 256;   // one workgroup per CU (one wave per SIMD: the matrix takes up to 420 registers)
    hipLaunchKernelGGL((tq::thsolve_quadn_kernel<NG, NMIN>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, (const float*)p, ldp,
                       (const float*)q, ldq, (const float*)r, ldr, (const float*)sub, (const float*)add, (long)F, n, (float*)g);
    return check_launch("th_solve_quadn_fwd");
}

// float32, 2 <= n <= 55.  p:(F, n) row stride ldp, q:(F, 2n-1) stride ldq, r:(F, n) stride ldr; sub: NULL or (n), subtracted from
// every right-hand side; add: NULL or (F, n) contiguous, added to the solution; g:(F, n) contiguous.
int thsolve_quadn_fwd(const void* p, int ldp, const void* q, int ldq, const void* r, int ldr, const void* sub, const void* add, int64_t F,
                      int n, void* g, hipStream_t st)
{
    if (n < 2) return fail(DSA_ERR_UNSUPPORTED, "thsolve_quad: order below 2%s");
    if (n <= 27) return thsolve_quadn_launch<7, 2>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
    if (n <= 35) return thsolve_quadn_launch<9, 28>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
    if (n <= 43) return thsolve_quadn_launch<11, 36>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
    if (n <= 51) return thsolve_quadn_launch<13, 44>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
    if (n <= 55) return thsolve_quadn_launch<14, 52>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
    return fail(DSA_ERR_UNSUPPORTED, "thsolve_quad: order above 55%s");
}

}  // namespace dsa
