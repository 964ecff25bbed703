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

// LDS record of a system (floats): q window [0, QW) | mirrored p window pm[d + n - 1] = p[|d|], d in (-n, n) at [QW, 2 QW) | rhs [2 QW, 2 QW + RW)
template <int NG>
__global__ __launch_bounds__(256, 1) void thsolve_quadn_kernel(const float* __restrict__ p, int ldp, const float* __restrict__ q, int ldq,
                                                               const float* __restrict__ r, int ldr, const float* __restrict__ sub,
                                                               const float* __restrict__ add, long F, int n, float* __restrict__ g)
{
    using B = Blk<NG>;
    constexpr int NMAX = 4 * NG - 1;            // largest order: the right-hand side takes the last of the 4 NG columns
    constexpr int QW = 2 * NMAX + 1;            // >= 2 n - 1, odd: consecutive records start on different banks
    constexpr int REC = 2 * QW + 4 * NG;        // floats per system
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* wl = lds + wave * 16 * REC;
    const int nq = lane >> 2, gs = lane & 3;
    const long ntiles = (F + 15) / 16;
    for (long tile = (long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long)gridDim.x * 4) {
        __builtin_amdgcn_wave_barrier();
        const long fbase = tile * 16;
        const int nvalid = (int)((F - fbase) < 16 ? (F - fbase) : 16);
        // stage the 16 records; a missing system is the identity with a zero right-hand side
        for (int e = lane; e < 16 * REC; e += 64) wl[e] = 0.f;
        __builtin_amdgcn_wave_barrier();
        for (int idx = lane; idx < 16 * (2 * n - 1); idx += 64) {
            const int s = idx / (2 * n - 1), k = idx - s * (2 * n - 1);
            if (s < nvalid) wl[s * REC + k] = q[(fbase + s) * (long)ldq + k];
        }
        for (int idx = lane; idx < 16 * n; idx += 64) {
            const int s = idx / n, k = idx - s * n;
            const bool ok = s < nvalid;
            const float pv = ok ? p[(fbase + s) * (long)ldp + k] : (k == 0 ? 1.f : 0.f);
            wl[s * REC + QW + (n - 1) + k] = pv;
            wl[s * REC + QW + (n - 1) - k] = pv;
            float rv = ok ? r[(fbase + s) * (long)ldr + k] : 0.f;
            if (ok && sub) rv -= sub[k];
            wl[s * REC + 2 * QW + k] = rv;
        }
        __builtin_amdgcn_wave_barrier();
        const float* qs = wl + nq * REC;
        const float* pm = qs + QW + (n - 1);
        const float* rs = qs + 2 * QW;
        f32x4 a[B::N];
        // rows of T + H; the right-hand side in the last column CN = 4 NG - 1; rows / columns n .. CN - 1: the identity
        constexpr int CN = 4 * NG - 1;
#pragma unroll
        for (int rg = 0; rg < NG; ++rg) {
#pragma unroll
            for (int cg = rg; cg < NG; ++cg) {
                const int col = 4 * cg + gs;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 4 * rg + i;
                    float v;
                    if (row < n) {   // uniform
                        const float tv = pm[(col < n ? col : row) - row] + qs[col < n ? row + col : 0];
                        v = col < n ? tv : (col == CN ? rs[row] : 0.f);
                    } else {
                        v = (col == row && row < CN) ? 1.f : 0.f;
                    }
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
            const float* qs2 = wl + sy * REC;
            const float* ps2 = qs2 + QW + (n - 1);          // ps2[d] = p[d]
            const float rhs = lane < n ? qs2[2 * QW + lane] : 0.f;
            int col;
            float sol;
            th_solve_reg<float, NMAX <= 32 ? 32 : (NMAX <= 48 ? 48 : 64)>(ps2, qs2, rhs, n, lane, col, sol);
            const long fs = fbase + sy;
            if (lane < n) g[fs * (long)n + col] = add ? add[fs * (long)n + col] + sol : sol;
        }
    }
}

}  // namespace tq

template <int NG>
static int thsolve_quadn_launch(const void* p, int ldp, const void* q, int ldq, const void* r, int ldr, const void* sub, const void* add,
                                int64_t F, int n, void* g, hipStream_t st)
{
    constexpr int NMAX = 4 * NG - 1, REC = 2 * (2 * NMAX + 1) + 4 * NG;
    const int lds_bytes = 4 * 16 * REC * (int)sizeof(float);
    static std::atomic<uint64_t> attr{0};
    if (lds_bytes > 48 * 1024 && !ensure_dynamic_lds((const void*)tq::thsolve_quadn_kernel<NG>, lds_bytes, attr))
        return fail(DSA_ERR_LAUNCH, "thsolve_quad: cannot reserve LDS%s");
    long blocks = ((F + 15) / 16 + 3) / 4;
    if (blocks > 256) blocks = 256;   // one workgroup per CU (one wave per SIMD: the matrix takes up to 420 registers)
    hipLaunchKernelGGL((tq::thsolve_quadn_kernel<NG>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, (const float*)p, ldp,
                       (const float*)q, ldq, (const float*)r, ldr, (const float*)sub, (const float*)add, (long)F, n, (float*)g);
    return check_launch("th_solve_quadn_fwd");
}

// float32, 2 <= n <= 55.  p:(F, n) row stride ldp, q:(F, 2n-1) stride ldq, r:(F, n) stride ldr; sub: NULL or (n), subtracted from
// every right-hand side; add: NULL or (F, n) contiguous, added to the solution; g:(F, n) contiguous.
int thsolve_quadn_fwd(const void* p, int ldp, const void* q, int ldq, const void* r, int ldr, const void* sub, const void* add, int64_t F,
                      int n, void* g, hipStream_t st)
{
    if (n < 2) return fail(DSA_ERR_UNSUPPORTED, "thsolve_quad: order below 2%s");
    if (n <= 27) return thsolve_quadn_launch<7>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
    if (n <= 35) return thsolve_quadn_launch<9>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
    if (n <= 43) return thsolve_quadn_launch<11>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
    if (n <= 51) return thsolve_quadn_launch<13>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
    if (n <= 55) return thsolve_quadn_launch<14>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
    return fail(DSA_ERR_UNSUPPORTED, "thsolve_quad: order above 55%s");
}

}  // namespace dsa
