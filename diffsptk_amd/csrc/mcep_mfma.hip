// Tuned mel-cepstral analysis for gfx950: float32, fft_length 512, cep_order 24.
// (MelCepstralAnalysis._forward, diffsptk/modules/mcep.py:189-224, in the composed-matrix form
//  described in mcep.hip.)  tools/proto_mcep_mfma.py is a lane-level numpy model of this file.
//
// Mapping.  One wave64 owns 16 frames for the whole Newton iteration; a workgroup is 4 such
// waves sharing the operand images of D and E in LDS (one workgroup per CU, persistent over
// frame tiles).  Lane l = (n = l & 15: frame, g = l >> 4: lane group).
//   * the frames are the N (column) dimension of v_mfma_f32_16x16x4_f32, so products come out
//     TRANSPOSED:  d^T (256 x 16) = D^T (256 x 28) mc^T (28 x 16)   -> 16 tiles x 7 k-steps
//                  rt^T (48 x 16) = E^T (48 x 256) e^T (256 x 16)   ->  3 tiles x 64 k-steps
//     In the C/D layout lane (n, g) register r of tile mt holds bin mt*16 + 4g + r of frame n --
//     which is exactly a B operand (k-slot g, column n) if the k-steps of the second product are
//     enumerated as (mt, r).  The E^T operand image is laid out in that order, so e = exp(log X
//     - 2 d) feeds the second MFMA chain straight from the accumulator registers: no transpose,
//     no LDS round trip, log X stays in 64 VGPRs for all 10 iterations.
//   * bin 256 (Nyquist) and output rt[48] do not fit the 16-wide tiles; they are one extra
//     k-step / one VALU dot product instead of a whole padded tile each.
//   * the 25 x 25 system (Toeplitz + Hankel, SPD) is eliminated row-cyclically by the 4 lanes of
//     a frame: lane group g owns rows g, g+4, ...; the pivot row is broadcast by ds_bpermute;
//     everything is statically indexed registers.  The rows are assembled from two shifted
//     windows of rt kept in LDS (rt itself and a reflected copy).
// Bound: VALU + MFMA issue (fp32 MFMA rate = fp32 vector rate = 157.3 TFLOP/s); HBM traffic is
// 1028 B in + 100 B out per frame.
#include "common.h"

#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <utility>

namespace dsa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// exp(x) for |x| < 87 without control flow: x log2(e) = n + r with the product carried in two
// floats (Cody-Waite), v_exp_f32 on the reduced argument, v_ldexp_f32 for 2^n.  ~1 ulp.
__device__ __forceinline__ float exp_nobranch(float x)
{
    constexpr float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f;
    const float nn = __builtin_rintf(x * L2E_HI);
    float r = __builtin_fmaf(x, L2E_HI, -nn);
    r = __builtin_fmaf(x, L2E_LO, r);
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(r), (int)nn);
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- statically unrolled elimination steps (template recursion keeps every index a constant,
// so the 7 x 25 local rows stay in registers) ----
template <int k>
__device__ __forceinline__ void elim_step(float (&a)[mm::NR][mm::M1], float (&b)[mm::NR], int n, int g)
{
    using namespace mm;
    constexpr int gk = k & 3, mk = k >> 2;
    const int src = n + 16 * gk;
    float prow[M1];
#pragma unroll
    for (int j = k; j < M1; ++j) prow[j] = __shfl(a[mk][j], src, 64);
    const float pb = __shfl(b[mk], src, 64);
    const float inv = 1.f / prow[k];
#pragma unroll
    for (int m = mk; m < NR; ++m) {
        float fct = a[m][k] * inv;
        if (m == mk) fct = g > gk ? fct : 0.f;  // rows at or above the pivot stay
#pragma unroll
        for (int j = k + 1; j < M1; ++j) a[m][j] -= fct * prow[j];
        b[m] -= fct * pb;
    }
}
template <int k>
__device__ __forceinline__ void backsub_step(float (&a)[mm::NR][mm::M1], float (&b)[mm::NR],
                                             float (&xs)[mm::M1], int n)
{
    constexpr int gk = k & 3, mk = k >> 2;
    const float xk = __shfl(b[mk] / a[mk][k], n + 16 * gk, 64);
    xs[k] = xk;
#pragma unroll
    for (int m = 0; m <= mk; ++m) b[m] -= a[m][k] * xk;
}
template <int... Ks>
__device__ __forceinline__ void elim_all(float (&a)[mm::NR][mm::M1], float (&b)[mm::NR], int n, int g,
                                         std::integer_sequence<int, Ks...>)
{
    (elim_step<Ks>(a, b, n, g), ...);
}
template <int... Ks>
__device__ __forceinline__ void backsub_all(float (&a)[mm::NR][mm::M1], float (&b)[mm::NR],
                                            float (&xs)[mm::M1], int n, std::integer_sequence<int, Ks...>)
{
    (backsub_step<mm::M1 - 1 - Ks>(a, b, xs, n), ...);
}

__global__ __launch_bounds__(256, 1) void mcep_mfma_fwd_kernel(
    const float* __restrict__ X, long F, int n_iter, const float* __restrict__ G,
    const float* __restrict__ D, const float* __restrict__ E, const float* __restrict__ av,
    float* __restrict__ mc_out, float* __restrict__ hist, long ntiles)
{
    using namespace mm;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;

    // ---------------- operand images: built once per workgroup ----------------
    for (int idx = tid; idx < 16 * 2 * 64 * 4; idx += 256) {
        int q = idx & 3, l = (idx >> 2) & 63, half = (idx >> 8) & 1, mt = idx >> 9;
        int k = 4 * (half * 4 + q) + (l >> 4);
        lds[DT_OFF + idx] = k < M1 ? D[k * K + mt * 16 + (l & 15)] : 0.f;
    }
    for (int idx = tid; idx < 3 * 16 * 64 * 4; idx += 256) {
        int r = idx & 3, l = (idx >> 2) & 63, mt = (idx >> 8) & 15, it = idx >> 12;
        lds[ET_OFF + idx] = E[(mt * 16 + (l >> 4) * 4 + r) * M2 + it * 16 + (l & 15)];
    }
    {
        int r = tid & 3, gg = (tid >> 2) & 3, mt = tid >> 4;  // 256 entries
        lds[E48_OFF + tid] = E[(mt * 16 + gg * 4 + r) * M2 + 48];
    }
    if (tid < M2) lds[E256_OFF + tid] = E[H * M2 + tid];
    if (tid < 28) {
        lds[D256_OFF + tid] = tid < M1 ? D[tid * K + H] : 0.f;
        lds[AV_OFF + tid] = tid < M1 ? av[tid] : 0.f;
    }
    __syncthreads();

    float* rt_lds = lds + WAVE_OFF + wave * WAVE_FLOATS + n * RS;  // this lane's frame window
    float* rr_lds = rt_lds + 16 * RS;
    const f32x4* Dt4 = reinterpret_cast<const f32x4*>(lds + DT_OFF);
    const f32x4* Et4 = reinterpret_cast<const f32x4*>(lds + ET_OFF);
    const f32x4* E484 = reinterpret_cast<const f32x4*>(lds + E48_OFF);

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long f_raw = tile * 64 + wave * 16 + n;
        const bool f_ok = f_raw < F;
        const long f = f_ok ? f_raw : F - 1;  // tail lanes recompute the last frame, never store
        const float* xf = X + f * K;

        // ---------------- log spectrum into the accumulator layout ----------------
        f32x4 logx[16];
#pragma unroll
        for (int mt = 0; mt < 16; ++mt) {
            const float* p = xf + mt * 16 + 4 * g;
            logx[mt] = f32x4{logf(p[0]), logf(p[1]), logf(p[2]), logf(p[3])};  // mcep.py:203
        }
        const float logx256 = logf(xf[H]);

        // ---------------- mc0^T = G^T logx^T  (mcep.py:204-207) ----------------
        f32x4 accG[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int out = it * 16 + n;
            const bool ov = out < M1;
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
                const float* gp = G + (mt * 16 + 4 * g) * M1 + out;
#pragma unroll
                for (int r = 0; r < 4; ++r) accG[it] = mfma4(ov ? gp[r * M1] : 0.f, logx[mt][r], accG[it]);
            }
            accG[it] = mfma4((ov && g == 0) ? G[H * M1 + out] : 0.f, g == 0 ? logx256 : 0.f, accG[it]);
        }
        // accG[it][r] = mc0[coef it*16 + 4g + r] of frame n; re-distribute to mcB[ks] = mc[4ks + g]
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) rt_lds[it * 16 + 4 * g + r] = accG[it][r];
        __syncthreads();
        float mcB[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) mcB[ks] = (4 * ks + g < M1) ? rt_lds[4 * ks + g] : 0.f;
        if (hist && f_ok)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                if (4 * ks + g < M1) hist[f * M1 + 4 * ks + g] = mcB[ks];

        for (int iter = 0; iter < n_iter; ++iter) {
            // ------------- d^T = D^T mc^T ; e = exp(log X - 2 d)  (mcep.py:210-212) -------------
            f32x4 e[16];
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
                f32x4 a0 = Dt4[(mt * 2 + 0) * 64 + lane];
                f32x4 a1 = Dt4[(mt * 2 + 1) * 64 + lane];
                f32x4 acc = {0, 0, 0, 0};
                acc = mfma4(a0[0], mcB[0], acc);
                acc = mfma4(a0[1], mcB[1], acc);
                acc = mfma4(a0[2], mcB[2], acc);
                acc = mfma4(a0[3], mcB[3], acc);
                acc = mfma4(a1[0], mcB[4], acc);
                acc = mfma4(a1[1], mcB[5], acc);
                acc = mfma4(a1[2], mcB[6], acc);
                e[mt] = acc;
            }
#pragma unroll
            for (int mt = 0; mt < 16; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) e[mt][r] = expf(logx[mt][r] - 2.f * e[mt][r]);
            float d256 = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) d256 += mcB[ks] * lds[D256_OFF + 4 * ks + g];
            d256 += __shfl_xor(d256, 16, 64);
            d256 += __shfl_xor(d256, 32, 64);
            const float e256 = expf(logx256 - 2.f * d256);

            // ------------- rt^T = E^T e^T  (mcep.py:214-215) -------------
            f32x4 accB[3];
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                f32x4 acc = {0, 0, 0, 0};
#pragma unroll
                for (int mt = 0; mt < 16; ++mt) {
                    f32x4 a = Et4[(it * 16 + mt) * 64 + lane];
                    acc = mfma4(a[0], e[mt][0], acc);
                    acc = mfma4(a[1], e[mt][1], acc);
                    acc = mfma4(a[2], e[mt][2], acc);
                    acc = mfma4(a[3], e[mt][3], acc);
                }
                acc = mfma4(g == 0 ? lds[E256_OFF + it * 16 + n] : 0.f, g == 0 ? e256 : 0.f, acc);
                accB[it] = acc;
            }
            float rt48 = 0.f;
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
                f32x4 c48 = E484[mt * 4 + g];
                rt48 += e[mt][0] * c48[0] + e[mt][1] * c48[1] + e[mt][2] * c48[2] + e[mt][3] * c48[3];
            }
            rt48 += __shfl_xor(rt48, 16, 64);
            rt48 += __shfl_xor(rt48, 32, 64);
            rt48 += e256 * lds[E256_OFF + 48];

            // ------------- rt and its reflection into this frame's LDS windows -------------
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 3; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = it * 16 + 4 * g + r;
                    const float v = accB[it][r];
                    rt_lds[idx] = v;
                    if (idx <= 27) {  // rr[27 + d] = r[|d|]
                        rr_lds[27 + idx] = v;
                        rr_lds[27 - idx] = v;
                    }
                }
            if (g == 0) rt_lds[48] = rt48;
            __syncthreads();

            // ------------- local rows i = g + 4m of R + Q and of r - alpha_vector -------------
            float a[NR][M1], b[NR];
            {
                // S[v] = rt[g + v], v = 0..48 ; Rw[t + 24] = r[|g + t|], t = -24..24
                float S[49], Rw[49];
#pragma unroll
                for (int v = 0; v < 49; ++v) {
                    S[v] = rt_lds[g + v];
                    Rw[v] = rr_lds[27 + g + v - 24];
                }
#pragma unroll
                for (int m = 0; m < NR; ++m) {
                    const bool valid = g + 4 * m < M1;
#pragma unroll
                    for (int j = 0; j < M1; ++j) {
                        // mcep.py:219-221: R[i][j] = r[|i-j|], Q[i][j] = rt[i+j]
                        float v = S[4 * m + j] + Rw[4 * m - j + 24];
                        a[m][j] = valid ? v : 0.f;
                    }
                    // mcep.py:216-217; S[4m] = rt[i]
                    b[m] = valid ? S[4 * m] - lds[AV_OFF + g + 4 * m] : 0.f;
                }
            }
            // ------------- forward elimination (no pivoting: the system is SPD), then back
            // substitution: x_k from its owner lane, broadcast, column update -------------
            float xs[M1];
            elim_all(a, b, n, g, std::make_integer_sequence<int, M1>{});
            backsub_all(a, b, xs, n, std::make_integer_sequence<int, M1>{});
            // ------------- mc += solution (mcep.py:222), back in the B-operand layout -------------
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float s0 = xs[4 * ks];
                float s1 = 4 * ks + 1 < M1 ? xs[4 * ks + 1 < M1 ? 4 * ks + 1 : 0] : 0.f;
                float s2 = 4 * ks + 2 < M1 ? xs[4 * ks + 2 < M1 ? 4 * ks + 2 : 0] : 0.f;
                float s3 = 4 * ks + 3 < M1 ? xs[4 * ks + 3 < M1 ? 4 * ks + 3 : 0] : 0.f;
                mcB[ks] += g == 0 ? s0 : (g == 1 ? s1 : (g == 2 ? s2 : s3));
            }
            if (hist && f_ok)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    if (4 * ks + g < M1) hist[((long)(iter + 1) * F + f) * M1 + 4 * ks + g] = mcB[ks];
        }
        if (f_ok)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                if (4 * ks + g < M1) mc_out[f * M1 + 4 * ks + g] = mcB[ks];
    }
}

// =====================================================================================
// v2: 8 waves per workgroup (two per SIMD), no workgroup barriers inside the iteration, and a
// SYMMETRIC row-cyclic elimination: only columns j >= 4m of local row m are kept (91 registers
// instead of 175), multipliers come from the broadcast pivot row (a[i][k] = a[k][i]), and the
// solution updates mc as soon as each x_k is known.  With <= 256 registers per wave the second
// wave of a SIMD runs its MFMA chains while the first is in the (latency-bound) solve.
// =====================================================================================
#ifdef DSA_MCEP_TIMING
__device__ unsigned long long g_mcep_stamps[64];
#define DSA_STAMP(i)                                                                   \
    do {                                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0 && tile == 0 && iter == 1)             \
            g_mcep_stamps[i] = __builtin_readcyclecounter();                            \
    } while (0)
#else
#define DSA_STAMP(i)
#endif

namespace mm2 {
using namespace mm;
constexpr int lds_floats(int waves) { return WAVE_OFF + waves * WAVE_FLOATS; }
}  // namespace mm2

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

// value of lane Q of this lane's quad (DPP quad_perm broadcast: a plain VALU move, no LDS)
template <int Q>
__device__ __forceinline__ float quad_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), Q * 0x55, 0xf, 0xf, true));
}

// local symmetric row block: row m keeps columns 4m .. 24
struct SymRows {
    float r0[25], r1[21], r2[17], r3[13], r4[9], r5[5], r6[1];
};
template <int m>
__device__ __forceinline__ float* sym_row(SymRows& a)
{
    if constexpr (m == 0) return a.r0;
    else if constexpr (m == 1) return a.r1;
    else if constexpr (m == 2) return a.r2;
    else if constexpr (m == 3) return a.r3;
    else if constexpr (m == 4) return a.r4;
    else if constexpr (m == 5) return a.r5;
    else return a.r6;
}

template <int k, int m>
__device__ __forceinline__ void sym_update_row(SymRows& a, float (&b)[mm::NR], const float (&prow)[mm::M1],
                                               float pb, float inv, const GroupMask& gq)
{
    using namespace mm;
    constexpr int gk = k & 3, mk = k >> 2;
    if constexpr (m >= mk && m < NR) {
        // multiplier a[i][k] / a[k][k] with a[i][k] = a[k][i] = prow[i], i = 4m + g
        constexpr int i0 = 4 * m;
        const float c0 = (i0 + 0 > k && i0 + 0 < M1) ? prow[i0 + 0 < M1 ? i0 + 0 : 0] : 0.f;
        const float c1 = (i0 + 1 > k && i0 + 1 < M1) ? prow[i0 + 1 < M1 ? i0 + 1 : 0] : 0.f;
        const float c2 = (i0 + 2 > k && i0 + 2 < M1) ? prow[i0 + 2 < M1 ? i0 + 2 : 0] : 0.f;
        const float c3 = (i0 + 3 > k && i0 + 3 < M1) ? prow[i0 + 3 < M1 ? i0 + 3 : 0] : 0.f;
        (void)gk;
        const float fct = sel4(gq, c0, c1, c2, c3) * inv;
        float* row = sym_row<m>(a);
        constexpr int j0 = (4 * m > k + 1) ? 4 * m : k + 1;
#pragma unroll
        for (int j = j0; j < M1; ++j) row[j - 4 * m] -= fct * prow[j];
        b[m] -= fct * pb;
        sym_update_row<k, m + 1>(a, b, prow, pb, inv, gq);
    }
}

template <int k>
__device__ __forceinline__ void sym_elim_step(SymRows& a, float (&b)[mm::NR], int n, const GroupMask& gq)
{
    using namespace mm;
    constexpr int gk = k & 3, mk = k >> 2;
    (void)n;
    float prow[M1];
    float* prw = sym_row<mk>(a);
#pragma unroll
    for (int j = k; j < M1; ++j) prow[j] = quad_bcast<gk>(prw[j - 4 * mk]);
    const float pb = quad_bcast<gk>(b[mk]);
    const float inv = rcp_nr(prow[k]);
    sym_update_row<k, mk>(a, b, prow, pb, inv, gq);
}

template <int k, int m>
__device__ __forceinline__ void sym_backsub_rows(SymRows& a, float (&b)[mm::NR], float xk)
{
    constexpr int mk = k >> 2;
    if constexpr (m <= mk) {
        b[m] -= sym_row<m>(a)[k - 4 * m] * xk;
        sym_backsub_rows<k, m + 1>(a, b, xk);
    }
}

template <int k>
__device__ __forceinline__ void sym_backsub_step(SymRows& a, float (&b)[mm::NR], float (&mcB)[mm::KS], int n,
                                                 const GroupMask& gq)
{
    constexpr int gk = k & 3, mk = k >> 2;
    (void)n;
    const float xk = quad_bcast<gk>(b[mk] * rcp_nr(sym_row<mk>(a)[k - 4 * mk]));
    mcB[mk] = __uint_as_float(__float_as_uint(mcB[mk]) | (__float_as_uint(xk) & gq.m[gk]));  // x[4 mk + g']
    sym_backsub_rows<k, 0>(a, b, xk);
}

template <int... Ks>
__device__ __forceinline__ void sym_elim_all(SymRows& a, float (&b)[mm::NR], int n, const GroupMask& gq,
                                             std::integer_sequence<int, Ks...>)
{
    (sym_elim_step<Ks>(a, b, n, gq), ...);
}
template <int... Ks>
__device__ __forceinline__ void sym_backsub_all(SymRows& a, float (&b)[mm::NR], float (&mcB)[mm::KS], int n,
                                                const GroupMask& gq, std::integer_sequence<int, Ks...>)
{
    (sym_backsub_step<mm::M1 - 1 - Ks>(a, b, mcB, n, gq), ...);
}

template <int m>
__device__ __forceinline__ void sym_build_rows(SymRows& a, float (&b)[mm::NR], const float* rt_g, const float* rr_g,
                                               const float* avs, int g)
{
    using namespace mm;
    if constexpr (m < NR) {
        // row i = g + 4m, columns j = 4m..24:  R[i][j] = r[|i-j|] = rr[27 + i - j],  Q[i][j] = rt[i + j]
        // (mcep.py:219-221); rt_g = rt + g, rr_g = rr + 27 + g are this lane's shifted windows
        // rows 4m .. 4m+3 all exist unless this is the last block (only g = 0 has row 24 when M1 = 25)
        constexpr bool all_valid = 4 * m + 3 < M1;
        const unsigned vmask = (all_valid || g + 4 * m < M1) ? 0xffffffffu : 0u;
        float* row = sym_row<m>(a);
#pragma unroll
        for (int j = 4 * m; j < M1; ++j) {
            const float v = rt_g[4 * m + j] + rr_g[4 * m - j];
            row[j - 4 * m] = all_valid ? v : keep_if(vmask, v);
        }
        const float bv = rt_g[4 * m] - avs[g + 4 * m];  // mcep.py:216-217
        b[m] = all_valid ? bv : keep_if(vmask, bv);
        sym_build_rows<m + 1>(a, b, rt_g, rr_g, avs, g);
    }
}

// WAVES = 4: one wave per SIMD with the whole 512-entry register file (no spills);
// WAVES = 8: two waves per SIMD at <= 256 registers (log X spills to scratch).
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
    const float ninv = -rcp_nr(quad_bcast<(k & 3)>(COL_AT(a, k, k >> 2)));
    float pneg[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = k >> 2; c < 7; ++c) pneg[c - (k >> 2)] = COL_AT(a, k, c) * ninv;
    // VALU write -> DPP read of pneg needs 2 wait states, which hipcc cannot see inside inline asm; the
    // dummy in/out operands pin the nop after the multiplies
    asm volatile("s_nop 1"
                 : "+v"(pneg[0]), "+v"(pneg[1]), "+v"(pneg[2]), "+v"(pneg[3]), "+v"(pneg[4]), "+v"(pneg[5]), "+v"(pneg[6]));
    col_update_rows<k, k + 1>(a, pneg);
}
template <int... Ks>
__device__ __forceinline__ void col_elim_all(float (&a)[colm::TOTAL], std::integer_sequence<int, Ks...>)
{
    (col_elim_step<Ks>(a), ...);
}

// back substitution of right-hand side RHS (1 or 2): xq[c] accumulates x[gs + 4c]; the slot of the
// right-hand-side column is preset to -1 on its owner lane, so  sum_j U[k][j] x_j - b_k  is one dot product
template <int k>
__device__ __forceinline__ float col_backsub_step(const float (&a)[colm::TOTAL], float (&xq)[mm::KS],
                                                  const GroupMask& gq)
{
    float sl = 0.f;
#pragma unroll
    for (int c = k >> 2; c < 7; ++c) sl = __builtin_fmaf(COL_AT(a, k, c), xq[c], sl);
    sl += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sl), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
    sl += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(sl), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
    const float xk = -sl * rcp_nr(quad_bcast<(k & 3)>(COL_AT(a, k, k >> 2)));
    xq[k >> 2] = __uint_as_float(__float_as_uint(xq[k >> 2]) | (__float_as_uint(xk) & gq.m[k & 3]));
    return xk;
}
template <int... Ks>
__device__ __forceinline__ void col_backsub_all(const float (&a)[colm::TOTAL], float (&xq)[mm::KS],
                                                const GroupMask& gq, std::integer_sequence<int, Ks...>)
{
    ((void)col_backsub_step<mm::M1 - 1 - Ks>(a, xq, gq), ...);
}
template <int... Ks>
__device__ __forceinline__ void col_backsub_full(const float (&a)[colm::TOTAL], float (&xq)[mm::KS],
                                                 float (&xv)[mm::M1], const GroupMask& gq,
                                                 std::integer_sequence<int, Ks...>)
{
    ((xv[mm::M1 - 1 - Ks] = col_backsub_step<mm::M1 - 1 - Ks>(a, xq, gq)), ...);
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void mcep_mfma_fwd_kernel_v2(
    const float* __restrict__ X, long F, int n_iter, const float* __restrict__ G,
    const float* __restrict__ D, const float* __restrict__ E, const float* __restrict__ av,
    float* __restrict__ mc_out, float* __restrict__ hist, long ntiles16, unsigned int* __restrict__ queue)
{
    using namespace mm2;
    // exp(ln X - 2 d) = exp2(log2 X - 2 log2(e) d): log2 / exp2 are single gfx950 instructions, so
    // the D operand image is pre-scaled by -2 log2(e) and log2 X is what stays in registers;
    // mc0 = ln X . G becomes log2 X . (ln 2 G).
    constexpr float kNeg2Log2e = -2.885390081777926815f, kLn2 = 0.693147180559945309f;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;

    // ---------------- operand images: built once per workgroup ----------------
    for (int idx = tid; idx < 16 * 2 * 64 * 4; idx += WAVES * 64) {
        int q = idx & 3, l = (idx >> 2) & 63, half = (idx >> 8) & 1, mt = idx >> 9;
        int k = 4 * (half * 4 + q) + (l >> 4);
        lds[DT_OFF + idx] = k < M1 ? kNeg2Log2e * D[k * K + mt * 16 + (l & 15)] : 0.f;
    }
    for (int idx = tid; idx < 3 * 16 * 64 * 4; idx += WAVES * 64) {
        int r = idx & 3, l = (idx >> 2) & 63, mt = (idx >> 8) & 15, it = idx >> 12;
        lds[ET_OFF + idx] = E[(mt * 16 + (l >> 4) * 4 + r) * M2 + it * 16 + (l & 15)];
    }
    {
        const int t2 = tid & 255;
        int r = t2 & 3, gg = (t2 >> 2) & 3, mt = t2 >> 4;
        lds[E48_OFF + t2] = E[(mt * 16 + gg * 4 + r) * M2 + 48];
    }
    if (tid < M2) lds[E256_OFF + tid] = E[H * M2 + tid];
    if (tid < 28) {
        lds[D256_OFF + tid] = tid < M1 ? kNeg2Log2e * D[tid * K + H] : 0.f;
        lds[AV_OFF + tid] = tid < M1 ? av[tid] : 0.f;
    }
    __syncthreads();  // the only workgroup barrier: from here on every wave runs on its own

    float* rt_lds = lds + WAVE_OFF + wave * WAVE_FLOATS + n * RS;  // this lane's frame windows
    float* rr_lds = rt_lds + 16 * RS;
    const f32x4* Dt4 = reinterpret_cast<const f32x4*>(lds + DT_OFF);
    const f32x4* Et4 = reinterpret_cast<const f32x4*>(lds + ET_OFF);
    const f32x4* E484 = reinterpret_cast<const f32x4*>(lds + E48_OFF);
    const long wave_id = (long)blockIdx.x * WAVES + wave;
    const long wave_stride = (long)gridDim.x * WAVES;
    // solve layout: the 4 lanes of a QUAD share a frame (nq = lane >> 2, gs = lane & 3), so the
    // pivot-row broadcasts of the elimination are DPP quad_perm moves instead of ds_bpermute
    const int nq = lane >> 2, gs = lane & 3;
    const GroupMask gq = make_group_mask(gs);
#ifdef DSA_MCEP_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0) g_mcep_stamps[8] = __builtin_readcyclecounter();
#endif
    float* rt_q = lds + WAVE_OFF + wave * WAVE_FLOATS + nq * RS;
    float* rr_q = rt_q + 16 * RS;

    // dynamic tile queue: the first round is static (tile = wave slot), later tiles are drawn from a
    // device counter, so the tail of a launch is one tile long instead of a whole static round
    for (long tile = wave_id; tile < ntiles16;) {
        const long f_raw = tile * 16 + n;
        const bool f_ok = f_raw < F;
        const long f = f_ok ? f_raw : F - 1;  // tail lanes recompute the last frame, never store
        const float* xf = X + f * K;

        f32x4 logx[16];
#pragma unroll
        for (int mt = 0; mt < 16; ++mt) {
            const float* p = xf + mt * 16 + 4 * g;
            logx[mt] = f32x4{__log2f(p[0]), __log2f(p[1]), __log2f(p[2]), __log2f(p[3])};  // mcep.py:203 (base 2)
        }
        const float logx256 = __log2f(xf[H]);

        // ---------------- mc0^T = G^T logx^T  (mcep.py:204-207) ----------------
        float mcB[KS];
        {
            f32x4 accG[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int out = it * 16 + n;
                const bool ov = out < M1;
#pragma unroll
                for (int mt = 0; mt < 16; ++mt) {
                    const float* gp = G + (mt * 16 + 4 * g) * M1 + out;
#pragma unroll
                    for (int r = 0; r < 4; ++r) accG[it] = mfma4(ov ? kLn2 * gp[r * M1] : 0.f, logx[mt][r], accG[it]);
                }
                accG[it] = mfma4((ov && g == 0) ? kLn2 * G[H * M1 + out] : 0.f, g == 0 ? logx256 : 0.f, accG[it]);
            }
            // accG[it][r] = mc0[coef it*16 + 4g + r]; re-distribute through this wave's LDS window
            // (wave-private data: LDS executes a wave's accesses in program order)
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) rt_lds[it * 16 + 4 * g + r] = accG[it][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) mcB[ks] = (4 * ks + g < M1) ? rt_lds[4 * ks + g] : 0.f;
            __builtin_amdgcn_wave_barrier();
        }
        if (hist && f_ok)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                if (4 * ks + g < M1) hist[f * M1 + 4 * ks + g] = mcB[ks];

        for (int iter = 0; iter < n_iter; ++iter) {
            // ------------- per 16-bin tile: d^T = D^T mc^T, e = exp(log X - 2 d) (mcep.py:210-212),
            // and straight on into rt^T += E^T e^T (mcep.py:214-215): e never leaves 4 registers ----
            DSA_STAMP(0);
            f32x4 accB[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
            float rt48 = 0.f;
            // Software pipeline, pinned with sched_group_barrier: per 16-bin tile mt the matrix pipe
            // gets 7 MFMAs of the D-chain of tile mt+1 (two accumulators) and 12 of the E-chain of
            // tile mt (three accumulators, round-robin: no dependent-MFMA stall); the ~36 VALU
            // instructions of exp(tile mt) and the operand ds_reads are slotted into the MFMA
            // issue gaps (one wave cannot overlap MFMA and VALU unless they alternate in program order).
            f32x4 pa = {0, 0, 0, 0}, qa = {0, 0, 0, 0};
            {
                const f32x4 a0 = Dt4[lane], a1 = Dt4[64 + lane];
                pa = mfma4(a0[0], mcB[0], pa);
                qa = mfma4(a1[0], mcB[4], qa);
                pa = mfma4(a0[1], mcB[1], pa);
                qa = mfma4(a1[1], mcB[5], qa);
                pa = mfma4(a0[2], mcB[2], pa);
                qa = mfma4(a1[2], mcB[6], qa);
                pa = mfma4(a0[3], mcB[3], pa);
            }
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
                const int mn = mt + 1 < 16 ? mt + 1 : 15;
                const f32x4 a0 = Dt4[(mn * 2 + 0) * 64 + lane];
                const f32x4 a1 = Dt4[(mn * 2 + 1) * 64 + lane];
                const f32x4 ea0 = Et4[(0 * 16 + mt) * 64 + lane];
                const f32x4 ea1 = Et4[(1 * 16 + mt) * 64 + lane];
                const f32x4 ea2 = Et4[(2 * 16 + mt) * 64 + lane];
                const f32x4 c48 = E484[mt * 4 + g];
                const f32x4 acc = pa + qa;
                f32x4 pn = {0, 0, 0, 0}, qn = {0, 0, 0, 0};
                f32x4 e;
                // D-chain of the next tile: issued while exp of this tile runs on the VALU
#if defined(DSA_MCEP_ABL) && DSA_MCEP_ABL == 2
                if (mt + 1 < 16) { pn = a0 * mcB[0]; qn = a1 * mcB[1]; } else
#endif
                if (mt + 1 < 16) {
                    pn = mfma4(a0[0], mcB[0], pn);
                    qn = mfma4(a1[0], mcB[4], qn);
                    pn = mfma4(a0[1], mcB[1], pn);
                    qn = mfma4(a1[1], mcB[5], qn);
                    pn = mfma4(a0[2], mcB[2], pn);
                    qn = mfma4(a1[2], mcB[6], qn);
                    pn = mfma4(a0[3], mcB[3], pn);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#if defined(DSA_MCEP_ABL) && DSA_MCEP_ABL == 1
                    e[r] = logx[mt][r] + acc[r];
#else
                    e[r] = __builtin_amdgcn_exp2f(logx[mt][r] + acc[r]);  // mcep.py:212
#endif
                }
#if defined(DSA_MCEP_ABL) && DSA_MCEP_ABL == 3
                accB[0] += e * ea0; accB[1] += e * ea1; accB[2] += e * ea2;
#else
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    accB[0] = mfma4(ea0[r], e[r], accB[0]);
                    accB[1] = mfma4(ea1[r], e[r], accB[1]);
                    accB[2] = mfma4(ea2[r], e[r], accB[2]);
                }
#endif
                rt48 += e[0] * c48[0] + e[1] * c48[1] + e[2] * c48[2] + e[3] * c48[3];
                pa = pn;
                qa = qn;
                // issue pattern for this tile: (1 MFMA, 5 VALU) x 7 for the D-chain + exp, then
                // (1 MFMA, 1 VALU/DS) x 12 for the E-chain and the next tile's operand reads
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
            }
            float d256 = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) d256 += mcB[ks] * lds[D256_OFF + 4 * ks + g];
            d256 += __shfl_xor(d256, 16, 64);
            d256 += __shfl_xor(d256, 32, 64);
            const float e256 = __builtin_amdgcn_exp2f(logx256 + d256);
#pragma unroll
            for (int it = 0; it < 3; ++it)
                accB[it] = mfma4(g == 0 ? lds[E256_OFF + it * 16 + n] : 0.f, g == 0 ? e256 : 0.f, accB[it]);
            rt48 += __shfl_xor(rt48, 16, 64);
            rt48 += __shfl_xor(rt48, 32, 64);
            rt48 += e256 * lds[E256_OFF + 48];

            // ------------- rt and its reflection into this frame's LDS windows -------------
            DSA_STAMP(1);
#pragma unroll
            for (int it = 0; it < 3; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = it * 16 + 4 * g + r;
                    const float v = accB[it][r];
                    rt_lds[idx] = v;
                    if (idx <= 27) {  // rr[27 + d] = r[|d|]
                        rr_lds[27 + idx] = v;
                        rr_lds[27 - idx] = v;
                    }
                }
            if (g == 0) rt_lds[48] = rt48;
            __builtin_amdgcn_wave_barrier();
            DSA_STAMP(2);

            // ------------- local rows of R + Q, symmetric elimination, back substitution -------------
            float a[colm::TOTAL];
            col_build_rows<0>(a, rt_q, rr_q, lds + AV_OFF, (const float*)nullptr, gs, gq);
            __builtin_amdgcn_wave_barrier();
            DSA_STAMP(3);
            col_elim_all(a, std::make_integer_sequence<int, M1>{});
            DSA_STAMP(4);
            // xq[ks] = x[4 ks + gs] of frame nq; slot 6 of lane 1 is the right-hand-side column (x = -1)
            float xq[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, keep_if(gq.m[1], -1.f)};
            col_backsub_all(a, xq, gq, std::make_integer_sequence<int, M1>{});
            xq[6] = keep_if(gq.m[0], xq[6]);  // only lane 0's slot 6 is a solution component (x[24])
            // back to the MFMA layout through the (now free) rt window: mc += x  (mcep.py:222)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) rt_q[4 * ks + gs] = xq[ks];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) mcB[ks] += rt_lds[4 * ks + g];
            __builtin_amdgcn_wave_barrier();
            DSA_STAMP(5);
            if (hist && f_ok)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    if (4 * ks + g < M1) hist[((long)(iter + 1) * F + f) * M1 + 4 * ks + g] = mcB[ks];
        }
        if (f_ok)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                if (4 * ks + g < M1) mc_out[f * M1 + 4 * ks + g] = mcB[ks];
        unsigned int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1u);
        tile = wave_stride + (long)__builtin_amdgcn_readfirstlane((int)nxt);
#ifdef DSA_MCEP_TIMING
        if (blockIdx.x == 0 && threadIdx.x == 0) { g_mcep_stamps[9] = __builtin_readcyclecounter(); g_mcep_stamps[10] += 1; }
#endif
    }
}

int mcep_mfma_supported(int nfft, int M, int dtype) { return dtype == DSA_F32 && nfft == 512 && M == 24; }

// =====================================================================================
// Backward of the unrolled Newton iteration (what autograd gives the reference: SURVEY.md
// section 3.5), float32 / fft_length 512 / cep_order 24, same wave mapping as the forward.
// For step k (cotangent mbar of mc_{k+1}, saved iterate mc_k):
//   re-form e = exp(ln X - 2 mc_k D), rt = e E and the system A = T(rt[:25]) + H(rt)   (forward code)
//   solve A [g | u] = [rt[:25] - alpha | mbar]      (two right-hand sides in one elimination)
//   rtbar[m] = -sum_{i+j=m} u_i g_j - [m<=24] (sum_{|i-j|=m} u_i g_j - u_m)
//   ebar^T = E rtbar^T  (MFMA; operand gathered from the forward's E^T image);  zbar = ebar * e
//   lbar += zbar ;  mbar <- mbar - 2 D zbar^T   (MFMA on the D^T image; C/D registers are the B operand)
// and finally lbar += G mbar_0, gX = lbar / X.
// =====================================================================================
template <int k, int m>
__device__ __forceinline__ void sym2_update_row(SymRows& a, float (&b)[mm::NR], float (&b2)[mm::NR],
                                                const float (&prow)[mm::M1], float pb, float pb2, float inv,
                                                const GroupMask& gq)
{
    using namespace mm;
    constexpr int mk = k >> 2;
    if constexpr (m >= mk && m < NR) {
        constexpr int i0 = 4 * m;
        const float c0 = (i0 + 0 > k && i0 + 0 < M1) ? prow[i0 + 0 < M1 ? i0 + 0 : 0] : 0.f;
        const float c1 = (i0 + 1 > k && i0 + 1 < M1) ? prow[i0 + 1 < M1 ? i0 + 1 : 0] : 0.f;
        const float c2 = (i0 + 2 > k && i0 + 2 < M1) ? prow[i0 + 2 < M1 ? i0 + 2 : 0] : 0.f;
        const float c3 = (i0 + 3 > k && i0 + 3 < M1) ? prow[i0 + 3 < M1 ? i0 + 3 : 0] : 0.f;
        const float fct = sel4(gq, c0, c1, c2, c3) * inv;
        float* row = sym_row<m>(a);
        constexpr int j0 = (4 * m > k + 1) ? 4 * m : k + 1;
#pragma unroll
        for (int j = j0; j < M1; ++j) row[j - 4 * m] -= fct * prow[j];
        b[m] -= fct * pb;
        b2[m] -= fct * pb2;
        sym2_update_row<k, m + 1>(a, b, b2, prow, pb, pb2, inv, gq);
    }
}
template <int k>
__device__ __forceinline__ void sym2_elim_step(SymRows& a, float (&b)[mm::NR], float (&b2)[mm::NR],
                                               const GroupMask& gq)
{
    using namespace mm;
    constexpr int gk = k & 3, mk = k >> 2;
    float prow[M1];
    float* prw = sym_row<mk>(a);
#pragma unroll
    for (int j = k; j < M1; ++j) prow[j] = quad_bcast<gk>(prw[j - 4 * mk]);
    const float pb = quad_bcast<gk>(b[mk]), pb2 = quad_bcast<gk>(b2[mk]);
    const float inv = rcp_nr(prow[k]);
    sym2_update_row<k, mk>(a, b, b2, prow, pb, pb2, inv, gq);
}
template <int k, int m>
__device__ __forceinline__ void sym2_backsub_rows(SymRows& a, float (&b)[mm::NR], float (&b2)[mm::NR], float xk,
                                                  float uk)
{
    constexpr int mk = k >> 2;
    if constexpr (m <= mk) {
        const float aik = sym_row<m>(a)[k - 4 * m];
        b[m] -= aik * xk;
        b2[m] -= aik * uk;
        sym2_backsub_rows<k, m + 1>(a, b, b2, xk, uk);
    }
}
// full solution vectors on every lane of the quad: gv = A^{-1} b, uv = A^{-1} b2
template <int k>
__device__ __forceinline__ void sym2_backsub_step(SymRows& a, float (&b)[mm::NR], float (&b2)[mm::NR],
                                                  float (&gv)[mm::M1], float (&uv)[mm::M1])
{
    constexpr int gk = k & 3, mk = k >> 2;
    const float rinv = rcp_nr(sym_row<mk>(a)[k - 4 * mk]);
    const float xk = quad_bcast<gk>(b[mk] * rinv), uk = quad_bcast<gk>(b2[mk] * rinv);
    gv[k] = xk;
    uv[k] = uk;
    sym2_backsub_rows<k, 0>(a, b, b2, xk, uk);
}
template <int... Ks>
__device__ __forceinline__ void sym2_elim_all(SymRows& a, float (&b)[mm::NR], float (&b2)[mm::NR],
                                              const GroupMask& gq, std::integer_sequence<int, Ks...>)
{
    (sym2_elim_step<Ks>(a, b, b2, gq), ...);
}
template <int... Ks>
__device__ __forceinline__ void sym2_backsub_all(SymRows& a, float (&b)[mm::NR], float (&b2)[mm::NR],
                                                 float (&gv)[mm::M1], float (&uv)[mm::M1],
                                                 std::integer_sequence<int, Ks...>)
{
    (sym2_backsub_step<mm::M1 - 1 - Ks>(a, b, b2, gv, uv), ...);
}
// rtbar[m] for compile-time m (every lane of the quad computes all of them: static registers only)
template <int m>
__device__ __forceinline__ float rtbar_at(const float (&gv)[mm::M1], const float (&uv)[mm::M1])
{
    using namespace mm;
    float acc = 0.f;
    constexpr int ilo = m - (M1 - 1) > 0 ? m - (M1 - 1) : 0;
    constexpr int ihi = m < M1 - 1 ? m : M1 - 1;
#pragma unroll
    for (int i = ilo; i <= ihi; ++i) acc -= uv[i] * gv[m - i];  // Hankel diagonals: i + j = m
    if constexpr (m < M1) {
#pragma unroll
        for (int i = 0; i + m < M1; ++i) {  // Toeplitz diagonals: |i - j| = m
            acc -= uv[i] * gv[i + m];
            if (m > 0) acc -= uv[i + m] * gv[i];
        }
        acc += uv[m];  // through the right-hand side rt[:25] - alpha
    }
    return acc;
}
template <int... Ms>
__device__ __forceinline__ void rtbar_store(float* dst, const float (&gv)[mm::M1], const float (&uv)[mm::M1],
                                            std::integer_sequence<int, Ms...>)
{
    ((dst[Ms] = rtbar_at<Ms>(gv, uv)), ...);
}

namespace mmb {
using namespace mm;
constexpr int WAVES = 4;
constexpr int WAVE_FLOATS_B = 3 * 16 * RS;  // rt, rr and an exchange window per frame
constexpr int LDS_FLOATS = WAVE_OFF + WAVES * WAVE_FLOATS_B;
}  // namespace mmb

__global__ __launch_bounds__(256, 1) void mcep_mfma_bwd_kernel(
    const float* __restrict__ gmc, const float* __restrict__ X, const float* __restrict__ hist, long F, int n_iter,
    const float* __restrict__ G, const float* __restrict__ D, const float* __restrict__ E,
    const float* __restrict__ av, float* __restrict__ gX, long ntiles16)
{
    using namespace mmb;
    constexpr float kNeg2Log2e = -2.885390081777926815f, kLn2 = 0.693147180559945309f;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;

    // ---------------- operand images (identical to the forward kernel's) ----------------
    for (int idx = tid; idx < 16 * 2 * 64 * 4; idx += WAVES * 64) {
        int q = idx & 3, l = (idx >> 2) & 63, half = (idx >> 8) & 1, mt = idx >> 9;
        int k = 4 * (half * 4 + q) + (l >> 4);
        lds[DT_OFF + idx] = k < M1 ? kNeg2Log2e * D[k * K + mt * 16 + (l & 15)] : 0.f;
    }
    for (int idx = tid; idx < 3 * 16 * 64 * 4; idx += WAVES * 64) {
        int r = idx & 3, l = (idx >> 2) & 63, mt = (idx >> 8) & 15, it = idx >> 12;
        lds[ET_OFF + idx] = E[(mt * 16 + (l >> 4) * 4 + r) * M2 + it * 16 + (l & 15)];
    }
    {
        int r = tid & 3, gg = (tid >> 2) & 3, mt = tid >> 4;
        lds[E48_OFF + tid] = E[(mt * 16 + gg * 4 + r) * M2 + 48];
    }
    if (tid < 52) lds[E256_OFF + tid] = tid < M2 ? E[H * M2 + tid] : 0.f;
    if (tid < 28) {
        lds[D256_OFF + tid] = tid < M1 ? kNeg2Log2e * D[tid * K + H] : 0.f;
        lds[AV_OFF + tid] = tid < M1 ? av[tid] : 0.f;
    }
    __syncthreads();

    float* wave_lds = lds + WAVE_OFF + wave * WAVE_FLOATS_B;
    float* win_n = wave_lds + n * RS;          // rt window of frame n (MFMA-layout view)
    float* rr_n = win_n + 16 * RS;
    float* aux_n = win_n + 32 * RS;            // exchange window
    const int nq = lane >> 2, gs = lane & 3;   // solve layout: a quad per frame
    float* win_q = wave_lds + nq * RS;
    float* rr_q = win_q + 16 * RS;
    float* aux_q = win_q + 32 * RS;
    const GroupMask gq = make_group_mask(gs);
    const f32x4* Dt4 = reinterpret_cast<const f32x4*>(lds + DT_OFF);
    const f32x4* Et4 = reinterpret_cast<const f32x4*>(lds + ET_OFF);
    const f32x4* E484 = reinterpret_cast<const f32x4*>(lds + E48_OFF);
    const long wave_id = (long)blockIdx.x * WAVES + wave;
    const long wave_stride = (long)gridDim.x * WAVES;
    // gathers of the transposed-role operands out of the forward images:
    //   E[bin = mt*16 + bl][out = 4ks + g]        = Et[((ks>>2)*16 + mt)*256 + 16*(ks&3) + eb_lane]
    //   -2log2e D[coef = it2*16 + bl][bin = mt*16 + 4g + r] = Dt[(mt*2 + it2)*256 + db_lane + 4r]
    const int bl = lane & 15;
    const int eb_lane = ((bl >> 2) * 16 + g) * 4 + (bl & 3);
    const int db_lane = ((bl & 3) * 16 + 4 * g) * 4 + (bl >> 2);
    const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)G, 0, K * M1 * 4, 0x00020000);

    for (long tile = wave_id; tile < ntiles16; tile += wave_stride) {
        const long f_raw = tile * 16 + n;
        const bool f_ok = f_raw < F;
        const long f = f_ok ? f_raw : F - 1;
        const float* xf = X + f * K;
        f32x4 logx[16], lbar[16];
#pragma unroll
        for (int mt = 0; mt < 16; ++mt) {
            const float* p = xf + mt * 16 + 4 * g;
            logx[mt] = f32x4{__log2f(p[0]), __log2f(p[1]), __log2f(p[2]), __log2f(p[3])};
            lbar[mt] = f32x4{0, 0, 0, 0};
        }
        const float logx256 = __log2f(xf[H]);
        float lbar256 = 0.f;
        // mbar in the C/D layout of a 28-row product: tile it2, reg r <-> coefficient it2*16 + 4g + r
        f32x4 mbarC[2];
#pragma unroll
        for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = it2 * 16 + 4 * g + r;
                mbarC[it2][r] = c < M1 ? gmc[f * M1 + c] : 0.f;
            }

        for (int iter = n_iter - 1; iter >= 0; --iter) {
            float mcB[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                mcB[ks] = (4 * ks + g < M1) ? hist[((long)iter * F + f) * M1 + 4 * ks + g] : 0.f;
            // ---- forward quantities of this step: e (kept in registers), rt -> LDS windows ----
            f32x4 e[16];
            f32x4 accB[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
            float rt48 = 0.f;
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
                const f32x4 a0 = Dt4[(mt * 2 + 0) * 64 + lane];
                const f32x4 a1 = Dt4[(mt * 2 + 1) * 64 + lane];
                f32x4 pa = {0, 0, 0, 0}, qa = {0, 0, 0, 0};
                pa = mfma4(a0[0], mcB[0], pa);
                qa = mfma4(a1[0], mcB[4], qa);
                pa = mfma4(a0[1], mcB[1], pa);
                qa = mfma4(a1[1], mcB[5], qa);
                pa = mfma4(a0[2], mcB[2], pa);
                qa = mfma4(a1[2], mcB[6], qa);
                pa = mfma4(a0[3], mcB[3], pa);
                const f32x4 acc = pa + qa;
#pragma unroll
                for (int r = 0; r < 4; ++r) e[mt][r] = __builtin_amdgcn_exp2f(logx[mt][r] + acc[r]);
                const f32x4 ea0 = Et4[(0 * 16 + mt) * 64 + lane];
                const f32x4 ea1 = Et4[(1 * 16 + mt) * 64 + lane];
                const f32x4 ea2 = Et4[(2 * 16 + mt) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    accB[0] = mfma4(ea0[r], e[mt][r], accB[0]);
                    accB[1] = mfma4(ea1[r], e[mt][r], accB[1]);
                    accB[2] = mfma4(ea2[r], e[mt][r], accB[2]);
                }
                const f32x4 c48 = E484[mt * 4 + g];
                rt48 += e[mt][0] * c48[0] + e[mt][1] * c48[1] + e[mt][2] * c48[2] + e[mt][3] * c48[3];
            }
            float d256 = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) d256 += mcB[ks] * lds[D256_OFF + 4 * ks + g];
            d256 += __shfl_xor(d256, 16, 64);
            d256 += __shfl_xor(d256, 32, 64);
            const float e256 = __builtin_amdgcn_exp2f(logx256 + d256);
#pragma unroll
            for (int it = 0; it < 3; ++it)
                accB[it] = mfma4(g == 0 ? lds[E256_OFF + it * 16 + n] : 0.f, g == 0 ? e256 : 0.f, accB[it]);
            rt48 += __shfl_xor(rt48, 16, 64);
            rt48 += __shfl_xor(rt48, 32, 64);
            rt48 += e256 * lds[E256_OFF + 48];
#pragma unroll
            for (int it = 0; it < 3; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = it * 16 + 4 * g + r;
                    const float v = accB[it][r];
                    win_n[idx] = v;
                    if (idx <= 27) {
                        rr_n[27 + idx] = v;
                        rr_n[27 - idx] = v;
                    }
                }
            if (g == 0) win_n[48] = rt48;
            // mbar to the exchange window (C/D layout writer -> quad-layout reader)
#pragma unroll
            for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                for (int r = 0; r < 4; ++r) aux_n[it2 * 16 + 4 * g + r] = mbarC[it2][r];
            __builtin_amdgcn_wave_barrier();

            // ---- solve A [gv | uv] = [rt[:25] - alpha | mbar] in the quad layout (column-cyclic) ----
            float gv[M1], uv[M1];
            {
                float a[colm::TOTAL];
                col_build_rows<0>(a, win_q, rr_q, lds + AV_OFF, aux_q, gs, gq);
                __builtin_amdgcn_wave_barrier();
                col_elim_all(a, std::make_integer_sequence<int, M1>{});
                float xq1[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, keep_if(gq.m[1], -1.f)};
                col_backsub_full(a, xq1, gv, gq, std::make_integer_sequence<int, M1>{});
                float xq2[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, keep_if(gq.m[2], -1.f)};
                col_backsub_full(a, xq2, uv, gq, std::make_integer_sequence<int, M1>{});
            }
            // ---- rtbar (all 49 entries on every lane; lane 0 of the quad publishes them) ----
            if (gs == 0) {
                rtbar_store(aux_q, gv, uv, std::make_integer_sequence<int, M2>{});
                aux_q[49] = 0.f;
                aux_q[50] = 0.f;
                aux_q[51] = 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            float rtbB[13];
#pragma unroll
            for (int ks = 0; ks < 13; ++ks) rtbB[ks] = aux_n[4 * ks + g];
            __builtin_amdgcn_wave_barrier();

            // ---- ebar^T = E rtbar^T ; zbar = ebar * e ; lbar += zbar ; mbar += (-2 D) zbar^T ----
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
                f32x4 acc = {0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < 12; ++ks) {
                    const float av_ = lds[ET_OFF + ((ks >> 2) * 16 + mt) * 256 + 16 * (ks & 3) + eb_lane];
                    acc = mfma4(av_, rtbB[ks], acc);
                }
                acc = mfma4(g == 0 ? lds[E48_OFF + mt * 16 + bl] : 0.f, rtbB[12], acc);  // out = 48 (only g = 0 slot)
                f32x4 zb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    zb[r] = acc[r] * e[mt][r];
                    lbar[mt][r] += zb[r];
                    zb[r] *= kLn2;  // the D image is scaled by -2 log2(e): (-2 log2e D)(ln2 zbar) = -2 D zbar
                }
#pragma unroll
                for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        mbarC[it2] = mfma4(lds[DT_OFF + (mt * 2 + it2) * 256 + db_lane + 4 * r], zb[r], mbarC[it2]);
            }
            // Nyquist bin
            float eb256 = 0.f;
#pragma unroll
            for (int ks = 0; ks < 13; ++ks) eb256 += rtbB[ks] * lds[E256_OFF + 4 * ks + g];  // slots 49..51 are 0
            eb256 += __shfl_xor(eb256, 16, 64);
            eb256 += __shfl_xor(eb256, 32, 64);
            const float zb256 = eb256 * e256;
            lbar256 += zb256;
#pragma unroll
            for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = it2 * 16 + 4 * g + r;
                    if (c < 28) mbarC[it2][r] += (zb256 * kLn2) * lds[D256_OFF + c];
                }
        }

        // ---- lbar += G mbar_0 (mcep.py:204-207 adjoint); gX = lbar / X ----
#pragma unroll
        for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
            for (int r = 0; r < 4; ++r) aux_n[it2 * 16 + 4 * g + r] = mbarC[it2][r];
        __builtin_amdgcn_wave_barrier();
        float m0B[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) m0B[ks] = (4 * ks + g < M1) ? aux_n[4 * ks + g] : 0.f;
        __builtin_amdgcn_wave_barrier();
        const int gvoff = (bl * M1 + g) * 4;
        float part256 = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c = 4 * ks + g;
            const float g256 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, (c < M1 ? c : 0) * 4, H * M1 * 4, 0));
            part256 += (c < M1 ? g256 : 0.f) * m0B[ks];
        }
        part256 += __shfl_xor(part256, 16, 64);
        part256 += __shfl_xor(part256, 32, 64);
        lbar256 += part256;
#pragma unroll
        for (int mt = 0; mt < 16; ++mt) {
            f32x4 acc = lbar[mt];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float gvv = __builtin_bit_cast(
                    float, __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, gvoff, (mt * 16 * M1 + 4 * ks) * 4, 0));
                acc = mfma4((4 * ks + g < M1) ? gvv : 0.f, m0B[ks], acc);
            }
            if (f_ok) {
                float* dst = gX + f * K + mt * 16 + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r] = acc[r] * __builtin_amdgcn_exp2f(-logx[mt][r]);
            }
        }
        if (f_ok && g == 0) gX[f * K + H] = lbar256 * __builtin_amdgcn_exp2f(-logx256);
    }
}

int mcep_mfma_bwd_h(const void* gmc, const void* X, const void* hist, int64_t F, int n_iter, const void* G, const void* D,
                    const void* E, const void* av, void* gX, hipStream_t st);

int mcep_mfma_bwd(const void* gmc, const void* X, const void* hist, int64_t F, int n_iter, const void* G, const void* D,
                  const void* E, const void* av, void* gX, hipStream_t st)
{
    // DSA_MCEP_BWD_VARIANT (A/B knob): 16 = split-precision binary16 MFMA chains (default), 8 = float32 MFMA chains
    static const int variant = [] {
        const char* e = getenv("DSA_MCEP_BWD_VARIANT");
        return e ? atoi(e) : 16;
    }();
    if (variant == 16) return mcep_mfma_bwd_h(gmc, X, hist, F, n_iter, G, D, E, av, gX, st);
    const int lds_bytes = mmb::LDS_FLOATS * 4;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)mcep_mfma_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_bytes) != hipSuccess)
            return fail(DSA_ERR_LAUNCH, "mcep_mfma_bwd: cannot reserve the LDS operand images%s");
        attr_set = true;
    }
    long ntiles16 = (long)((F + 15) / 16);
    long blocks = (ntiles16 + mmb::WAVES - 1) / mmb::WAVES;
    long grid = blocks < 256 ? blocks : 256;
    hipLaunchKernelGGL(mcep_mfma_bwd_kernel, dim3((unsigned)grid), dim3(256), lds_bytes, st, (const float*)gmc,
                       (const float*)X, (const float*)hist, (long)F, n_iter, (const float*)G, (const float*)D,
                       (const float*)E, (const float*)av, (float*)gX, ntiles16);
    return check_launch("mcep_mfma_bwd");
}

// =====================================================================================
// v3: role-split workgroup.  Waves 0-3 ("matrix" role) keep log X of TWO 16-frame groups in
// registers and run only the MFMA chains + exp; waves 4-7 ("solver" role) own mc and run only
// the build / elimination / back-substitution.  Wave w and wave w+4 sit on the same SIMD, whose
// matrix pipe and VALU are then busy at the same time: while the matrix wave forms rt for group A
// the solver wave eliminates group B, and they swap every phase (one workgroup barrier per
// phase).  Neither role needs more than 256 registers, so nothing spills.
//   phase p: matrix wave -> group p & 1, Newton step p >> 1;  solver wave -> group (p-1) & 1, step (p-1) >> 1
// Hand-off through LDS: rt windows (matrix -> solver), mc in the same window (solver -> matrix);
// mc0 travels through the mc_out buffer in global memory.
// =====================================================================================
namespace mm3 {
using namespace mm;
constexpr int GROUP_FLOATS = 2 * 16 * RS;                 // rt + rr windows of one 16-frame group
constexpr int PAIR_FLOATS = 2 * GROUP_FLOATS;             // two groups per matrix/solver pair
constexpr int LDS_FLOATS = WAVE_OFF + 4 * PAIR_FLOATS;    // operand images + 4 pairs
}  // namespace mm3

__global__ __launch_bounds__(512, 2) void mcep_mfma_fwd_kernel_v3(
    const float* __restrict__ X, long F, int n_iter, const float* __restrict__ G,
    const float* __restrict__ D, const float* __restrict__ E, const float* __restrict__ av,
    float* __restrict__ mc_out, float* __restrict__ hist, long nbt)
{
    using namespace mm3;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int pair = wave & 3;
    const bool matrix_role = wave < 4;

    // ---------------- operand images: built once per workgroup ----------------
    for (int idx = tid; idx < 16 * 2 * 64 * 4; idx += 512) {
        int q = idx & 3, l = (idx >> 2) & 63, half = (idx >> 8) & 1, mt = idx >> 9;
        int k = 4 * (half * 4 + q) + (l >> 4);
        lds[DT_OFF + idx] = k < M1 ? D[k * K + mt * 16 + (l & 15)] : 0.f;
    }
    for (int idx = tid; idx < 3 * 16 * 64 * 4; idx += 512) {
        int r = idx & 3, l = (idx >> 2) & 63, mt = (idx >> 8) & 15, it = idx >> 12;
        lds[ET_OFF + idx] = E[(mt * 16 + (l >> 4) * 4 + r) * M2 + it * 16 + (l & 15)];
    }
    if (tid < 256) {
        int r = tid & 3, gg = (tid >> 2) & 3, mt = tid >> 4;
        lds[E48_OFF + tid] = E[(mt * 16 + gg * 4 + r) * M2 + 48];
    }
    if (tid < M2) lds[E256_OFF + tid] = E[H * M2 + tid];
    if (tid < 28) {
        lds[D256_OFF + tid] = tid < M1 ? D[tid * K + H] : 0.f;
        lds[AV_OFF + tid] = tid < M1 ? av[tid] : 0.f;
    }
    __syncthreads();

    float* pair_lds = lds + WAVE_OFF + pair * PAIR_FLOATS;
    const int nphase = 2 * n_iter + 1;

    if (matrix_role) {
        // =============================== matrix role ===============================
        const int n = lane & 15, g = lane >> 4;
        const f32x4* Dt4 = reinterpret_cast<const f32x4*>(lds + DT_OFF);
        const f32x4* Et4 = reinterpret_cast<const f32x4*>(lds + ET_OFF);
        const f32x4* E484 = reinterpret_cast<const f32x4*>(lds + E48_OFF);
        const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)G, 0, K * M1 * 4, 0x00020000);

        // one phase of one group: (step 0: load log X, mc0 = G^T log X) ; rt = E^T exp(log X - 2 D^T mc)
        auto phase = [&](f32x4(&logx)[16], float& logx256, int grp, int step, long f0) __attribute__((always_inline)) {
            float* rt_w = pair_lds + grp * GROUP_FLOATS + n * RS;  // this lane's frame windows
            float* rr_w = rt_w + 16 * RS;
            const long f_raw = f0 + n;
            const bool f_ok = f_raw < F;
            const long f = f_ok ? f_raw : F - 1;
            float mcB[KS];
            if (step == 0) {
                const float* xf = X + f * K;
#pragma unroll
                for (int mt = 0; mt < 16; ++mt) {
                    const float* p = xf + mt * 16 + 4 * g;
                    logx[mt] = f32x4{logf(p[0]), logf(p[1]), logf(p[2]), logf(p[3])};  // mcep.py:203
                }
                logx256 = logf(xf[H]);
                f32x4 accG[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
                // G operands by buffer loads: one VGPR offset per output tile, the (mt, r) part of
                // the address is a scalar immediate -- keeps the address arithmetic out of VGPRs
#pragma unroll
                for (int it = 0; it < 2; ++it) {  // mc0^T = G^T logx^T  (mcep.py:204-207)
                    const int out = it * 16 + n;
                    const bool ov = out < M1;
                    const int voff = ((4 * g) * M1 + (ov ? out : 0)) * 4;
#pragma unroll
                    for (int mt = 0; mt < 16; ++mt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float gv = __builtin_bit_cast(
                                float, __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, voff, (mt * 16 + r) * M1 * 4, 0));
                            accG[it] = mfma4(ov ? gv : 0.f, logx[mt][r], accG[it]);
                        }
                    }
                    const float g256 = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, (ov ? out : 0) * 4, H * M1 * 4, 0));
                    accG[it] = mfma4((ov && g == 0) ? g256 : 0.f, g == 0 ? logx256 : 0.f, accG[it]);
                }
                // accG[it][r] = mc0[it*16 + 4g + r]: to the solver through mc_out (and hist[0]),
                // to this wave's B-operand layout through the (free) rt window
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = it * 16 + 4 * g + r;
                        rt_w[c] = accG[it][r];
                        if (f_ok && c < M1) {
                            mc_out[f * M1 + c] = accG[it][r];
                            if (hist) hist[f * M1 + c] = accG[it][r];
                        }
                    }
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) mcB[ks] = (4 * ks + g < M1) ? rt_w[4 * ks + g] : 0.f;
            __builtin_amdgcn_wave_barrier();

            f32x4 accB[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
            float rt48 = 0.f;
            auto d_chain = [&](int mt) __attribute__((always_inline)) -> f32x4 {  // d^T = D^T mc^T, two accumulators (mcep.py:210-211)
                const f32x4 a0 = Dt4[(mt * 2 + 0) * 64 + lane];
                const f32x4 a1 = Dt4[(mt * 2 + 1) * 64 + lane];
                f32x4 pa = {0, 0, 0, 0}, qa = {0, 0, 0, 0};
                pa = mfma4(a0[0], mcB[0], pa);
                qa = mfma4(a1[0], mcB[4], qa);
                pa = mfma4(a0[1], mcB[1], pa);
                qa = mfma4(a1[1], mcB[5], qa);
                pa = mfma4(a0[2], mcB[2], pa);
                qa = mfma4(a1[2], mcB[6], qa);
                pa = mfma4(a0[3], mcB[3], pa);
                return pa + qa;
            };
            f32x4 acc = d_chain(0);
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
                f32x4 acc_next = acc;
                if (mt + 1 < 16) acc_next = d_chain(mt + 1);
                f32x4 e;
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = exp_nobranch(logx[mt][r] - 2.f * acc[r]);  // mcep.py:212
#pragma unroll
                for (int it = 0; it < 3; ++it) {  // rt^T += E^T e^T  (mcep.py:214-215)
                    const f32x4 a = Et4[(it * 16 + mt) * 64 + lane];
                    accB[it] = mfma4(a[0], e[0], accB[it]);
                    accB[it] = mfma4(a[1], e[1], accB[it]);
                    accB[it] = mfma4(a[2], e[2], accB[it]);
                    accB[it] = mfma4(a[3], e[3], accB[it]);
                }
                const f32x4 c48 = E484[mt * 4 + g];
                rt48 += e[0] * c48[0] + e[1] * c48[1] + e[2] * c48[2] + e[3] * c48[3];
                acc = acc_next;
            }
            float d256 = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) d256 += mcB[ks] * lds[D256_OFF + 4 * ks + g];
            d256 += __shfl_xor(d256, 16, 64);
            d256 += __shfl_xor(d256, 32, 64);
            const float e256 = exp_nobranch(logx256 - 2.f * d256);
#pragma unroll
            for (int it = 0; it < 3; ++it)
                accB[it] = mfma4(g == 0 ? lds[E256_OFF + it * 16 + n] : 0.f, g == 0 ? e256 : 0.f, accB[it]);
            rt48 += __shfl_xor(rt48, 16, 64);
            rt48 += __shfl_xor(rt48, 32, 64);
            rt48 += e256 * lds[E256_OFF + 48];
#pragma unroll
            for (int it = 0; it < 3; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = it * 16 + 4 * g + r;
                    const float v = accB[it][r];
                    rt_w[idx] = v;
                    if (idx <= 27) {  // rr[27 + d] = r[|d|]
                        rr_w[27 + idx] = v;
                        rr_w[27 - idx] = v;
                    }
                }
            if (g == 0) rt_w[48] = rt48;
        };

        f32x4 logxA[16], logxB[16];
        float l256A = 0.f, l256B = 0.f;
        for (long bt = blockIdx.x; bt < nbt; bt += gridDim.x) {
            const long fbase = bt * 128 + pair * 32;
            for (int p = 0; p < nphase; ++p) {
                if (p < 2 * n_iter) {
                    if ((p & 1) == 0) phase(logxA, l256A, 0, p >> 1, fbase);
                    else phase(logxB, l256B, 1, p >> 1, fbase + 16);
                }
                __syncthreads();
            }
        }
    } else {
        // =============================== solver role ===============================
        const int nq = lane >> 2, gs = lane & 3;  // the 4 lanes of a quad share a frame
        const GroupMask gq = make_group_mask(gs);
        float mcqA[KS], mcqB[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) mcqA[ks] = mcqB[ks] = 0.f;

        auto solve = [&](float(&mcq)[KS], int grp, int step, long f0) __attribute__((always_inline)) {
            float* rt_q = pair_lds + grp * GROUP_FLOATS + nq * RS;
            float* rr_q = rt_q + 16 * RS;
            const long f = f0 + nq;
            const bool f_ok = f < F;
            if (step == 0) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    mcq[ks] = (f_ok && 4 * ks + gs < M1) ? mc_out[f * M1 + 4 * ks + gs] : 0.f;
            }
            SymRows a;
            float b[NR];
            sym_build_rows<0>(a, b, rt_q + gs, rr_q + 27 + gs, lds + AV_OFF, gs);
            __builtin_amdgcn_wave_barrier();
            sym_elim_all(a, b, nq, gq, std::make_integer_sequence<int, M1>{});
            float xq[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            sym_backsub_all(a, b, xq, nq, gq, std::make_integer_sequence<int, M1>{});
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                mcq[ks] += xq[ks];  // mcep.py:222
                rt_q[4 * ks + gs] = mcq[ks];  // next step's B operand for the matrix wave
                if (f_ok && 4 * ks + gs < M1) {
                    if (hist) hist[((long)(step + 1) * F + f) * M1 + 4 * ks + gs] = mcq[ks];
                    if (step == n_iter - 1) mc_out[f * M1 + 4 * ks + gs] = mcq[ks];
                }
            }
        };

        for (long bt = blockIdx.x; bt < nbt; bt += gridDim.x) {
            const long fbase = bt * 128 + pair * 32;
            for (int p = 0; p < nphase; ++p) {
                if (p >= 1) {
                    const int q = p - 1;
                    if ((q & 1) == 0) solve(mcqA, 0, q >> 1, fbase);
                    else solve(mcqB, 1, q >> 1, fbase + 16);
                }
                __syncthreads();
            }
        }
    }
}

}  // namespace dsa
#include "mcep_mfma_f16.h"
#include "mcep_mfma_bwd_f16.h"
namespace dsa {

static int launch_v3(const void* X, int64_t F, int n_iter, const void* G, const void* D, const void* E,
                     const void* av, void* mc, void* hist, hipStream_t st)
{
    const int lds_bytes = mm3::LDS_FLOATS * 4;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)mcep_mfma_fwd_kernel_v3, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_bytes) != hipSuccess)
            return fail(DSA_ERR_LAUNCH, "mcep_mfma: cannot reserve the LDS operand images%s");
        attr_set = true;
    }
    long nbt = (long)((F + 127) / 128);
    long grid = nbt < 256 ? nbt : 256;  // one persistent workgroup per CU
    hipLaunchKernelGGL(mcep_mfma_fwd_kernel_v3, dim3((unsigned)grid), dim3(512), lds_bytes, st, (const float*)X,
                       (long)F, n_iter, (const float*)G, (const float*)D, (const float*)E, (const float*)av,
                       (float*)mc, (float*)hist, nbt);
    return check_launch("mcep_mfma_fwd_split");
}

// Rotating pools of per-launch scratch (tile-queue counters, operand images of the split-precision
// kernel).  Allocated once under std::call_once, slots handed out by an atomic counter: launches
// from several host threads / streams never share a slot unless more than kSlots are in flight.
static unsigned int* queue_slot(hipStream_t st)
{
    static unsigned int* pool = nullptr;
    static std::once_flag once;
    static std::atomic<unsigned int> next{0};
    constexpr unsigned int kSlots = 256;
    std::call_once(once, [] {
        if (hipMalloc((void**)&pool, 2 * kSlots * sizeof(unsigned int)) != hipSuccess) pool = nullptr;
    });
    if (!pool) return nullptr;
    unsigned int* q = pool + 2 * (next.fetch_add(1, std::memory_order_relaxed) % kSlots);  // two counters per launch
    if (hipMemsetAsync(q, 0, 2 * sizeof(unsigned int), st) != hipSuccess) return nullptr;
    return q;
}

static _Float16* image_slot()
{
    static _Float16* pool = nullptr;
    static std::once_flag once;
    static std::atomic<unsigned int> next{0};
    constexpr unsigned int kSlots = 64;  // 112 KB each
    std::call_once(once, [] {
        if (hipMalloc((void**)&pool, (size_t)kSlots * mh::IMG_BYTES) != hipSuccess) pool = nullptr;
    });
    if (!pool) return nullptr;
    return pool + (size_t)(next.fetch_add(1, std::memory_order_relaxed) % kSlots) * (mh::IMG_BYTES / 2);
}

template <int WAVES>
static int launch_v2(const void* X, int64_t F, int n_iter, const void* G, const void* D, const void* E,
                     const void* av, void* mc, void* hist, hipStream_t st, const char* name)
{
    const int lds_bytes = mm2::lds_floats(WAVES) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)mcep_mfma_fwd_kernel_v2<WAVES>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return fail(DSA_ERR_LAUNCH, "mcep_mfma: cannot reserve the LDS operand images%s");
        attr_set = true;
    }
    long ntiles16 = (long)((F + 15) / 16);
    long blocks = (ntiles16 + WAVES - 1) / WAVES;
    long grid = blocks < 256 ? blocks : 256;  // one persistent workgroup per CU
    unsigned int* queue = queue_slot(st);
    if (!queue) return fail(DSA_ERR_LAUNCH, "mcep_mfma: cannot set up the tile queue%s");
    hipLaunchKernelGGL((mcep_mfma_fwd_kernel_v2<WAVES>), dim3((unsigned)grid), dim3(WAVES * 64), lds_bytes, st,
                       (const float*)X, (long)F, n_iter, (const float*)G, (const float*)D, (const float*)E,
                       (const float*)av, (float*)mc, (float*)hist, ntiles16, queue);
    return check_launch(name);
}

template <int WAVES>
static int launch_h(const void* X, int64_t F, int n_iter, const void* G, const void* D, const void* E,
                    const void* av, void* mc, void* hist, hipStream_t st, const char* name)
{
    const int lds_bytes = mh::h_lds_floats(WAVES) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)mcep_mfma_fwd_kernel_h<WAVES>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return fail(DSA_ERR_LAUNCH, "mcep_mfma: cannot reserve the LDS operand images%s");
        attr_set = true;
    }
    long ntiles16 = (long)((F + 15) / 16);
    long blocks = (ntiles16 + WAVES - 1) / WAVES;
    long grid = blocks < 256 ? blocks : 256;  // one persistent workgroup per CU
    unsigned int* queue = queue_slot(st);
    _Float16* img = image_slot();
    if (!queue || !img) return fail(DSA_ERR_LAUNCH, "mcep_mfma: cannot set up the tile queue / operand images%s");
    hipLaunchKernelGGL(mcep_h_prep_kernel, dim3((mh::IMG_D + mh::IMG_E + mh::IMG_G + 255) / 256), dim3(256), 0, st,
                       (const float*)G, (const float*)D, (const float*)E, img);
    // see the ticket comment in the kernel: a short last round goes to one wave per SIMD pair
    const long slots = grid * WAVES, full = ntiles16 / slots * slots, rest = ntiles16 - full;
    const long tiles_shared = (full > 0 && rest > 0 && rest <= slots / 2) ? full : ntiles16;
    hipLaunchKernelGGL((mcep_mfma_fwd_kernel_h<WAVES>), dim3((unsigned)grid), dim3(WAVES * 64), lds_bytes, st,
                       (const float*)X, (long)F, n_iter, (const float*)G, (const float*)D, (const float*)E,
                       (const float*)av, (float*)mc, (float*)hist, ntiles16, tiles_shared, queue, (const _Float16*)img);
    return check_launch(name);
}

static _Float16* image_slot_b()
{
    static _Float16* pool = nullptr;
    static std::once_flag once;
    static std::atomic<unsigned int> next{0};
    constexpr unsigned int kSlots = 32;  // 240 KB each
    constexpr size_t kStride = ((size_t)mhb::IMG_B_BYTES + 255) & ~(size_t)255;
    std::call_once(once, [] {
        if (hipMalloc((void**)&pool, kSlots * kStride) != hipSuccess) pool = nullptr;
    });
    if (!pool) return nullptr;
    return pool + (size_t)(next.fetch_add(1, std::memory_order_relaxed) % kSlots) * (kStride / 2);
}

int mcep_mfma_bwd_h(const void* gmc, const void* X, const void* hist, int64_t F, int n_iter, const void* G, const void* D,
                    const void* E, const void* av, void* gX, hipStream_t st)
{
    const int lds_bytes = mhb::B_LDS_FLOATS * 4;
    static std::once_flag once;
    static bool attr_ok = true;
    std::call_once(once, [&] {
        attr_ok = hipFuncSetAttribute((const void*)mcep_mfma_bwd_kernel_h, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      lds_bytes) == hipSuccess;
    });
    if (!attr_ok) return fail(DSA_ERR_LAUNCH, "mcep_mfma_bwd: cannot reserve the LDS operand images%s");
    unsigned int* queue = queue_slot(st);
    _Float16* img = image_slot_b();
    if (!queue || !img) return fail(DSA_ERR_LAUNCH, "mcep_mfma_bwd: cannot set up the tile queue / operand images%s");
    hipLaunchKernelGGL(mcep_h_prep_kernel, dim3((mh::IMG_D + mh::IMG_E + mh::IMG_G + 255) / 256), dim3(256), 0, st,
                       (const float*)G, (const float*)D, (const float*)E, img);
    hipLaunchKernelGGL(mcep_hb_prep_kernel, dim3((mhb::IMG_EB + mhb::IMG_DB + mhb::IMG_GB + 255) / 256), dim3(256), 0, st,
                       (const float*)G, (const float*)D, (const float*)E, img);
    long ntiles16 = (long)((F + 15) / 16);
    long blocks = (ntiles16 + mhb::WAVES_B - 1) / mhb::WAVES_B;
    long grid = blocks < 256 ? blocks : 256;
    hipLaunchKernelGGL(mcep_mfma_bwd_kernel_h, dim3((unsigned)grid), dim3(256), lds_bytes, st, (const float*)gmc,
                       (const float*)X, (const float*)hist, (long)F, n_iter, (const float*)av, (float*)gX, ntiles16, queue,
                       (const _Float16*)img);
    return check_launch("mcep_mfma_bwd");
}

int mcep_mfma_fwd(const void* X, int64_t F, int nfft, int M, int n_iter, const void* G, const void* D,
                  const void* E, const void* av, void* mc, void* hist, hipStream_t st)
{
    (void)nfft;
    (void)M;
    // DSA_MCEP_VARIANT (A/B knob): 16 = split-precision binary16 MFMA chains, two waves per SIMD
    // (default, fastest measured); 8 = float32 MFMA chains, two waves per SIMD; 4 = one wave per
    // SIMD; 3 = role-split matrix/solver waves; 1 = first kernel (full elimination, ds_bpermute)
    static const int variant = [] {
        const char* e = getenv("DSA_MCEP_VARIANT");
        return e ? atoi(e) : 16;
    }();
    if (variant == 1) {
        const int lds_bytes = mm::LDS_FLOATS * 4;
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)mcep_mfma_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    lds_bytes) != hipSuccess)
                return fail(DSA_ERR_LAUNCH, "mcep_mfma: cannot reserve %s of LDS", "117 KB");
            attr_set = true;
        }
        long ntiles = (long)((F + 63) / 64);
        long grid = ntiles < 256 ? ntiles : 256;
        hipLaunchKernelGGL(mcep_mfma_fwd_kernel, dim3((unsigned)grid), dim3(256), lds_bytes, st, (const float*)X,
                           (long)F, n_iter, (const float*)G, (const float*)D, (const float*)E, (const float*)av,
                           (float*)mc, (float*)hist, ntiles);
        return check_launch("mcep_mfma_fwd_v1");
    }
    if (variant == 8) return launch_v2<8>(X, F, n_iter, G, D, E, av, mc, hist, st, "mcep_mfma_fwd_f32");
    if (variant == 4) return launch_v2<4>(X, F, n_iter, G, D, E, av, mc, hist, st, "mcep_mfma_fwd_w4");
    if (variant == 3 && n_iter >= 1) return launch_v3(X, F, n_iter, G, D, E, av, mc, hist, st);
    return launch_h<8>(X, F, n_iter, G, D, E, av, mc, hist, st, "mcep_mfma_fwd");
}

}  // namespace dsa
