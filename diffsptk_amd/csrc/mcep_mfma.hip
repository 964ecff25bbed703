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

#include <utility>

namespace dsa {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace mm {
constexpr int H = 256, K = 257;   // nfft = 512
constexpr int M1 = 25, M2 = 49;   // cep_order 24
constexpr int KS = 7;             // k-steps of the first product (28 >= M1 coefficients)
constexpr int NR = 7;             // local rows per lane group (ceil(M1 / 4))
constexpr int RS = 66;            // per-frame stride of the rt / rr windows in LDS (floats)
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

int mcep_mfma_supported(int nfft, int M, int dtype) { return dtype == DSA_F32 && nfft == 512 && M == 24; }

int mcep_mfma_fwd(const void* X, int64_t F, int nfft, int M, int n_iter, const void* G, const void* D,
                  const void* E, const void* av, void* mc, void* hist, hipStream_t st)
{
    (void)nfft;
    (void)M;
    const int lds_bytes = mm::LDS_FLOATS * 4;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)mcep_mfma_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_bytes) != hipSuccess)
            return fail(DSA_ERR_LAUNCH, "mcep_mfma: cannot reserve %s of LDS", "117 KB");
        attr_set = true;
    }
    long ntiles = (long)((F + 63) / 64);
    long grid = ntiles < 256 ? ntiles : 256;  // one persistent workgroup per CU
    hipLaunchKernelGGL(mcep_mfma_fwd_kernel, dim3((unsigned)grid), dim3(256), lds_bytes, st, (const float*)X,
                       (long)F, n_iter, (const float*)G, (const float*)D, (const float*)E, (const float*)av,
                       (float*)mc, (float*)hist, ntiles);
    return check_launch("mcep_mfma_fwd");
}

}  // namespace dsa
