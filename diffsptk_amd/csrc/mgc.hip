// Toeplitz-plus-Hankel solve of the mel-generalized cepstral analysis:
//   MelGeneralizedCepstralAnalysis.forward, mgcep.py:226-229:  R = symmetric_toeplitz(pt), Q = hankel(qt),
//   gradient = torch.linalg.solve(R + Q, rt)     (utils/private.py:291-302 for the two builders).
// One wave per frame: the M x (M + 1) augmented system lives in LDS, lane i owns row i; Gauss-Jordan elimination
// with row pivoting by magnitude (the reference's LAPACK call pivots too; the system is not guaranteed positive
// definite for gamma != 0).  Backward: with A = T(p) + H(q) symmetric, u = A^{-1} gbar, rbar = u, Abar = -u g^T,
// pbar[k] = sum over |i - j| = k of Abar[i][j], qbar[k] = sum over i + j = k.  float32 and float64; M <= 64.
// The rest of the analysis (warping / FFT stages composed into row products, pointwise spectrum arithmetic) is
// assembled by the host layer from the library's row-product kernel (modules/mgcep.py).
#include "common.h"

namespace dsa {

constexpr int kThMax = 64;

// Solves the n x n system in LDS (row stride W >= n + nrhs) for nrhs right-hand sides; on return column n + c of row
// piv_row[k] divided by its pivot is x_c[k].  One wave, lane i owns row i (n <= 64).
template <typename T>
__device__ void th_gauss_jordan(T* Aug, int n, int W, int nrhs, int* rowof, int lane)
{
    unsigned long long used = 0ull;   // rows already chosen as pivots (uniform)
    for (int k = 0; k < n; ++k) {
        // pivot: the unused row with the largest |Aug[i][k]|
        T mag = (lane < n && !((used >> lane) & 1ull)) ? (Aug[lane * W + k] < T(0) ? -Aug[lane * W + k] : Aug[lane * W + k]) : T(-1);
        int arg = lane;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const T m2 = __shfl_xor(mag, o, 64);
            const int a2 = __shfl_xor(arg, o, 64);
            if (m2 > mag || (m2 == mag && a2 < arg)) {
                mag = m2;
                arg = a2;
            }
        }
        const int p = arg;   // uniform
        used |= 1ull << p;
        if (lane == 0) rowof[k] = p;
        const T inv = T(1) / Aug[p * W + k];
        const T fac = (lane < n && lane != p) ? Aug[lane * W + k] * inv : T(0);
        __builtin_amdgcn_wave_barrier();
        if (lane < n && lane != p)
            for (int j = k + 1; j < n + nrhs; ++j) Aug[lane * W + j] -= fac * Aug[p * W + j];
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T>
__device__ void th_build(T* Aug, const T* p, const T* q, int n, int W, int lane)
{
    if (lane < n)
        for (int j = 0; j < n; ++j) {
            const int d = lane > j ? lane - j : j - lane;
            Aug[lane * W + j] = p[d] + q[lane + j];
        }
}

template <typename T>
__global__ __launch_bounds__(64) void th_solve_fwd_kernel(const T* __restrict__ p, const T* __restrict__ q,
                                                          const T* __restrict__ r, long F, int n, T* __restrict__ g)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Aug = reinterpret_cast<T*>(smem_raw);
    const int W = n + 1;
    int* rowof = reinterpret_cast<int*>(Aug + (size_t)n * W);
    const int lane = threadIdx.x;
    for (long f = blockIdx.x; f < F; f += gridDim.x) {
        __builtin_amdgcn_wave_barrier();
        th_build(Aug, p + f * n, q + f * (2 * n - 1), n, W, lane);
        if (lane < n) Aug[lane * W + n] = r[f * n + lane];
        __builtin_amdgcn_wave_barrier();
        th_gauss_jordan(Aug, n, W, 1, rowof, lane);
        if (lane < n) {
            const int row = rowof[lane];
            g[f * n + lane] = Aug[row * W + n] / Aug[row * W + lane];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(64) void th_solve_bwd_kernel(const T* __restrict__ gg, const T* __restrict__ p,
                                                          const T* __restrict__ q, const T* __restrict__ g, long F, int n,
                                                          T* __restrict__ gp, T* __restrict__ gq, T* __restrict__ gr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Aug = reinterpret_cast<T*>(smem_raw);
    const int W = n + 1;
    int* rowof = reinterpret_cast<int*>(Aug + (size_t)n * W);
    T* u = reinterpret_cast<T*>(rowof + kThMax);
    T* gs = u + kThMax;
    const int lane = threadIdx.x;
    for (long f = blockIdx.x; f < F; f += gridDim.x) {
        __builtin_amdgcn_wave_barrier();
        th_build(Aug, p + f * n, q + f * (2 * n - 1), n, W, lane);
        if (lane < n) {
            Aug[lane * W + n] = gg[f * n + lane];   // A is symmetric: u = A^{-T} gbar = A^{-1} gbar
            gs[lane] = g[f * n + lane];
        }
        __builtin_amdgcn_wave_barrier();
        th_gauss_jordan(Aug, n, W, 1, rowof, lane);
        if (lane < n) {
            const int row = rowof[lane];
            const T ul = Aug[row * W + n] / Aug[row * W + lane];
            u[lane] = ul;
            gr[f * n + lane] = ul;
        }
        __builtin_amdgcn_wave_barrier();
        // Abar = -u g^T on the Toeplitz diagonals |i - j| = k (k < n) and the Hankel anti-diagonals i + j = k (k < 2n-1)
        for (int k = lane; k < 2 * n - 1; k += 64) {
            T sq = 0;
            const int lo = k - (n - 1) > 0 ? k - (n - 1) : 0, hi = k < n - 1 ? k : n - 1;
            for (int i = lo; i <= hi; ++i) sq -= u[i] * gs[k - i];
            gq[f * (2 * n - 1) + k] = sq;
            if (k < n) {
                T sp = 0;
                for (int i = 0; i + k < n; ++i) sp -= u[i] * gs[i + k] + (k > 0 ? u[i + k] * gs[i] : T(0));
                gp[f * n + k] = sp;
            }
        }
    }
}

template <typename T>
static int th_launch(bool bwd, const void* gg, const void* p, const void* q, const void* r_or_g, int64_t F, int n, void* o1,
                     void* o2, void* o3, hipStream_t st)
{
    const size_t lds = sizeof(T) * ((size_t)n * (n + 1) + 2 * kThMax) + sizeof(int) * kThMax;
    long grid = F < 256L * 16 ? (long)F : 256L * 16;
    if (!bwd)
        hipLaunchKernelGGL((th_solve_fwd_kernel<T>), dim3((unsigned)grid), dim3(64), lds, st, (const T*)p, (const T*)q,
                           (const T*)r_or_g, (long)F, n, (T*)o1);
    else
        hipLaunchKernelGGL((th_solve_bwd_kernel<T>), dim3((unsigned)grid), dim3(64), lds, st, (const T*)gg, (const T*)p,
                           (const T*)q, (const T*)r_or_g, (long)F, n, (T*)o1, (T*)o2, (T*)o3);
    return check_launch(bwd ? "th_solve_bwd" : "th_solve_fwd");
}

}  // namespace dsa

using namespace dsa;

DSA_EXPORT int dsa_thsolve_fwd(const void* p, const void* q, const void* r, int64_t F, int32_t n, int32_t dtype, void* g,
                               void* stream)
{
    DSA_REQUIRE(n >= 1 && n <= kThMax && F >= 0, "thsolve: order must be in [1, 64]");
    if (F == 0) return DSA_OK;
    if (dtype == DSA_F32) return th_launch<float>(false, nullptr, p, q, r, F, n, g, nullptr, nullptr, (hipStream_t)stream);
    if (dtype == DSA_F64) return th_launch<double>(false, nullptr, p, q, r, F, n, g, nullptr, nullptr, (hipStream_t)stream);
    return fail(DSA_ERR_UNSUPPORTED, "thsolve: unsupported dtype%s");
}

DSA_EXPORT int dsa_thsolve_bwd(const void* gg, const void* p, const void* q, const void* g, int64_t F, int32_t n,
                               int32_t dtype, void* gp, void* gq, void* gr, void* stream)
{
    DSA_REQUIRE(n >= 1 && n <= kThMax && F >= 0, "thsolve_bwd: order must be in [1, 64]");
    if (F == 0) return DSA_OK;
    if (dtype == DSA_F32) return th_launch<float>(true, gg, p, q, g, F, n, gp, gq, gr, (hipStream_t)stream);
    if (dtype == DSA_F64) return th_launch<double>(true, gg, p, q, g, F, n, gp, gq, gr, (hipStream_t)stream);
    return fail(DSA_ERR_UNSUPPORTED, "thsolve_bwd: unsupported dtype%s");
}
