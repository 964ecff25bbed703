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
#include "th_solve_reg.h"

#include <cstdlib>

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

template <typename T, int NMAX = 0>   // NMAX > 0: register version (n <= NMAX)
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
        if (NMAX > 0) {
            T* ps = Aug;          // [n]
            T* qs = Aug + n;      // [2n - 1]
            if (lane < n) ps[lane] = p[f * n + lane];
            for (int i = lane; i < 2 * n - 1; i += 64) qs[i] = q[f * (2 * n - 1) + i];
            const T rhs = lane < n ? r[f * n + lane] : T(0);
            __builtin_amdgcn_wave_barrier();
            int col;
            T sol;
            th_solve_reg<T, (NMAX > 0 ? NMAX : 1)>(ps, qs, rhs, n, lane, col, sol);
            if (lane < n) g[f * n + col] = sol;
            continue;
        }
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

template <typename T, int NMAX = 0>
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
        if (NMAX > 0) {
            T* ps = Aug;
            T* qs = Aug + n;
            if (lane < n) {
                ps[lane] = p[f * n + lane];
                gs[lane] = g[f * n + lane];
            }
            for (int i = lane; i < 2 * n - 1; i += 64) qs[i] = q[f * (2 * n - 1) + i];
            const T rhs = lane < n ? gg[f * n + lane] : T(0);   // A is symmetric: u = A^{-T} gbar = A^{-1} gbar
            __builtin_amdgcn_wave_barrier();
            int col;
            T sol;
            th_solve_reg<T, (NMAX > 0 ? NMAX : 1)>(ps, qs, rhs, n, lane, col, sol);
            if (lane < n) {
                u[col] = sol;
                gr[f * n + col] = sol;
            }
        } else {
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

// Cotangents of the Toeplitz column p and the Hankel sequence q from u = A^{-1} gbar and the forward's solution g:
// Abar = -u g^T summed along the diagonals |i - j| = k and the anti-diagonals i + j = k.  64 threads per system.
__global__ __launch_bounds__(256) void th_bwd_sums_kernel(const float* __restrict__ u, const float* __restrict__ g, long F, int n,
                                                         float* __restrict__ gp, float* __restrict__ gq)
{
    __shared__ float us[4][64], gs[4][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long f = (long)blockIdx.x * 4 + w;
    const bool ok = f < F;
    us[w][lane] = ok && lane < n ? u[f * n + lane] : 0.f;
    gs[w][lane] = ok && lane < n ? g[f * n + lane] : 0.f;
    __syncthreads();
    if (!ok) return;
    for (int k = lane; k < 2 * n - 1; k += 64) {
        float sq = 0.f;
        const int lo = k - (n - 1) > 0 ? k - (n - 1) : 0, hi = k < n - 1 ? k : n - 1;
        for (int i = lo; i <= hi; ++i) sq -= us[w][i] * gs[w][k - i];
        gq[f * (2 * n - 1) + k] = sq;
        if (k < n) {
            float sp = 0.f;
            for (int i = 0; i + k < n; ++i) sp -= us[w][i] * gs[w][i + k] + (k > 0 ? us[w][i + k] * gs[w][i] : 0.f);
            gp[f * n + k] = sp;
        }
    }
}

template <typename T>
static int th_launch(bool bwd, const void* gg, const void* p, const void* q, const void* r_or_g, int64_t F, int n, void* o1,
                     void* o2, void* o3, hipStream_t st)
{
    const size_t lds = sizeof(T) * ((size_t)n * (n + 1) + 2 * kThMax) + sizeof(int) * kThMax;
    long grid = F < 256L * 16 ? (long)F : 256L * 16;
    static const bool lds_only = [] {
        const char* e = getenv("DSA_THSOLVE_LDS");
        return e && atoi(e) != 0;
    }();
#define DSA_TH_LAUNCH(NM)                                                                                                  \
    do {                                                                                                                   \
        if (!bwd)                                                                                                          \
            hipLaunchKernelGGL((th_solve_fwd_kernel<T, NM>), dim3((unsigned)grid), dim3(64), lds, st, (const T*)p, (const T*)q, \
                               (const T*)r_or_g, (long)F, n, (T*)o1);                                                      \
        else                                                                                                               \
            hipLaunchKernelGGL((th_solve_bwd_kernel<T, NM>), dim3((unsigned)grid), dim3(64), lds, st, (const T*)gg, (const T*)p, \
                               (const T*)q, (const T*)r_or_g, (long)F, n, (T*)o1, (T*)o2, (T*)o3);                         \
    } while (0)
    if (!lds_only && n <= 24) DSA_TH_LAUNCH(24);
    else if (!lds_only && n <= 32) DSA_TH_LAUNCH(32);
    else if (!lds_only && n <= 48) DSA_TH_LAUNCH(48);   // the orders of the 48 kHz set-ups (34 .. 60): rows in registers too
    else if (!lds_only && n <= 64) DSA_TH_LAUNCH(64);
    else DSA_TH_LAUNCH(0);
#undef DSA_TH_LAUNCH
    return check_launch(bwd ? "th_solve_bwd" : "th_solve_fwd");
}

// ------------------------------------------------------------------------------------------------------------------
// Time-variant all-zero filter: AllZeroDigitalFilter._forward_efficient, zerodf.py:207-243 -- the
// FIR core of the multi-stage / single-stage MLSA filter (mglsadf.py:254-527).
//   y[t] = sum_{k=0}^{M} h_t[k] x[t - k + z0],   h_t = (1 - w) b[n] + w b[min(n + 1, N - 1)],  n = t / P, w = (t % P) / P
// (x is zero outside [0, T); z0 = zeroth_index: taps k < z0 look ahead).  ignore_gain divides by the interpolated b[.][0]
// (z0 < M) or b[.][M] (z0 = M).  One workgroup per frame: both coefficient rows and the frame's stretch of x in LDS.
template <typename T>
__global__ __launch_bounds__(256) void zerodf_fwd_kernel(const T* __restrict__ x, const T* __restrict__ b, long Tlen, long N,
                                                         int M, int P, int z0, int ignore_gain, T* __restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* b0 = reinterpret_cast<T*>(smem_raw);   // [M + 1]
    T* b1 = b0 + (M + 1);                     // [M + 1]
    T* xs = b1 + (M + 1);                     // [P + M]: x[t0 - M + z0 .. t0 + P - 1 + z0]
    const long f = blockIdx.x;                // flattened (utterance, frame)
    const long u = f / N, n = f - u * N;
    const long n1 = n + 1 < N ? n + 1 : N - 1;
    const T* br0 = b + (u * N + n) * (M + 1);
    const T* br1 = b + (u * N + n1) * (M + 1);
    for (int k = threadIdx.x; k <= M; k += blockDim.x) {
        b0[k] = br0[k];
        b1[k] = br1[k];
    }
    const long t0 = n * P;
    const T* xu = x + u * Tlen;
    for (int i = threadIdx.x; i < P + M; i += blockDim.x) {
        const long s = t0 - M + z0 + i;
        xs[i] = (s >= 0 && s < Tlen) ? xu[s] : T(0);
    }
    __syncthreads();
    const int gk = z0 == M ? M : 0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const T w = (T)i / (T)P;
        T a0 = 0, a1 = 0;
        // x[t - k + z0] = xs[i + M - k]
        for (int k = 0; k <= M; ++k) {
            const T xv = xs[i + M - k];
            a0 += b0[k] * xv;
            a1 += b1[k] * xv;
        }
        T v = a0 + w * (a1 - a0);             // torch.lerp(y1, y2, ramp)
        if (ignore_gain) v /= b0[gk] + w * (b1[gk] - b0[gk]);
        y[u * Tlen + t0 + i] = v;
    }
}

// gx[s] = sum_k gyn[t] h_t[k], t = s - z0 + k (gather: deterministic); gyn = gy / gain when ignore_gain
template <typename T>
__global__ __launch_bounds__(256) void zerodf_bwd_x_kernel(const T* __restrict__ gy, const T* __restrict__ b, long B, long Tlen,
                                                           long N, int M, int P, int z0, int ignore_gain, T* __restrict__ gx)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long u = idx / Tlen, s = idx - u * Tlen;
    if (u >= B) return;
    const int gk = z0 == M ? M : 0;
    T acc = 0;
    for (int k = 0; k <= M; ++k) {
        const long t = s - z0 + k;
        if (t < 0 || t >= Tlen) continue;
        const long n = t / P;
        const long n1 = n + 1 < N ? n + 1 : N - 1;
        const T w = (T)(t - n * P) / (T)P;
        const T* r0 = b + (u * N + n) * (M + 1);
        const T* r1 = b + (u * N + n1) * (M + 1);
        T g = gy[u * Tlen + t];
        if (ignore_gain) g /= r0[gk] + w * (r1[gk] - r0[gk]);
        acc += g * (r0[k] + w * (r1[k] - r0[k]));
    }
    gx[idx] = acc;
}

// gb[n][k] = sum over the samples of frame n (weight 1 - w) and of frame n - 1 (weight w; the last frame also takes its
// own w part) of gyn[t] x[t - k + z0]; with ignore_gain the gain tap additionally receives -gy y / gain.
template <typename T>
__global__ __launch_bounds__(256) void zerodf_bwd_b_kernel(const T* __restrict__ gy, const T* __restrict__ x, const T* __restrict__ b,
                                                           const T* __restrict__ y, long Tlen, long N, int M, int P, int z0,
                                                           int ignore_gain, T* __restrict__ gb)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* gs = reinterpret_cast<T*>(smem_raw);   // [2P]: normalised cotangent times the frame weight, frames n - 1 and n
    T* xs = gs + 2 * P;                       // [2P + M]
    T* red = xs + 2 * P + M;                  // [blockDim.x] reduction scratch for the gain tap
    const long f = blockIdx.x;
    const long u = f / N, n = f - u * N;
    const int gk = z0 == M ? M : 0;
    const long tbase = (n - 1) * P;           // first sample of frame n - 1
    T gain_part = 0;
    for (int i = threadIdx.x; i < 2 * P; i += blockDim.x) {
        const long t = tbase + i;
        T v = 0;
        if (t >= 0 && t < Tlen) {
            const long nt = t / P;            // n - 1 or n
            const long nt1 = nt + 1 < N ? nt + 1 : N - 1;
            const T w = (T)(t - nt * P) / (T)P;
            T wt = 0;                         // weight with which b[n] enters h_t
            if (nt == n) wt += T(1) - w;
            if (nt1 == n) wt += w;
            T g = gy[u * Tlen + t];
            if (ignore_gain) {
                const T* r0 = b + (u * N + nt) * (M + 1);
                const T* r1 = b + (u * N + nt1) * (M + 1);
                const T gain = r0[gk] + w * (r1[gk] - r0[gk]);
                g /= gain;
                gain_part -= wt * g * y[u * Tlen + t];   // d/d gain of (u / gain) = -y / gain, gain = sum wt b[.][gk]
            }
            v = wt * g;
        }
        gs[i] = v;
    }
    const T* xu = x + u * Tlen;
    for (int i = threadIdx.x; i < 2 * P + M; i += blockDim.x) {
        const long s = tbase - M + z0 + i;
        xs[i] = (s >= 0 && s < Tlen) ? xu[s] : T(0);
    }
    red[threadIdx.x] = gain_part;
    __syncthreads();
    for (int k = threadIdx.x; k <= M; k += blockDim.x) {
        T acc = 0;
        for (int i = 0; i < 2 * P; ++i) acc += gs[i] * xs[i + M - k];
        if (ignore_gain && k == gk)
            for (int q = 0; q < (int)blockDim.x; ++q) acc += red[q];
        gb[(u * N + n) * (M + 1) + k] = acc;
    }
}

// Long filters (M >= 64: the 200-tap cepstra of the multi-stage MLSA filter, the 2000-tap impulse responses of the
// single-stage one) with P <= 128: the taps are dealt to 8 slices of 32 threads, a thread keeps up to four output samples
// (i = l, l + 32, ..) in registers -- two coefficient reads feed eight multiply-adds instead of two, and all 256 threads work
// where the kernel above keeps P of them busy -- and the slices' partial sums meet in LDS (fixed order: deterministic).
template <typename T>
__global__ __launch_bounds__(256) void zerodf_fwd_sliced_kernel(const T* __restrict__ x, const T* __restrict__ b, long Tlen, long N,
                                                                int M, int P, int z0, int ignore_gain, T* __restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* b0 = reinterpret_cast<T*>(smem_raw);   // [M + 1]
    T* b1 = b0 + (M + 1);                     // [M + 1]
    T* xs = b1 + (M + 1);                     // [128 + M]: x[t0 - M + z0 ..], zero beyond the frame's stretch
    T* part = xs + (128 + M);                 // [8][2][128]
    const long f = blockIdx.x;
    const long u = f / N, n = f - u * N;
    const long n1 = n + 1 < N ? n + 1 : N - 1;
    const T* br0 = b + (u * N + n) * (M + 1);
    const T* br1 = b + (u * N + n1) * (M + 1);
    for (int k = threadIdx.x; k <= M; k += blockDim.x) {
        b0[k] = br0[k];
        b1[k] = br1[k];
    }
    const long t0 = n * P;
    const T* xu = x + u * Tlen;
    for (int i = threadIdx.x; i < 128 + M; i += blockDim.x) {
        const long s = t0 - M + z0 + i;
        xs[i] = (i < P + M && s >= 0 && s < Tlen) ? xu[s] : T(0);
    }
    __syncthreads();
    const int g = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int per = (M + 8) / 8;              // ceil((M + 1) / 8)
    const int k0 = g * per, k1 = (k0 + per < M + 1) ? k0 + per : M + 1;
    T a0[4] = {T(0), T(0), T(0), T(0)}, a1[4] = {T(0), T(0), T(0), T(0)};
    for (int k = k0; k < k1; ++k) {
        const T c0 = b0[k], c1 = b1[k];
        const T* xp = xs + (M - k) + l;       // x[t - k + z0] = xs[i + M - k]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T xv = xp[32 * j];
            a0[j] += c0 * xv;
            a1[j] += c1 * xv;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        part[(g * 2 + 0) * 128 + l + 32 * j] = a0[j];
        part[(g * 2 + 1) * 128 + l + 32 * j] = a1[j];
    }
    __syncthreads();
    const int gk = z0 == M ? M : 0;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        T s0 = 0, s1 = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            s0 += part[(q * 2 + 0) * 128 + i];
            s1 += part[(q * 2 + 1) * 128 + i];
        }
        const T w = (T)i / (T)P;
        T v = s0 + w * (s1 - s0);             // torch.lerp(y1, y2, ramp)
        if (ignore_gain) v /= b0[gk] + w * (b1[gk] - b0[gk]);
        y[u * Tlen + t0 + i] = v;
    }
}

// The same filters with the taps AND the samples blocked by four: a thread owns four consecutive output samples and a
// contiguous range of 4-tap blocks; with the taps stored reversed (br[kk] = b[M - kk]) a block needs the eight samples
// xs[4 (l + m) .. + 7] -- two aligned 16-byte reads, one of them carried over from the previous block -- and two
// 16-byte coefficient reads (broadcasts): 3 LDS reads per 32 multiply-adds (the kernel above: 6 per 8, which bound it).
// The tap ranges of the 256 / ceil(P / 4) thread groups meet in LDS in a fixed order (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void zerodf_fwd_blocked_kernel(const T* __restrict__ x, const T* __restrict__ b, long Tlen, long N,
                                                                 int M, int P, int z0, int ignore_gain, T* __restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int NB = (M + 4) / 4;               // 4-tap blocks: ceil((M + 1) / 4)
    const int nt = (P + 3) / 4;               // threads per group (four samples each)
    const int G = 256 / nt;                   // tap-range groups
    const int PP = nt * 4;
    T* br0 = reinterpret_cast<T*>(smem_raw);  // [4 NB] reversed taps of frame n, zero-padded
    T* br1 = br0 + 4 * NB;                    // [4 NB] ... of frame n + 1
    T* xs = br1 + 4 * NB;                     // [PP + 4 NB + 4]: x[t0 - M + z0 ..], zero beyond the frame's stretch
    T* part = xs + (PP + 4 * NB + 4);         // [G][2][PP]
    const long f = blockIdx.x;
    const long u = f / N, n = f - u * N;
    const long n1 = n + 1 < N ? n + 1 : N - 1;
    const T* r0 = b + (u * N + n) * (M + 1);
    const T* r1 = b + (u * N + n1) * (M + 1);
    // (four independent loads per round: a load -> store loop waits out one round trip to memory per element)
    const long t0 = n * P;
    const T* xu = x + u * Tlen;
    for (int kb = threadIdx.x; kb < 4 * NB; kb += 4 * blockDim.x) {
        T v0[4], v1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kk = kb + q * blockDim.x;
            const bool ok = kk <= M;
            v0[q] = ok ? r0[M - (ok ? kk : M)] : T(0);
            v1[q] = ok ? r1[M - (ok ? kk : M)] : T(0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kk = kb + q * blockDim.x;
            if (kk < 4 * NB) {
                br0[kk] = v0[q];
                br1[kk] = v1[q];
            }
        }
    }
    for (int ib = threadIdx.x; ib < PP + 4 * NB + 4; ib += 4 * blockDim.x) {
        T v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = ib + q * blockDim.x;
            const long sidx = t0 - M + z0 + i;
            const bool ok = i < P + M && sidx >= 0 && sidx < Tlen;
            v[q] = ok ? xu[ok ? sidx : 0] : T(0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = ib + q * blockDim.x;
            if (i < PP + 4 * NB + 4) xs[i] = v[q];
        }
    }
    __syncthreads();
    const int g = threadIdx.x / nt, l = threadIdx.x - g * nt;
    if (g < G) {
        const int per = (NB + G - 1) / G;
        const int m0 = g * per, m1 = (m0 + per < NB) ? m0 + per : NB;
        T a0[4] = {T(0), T(0), T(0), T(0)}, a1[4] = {T(0), T(0), T(0), T(0)};
        T wv[8];
        if (m0 < m1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[q] = xs[4 * (l + m0) + q];
        }
        for (int m = m0; m < m1; ++m) {
            T c0[4], c1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                wv[4 + q] = xs[4 * (l + m + 1) + q];
                c0[q] = br0[4 * m + q];
                c1[q] = br1[4 * m + q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a0[q] += c0[r] * wv[q + r];
                    a1[q] += c1[r] * wv[q + r];
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[q] = wv[4 + q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            part[(g * 2 + 0) * PP + 4 * l + q] = a0[q];
            part[(g * 2 + 1) * PP + 4 * l + q] = a1[q];
        }
    }
    __syncthreads();
    const int gk = z0 == M ? M : 0;
    const T g0 = r0[gk], g1 = r1[gk];
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        T s0 = 0, s1 = 0;
        for (int q = 0; q < G; ++q) {
            s0 += part[(q * 2 + 0) * PP + i];
            s1 += part[(q * 2 + 1) * PP + i];
        }
        const T w = (T)i / (T)P;
        T v = s0 + w * (s1 - s0);             // torch.lerp(y1, y2, ramp)
        if (ignore_gain) v /= g0 + w * (g1 - g0);
        y[u * Tlen + t0 + i] = v;
    }
}

// Round 3: several frames per workgroup, every tap of a sample block in ONE thread, float32 on packed multiply-adds.
// The blocked kernel above spends most of a launch around its inner loop (one frame per workgroup: two barriers, the
// partial sums of twelve tap ranges through LDS, ~5 blocks of taps per thread).  Here a workgroup takes `nf` consecutive
// frames of one utterance: a thread owns four consecutive output samples of one frame and (G = 1) all of its taps, so the
// sums stay in registers; the rows of frame n and n + 1 are stored INTERLEAVED in LDS -- (b_n[k], b_n+1[k]) as one 8-byte
// pair -- so that the two filters of the interpolation are the two halves of one v_pk_fma_f32 whose other factor is the
// sample, broadcast by op_sel: 16 packed instructions per 4 taps x 4 samples x 2 rows instead of 32 v_fma_f32 (the packed
// form is the only one that issues two float32 multiply-adds per lane in 4 cycles: DESIGN 3.2).  Long filters (the
// 2000-tap impulse responses of the single-stage form) split the taps over G groups of threads that meet in LDS in a fixed
// order.  Optional epilogue for the Taylor stages of the multi-stage form: y = scale * filter(x), ysum = acc + y.
// NaN containment: taps beyond M (padding of the last block of four) are skipped, not multiplied by zero.
typedef float zd_v2f __attribute__((ext_vector_type(2)));
typedef float zd_v4f __attribute__((ext_vector_type(4)));
typedef float zd_v4f_u __attribute__((ext_vector_type(4), aligned(4)));
typedef double zd_v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void zd_fma_lo(zd_v2f& acc, zd_v2f c, zd_v2f w)   // acc += c * w.x (both halves)
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(c), "v"(w));
}
__device__ __forceinline__ void zd_fma_hi(zd_v2f& acc, zd_v2f c, zd_v2f w)   // acc += c * w.y (both halves)
{
    // the odd sample as the LOW half of its own pair (a move the compiler shares between the uses of a ring slot), then the
    // low-half broadcast of zd_fma_lo: the one-instruction form "op_sel:[0,1,0] op_sel_hi:[1,1,1]" has a set op_sel bit -- its
    // low result reads a high source half -- and no shipped kernel executes that class (pk_math.h, DSA_PK_CROSSED)
    const zd_v2f wh = __builtin_shufflevector(w, w, 1, 1);
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(c), "v"(wh));
}
// two adjacent pairs from a 16-byte aligned LDS address (float: one ds_read_b128); `both` false: only the first is wanted
__device__ __forceinline__ void zd_load2(const zd_v2f* p, zd_v2f& a, zd_v2f& b, bool both)
{
    if (both) {
        const zd_v4f v = *reinterpret_cast<const zd_v4f*>(p);
        a = zd_v2f{v.x, v.y};
        b = zd_v2f{v.z, v.w};
    } else {
        a = p[0];
    }
}
__device__ __forceinline__ void zd_load2(const zd_v2d* p, zd_v2d& a, zd_v2d& b, bool both)
{
    a = p[0];
    if (both) b = p[1];
}
__device__ __forceinline__ void zd_fma2(zd_v2f& acc, zd_v2f c, zd_v2f w)     // acc += c * w, half by half
{
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(c), "v"(w));
}
__device__ __forceinline__ void zd_fma2(zd_v2d& acc, zd_v2d c, zd_v2d w) { acc += c * w; }
__device__ __forceinline__ void zd_fma_lo(zd_v2d& acc, zd_v2d c, zd_v2d w) { acc += c * w.x; }
__device__ __forceinline__ void zd_fma_hi(zd_v2d& acc, zd_v2d c, zd_v2d w) { acc += c * w.y; }

template <typename T, int S>
__global__ __launch_bounds__(256) DSA_PK_TARGET void zerodf_fwd_rows_kernel(const T* __restrict__ x, const T* __restrict__ b, long Tlen, long N,
                                                              int M, int P, int z0, int ignore_gain, int nf, int G, T scale,
                                                              const T* acc, T* __restrict__ y, T* ysum)
{
    using V2 = T __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int NB = (M + 4) / 4;                 // 4-tap blocks
    const int nt = P / S;                       // threads per (frame, tap group): S consecutive samples each
    V2* brp = reinterpret_cast<V2*>(smem_raw);  // [nf][4 NB]: (b[n][M - kk], b[n + 1][M - kk]), zero beyond kk = M
    T* xs = reinterpret_cast<T*>(brp + (size_t)nf * 4 * NB);   // [nf P + 4 NB + 8]: x[t0 - M + z0 ..]
    V2* part = reinterpret_cast<V2*>(xs + ((size_t)nf * P + 4 * NB + 8));   // [nf][G][P] when G > 1
    const long chunks = (N + nf - 1) / nf;
    const long u = blockIdx.x / chunks, n0 = (blockIdx.x - u * chunks) * nf;
    const int frames = (int)((N - n0 < nf) ? N - n0 : nf);
    const T* bu = b + u * N * (M + 1);
    // (all loads of a batch first, then the stores: a load -> store loop waits out one trip to memory per element)
    for (int kk = threadIdx.x; kk < 4 * NB; kk += blockDim.x) {   // a thread walks down one tap: every row is read once
        const bool tap = kk <= M;
        const T* col = bu + (M - (tap ? kk : M));
        T cv[17];
#pragma unroll
        for (int p = 0; p <= 16; ++p) {
            const long row = n0 + p < N ? n0 + p : N - 1;
            cv[p] = (tap && p <= frames) ? col[row * (M + 1)] : T(0);
        }
#pragma unroll
        for (int p = 0; p < 16; ++p)
            if (p < frames) brp[(size_t)p * 4 * NB + kk] = V2{cv[p], cv[p + 1]};
    }
    const long t0 = n0 * P;
    const T* xu = x + u * Tlen;
    const int xlen = frames * P + 4 * NB + 8;
    for (int i0 = threadIdx.x; i0 < xlen; i0 += 8 * blockDim.x) {
        T xv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const long sidx = t0 - M + z0 + i0 + q * (int)blockDim.x;
            xv[q] = (i0 + q * (int)blockDim.x < xlen && sidx >= 0 && sidx < Tlen) ? xu[sidx] : T(0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (i0 + q * (int)blockDim.x < xlen) xs[i0 + q * (int)blockDim.x] = xv[q];
    }
    __syncthreads();
    const int grp = threadIdx.x / nt, l = threadIdx.x - grp * nt;   // grp = fr * G + g
    const int fr = grp / G, g = grp - fr * G;
    const bool active = fr < frames;
    V2 a[S];
#pragma unroll
    for (int q = 0; q < S; ++q) a[q] = V2{0, 0};
    if (active) {
        // (host: G divides the number of full blocks, so the loop count is the same for every thread of the launch)
        const int rem = (M + 1) & 3;                       // taps in the last block when it is a partial one
        const int per = (rem ? NB - 1 : NB) / G;
        const int m0 = g * per, m_full = m0 + per;
        const int m1 = (rem != 0 && g == G - 1) ? m_full + 1 : m_full;
        const V2* cp = brp + (size_t)fr * 4 * NB;
        const T* xf = xs + fr * P + S * l;
        // A block of four taps on S samples reads the S + 4 samples xf[4 m .. 4 m + S + 3]: NP = (S + 4) / 2 pairs kept in a
        // ring of NP registers pairs that advances by two pairs per block -- after NP / 2 blocks (a trip, unrolled) every pair
        // is back in its slot, so nothing is copied; per block two 16-byte tap reads (broadcasts) and one 16-byte sample read
        // feed 4 S packed multiply-adds (S = 8: the LDS pipe, which the S = 4 form loads as much as the vector unit, idles).
        constexpr int NP = (S + 4) / 2, TRIP = NP / 2;
        V2 R[NP];
#pragma unroll
        for (int i = 0; i < NP - 2; ++i) R[i] = *reinterpret_cast<const V2*>(xf + 4 * m0 + 2 * i);
        auto block = [&](int m, int b) __attribute__((always_inline)) {   // b = (m - m0) % TRIP: the ring's phase
            V2 c[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = cp[4 * m + r];
            zd_load2(reinterpret_cast<const V2*>(xf + 4 * m + 2 * (NP - 2)), R[(2 * b + NP - 2) % NP], R[(2 * b + NP - 1) % NP], true);
#pragma unroll
            for (int q = 0; q < S; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if ((q + r) & 1) zd_fma_hi(a[q], c[r], R[(2 * b + ((q + r) >> 1)) % NP]);
                    else zd_fma_lo(a[q], c[r], R[(2 * b + ((q + r) >> 1)) % NP]);
                }
        };
        int j = 0;
        for (; j + TRIP <= per; j += TRIP) {   // (uniform trip count: a scalar loop)
#pragma unroll
            for (int b = 0; b < TRIP; ++b) block(m0 + j + b, b);
        }
        int tail_b = 0;   // blocks left after the last whole trip (the ring's phase restarts at 0 there)
#pragma unroll
        for (int b = 0; b < TRIP - 1; ++b)
            if (j + b < per) {
                block(m0 + j + b, b);
                tail_b = b + 1;
            }
        if (m_full < m1) {   // the partial last block: only its real taps
            const int m = m_full;
            // bring the ring back to phase 0 (at most once per thread)
            V2 Wn[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) Wn[i] = tail_b == 0 ? R[i] : (tail_b == 1 ? R[(i + 2) % NP] : R[(i + 4) % NP]);
            Wn[NP - 2] = *reinterpret_cast<const V2*>(xf + 4 * m + 2 * (NP - 2));
#pragma unroll
            for (int r = 0; r < 3; ++r)
                if (r < rem) {
                    const V2 cr = cp[4 * m + r];
#pragma unroll
                    for (int q = 0; q < S; ++q) {
                        if ((q + r) & 1) zd_fma_hi(a[q], cr, Wn[(q + r) >> 1]);
                        else zd_fma_lo(a[q], cr, Wn[(q + r) >> 1]);
                    }
                }
        }
    }
    if (G > 1) {
        if (active) {
#pragma unroll
            for (int q = 0; q < S; ++q) part[((size_t)fr * G + g) * P + S * l + q] = a[q];
        }
        __syncthreads();
        if (active && g == 0) {
#pragma unroll
            for (int q = 0; q < S; ++q) {
                V2 sacc = part[((size_t)fr * G) * P + S * l + q];
                for (int gg = 1; gg < G; ++gg) sacc += part[((size_t)fr * G + gg) * P + S * l + q];
                a[q] = sacc;
            }
        }
    }
    if (active && g == 0) {
        const int gk = z0 == M ? M : 0;
        const V2 gain = brp[(size_t)fr * 4 * NB + (M - gk)];
        const long o = u * Tlen + t0 + (long)fr * P + S * l;
        T v[S];
#pragma unroll
        for (int q = 0; q < S; ++q) {
            const T wt = (T)(S * l + q) / (T)P;
            T r = a[q].x + wt * (a[q].y - a[q].x);          // torch.lerp(y1, y2, ramp)
            if (ignore_gain) r /= gain.x + wt * (gain.y - gain.x);
            v[q] = r * scale;
        }
        if (y) {
#pragma unroll
            for (int q = 0; q < S; ++q) y[o + q] = v[q];
        }
        if (ysum) {
#pragma unroll
            for (int q = 0; q < S; ++q) ysum[o + q] = acc[o + q] + v[q];
        }
    }
}

// nf frames x G tap groups of P / S threads per 256-thread workgroup within 64 KB of LDS; false: shape not covered
static bool zerodf_rows_plan(int M, int P, size_t elt, int& S, int& nf, int& G, size_t& lds)
{
    if (P % 4 != 0 || P / 4 > 256 || M < 16) return false;
    S = 4;   // (S = 8 -- half the LDS reads per multiply-add, 160 of 256 threads busy at P = 80 -- measured the same: 1.50 vs 1.46 ms)
    const int NB = (M + 4) / 4, nt = P / S, groups = 256 / nt;
    const int nb_full = ((M + 1) & 3) ? NB - 1 : NB;
    for (nf = groups < 16 ? groups : 16; nf >= 1; --nf) {
        G = groups / nf;
        while (G > 1 && nb_full % G != 0) --G;   // equal tap ranges: one loop count for the whole launch
        lds = (size_t)nf * 4 * NB * 2 * elt + ((size_t)nf * P + 4 * NB + 8) * elt + (G > 1 ? (size_t)nf * G * P * 2 * elt : 0);
        lds = (lds + 15) & ~(size_t)15;
        if (lds <= 64 * 1024) return true;
    }
    return false;
}

template <typename T>
static int zerodf_launch_fwd(const void* x, const void* b, int64_t B, int64_t Tlen, int64_t N, int M, int P, int z0, int ig,
                             void* y, hipStream_t st, double scale = 1.0, const void* acc = nullptr, void* ysum = nullptr)
{
    static const int variant = [] { const char* e = getenv("DSA_ZERODF"); return e ? atoi(e) : 0; }();   // 1: round-2 kernels (A/B)
    {
        int S, nf, G;
        size_t lds_r;
        if ((variant == 0 || ysum || scale != 1.0) && zerodf_rows_plan(M, P, sizeof(T), S, nf, G, lds_r)) {
            const long chunks = (N + nf - 1) / nf;
            // (the kernel is written for S = 4 or 8 samples per thread; 8 measured the same at P = 80 and is not instantiated)
            hipLaunchKernelGGL((zerodf_fwd_rows_kernel<T, 4>), dim3((unsigned)(B * chunks)), dim3(256), lds_r, st, (const T*)x,
                               (const T*)b, (long)Tlen, (long)N, M, P, z0, ig, nf, G, (T)scale, (const T*)acc, (T*)y, (T*)ysum);
            return check_launch("zerodf_rows_fwd");
        }
        if (ysum || scale != 1.0) return fail(DSA_ERR_UNSUPPORTED, "zerodf: the scaled / accumulating form needs P % 4 == 0 and M >= 16%s");
    }
    {   // long filters: taps and samples blocked by four (DSA_ZERODF_SLICED=1 keeps the older sliced kernel: A/B)
        static const bool sliced_only = [] {
            const char* e = getenv("DSA_ZERODF_SLICED");
            return e && atoi(e) != 0;
        }();
        const int NB = (M + 4) / 4, nt = (P + 3) / 4;
        const size_t lds_b = sizeof(T) * ((size_t)8 * NB + (size_t)(4 * nt + 4 * NB + 4) + (size_t)(256 / (nt > 0 ? nt : 1)) * 2 * 4 * nt);
        if (!sliced_only && M >= 64 && P >= 4 && P <= 128 && lds_b <= 64 * 1024) {
            hipLaunchKernelGGL((zerodf_fwd_blocked_kernel<T>), dim3((unsigned)(B * N)), dim3(256), lds_b, st, (const T*)x, (const T*)b,
                               (long)Tlen, (long)N, M, P, z0, ig, (T*)y);
            return check_launch("zerodf_blocked_fwd");
        }
    }
    const size_t lds_s = sizeof(T) * (2 * (size_t)(M + 1) + 128 + M + 8 * 2 * 128);
    if (M >= 64 && P <= 128 && lds_s <= 64 * 1024) {
        hipLaunchKernelGGL((zerodf_fwd_sliced_kernel<T>), dim3((unsigned)(B * N)), dim3(256), lds_s, st, (const T*)x, (const T*)b,
                           (long)Tlen, (long)N, M, P, z0, ig, (T*)y);
        return check_launch("zerodf_sliced_fwd");
    }
    const size_t lds = sizeof(T) * (2 * (size_t)(M + 1) + P + M);
    if (lds > 64 * 1024) return fail(DSA_ERR_UNSUPPORTED, "zerodf: filter too long for LDS%s");
    hipLaunchKernelGGL((zerodf_fwd_kernel<T>), dim3((unsigned)(B * N)), dim3(P >= 192 ? 256 : (P >= 96 ? 128 : 64)), lds, st,
                       (const T*)x, (const T*)b, (long)Tlen, (long)N, M, P, z0, ig, (T*)y);
    return check_launch("zerodf_fwd");
}

// Round 3: the backward of the time-variant FIR on the forward's pattern (rows of frames n and n + 1 interleaved as pairs,
// several frames per workgroup, four consecutive samples / taps per thread).  Without ignore_gain:
//   gx[s] = sum_k gy[t] h_t[k],  t = s - z0 + k  =  sum_t (b[n(t)][k], b[n(t) + 1][k]) . ((1 - w_t) gy[t], w_t gy[t])
// -- the dot product of two pairs, i.e. ONE packed multiply-add into a pair accumulator whose halves are added at the end.
// A thread owns four consecutive s and walks t in blocks of four that never straddle a frame (P % 4 == 0; z0 is rounded up
// to a multiple of 4 by shifting the taps): block m needs the tap pairs 4 m - 3 .. 4 m + 3 of ITS frame's row (stored from
// position 3, so the window starts 16-byte aligned) and the four weighted cotangent pairs.  Lanes cross frame boundaries at
// different m, so the window is re-read every block (the kernel is bound by LDS reads, ~1.4 x the multiply-adds).
// The old kernel: a thread per sample over all taps with two row reads from memory per tap.
template <typename T, int S>
__global__ __launch_bounds__(256) DSA_PK_TARGET void zerodf_bwd_x_rows_kernel(const T* __restrict__ gy, const T* __restrict__ b, long Tlen, long N,
                                                                int M, int P, int z0, int nf, int nrows, int ldb, int accumulate,
                                                                T scale, const T* add, T* gx)
{
    // gx = (accumulate ? gx : (add ? add : 0)) + scale * (the sum): `add` / `scale` serve the Taylor stages of the multi-stage
    // MLSA filter's backward (G_{i-1} = gy + F^T G_i / i)
    // (`b` may point at a run of M + 1 taps inside rows of ldb coefficients -- long filters are handled as a sum of
    // 200-tap pieces: piece c has z0 - c KC as its (possibly negative) zeroth index and accumulates into gx)
    using V2 = T __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int dz = (4 - (((z0 % 4) + 4) & 3)) & 3, Mp = M + dz, z0p = z0 + dz;
    const int NBt = (Mp + S - 1) / 4 + 1;              // blocks of four t per block of S output samples
    const int RW = (Mp + S + 6 + 3) & ~3;              // pairs per row: tap k' at position k' + S - 1, zeros around
    const int nt = P / S;
    V2* rows = reinterpret_cast<V2*>(smem_raw);        // [nrows][RW]
    V2* up = rows + (size_t)nrows * RW;                // [nf P + 4 NBt]: ((1 - w) gy, w gy) of t = Tstart + j
    const long chunks = (N + nf - 1) / nf;
    const long u = blockIdx.x / chunks, n0 = (blockIdx.x - u * chunks) * nf;
    const int frames = (int)((N - n0 < nf) ? N - n0 : nf);
    const long Tstart = n0 * P - z0p;                  // t of up[0]
    // floor division by P for a possibly negative Tstart
    const long nlo = Tstart >= 0 ? Tstart / P : -((-Tstart + P - 1) / P);
    const int r0 = (int)(Tstart - nlo * P);            // in [0, P)
    const T* bu = b + u * N * ldb;
    for (int pos = threadIdx.x; pos < RW; pos += blockDim.x) {
        const int k = pos - (S - 1) - dz;
        const bool tap = k >= 0 && k <= M;
        T cv[25];
#pragma unroll
        for (int i = 0; i <= 24; ++i) {
            const long nn = nlo + i;
            const long row = nn < 0 ? 0 : (nn < N ? nn : N - 1);
            cv[i] = (tap && i <= nrows) ? bu[row * ldb + (tap ? k : 0)] : T(0);
        }
#pragma unroll
        for (int i = 0; i < 24; ++i)
            if (i < nrows) rows[(size_t)i * RW + pos] = V2{cv[i], cv[i + 1]};
    }
    const int ulen = frames * P + 4 * NBt;
    const T* gyu = gy + u * Tlen;
    for (int j0 = threadIdx.x; j0 < ulen; j0 += 8 * blockDim.x) {
        T gv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = j0 + q * (int)blockDim.x;
            const long t = Tstart + j;
            gv[q] = (j < ulen && t >= 0 && t < Tlen) ? gyu[t] : T(0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = j0 + q * (int)blockDim.x;
            if (j < ulen) {
                const int ph = (r0 + j) % P;
                const T w = (T)ph / (T)P;
                up[j] = V2{gv[q] - w * gv[q], w * gv[q]};
            }
        }
    }
    __syncthreads();
    const int fr = threadIdx.x / nt, l = threadIdx.x - fr * nt;
    if (fr >= frames) return;
    const int jb0 = fr * P + S * l;                    // this thread's samples are s = n0 P + jb0 + q; block m reads up[jb0 + 4 m ..]
    int ph = (r0 + jb0) % P;
    const V2* rp = rows + (size_t)((r0 + jb0) / P) * RW;
    const V2* upp = up + jb0;
    V2 a[S];
#pragma unroll
    for (int q = 0; q < S; ++q) a[q] = V2{0, 0};
    for (int m = 0; m < NBt; ++m) {                    // (uniform trip count: a scalar loop)
        // (two pairs per 16-byte aligned read: the rows and `up` start 32-byte aligned and advance by four pairs a block;
        // left as pair reads the compiler emits ds_read2_b64, which moves half as many bytes per LDS cycle as ds_read_b128)
        V2 w[S + 4], uu[4];
#pragma unroll
        for (int i = 0; i < S + 3; i += 2) zd_load2(rp + 4 * m + i, w[i], w[i + 1], i + 1 < S + 3);
#pragma unroll
        for (int i = 0; i < 4; i += 2) zd_load2(upp + 4 * m + i, uu[i], uu[i + 1], true);
#pragma unroll
        for (int q = 0; q < S; ++q)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) zd_fma2(a[q], w[jt - q + S - 1], uu[jt]);
        ph += 4;
        if (ph >= P) {
            ph -= P;
            rp += RW;
        }
    }
    T* dst = gx + u * Tlen + n0 * P + jb0;
#pragma unroll
    for (int q = 0; q < S; ++q) {
        const T base = accumulate ? dst[q] : (add ? add[u * Tlen + n0 * P + jb0 + q] : T(0));
        dst[q] = base + scale * (a[q].x + a[q].y);
    }
}

// gb[n][k] = sum over the samples i of frames n - 1 and n of gs[i] x[t - k + z0], gs = the frame weight of b[n] in h_t times gy
// (frame n: 1 - w, frame n - 1: w; the last frame also takes its own w part).  A thread owns four consecutive taps and walks
// the 2 P samples in blocks of four on a sliding window of seven x values (one aligned 16-byte read of x and one of gs per 16
// multiply-adds); 256 / ceil((M + 1) / 4) frames per workgroup.  The old kernel: a thread per tap, two LDS reads per
// multiply-add, one frame per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void zerodf_bwd_b_rows_kernel(const T* __restrict__ gy, const T* __restrict__ x, long Tlen, long N,
                                                                long BN, int M, int P, int z0, int nfw, int ldb, T scale, int accumulate,
                                                                T* __restrict__ gb)
{
    using V4 = T __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int NBk = (M + 4) / 4;                       // 4-tap blocks
    const int o = (3 - M) & 3;                         // shift that aligns the x window: (M - k0 - 3 + o) % 4 == 0
    const int XL = (2 * P + M + 8 + 3) & ~3;           // floats of x per frame
    T* gs = reinterpret_cast<T*>(smem_raw);            // [nfw][2 P]
    T* xs = gs + (size_t)nfw * 2 * P;                  // [nfw][XL]: xs[j + o] = x[(n - 1) P - M + z0 + j]
    const long f0 = (long)blockIdx.x * nfw;
    for (int fw = 0; fw < nfw; ++fw) {                 // (frame indices per frame, not per element: 64-bit divisions)
        const long f = f0 + fw;
        const bool fok = f < BN;
        const long u = fok ? f / N : 0, n = fok ? f - u * N : 0;
        const T* gyu = gy + u * Tlen;
        const T* xu = x + u * Tlen;
        for (int i = threadIdx.x; i < 2 * P; i += blockDim.x) {
            const long t = (n - 1) * P + i;
            T v = 0;
            if (fok && t >= 0) {
                // frame of t: n - 1 for i < P, n otherwise; the row b[n] enters h_t with 1 - w in its own frame, with w in the
                // frame before, and the clamped last frame takes both
                const int ph = i < P ? i : i - P;
                const T w = (T)ph / (T)P;
                const T wt = i < P ? w : ((n == N - 1) ? T(1) : T(1) - w);
                v = wt * gyu[t];
            }
            gs[(size_t)fw * 2 * P + i] = v;
        }
        for (int jj = threadIdx.x; jj < XL; jj += blockDim.x) {
            const int j = jj - o;
            const long sidx = (n - 1) * P - M + z0 + j;
            xs[(size_t)fw * XL + jj] = (fok && j >= 0 && sidx >= 0 && sidx < Tlen) ? xu[sidx] : T(0);
        }
    }
    __syncthreads();
    const int fw = threadIdx.x / NBk, kb = threadIdx.x - fw * NBk;
    const long f = f0 + fw;
    if (fw >= nfw || f >= BN) return;
    const int k0 = 4 * kb;
    // acc[q] (tap k0 + q) += gs[i + j] xs[i + j + M - k0 - q]: window xw[c] = xs[i + e + c], e = M - k0 - 3, c = j - q + 3
    const T* gp = gs + (size_t)fw * 2 * P;
    const T* xp = xs + (size_t)fw * XL + (M - k0 - 3 + o);   // 16-byte aligned
    T acc[4] = {T(0), T(0), T(0), T(0)};
    V4 lo = *reinterpret_cast<const V4*>(xp);
    for (int i = 0; i < 2 * P; i += 4) {
        const V4 hi = *reinterpret_cast<const V4*>(xp + i + 4);
        const V4 gv = *reinterpret_cast<const V4*>(gp + i);
        const T xw[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[q] += gv[j] * xw[j - q + 3];
        lo = hi;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (k0 + q <= M) gb[f * ldb + k0 + q] = (accumulate ? gb[f * ldb + k0 + q] : T(0)) + scale * acc[q];
}

template <typename T>
static int zerodf_launch_bwd(const void* gy, const void* x, const void* b, const void* y, int64_t B, int64_t Tlen, int64_t N, int M,
                             int P, int z0, int ig, void* gx, void* gb, hipStream_t st, double scale = 1.0, const void* gx_add = nullptr,
                             bool gb_accumulate = false)
{
    const bool plain = scale == 1.0 && gx_add == nullptr && !gb_accumulate;
    static const int variant = [] { const char* e = getenv("DSA_ZERODF"); return e ? atoi(e) : 0; }();   // 1: round-2 kernels (A/B)
    const bool rows_ok = variant == 0 && !ig && P % 4 == 0 && P / 4 <= 64 && M >= 16;
    if (rows_ok) {
        // filters of more than ~200 taps as a sum of pieces (the kernels keep one piece's rows of a few frames in LDS): piece c
        // = taps [c KC, c KC + Mc], zeroth index z0 - c KC; gx accumulates over the pieces, gb's columns are disjoint
        const int npieces = (M + 1 + 199) / 200;
        const int KC = (((M + 1 + npieces - 1) / npieces) + 3) & ~3;
        // feasibility of BOTH kernels for EVERY piece is decided before anything is launched (round 3 launched gx first and could
        // then find that gb's rows did not fit LDS: P = 252 / 256 in float32, P >= 124 in float64 -- a half-written backward)
        struct PiecePlan { int Mc, z0c, nf, nrows, nfw; size_t lds_x, lds_b; };
        PiecePlan plan[64];
        bool ok = npieces <= 64;
        int np_used = 0;
        constexpr int S = 4;   // (eight samples per thread -- 2/3 of the LDS reads per multiply-add, 160 of 256 threads at P = 80 -- measured slower)
        for (int c = 0; c < npieces && ok; ++c) {
            PiecePlan& pl = plan[c];
            pl.Mc = ((M + 1 - c * KC) < KC ? (M + 1 - c * KC) : KC) - 1;
            pl.z0c = z0 - c * KC;
            if (pl.Mc < 0) break;
            np_used = c + 1;
            pl.nf = pl.nrows = pl.nfw = 0;
            pl.lds_x = pl.lds_b = 0;
            if (gx) {
                const int dz = (4 - (((pl.z0c % 4) + 4) & 3)) & 3, Mp = pl.Mc + dz, NBt = (Mp + S - 1) / 4 + 1, RW = (Mp + S + 6 + 3) & ~3, nt = P / S;
                int nf = 256 / nt;
                if (nf > 16) nf = 16;
                for (; nf >= 1; --nf) {
                    pl.nrows = nf + (Mp + P - 1) / P + 2;      // frames the t range of nf output frames can touch
                    pl.lds_x = sizeof(T) * 2 * ((size_t)pl.nrows * RW + (size_t)nf * P + 4 * NBt);
                    if (pl.nrows <= 24 && pl.lds_x <= 64 * 1024) break;
                }
                if (nf < 1) { ok = false; break; }
                pl.nf = nf;
            }
            if (gb) {
                const int NBk = (pl.Mc + 4) / 4;
                const int XL = (2 * P + pl.Mc + 8 + 3) & ~3;
                int nfw = 256 / NBk;
                if (nfw > 16) nfw = 16;
                for (; nfw >= 1; --nfw) {                      // fewer frames per workgroup until their rows fit LDS
                    pl.lds_b = sizeof(T) * (size_t)nfw * (2 * P + XL);
                    if (pl.lds_b <= 64 * 1024) break;
                }
                if (nfw < 1) { ok = false; break; }
                pl.nfw = nfw;
            }
        }
        for (int c = 0; c < np_used && ok; ++c) {
            const PiecePlan& pl = plan[c];
            if (gx) {
                const long chunks = (N + pl.nf - 1) / pl.nf;
                hipLaunchKernelGGL((zerodf_bwd_x_rows_kernel<T, S>), dim3((unsigned)(B * chunks)), dim3(256), pl.lds_x, st, (const T*)gy,
                                   (const T*)b + c * KC, (long)Tlen, (long)N, pl.Mc, P, pl.z0c, pl.nf, pl.nrows, M + 1, c > 0 ? 1 : 0, (T)scale,
                                   (const T*)gx_add, (T*)gx);
            }
            if (gb) {
                hipLaunchKernelGGL((zerodf_bwd_b_rows_kernel<T>), dim3((unsigned)((B * N + pl.nfw - 1) / pl.nfw)), dim3(256), pl.lds_b, st,
                                   (const T*)gy, (const T*)x, (long)Tlen, (long)N, (long)(B * N), pl.Mc, P, pl.z0c, pl.nfw, M + 1, (T)scale,
                                   gb_accumulate ? 1 : 0, (T*)gb + c * KC);
            }
        }
        // (nothing was launched unless every piece of both kernels fits)
        if (ok) return check_launch("zerodf_rows_bwd");
    }
    if (!plain) return fail(DSA_ERR_UNSUPPORTED, "zerodf_bwd: the scaled / accumulating form needs P % 4 == 0 and M >= 16%s");
    if (gx) {
        hipLaunchKernelGGL((zerodf_bwd_x_kernel<T>), dim3((unsigned)((B * Tlen + 255) / 256)), dim3(256), 0, st, (const T*)gy,
                           (const T*)b, (long)B, (long)Tlen, (long)N, M, P, z0, ig, (T*)gx);
    }
    if (gb) {
        const size_t lds = sizeof(T) * ((size_t)2 * P + 2 * P + M + 256);
        if (lds > 64 * 1024) return fail(DSA_ERR_UNSUPPORTED, "zerodf_bwd: filter too long for LDS%s");
        hipLaunchKernelGGL((zerodf_bwd_b_kernel<T>), dim3((unsigned)(B * N)), dim3(256), lds, st, (const T*)gy, (const T*)x,
                           (const T*)b, (const T*)y, (long)Tlen, (long)N, M, P, z0, ig, (T*)gb);
    }
    return check_launch("zerodf_bwd");
}

// ---------------------------------------------------------------------------------------------
// The spectrum arithmetic of one Newton step of MelGeneralizedCepstralAnalysis (mgcep.py:199-209), gamma not in {0, -1},
// in ONE pass over the (F, K) spectra (as stock element-wise operators it is ~20 passes and dominated the step):
//   C = b1 (Cr[1:], Ci[1:])   (b[0] = 0),  X = 1 + gamma Re C,  Y = gamma Im C,  D = X^2 + Y^2,
//   pp = x D^(-1/gamma - 1),  qq = pp / D,
//   out[0] = pp   out[1] = qq (X^2 - Y^2)   out[2] = qq 2 X Y   out[3] = pp X   out[4] = pp Y      (each (F, K))
// -- the inputs of the five row products against Pr, Qr, Qi, Rr, Ri.  A thread owns one bin: its 2 M matrix entries stay
// in registers for the workgroup's tile of frames, the frames' coefficients are broadcast from LDS.
constexpr int kMsFrames = 32, kMsMaxM = 64;
template <typename T, int MT>   // MT: compile-time bound on M (register-resident matrix columns)
__global__ __launch_bounds__(320) void mgcep_spectra_kernel(const T* __restrict__ x, const T* __restrict__ b1, long F, int K, int M,
                                                            const T* __restrict__ Cr, const T* __restrict__ Ci, T gamma,
                                                            T* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) T bs[kMsFrames * MT];   // row stride MT, zero-padded: static offsets, 16-byte reads
    const long f0 = (long)blockIdx.x * kMsFrames;
    const int nf = (int)((F - f0) < kMsFrames ? (F - f0) : kMsFrames);
    for (int i = threadIdx.x; i < kMsFrames * MT; i += blockDim.x) {
        const int fi = i / MT, m = i - fi * MT;
        bs[i] = (fi < nf && m < M) ? b1[(f0 + fi) * M + m] : T(0);
    }
    __syncthreads();
    const T ex = T(-1) / gamma - T(1);
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        T cr[MT], ci[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            cr[m] = m < M ? Cr[(long)(m + 1) * K + k] : T(0);
            ci[m] = m < M ? Ci[(long)(m + 1) * K + k] : T(0);
        }
        for (int fb = 0; fb < nf; fb += 8) {
        // (the eight spectrum values of a round are fetched together, ahead of the arithmetic: one round trip, not eight)
        T xv8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) xv8[q] = x[(f0 + (fb + q < nf ? fb + q : nf - 1)) * K + k];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int fi = fb + q;
            if (fi >= nf) break;
            T re = 0, im = 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const T b = bs[fi * MT + m];
                re += b * cr[m];
                im += b * ci[m];
            }
            const T X = T(1) + gamma * re, Y = gamma * im;
            const T XX = X * X, YY = Y * Y, D = XX + YY;
            T dp;
            if constexpr (sizeof(T) == 4) dp = __builtin_amdgcn_exp2f(ex * __builtin_amdgcn_logf(D));   // D > 0; 1 ulp each
            else dp = dsa_pow(D, ex);
            const T pp = xv8[q] * dp;
            const T qq = pp / D;
            const long o = (f0 + fi) * K + k, S = F * (long)K;
            out[o] = pp;
            out[S + o] = qq * (XX - YY);
            out[2 * S + o] = qq * (T(2) * X * Y);
            out[3 * S + o] = pp * X;
            out[4 * S + o] = pp * Y;
        }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// One Newton step's spectrum arithmetic AND its five row products in ONE launch (float32, fft_length 512, cep_order <= 24):
//   (re, im) = b1 (Cr, Ci)            first chain: 24 coefficients -> 257 bins, real and imaginary part   (mgcep.py:191-193, 199-201)
//   X = 1 + g re, Y = g im, D = X^2 + Y^2, pp = x D^(-1/g - 1), qq = pp / D                                 (mgcep.py:202-209)
//   pt = pp Pr,  qt = (1 + g) (qq (X^2 - Y^2) Qr + qq 2XY Qi),  r = pp X Rr + pp Y Ri                         (mgcep.py:212-220)
// on v_mfma_f32_16x16x4_f32 with the FRAMES as the N dimension, as in the mel-cepstral kernels: the first chain's result comes out
// with lane (n, g) register r holding bin 16 mt + 4 g + r of frame n -- exactly a B operand of the second chain if its k-steps are
// enumerated as (mt, r) with k-slot g <-> bin 16 mt + 4 g + r, so the five spectra feed the second chain from registers and never
// exist in memory (round 2: dsa_mgcep_spectra wrote them, 263 MB per step at 51 200 frames, and five launches of the matrix-core
// row product read them back: 0.14 + 5 x 0.05 ms per step).  One wave = 16 frames; the four waves of a workgroup share the
// operand images of a 16-bin tile through LDS (15 KB per tile, double-buffered, one barrier per tile).  Float32 products with
// float32 accumulation: nothing given up.  Bins 256 .. 271 are a seventeenth tile whose images are zero past bin 256.
// `images` (built by the caller once per configuration, tables.mgcep_step_images): per bin tile mt = 0 .. 16
//   [2 (Cr, Ci)][6 ks][64 l]      A of the first chain:  C[1 + 4 ks + (l >> 4)][16 mt + (l & 15)]                    (768 floats)
//   [12 c][64 l][4 r]             A of the second chain: W_c[16 mt + 4 (l >> 4) + r][16 tile_c + (l & 15)]           (3072 floats)
//   chains c: 0-1 Pr (input pp), 2-4 Qr (qq (X^2 - Y^2)), 5-7 Qi (qq 2XY), 8-9 Rr (pp X), 10-11 Ri (pp Y); 16-column tiles of each matrix.
// ---------------------------------------------------------------------------------------------------------------------------
typedef float ms_f4 __attribute__((ext_vector_type(4)));
typedef float ms_f4u __attribute__((ext_vector_type(4), aligned(4)));   // rows of 257 floats: 4-byte aligned only
constexpr int kMsTileFloats = 768 + 3072, kMsTiles = 17;
__global__ __launch_bounds__(256) void mgcep_step_kernel(const float* __restrict__ x, const float* __restrict__ b1, long F, int M, float gamma,
                                                        const float* __restrict__ img, float* __restrict__ pt, float* __restrict__ qt,
                                                        float* __restrict__ rr)
{
    __shared__ __attribute__((aligned(16))) float tile[2][kMsTileFloats];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const long f_raw = ((long)blockIdx.x * 4 + wave) * 16 + n;
    const bool f_ok = f_raw < F;
    const long f = f_ok ? f_raw : F - 1;
    // B operand of the first chain: b1[4 ks + g] of this lane's frame
    float bv[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) bv[ks] = 4 * ks + g < M ? b1[f * M + 4 * ks + g] : 0.f;
    const float ex = -1.f / gamma - 1.f;
    ms_f4 acc[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) acc[t] = ms_f4{0.f, 0.f, 0.f, 0.f};
    // staging: 3840 floats = 960 float4 per tile, 256 threads x 4 (threads 240 .. 255 idle on the last one)
    const ms_f4* img4 = reinterpret_cast<const ms_f4*>(img);
    ms_f4 st[4];
    auto fetch = [&](int mt) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 256 * q;
            st[q] = i < kMsTileFloats / 4 ? img4[(long)mt * (kMsTileFloats / 4) + i] : ms_f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        ms_f4* d = reinterpret_cast<ms_f4*>(tile[buf]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 256 * q;
            if (i < kMsTileFloats / 4) d[i] = st[q];
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    for (int mt = 0; mt < kMsTiles; ++mt) {
        const int buf = mt & 1;
        if (mt + 1 < kMsTiles) fetch(mt + 1);
        // this lane's four spectrum values of the tile: bins 16 mt + 4 g + r (only bin 256 exists in the last tile)
        ms_f4 xv = {0.f, 0.f, 0.f, 0.f};
        if (mt < 16) xv = *reinterpret_cast<const ms_f4u*>(x + f * 257 + 16 * mt + 4 * g);
        else if (g == 0) xv[0] = x[f * 257 + 256];
        const float* t1 = tile[buf];
        const ms_f4* t2 = reinterpret_cast<const ms_f4*>(tile[buf] + 768);
        ms_f4 re = {0.f, 0.f, 0.f, 0.f}, im = re;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            re = __builtin_amdgcn_mfma_f32_16x16x4f32(t1[ks * 64 + lane], bv[ks], re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_16x16x4f32(t1[384 + ks * 64 + lane], bv[ks], im, 0, 0, 0);
        }
        float s[5][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float X = 1.f + gamma * re[r], Y = gamma * im[r];
            const float XX = X * X, YY = Y * Y, D = XX + YY;
            const float dp = __builtin_amdgcn_exp2f(ex * __builtin_amdgcn_logf(D));   // D > 0; 1 ulp each (as dsa_mgcep_spectra)
            const float pp = xv[r] * dp;
            const float qq = pp / D;
            s[0][r] = pp;
            s[1][r] = qq * (XX - YY);
            s[2][r] = qq * (2.f * X * Y);
            s[3][r] = pp * X;
            s[4][r] = pp * Y;
        }
        // second chain: k-step (mt, r), k-slot g <-> bin 16 mt + 4 g + r: the values above ARE the B operands
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            const ms_f4 a = t2[c * 64 + lane];
            const int in = c < 2 ? 0 : (c < 5 ? 1 : (c < 8 ? 2 : (c < 10 ? 3 : 4)));
            const int t = c < 2 ? c : (c < 5 ? c : (c < 8 ? c - 3 : (c < 10 ? c - 3 : c - 5)));   // accumulators: 0-1 pt | 2-4 qt | 5-6 r
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], s[in][r], acc[t], 0, 0, 0);
        }
        if (mt + 1 < kMsTiles) {
            stage(buf ^ 1);      // the other buffer: its readers finished before the barrier that ended tile mt - 1
            __syncthreads();
        }
    }
    if (!f_ok) return;
    // C/D layout: lane (n, g) register r of tile t <-> column 16 t + 4 g + r of frame n
    const float og = 1.f + gamma;
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (t < 2) {
                const int col = 16 * t + 4 * g + r;
                if (col < M) pt[f * M + col] = acc[t][r];
            } else if (t < 5) {
                const int col = 16 * (t - 2) + 4 * g + r;
                if (col < 2 * M - 1) qt[f * (2 * M - 1) + col] = og * acc[t][r];
            } else {
                const int col = 16 * (t - 5) + 4 * g + r;
                if (col < M + 1) rr[f * (M + 1) + col] = acc[t][r];
            }
        }
}

// Backward of one Newton step's (pt, qt, r) = mgcep_step(x, b1) (above) in ONE launch, same tiling: a wave = 16 frames, a pass =
// 16 bins.  Per pass: re / im recomputed (first chain as above), the cotangents of the five spectra at these bins as row products
// of (gpt | (1 + gamma) gqt | gr) with the TRANSPOSED second-chain matrices (44 k-steps: the cotangent vectors are the B
// operands, held in registers for the whole launch), the element-wise chain rule, gx written, and the cotangent of (re, im)
// accumulated into gb1 = gamma (gX Cr^T + gY Ci^T) (16 k-steps).  72 products per pass (forward: 60).
// Image per tile (tables.mgcep_step_bwd_images): [0, 768) the forward's first-chain operands | [768, 3584) the 44 k-steps
// A[i = bin 16 mt + (lane & 15)][k = column 4 ks + (lane >> 4)] of Pr (6) | Qr (12) | Qi (12) | Rr (7) | Ri (7) |
// [3584, 4608) A[i = coefficient 1 + 16 t + (lane & 15)][k = bin 16 mt + 4 (lane >> 4) + r] of Cr (t, r) | Ci (t, r).
constexpr int kMbTileFloats = 768 + 44 * 64 + 16 * 64;
__global__ __launch_bounds__(256) void mgcep_step_bwd_kernel(const float* __restrict__ x, const float* __restrict__ b1,
                                                            const float* __restrict__ gpt, const float* __restrict__ gqt,
                                                            const float* __restrict__ grr, long F, int M, float gamma,
                                                            const float* __restrict__ img, const float* __restrict__ gx_in,
                                                            float* __restrict__ gx, float* __restrict__ gb1)
{
    __shared__ __attribute__((aligned(16))) float tile[2][kMbTileFloats];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const long f_raw = ((long)blockIdx.x * 4 + wave) * 16 + n;
    const bool f_ok = f_raw < F;
    const long f = f_ok ? f_raw : F - 1;
    const float og = 1.f + gamma;
    // B operands held for the whole launch: coefficient / column 4 ks + g of this lane's frame
    float bv[6], vp[6], vq[12], vr[7];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        bv[ks] = 4 * ks + g < M ? b1[f * M + 4 * ks + g] : 0.f;
        vp[ks] = 4 * ks + g < M ? gpt[f * M + 4 * ks + g] : 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) vq[ks] = 4 * ks + g < 2 * M - 1 ? og * gqt[f * (2 * M - 1) + 4 * ks + g] : 0.f;
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) vr[ks] = 4 * ks + g < M + 1 ? grr[f * (M + 1) + 4 * ks + g] : 0.f;
    const float ex = -1.f / gamma - 1.f;
    ms_f4 accb[2] = {ms_f4{0.f, 0.f, 0.f, 0.f}, ms_f4{0.f, 0.f, 0.f, 0.f}};
    const ms_f4* img4 = reinterpret_cast<const ms_f4*>(img);
    constexpr int kQ = (kMbTileFloats / 4 + 255) / 256;   // float4 per thread and tile
    ms_f4 st[kQ];
    auto fetch = [&](int mt) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            const int i = tid + 256 * q;
            st[q] = i < kMbTileFloats / 4 ? img4[(long)mt * (kMbTileFloats / 4) + i] : ms_f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        ms_f4* d = reinterpret_cast<ms_f4*>(tile[buf]);
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            const int i = tid + 256 * q;
            if (i < kMbTileFloats / 4) d[i] = st[q];
        }
    };
    fetch(0);
    stage(0);
    __syncthreads();
    for (int mt = 0; mt < kMsTiles; ++mt) {
        const int buf = mt & 1;
        if (mt + 1 < kMsTiles) fetch(mt + 1);
        ms_f4 xv = {0.f, 0.f, 0.f, 0.f};
        if (mt < 16) xv = *reinterpret_cast<const ms_f4u*>(x + f * 257 + 16 * mt + 4 * g);
        else if (g == 0) xv[0] = x[f * 257 + 256];
        const float* t1 = tile[buf];
        const float* t2 = tile[buf] + 768;
        const float* t3 = tile[buf] + 768 + 44 * 64;
        ms_f4 re = {0.f, 0.f, 0.f, 0.f}, im = re;
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            re = __builtin_amdgcn_mfma_f32_16x16x4f32(t1[ks * 64 + lane], bv[ks], re, 0, 0, 0);
            im = __builtin_amdgcn_mfma_f32_16x16x4f32(t1[384 + ks * 64 + lane], bv[ks], im, 0, 0, 0);
        }
        // cotangents of the five spectra at bins 16 mt + 4 g + r
        ms_f4 gs[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) gs[i] = ms_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) gs[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(t2[ks * 64 + lane], vp[ks], gs[0], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
            gs[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(t2[(6 + ks) * 64 + lane], vq[ks], gs[1], 0, 0, 0);
            gs[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(t2[(18 + ks) * 64 + lane], vq[ks], gs[2], 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            gs[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(t2[(30 + ks) * 64 + lane], vr[ks], gs[3], 0, 0, 0);
            gs[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(t2[(37 + ks) * 64 + lane], vr[ks], gs[4], 0, 0, 0);
        }
        float gre[4], gim[4], gxv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float X = 1.f + gamma * re[r], Y = gamma * im[r];
            const float XX = X * X, YY = Y * Y, D = XX + YY;
            const float dp = __builtin_amdgcn_exp2f(ex * __builtin_amdgcn_logf(D));
            const float pp = xv[r] * dp;
            const float rD = 1.f / D;
            const float qq = pp * rD;
            const float g_qq = gs[1][r] * (XX - YY) + gs[2][r] * (2.f * X * Y);
            const float g_pp = gs[0][r] + gs[3][r] * X + gs[4][r] * Y + g_qq * rD;
            const float g_D = (g_pp * ex * pp - g_qq * qq) * rD;
            const float gX = 2.f * qq * (gs[1][r] * X + gs[2][r] * Y) + gs[3][r] * pp + 2.f * X * g_D;
            const float gY = 2.f * qq * (gs[2][r] * X - gs[1][r] * Y) + gs[4][r] * pp + 2.f * Y * g_D;
            gre[r] = gamma * gX;
            gim[r] = gamma * gY;
            gxv[r] = g_pp * dp;
        }
        if (f_ok) {
            if (mt < 16) {
                float* dst = gx + f * 257 + 16 * mt + 4 * g;
                ms_f4 o = {gxv[0], gxv[1], gxv[2], gxv[3]};
                if (gx_in) o += *reinterpret_cast<const ms_f4u*>(gx_in + f * 257 + 16 * mt + 4 * g);
                *reinterpret_cast<ms_f4u*>(dst) = o;
            } else if (g == 0) {
                gx[f * 257 + 256] = gxv[0] + (gx_in ? gx_in[f * 257 + 256] : 0.f);
            }
        }
        // gb1[16 t + 4 g + r'] += sum over the tile's bins: k-step r, k-slot g <-> bin 16 mt + 4 g + r
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                accb[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(t3[(t * 4 + r) * 64 + lane], gre[r], accb[t], 0, 0, 0);
                accb[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(t3[(8 + t * 4 + r) * 64 + lane], gim[r], accb[t], 0, 0, 0);
            }
        if (mt + 1 < kMsTiles) {
            stage(buf ^ 1);
            __syncthreads();
        }
    }
    if (!f_ok) return;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 16 * t + 4 * g + r;
            if (col < M) gb1[f * M + col] = accb[t][r];
        }
}

template <typename T>
static int mgcep_spectra_launch(const void* x, const void* b1, int64_t F, int K, int M, const void* Cr, const void* Ci, double gamma,
                                void* out, hipStream_t st)
{
    const unsigned grid = (unsigned)((F + kMsFrames - 1) / kMsFrames);
#define DSA_MS_LAUNCH(MT)                                                                                                  \
    hipLaunchKernelGGL((mgcep_spectra_kernel<T, MT>), dim3(grid), dim3(K > 256 ? 320 : 256), 0, st, (const T*)x, (const T*)b1,  \
                       (long)F, K, M, (const T*)Cr, (const T*)Ci, (T)gamma, (T*)out)
    if (M <= 16) DSA_MS_LAUNCH(16);
    else if (M <= 32) DSA_MS_LAUNCH(32);
    else DSA_MS_LAUNCH(64);
#undef DSA_MS_LAUNCH
    return check_launch("mgcep_spectra");
}

}  // namespace dsa

using namespace dsa;

DSA_EXPORT int dsa_zerodf_fwd(const void* x, const void* b, int64_t B, int64_t T, int32_t M, int32_t P, int32_t zeroth_index,
                              int32_t ignore_gain, int32_t dtype, void* y, void* stream)
{
    DSA_REQUIRE(M >= 0 && P > 0 && B >= 0 && T >= 0 && zeroth_index >= 0 && zeroth_index <= M, "zerodf: invalid sizes");
    DSA_REQUIRE(T % P == 0, "zerodf: the sequence length must be frames x frame_period");
    if (B * T == 0) return DSA_OK;
    const int64_t N = T / P;
    if (dtype == DSA_F32) return zerodf_launch_fwd<float>(x, b, B, T, N, M, P, zeroth_index, ignore_gain, y, (hipStream_t)stream);
    if (dtype == DSA_F64) return zerodf_launch_fwd<double>(x, b, B, T, N, M, P, zeroth_index, ignore_gain, y, (hipStream_t)stream);
    return fail(DSA_ERR_UNSUPPORTED, "zerodf: unsupported dtype%s");
}

DSA_EXPORT int dsa_zerodf_taylor_fwd(const void* x, const void* b, int64_t B, int64_t T, int32_t M, int32_t P, int32_t zeroth_index,
                                     double scale, const void* acc, int32_t dtype, void* y, void* ysum, void* stream)
{
    DSA_REQUIRE(M >= 0 && P > 0 && B >= 0 && T >= 0 && zeroth_index >= 0 && zeroth_index <= M, "zerodf_taylor: invalid sizes");
    DSA_REQUIRE(T % P == 0, "zerodf_taylor: the sequence length must be frames x frame_period");
    if (B * T == 0) return DSA_OK;   // (an empty batch: its tensors have no storage)
    DSA_REQUIRE((acc != nullptr) == (ysum != nullptr), "zerodf_taylor: acc and ysum come together");
    DSA_REQUIRE(y != nullptr || ysum != nullptr, "zerodf_taylor: no output");
    const int64_t N = T / P;
    if (dtype == DSA_F32)
        return zerodf_launch_fwd<float>(x, b, B, T, N, M, P, zeroth_index, 0, y, (hipStream_t)stream, scale, acc, ysum);
    if (dtype == DSA_F64)
        return zerodf_launch_fwd<double>(x, b, B, T, N, M, P, zeroth_index, 0, y, (hipStream_t)stream, scale, acc, ysum);
    return fail(DSA_ERR_UNSUPPORTED, "zerodf_taylor: unsupported dtype%s");
}

DSA_EXPORT int dsa_zerodf_bwd(const void* gy, const void* x, const void* b, const void* y, int64_t B, int64_t T, int32_t M, int32_t P,
                              int32_t zeroth_index, int32_t ignore_gain, int32_t dtype, void* gx, void* gb, void* stream)
{
    DSA_REQUIRE(M >= 0 && P > 0 && B >= 0 && T >= 0 && zeroth_index >= 0 && zeroth_index <= M && T % P == 0, "zerodf_bwd: invalid sizes");
    if (B * T == 0) return DSA_OK;
    const int64_t N = T / P;
    if (dtype == DSA_F32)
        return zerodf_launch_bwd<float>(gy, x, b, y, B, T, N, M, P, zeroth_index, ignore_gain, gx, gb, (hipStream_t)stream);
    if (dtype == DSA_F64)
        return zerodf_launch_bwd<double>(gy, x, b, y, B, T, N, M, P, zeroth_index, ignore_gain, gx, gb, (hipStream_t)stream);
    return fail(DSA_ERR_UNSUPPORTED, "zerodf_bwd: unsupported dtype%s");
}

DSA_EXPORT int dsa_zerodf_taylor_bwd(const void* G, const void* x, const void* b, int64_t B, int64_t T, int32_t M, int32_t P,
                                     int32_t zeroth_index, double scale, const void* gy, int32_t dtype, void* G_out, void* gb,
                                     void* stream)
{
    DSA_REQUIRE(M >= 0 && P > 0 && B >= 0 && T >= 0 && zeroth_index >= 0 && zeroth_index <= M && T % P == 0, "zerodf_taylor_bwd: invalid sizes");
    if (B * T == 0) return DSA_OK;
    DSA_REQUIRE(G_out != nullptr && G_out != G, "zerodf_taylor_bwd: G_out must be a buffer of its own");
    const int64_t N = T / P;
    if (dtype == DSA_F32)
        return zerodf_launch_bwd<float>(G, x, b, nullptr, B, T, N, M, P, zeroth_index, 0, G_out, gb, (hipStream_t)stream, scale, gy, true);
    if (dtype == DSA_F64)
        return zerodf_launch_bwd<double>(G, x, b, nullptr, B, T, N, M, P, zeroth_index, 0, G_out, gb, (hipStream_t)stream, scale, gy, true);
    return fail(DSA_ERR_UNSUPPORTED, "zerodf_taylor_bwd: unsupported dtype%s");
}

DSA_EXPORT int dsa_thsolve_fwd(const void* p, const void* q, const void* r, int64_t F, int32_t n, int32_t dtype, void* g,
                               void* stream)
{
    DSA_REQUIRE(n >= 1 && n <= kThMax && F >= 0, "thsolve: order must be in [1, 64]");
    if (F == 0) return DSA_OK;
    // cepstral order 24, float32: the unpivoted quad-layout solve of the mel-cepstral kernels (DSA_THSOLVE_QUAD=0: A/B)
    static const bool quad = [] {
        const char* e = getenv("DSA_THSOLVE_QUAD");
        return !e || atoi(e) != 0;
    }();
    if (dtype == DSA_F32 && n == 24 && quad && F > 0) return thsolve_quad24_fwd(p, q, r, F, g, (hipStream_t)stream);
    // other orders up to 55, float32: the same scheme as a template over the size -- chosen from (n, dtype) ALONE, like
    // dsa_mcep_newton_update: a frame's rounding must not depend on how many frames travel with it (round 6: the F >= 64 test is gone)
    if (dtype == DSA_F32 && n >= 2 && n <= 55 && quad && F > 0)
        return thsolve_quadn_fwd(p, n, q, 2 * n - 1, r, n, nullptr, nullptr, F, n, g, (hipStream_t)stream);
    if (dtype == DSA_F32) return th_launch<float>(false, nullptr, p, q, r, F, n, g, nullptr, nullptr, (hipStream_t)stream);
    if (dtype == DSA_F64) return th_launch<double>(false, nullptr, p, q, r, F, n, g, nullptr, nullptr, (hipStream_t)stream);
    return fail(DSA_ERR_UNSUPPORTED, "thsolve: unsupported dtype%s");
}

// mcep.py:216-222 for the geometries without a tuned kernel: mc_out = mc_in + solve(T(rt[:n]) + H(rt), rt[:n] - alpha_vec), rt:(F, 2n-1)
DSA_EXPORT int dsa_mcep_newton_update(const void* rt, int64_t F, int32_t n, const void* alpha_vec, int32_t dtype, const void* mc_in,
                                      void* mc_out, void* stream)
{
    DSA_REQUIRE(n >= 2 && n <= 55 && F >= 0, "mcep_newton_update: order must be in [2, 55]");
    DSA_REQUIRE(F == 0 || (rt && alpha_vec && mc_out), "mcep_newton_update: null pointer");   // mc_in = NULL: the solution alone
    if (dtype != DSA_F32) return fail(DSA_ERR_UNSUPPORTED, "mcep_newton_update: float32 only%s");
    if (F == 0) return DSA_OK;
    return thsolve_quadn_fwd(rt, 2 * n - 1, rt, 2 * n - 1, rt, 2 * n - 1, alpha_vec, mc_in, F, n, mc_out, (hipStream_t)stream);
}

// Cotangent of rt from the cotangent of the solution s = solve(T(rt[:n]) + H(rt), rt[:n] - alpha_vec):
//   u = A^-1 gs (A is symmetric: the same batched solve), then per system
//   grt[k] = -sum_{i + j = k} u_i s_j  -  [k < n] sum_{|i - j| = k} u_i s_j  +  [k < n] u_k      (Hankel, Toeplitz, right-hand side)
namespace dsa {
__global__ __launch_bounds__(256) void newton_update_bwd_sums_kernel(const float* __restrict__ u, const float* __restrict__ s, long F, int n,
                                                                    float* __restrict__ grt)
{
    __shared__ float us[4][64], ss[4][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long f = (long)blockIdx.x * 4 + w;
    const bool ok = f < F;
    us[w][lane] = ok && lane < n ? u[f * n + lane] : 0.f;
    ss[w][lane] = ok && lane < n ? s[f * n + lane] : 0.f;
    __syncthreads();
    if (!ok) return;
    for (int k = lane; k < 2 * n - 1; k += 64) {
        float acc = 0.f;
        const int lo = k - (n - 1) > 0 ? k - (n - 1) : 0, hi = k < n - 1 ? k : n - 1;
        for (int i = lo; i <= hi; ++i) acc -= us[w][i] * ss[w][k - i];
        if (k < n) {
            for (int i = 0; i + k < n; ++i) acc -= us[w][i] * ss[w][i + k] + (k > 0 ? us[w][i + k] * ss[w][i] : 0.f);
            acc += us[w][k];
        }
        grt[f * (2 * n - 1) + k] = acc;
    }
}

// Round 6: the same sums with ONE FRAME PER LANE, everything in registers.  The kernel above gives a wave to a frame and a lane one or two
// of its 2 n - 1 sums, every multiply-add behind two LDS reads (143 us per 102 400 frames of order 49: a tenth of the 48 kHz analysis'
// forward + backward).  Here a lane loads its frame's u and s rows (zero-padded to NMAX), runs the three sums fully unrolled at compile
// time -- 2 NMAX^2 + NMAX multiply-adds, no memory access, no cross-lane operation -- and stores four results at a time.  The rows of a
// wave's 64 consecutive frames are one contiguous block, so the per-lane 16-byte accesses use every byte of the lines they touch.
// sums k = K4 .. K4 + 3 of a lane's frame, then the next group: a compile-time recursion (as a loop of 28 x 450 instructions the unroller
// gives up and the arrays live in private memory)
template <int NMAX, int K4>
__device__ __forceinline__ void sums_lane_groups(const float (&uu)[NMAX], const float (&sv)[NMAX], float* orow, int nout)
{
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    if constexpr (K4 < 2 * NMAX - 1) {
        if (K4 < nout) {   // (uniform)
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = K4 + e;
                float acc = 0.f;
                if (k < 2 * NMAX - 1) {
                    // Hankel: sum_{i + j = k} u_i s_j (zero padding makes the sum over the padded range the sum over the order's)
#pragma unroll
                    for (int i = (k - (NMAX - 1) > 0 ? k - (NMAX - 1) : 0); i <= (k < NMAX - 1 ? k : NMAX - 1); ++i) acc = __builtin_fmaf(-uu[i], sv[k - i], acc);
                    if (k < NMAX) {
                        // Toeplitz: sum_{|i - j| = k} u_i s_j, and the right-hand side's u_k (u_k = 0 from the order on)
#pragma unroll
                        for (int i = 0; i + k < NMAX; ++i) {
                            acc = __builtin_fmaf(-uu[i], sv[i + k], acc);
                            if (k > 0) acc = __builtin_fmaf(-uu[i + k], sv[i], acc);
                        }
                        acc += uu[k];
                    }
                }
                o[e] = acc;
            }
            if (K4 + 3 < nout) {
                *reinterpret_cast<f4u*>(orow + K4) = f4u{o[0], o[1], o[2], o[3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (K4 + e < nout) orow[K4 + e] = o[e];
            }
            sums_lane_groups<NMAX, K4 + 4>(uu, sv, orow, nout);
        }
    }
}

template <int NMAX>
__global__ __launch_bounds__(256) void newton_update_bwd_sums_lane_kernel(const float* __restrict__ u, const float* __restrict__ s, long F, int n,
                                                                         float* __restrict__ grt)
{
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const long f = (long)blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const float* ur = u + f * n;
    const float* sr = s + f * n;
    float uu[NMAX], sv[NMAX];
#pragma unroll
    for (int i4 = 0; i4 < NMAX; i4 += 4) {
        // (the group's values first, the array elements assigned unconditionally afterwards: element assignments on conditional paths keep
        //  the arrays in private memory)
        f4u a, b;
        if (i4 + 3 < n) {   // (uniform)
            a = *reinterpret_cast<const f4u*>(ur + i4);
            b = *reinterpret_cast<const f4u*>(sr + i4);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] = i4 + e < n ? ur[i4 + e] : 0.f;
                b[e] = i4 + e < n ? sr[i4 + e] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { uu[i4 + e] = a[e]; sv[i4 + e] = b[e]; }
    }
    float* orow = grt + f * (2 * n - 1);
    sums_lane_groups<NMAX, 0>(uu, sv, orow, 2 * n - 1);
}
}  // namespace dsa

DSA_EXPORT int dsa_mcep_newton_update_bwd(const void* gs, const void* rt, const void* sol, int64_t F, int32_t n, int32_t dtype, void* u,
                                          void* grt, void* stream)
{
    DSA_REQUIRE(n >= 2 && n <= 55 && F >= 0, "mcep_newton_update_bwd: order must be in [2, 55]");
    DSA_REQUIRE(F == 0 || (gs && rt && sol && u && grt), "mcep_newton_update_bwd: null pointer");
    if (dtype != DSA_F32) return fail(DSA_ERR_UNSUPPORTED, "mcep_newton_update_bwd: float32 only%s");
    if (F == 0) return DSA_OK;
    if (int rc = thsolve_quadn_fwd(rt, 2 * n - 1, rt, 2 * n - 1, gs, n, nullptr, nullptr, F, n, u, (hipStream_t)stream)) return rc;
    // one frame per lane (round 6), whatever the batch: the two kernels sum in different orders, and a frame's bits must not depend on
    // how many frames travel with it; DSA_SUMS_LANE=0: the wave-per-frame kernel (A/B)
    static const bool lane_on = [] { const char* e = getenv("DSA_SUMS_LANE"); return !(e && e[0] == '0'); }();
    if (lane_on) {
        const dim3 g((unsigned)((F + 255) / 256));
        if (n <= 36)
            hipLaunchKernelGGL((dsa::newton_update_bwd_sums_lane_kernel<36>), g, dim3(256), 0, (hipStream_t)stream, (const float*)u, (const float*)sol,
                               (long)F, (int)n, (float*)grt);
        else
            hipLaunchKernelGGL((dsa::newton_update_bwd_sums_lane_kernel<56>), g, dim3(256), 0, (hipStream_t)stream, (const float*)u, (const float*)sol,
                               (long)F, (int)n, (float*)grt);
        return dsa::check_launch("mcep_newton_update_bwd");
    }
    hipLaunchKernelGGL(dsa::newton_update_bwd_sums_kernel, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)u,
                       (const float*)sol, (long)F, (int)n, (float*)grt);
    return dsa::check_launch("mcep_newton_update_bwd");
}

DSA_EXPORT int dsa_thsolve_update_fwd(const void* p, const void* q, const void* r, int64_t r_stride, int64_t r_offset, int64_t F,
                                      int32_t n, int32_t dtype, const void* b_in, void* b_out, void* stream)
{
    DSA_REQUIRE(n >= 1 && n <= kThMax && F >= 0, "thsolve_update: order must be in [1, 64]");
    DSA_REQUIRE(r_offset >= 0 && r_stride >= r_offset + n, "thsolve_update: the right-hand side does not fit its row stride");
    DSA_REQUIRE(F == 0 || (b_in != nullptr && b_out != nullptr && b_in != b_out), "thsolve_update: b_in and b_out must be distinct buffers");
    if (F == 0) return DSA_OK;
    if (dtype == DSA_F32 && n == 24)
        return thsolve_quad24_fwd(p, q, r, F, b_out, (hipStream_t)stream, (int)r_stride, (int)r_offset, b_in);
    return fail(DSA_ERR_UNSUPPORTED, "thsolve_update: order 24 in float32 only (dsa_thsolve_fwd + an addition otherwise)%s");
}

DSA_EXPORT int dsa_thsolve_bwd(const void* gg, const void* p, const void* q, const void* g, int64_t F, int32_t n,
                               int32_t dtype, void* gp, void* gq, void* gr, void* stream)
{
    DSA_REQUIRE(n >= 1 && n <= kThMax && F >= 0, "thsolve_bwd: order must be in [1, 64]");
    if (F == 0) return DSA_OK;
    // order 24, float32: u = A^{-1} gbar on the quad-layout solve (A is symmetric; marked systems re-solved with pivoting as in
    // the forward), then the diagonal sums (DSA_THSOLVE_QUAD=0: the one-wave-per-system kernel, A/B)
    static const bool quad = [] {
        const char* e = getenv("DSA_THSOLVE_QUAD");
        return !e || atoi(e) != 0;
    }();
    if (dtype == DSA_F32 && n == 24 && quad && gp && gq && gr) {
        if (int rc = thsolve_quad24_fwd(p, q, gg, F, gr, (hipStream_t)stream)) return rc;
        hipLaunchKernelGGL(th_bwd_sums_kernel, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)gr,
                           (const float*)g, (long)F, (int)n, (float*)gp, (float*)gq);
        return check_launch("th_solve_quad_bwd");
    }
    // the other orders the batched forward covers (csrc/thsolve_quad.hip: 2 .. 55, batches from 64 systems): the same two launches.
    // (The one-wave-per-system backward -- a second pivoted elimination per system -- took 220 us per 12 800 systems of order 50,
    // 37 % of a forward + backward of the 48 kHz analysis; this takes 39 + 7.)
    if (dtype == DSA_F32 && n >= 2 && n <= 55 && n != 24 && F > 0 && quad && gp && gq && gr) {
        if (int rc = thsolve_quadn_fwd(p, n, q, 2 * n - 1, gg, n, nullptr, nullptr, F, n, gr, (hipStream_t)stream)) return rc;
        hipLaunchKernelGGL(th_bwd_sums_kernel, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)gr,
                           (const float*)g, (long)F, (int)n, (float*)gp, (float*)gq);
        return check_launch("th_solve_quadn_bwd");
    }
    if (dtype == DSA_F32) return th_launch<float>(true, gg, p, q, g, F, n, gp, gq, gr, (hipStream_t)stream);
    if (dtype == DSA_F64) return th_launch<double>(true, gg, p, q, g, F, n, gp, gq, gr, (hipStream_t)stream);
    return fail(DSA_ERR_UNSUPPORTED, "thsolve_bwd: unsupported dtype%s");
}

// ---- gain normalisation of generalized cepstra and its inverse as ONE launch each (gnorm.py:102-112, ignorm.py:99-109), forward only ----
//   forward:  (K, x1 / z),  z = 1 + gamma x0,  K = z^(1/gamma)      (gamma = 0: (exp x0, x1))
//   inverse:  ((z - 1) / gamma, y1 z),  z = K^gamma                   (gamma = 0: (log K, y1))
// As stock tensor operations (split, multiply-add, pow, divide, cat) a call was five to six launches of ~5 us; the mel-generalized
// analysis makes three such calls around its Newton steps.  (With a gradient wanted the modules keep the stock composition.)
namespace dsa {
template <typename T>
__global__ __launch_bounds__(256) void gnorm_rows_kernel(const T* __restrict__ x, long F, int n, T gamma, int inverse, T* __restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= F * n) return;
    // (32-bit division where the element count allows it; the launch is ~10 us per 1.3 M elements either way: its own latency)
    const long f = F * n < (1L << 31) ? (long)((unsigned)i / (unsigned)n) : i / n;
    const int m = (int)(i - f * n);
    const T x0 = x[f * n];
    T scale, head;
    if (!inverse) {
        if (gamma == T(0)) {
            head = dsa_exp(x0);
            scale = T(1);
        } else {
            const T z = T(1) + gamma * x0;
            head = dsa_pow(z, T(1) / gamma);
            scale = T(1) / z;
        }
    } else {
        if (gamma == T(0)) {
            head = dsa_log(x0);
            scale = T(1);
        } else {
            const T z = dsa_pow(x0, gamma);
            head = (z - T(1)) / gamma;
            scale = z;
        }
    }
    out[i] = m == 0 ? head : (!inverse && gamma != T(0) ? x[i] / (T(1) + gamma * x0) : x[i] * scale);
}
// b = (sqrt(r_0 + gamma sum_m r_{m+1} b_eps_m), b_join): the gain of a Newton step of the mel-generalized analysis joined to its
// coefficients (mgcep.py:213-215, 221, 231-233).  A row per wave (coalesced reads, the sum over the wave by DPP; one thread per row
// read 25 scattered words per lane: 19 us per 51 200 rows); orders above 64 loop.
template <typename T>
__global__ __launch_bounds__(256) void mgcep_gain_kernel(const T* __restrict__ r, const T* __restrict__ b_eps, const T* __restrict__ b_join,
                                                        long F, int M, T gamma, T* __restrict__ b)
{
    const int lane = threadIdx.x & 63;
    const long f = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= F) return;
    T acc = T(0);
    for (int m = lane; m < M; m += 64) {
        acc += r[f * (M + 1) + m + 1] * b_eps[f * M + m];
        b[f * (M + 1) + m + 1] = b_join[f * M + m];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) b[f * (M + 1)] = sqrt(r[f * (M + 1)] + gamma * acc);
}
}  // namespace dsa

DSA_EXPORT int dsa_gnorm_fwd(const void* x, int64_t F, int32_t n, double gamma, int32_t inverse, int32_t dtype, void* out, void* stream)
{
    DSA_REQUIRE(F >= 0 && n >= 1 && (F == 0 || (x && out)), "gnorm: invalid arguments");
    DSA_REQUIRE(gamma >= -1 && gamma <= 1, "gnorm: gamma must be in [-1, 1]");
    if (F == 0) return DSA_OK;
    const dim3 grid((unsigned)((F * n + 255) / 256));
    if (dtype == DSA_F32)
        hipLaunchKernelGGL(dsa::gnorm_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (long)F, (int)n,
                           (float)gamma, (int)inverse, (float*)out);
    else if (dtype == DSA_F64)
        hipLaunchKernelGGL(dsa::gnorm_rows_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double*)x, (long)F, (int)n, gamma,
                           (int)inverse, (double*)out);
    else return dsa::fail(DSA_ERR_UNSUPPORTED, "gnorm: unsupported dtype%s");
    return dsa::check_launch(inverse ? "ignorm_fwd" : "gnorm_fwd");
}

DSA_EXPORT int dsa_mgcep_gain(const void* r, const void* b_eps, const void* b_join, int64_t F, int32_t M, double gamma, int32_t dtype,
                              void* b, void* stream)
{
    DSA_REQUIRE(F >= 0 && M >= 1 && (F == 0 || (r && b_eps && b_join && b)), "mgcep_gain: invalid arguments");
    if (F == 0) return DSA_OK;
    const dim3 grid((unsigned)((F + 3) / 4));
    if (dtype == DSA_F32)
        hipLaunchKernelGGL(dsa::mgcep_gain_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)r, (const float*)b_eps,
                           (const float*)b_join, (long)F, (int)M, (float)gamma, (float*)b);
    else if (dtype == DSA_F64)
        hipLaunchKernelGGL(dsa::mgcep_gain_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, (const double*)r, (const double*)b_eps,
                           (const double*)b_join, (long)F, (int)M, gamma, (double*)b);
    else return dsa::fail(DSA_ERR_UNSUPPORTED, "mgcep_gain: unsupported dtype%s");
    return dsa::check_launch("mgcep_gain");
}

DSA_EXPORT int dsa_mgcep_step(const void* x, const void* b1, int64_t F, int32_t fft_length, int32_t M, double gamma, const void* images,
                              int32_t dtype, void* pt, void* qt, void* r, void* stream)
{
    DSA_REQUIRE(F >= 0 && M >= 1, "mgcep_step: sizes must be positive");
    DSA_REQUIRE(gamma != 0.0 && gamma >= -1.0 && gamma < 0.0, "mgcep_step: gamma must be in [-1, 0)");
    if (dtype != DSA_F32 || fft_length != 512 || M > 24) return fail(DSA_ERR_UNSUPPORTED, "mgcep_step: needs float32, fft_length 512, cep_order <= 24%s");
    if (F == 0) return DSA_OK;
    hipLaunchKernelGGL(mgcep_step_kernel, dim3((unsigned)((F + 63) / 64)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)b1,
                       (long)F, (int)M, (float)gamma, (const float*)images, (float*)pt, (float*)qt, (float*)r);
    return check_launch("mgcep_step");
}

DSA_EXPORT int dsa_mgcep_step_bwd_h(const void* x, const void* b1, const void* gpt, const void* gqt, const void* gr, int64_t F,
                                    int32_t fft_length, int32_t M, double gamma, const void* images_bwd_h, int32_t dtype, const void* gx_in,
                                    void* gx, void* gb1, void* stream)
{
    DSA_REQUIRE(F >= 0, "mgcep_step_bwd_h: sizes must be positive");
    DSA_REQUIRE(F == 0 || (x && b1 && gpt && gqt && gr && images_bwd_h && gx && gb1), "mgcep_step_bwd_h: null pointer");
    DSA_REQUIRE(gamma != 0.0 && gamma >= -1.0 && gamma < 0.0, "mgcep_step_bwd_h: gamma must be in [-1, 0)");
    if (dtype != DSA_F32 || fft_length != 512 || M != 24)
        return fail(DSA_ERR_UNSUPPORTED, "mgcep_step_bwd_h: needs float32, fft_length 512, cep_order 24%s");
    if (F == 0) return DSA_OK;
    return mgcep_step_bwd_h(x, b1, gpt, gqt, gr, F, gamma, images_bwd_h, gx_in, gx, gb1, (hipStream_t)stream);
}

DSA_EXPORT int dsa_mgcep_step_solve(const void* x, const void* b1, int64_t F, int32_t fft_length, int32_t M, double gamma,
                                    const void* images_h, int32_t dtype, void* b1_out, void* r, void* pt, void* qt, int32_t n_steps,
                                    void* b1_prev, void* stream)
{
    DSA_REQUIRE(F >= 0 && n_steps >= 1, "mgcep_step_solve: sizes must be positive");
    DSA_REQUIRE(F == 0 || (x && b1 && images_h && b1_out && r), "mgcep_step_solve: null pointer");
    DSA_REQUIRE(gamma != 0.0 && gamma > -1.0 && gamma < 0.0, "mgcep_step_solve: gamma must be in (-1, 0)");
    if (dtype != DSA_F32 || fft_length != 512 || M != 24)
        return fail(DSA_ERR_UNSUPPORTED, "mgcep_step_solve: needs float32, fft_length 512, cep_order 24%s");
    if (F == 0) return DSA_OK;
    return mgcep_step_solve_fwd(x, b1, F, gamma, images_h, b1_out, r, (hipStream_t)stream, pt, qt, n_steps, b1_prev);
}

DSA_EXPORT int dsa_mgcep_step_bwd(const void* x, const void* b1, const void* gpt, const void* gqt, const void* gr, int64_t F,
                                  int32_t fft_length, int32_t M, double gamma, const void* images_bwd, int32_t dtype, const void* gx_in,
                                  void* gx, void* gb1, void* stream)
{
    DSA_REQUIRE(F >= 0 && M >= 1, "mgcep_step_bwd: sizes must be positive");
    DSA_REQUIRE(gamma != 0.0 && gamma >= -1.0 && gamma < 0.0, "mgcep_step_bwd: gamma must be in [-1, 0)");
    if (dtype != DSA_F32 || fft_length != 512 || M > 24)
        return fail(DSA_ERR_UNSUPPORTED, "mgcep_step_bwd: needs float32, fft_length 512, cep_order <= 24%s");
    if (F == 0) return DSA_OK;
    hipLaunchKernelGGL(mgcep_step_bwd_kernel, dim3((unsigned)((F + 63) / 64)), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       (const float*)b1, (const float*)gpt, (const float*)gqt, (const float*)gr, (long)F, (int)M, (float)gamma,
                       (const float*)images_bwd, (const float*)gx_in, (float*)gx, (float*)gb1);
    return check_launch("mgcep_step_bwd");
}

DSA_EXPORT int dsa_mgcep_spectra(const void* x, const void* b1, int64_t F, int32_t fft_length, int32_t M, const void* Cr,
                                 const void* Ci, double gamma, int32_t dtype, void* out, void* stream)
{
    DSA_REQUIRE(F >= 0 && fft_length > 1 && fft_length % 2 == 0 && M >= 1, "mgcep_spectra: sizes must be positive");
    DSA_REQUIRE(gamma != 0.0 && gamma >= -1.0 && gamma < 0.0, "mgcep_spectra: gamma must be in [-1, 0)");
    if (M > kMsMaxM) return fail(DSA_ERR_UNSUPPORTED, "mgcep_spectra: cep_order above 64%s");
    if (F == 0) return DSA_OK;
    hipStream_t st = (hipStream_t)stream;
    const int K = fft_length / 2 + 1;
    if (dtype == DSA_F32) return mgcep_spectra_launch<float>(x, b1, F, K, M, Cr, Ci, gamma, out, st);
    if (dtype == DSA_F64) return mgcep_spectra_launch<double>(x, b1, F, K, M, Cr, Ci, gamma, out, st);
    return fail(DSA_ERR_UNSUPPORTED, "mgcep_spectra: unsupported dtype%s");
}
