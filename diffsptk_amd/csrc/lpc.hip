// LPC branch (a11-a13) for gfx950: autocorrelation, Levinson-Durbin, LPC, and the fused
// Frame -> Window -> LPC kernel.
//
// Reference: diffsptk/modules/acorr.py:110-120, levdur.py:113-127, lpc.py:137-139.
//
// MI355X-first restatement:
//  * the reference obtains r[0..M] as irfft(|rfft(x, L+M)|^2)[:M+1]; with the transform length
//    >= L+M there is no circular wrap, so this IS the direct lag sum r[m] = sum_l x[l] x[l+m],
//    computed here straight from LDS-resident frames (float64 accumulation: products of float32
//    samples are exact in float64);
//  * the reference solves the Yule-Walker system (toeplitz(r[:M]) + eps I) a = -r[1:] by dense
//    LU ("based on a simple matrix inversion", levdur.py:25); the same system is solved here by
//    the Levinson-Durbin recursion on (r0 + eps, r1, ..., rM) in float64, one wave per frame.
//    The gain keeps the reference's un-regularised r0 (levdur.py:124).
// Both choices are at least as accurate as the reference's float32 path; parity is judged
// against the float64 reference (tests/test_lpc_gpu.py states the tolerance).
#include <type_traits>

#include "common.h"

#include <atomic>
#include <mutex>
#include <vector>

#include <utility>

namespace dsa {

template <typename T>
__device__ __forceinline__ T acorr_format(double r_m, double r_0, int m, int L, int fmt)
{
    switch (fmt) {  // acorr.py:94-107
    case DSA_ACORR_NORMALIZED: return (T)(r_m / r_0);
    case DSA_ACORR_BIASED: return (T)(r_m / (double)L);
    case DSA_ACORR_UNBIASED: return (T)(r_m / (double)(L - m));
    default: return (T)r_m;
    }
}

// Autocorrelation._forward acorr.py:110-120.  One block (64..256 threads) per frame.
// dynamic LDS: L elements of T.
template <typename T>
__global__ void acorr_fwd_kernel(const T* __restrict__ x, long F, int L, int M, int fmt,
                                 T* __restrict__ r)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* xs = reinterpret_cast<T*>(smem_raw);
    __shared__ double r0s;
    long f = blockIdx.x;
    for (int l = threadIdx.x; l < L; l += blockDim.x) xs[l] = x[f * L + l];
    __syncthreads();
    // lag m handled by thread group; blockDim.x / 64 waves share the lags, each wave reduces
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    if (wave == 0) {
        double acc = 0;
        for (int l = lane; l < L; l += 64) acc += (double)xs[l] * (double)xs[l];
        acc = wave_sum(acc);
        if (lane == 0) r0s = acc;
    }
    __syncthreads();
    for (int m = wave; m <= M; m += nw) {
        double acc = 0;
        for (int l = lane; l + m < L; l += 64) acc += (double)xs[l] * (double)xs[l + m];
        acc = wave_sum(acc);
        if (lane == 0) r[f * (M + 1) + m] = acorr_format<T>(acc, r0s, m, L, fmt);
    }
}

// gx[l] = sum_m grr[m] (x[l+m] + x[l-m]), grr = cotangent of the raw lag sums
template <typename T>
__global__ void acorr_bwd_kernel(const T* __restrict__ gr, const T* __restrict__ x, long F, int L,
                                 int M, int fmt, T* __restrict__ gx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* xs = reinterpret_cast<T*>(smem_raw);
    double* g = reinterpret_cast<double*>(smem_raw + (((size_t)L * sizeof(T) + 7) & ~(size_t)7));
    __shared__ double red[8];
    long f = blockIdx.x;
    for (int l = threadIdx.x; l < L; l += blockDim.x) xs[l] = x[f * L + l];
    for (int m = threadIdx.x; m <= M; m += blockDim.x) {
        double v = (double)gr[f * (M + 1) + m];
        if (fmt == DSA_ACORR_BIASED) v /= (double)L;
        else if (fmt == DSA_ACORR_UNBIASED) v /= (double)(L - m);
        g[m] = v;
    }
    __syncthreads();
    if (fmt == DSA_ACORR_NORMALIZED) {
        // y_m = r_m / r_0:  rbar_m = ybar_m / r_0,  rbar_0 = (ybar_0 - sum_m ybar_m y_m) / r_0
        // (wave 0 recomputes the raw lags it needs)
        int lane = threadIdx.x & 63;
        if (threadIdx.x < 64) {
            double r0 = 0;
            for (int l = lane; l < L; l += 64) r0 += (double)xs[l] * (double)xs[l];
            r0 = wave_sum(r0);
            double corr = 0;
            for (int m = 1; m <= M; ++m) {
                double acc = 0;
                for (int l = lane; l + m < L; l += 64) acc += (double)xs[l] * (double)xs[l + m];
                acc = wave_sum(acc);
                corr += g[m] * acc / r0;
            }
            if (lane == 0) {
                red[0] = r0;
                red[1] = corr;
            }
        }
        __syncthreads();
        double r0 = red[0], corr = red[1];
        __syncthreads();
        for (int m = threadIdx.x; m <= M; m += blockDim.x) g[m] = (m == 0) ? (-corr) / r0 : g[m] / r0;
        // note: ybar_0 multiplies d(r0/r0) = 0, so it drops out
        __syncthreads();
    }
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        double acc = 2.0 * g[0] * (double)xs[l];
        for (int m = 1; m <= M; ++m) {
            double v = 0;
            if (l + m < L) v += (double)xs[l + m];
            if (l - m >= 0) v += (double)xs[l - m];
            acc += g[m] * v;
        }
        gx[f * L + l] = (T)acc;
    }
}

// Levinson-Durbin on (r0 + eps, r1..rM) for M <= 63: one wave per frame, lane j owns a_j and
// r_j in float64 registers, cross-lane traffic by shuffles only (no LDS, no barriers).
// Writes out = [K, a1..aM] (levdur.py:121-126).
template <typename T>
__device__ __forceinline__ void levinson_wave_reg(double r_lane, int M, double eps, T* out)
{
    const int lane = threadIdx.x & 63;
    const double r0 = __shfl(r_lane, 0, 64);
    double a = 0.0;
    double E = r0 + eps;
    for (int m = 1; m <= M; ++m) {
        const bool inner = lane >= 1 && lane < m;
        double rmj = __shfl(r_lane, (m - lane) & 63, 64);  // r[m-j] on lane j
        double acc = wave_sum(inner ? a * rmj : 0.0);
        double k = -(__shfl(r_lane, m, 64) + acc) / E;
        double amj = __shfl(a, (m - lane) & 63, 64);       // a[m-j] on lane j
        if (inner) a += k * amj;
        if (lane == m) a = k;
        E *= (1.0 - k * k);
    }
    double s = wave_sum((lane >= 1 && lane <= M) ? r_lane * a : 0.0) + r0;  // un-regularised r0, levdur.py:124
    if (lane == 0) out[0] = (T)sqrt(s);
    if (lane >= 1 && lane <= M) out[lane] = (T)a;
}

// LevinsonDurbin._forward levdur.py:113-127; one wave per frame, 4 frames per block.
template <typename T>
__global__ void levdur_fwd_kernel(const T* __restrict__ r, long F, int M, double eps, T* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    long f = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (f >= F) return;
    double r_lane = lane <= M ? (double)r[f * (M + 1) + lane] : 0.0;
    levinson_wave_reg<T>(r_lane, M, eps, out + f * (M + 1));
}

// Orders above 63: the same recursion with a[] in LDS; block = one wave per frame.
template <typename T>
__global__ void levdur_fwd_lds_kernel(const T* __restrict__ r, long F, int M, double eps, T* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* rr = reinterpret_cast<double*>(smem_raw);
    double* a = rr + (M + 1);
    double* tmp = a + (M + 1);
    const int lane = threadIdx.x;
    long f = blockIdx.x;
    for (int j = lane; j <= M; j += 64) {
        rr[j] = (double)r[f * (M + 1) + j];
        a[j] = 0;
    }
    __syncthreads();
    double E = rr[0] + eps;
    for (int m = 1; m <= M; ++m) {
        double acc = 0;
        for (int j = 1 + lane; j < m; j += 64) acc += a[j] * rr[m - j];
        acc = wave_sum(acc);
        double k = -(rr[m] + acc) / E;
        for (int j = 1 + lane; j < m; j += 64) tmp[j] = a[j] + k * a[m - j];
        __syncthreads();
        for (int j = 1 + lane; j < m; j += 64) a[j] = tmp[j];
        if (lane == 0) a[m] = k;
        __syncthreads();
        E *= (1.0 - k * k);
    }
    double s = 0;
    for (int j = 1 + lane; j <= M; j += 64) s += rr[j] * a[j];
    s = wave_sum(s) + rr[0];
    if (lane == 0) out[f * (M + 1)] = (T)sqrt(s);
    for (int j = 1 + lane; j <= M; j += 64) out[f * (M + 1) + j] = (T)a[j];
}

// Backward of levdur: out = [K, a], R a = -p, R = toeplitz(r[:M]) + eps I, p = r[1:],
// K = sqrt(p.a + r0).  With cotangents (Kbar, abar):
//   sbar = Kbar / (2K); abar += sbar p; pbar = sbar a; r0bar = sbar;
//   v = R^{-1} abar (R symmetric); pbar -= v; Rbar = -v a^T; rbar[m] += sum_{|i-j|=m} Rbar_ij.
// One block (64 threads) per frame, dense Gauss-Jordan in LDS (float64).
template <typename T>
__global__ void levdur_bwd_kernel(const T* __restrict__ gout, const T* __restrict__ r,
                                  const T* __restrict__ out, long F, int M, double eps,
                                  T* __restrict__ gr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* rr = reinterpret_cast<double*>(smem_raw);  // M+1
    double* a = rr + (M + 1);                          // M   (a[0..M-1] = a_1..a_M)
    double* ab = a + M;                                // M
    double* fac = ab + M;                              // M
    double* aug = fac + M;                             // M * (M+1)
    long f = blockIdx.x;
    const int W = M + 1;
    for (int j = threadIdx.x; j <= M; j += blockDim.x) rr[j] = (double)r[f * (M + 1) + j];
    for (int j = threadIdx.x; j < M; j += blockDim.x) a[j] = (double)out[f * (M + 1) + 1 + j];
    __syncthreads();
    double K = (double)out[f * (M + 1)];
    double sbar = (double)gout[f * (M + 1)] / (2.0 * K);
    for (int j = threadIdx.x; j < M; j += blockDim.x) ab[j] = (double)gout[f * (M + 1) + 1 + j] + sbar * rr[j + 1];
    __syncthreads();
    for (int idx = threadIdx.x; idx < M * W; idx += blockDim.x) {
        int i = idx / W, j = idx - i * W;
        aug[idx] = j < M ? rr[i > j ? i - j : j - i] + (i == j ? eps : 0.0) : ab[i];
    }
    // Gauss-Jordan, no pivoting (R is symmetric positive definite)
    for (int k = 0; k < M; ++k) {
        __syncthreads();
        double inv = 1.0 / aug[k * W + k];
        for (int i = threadIdx.x; i < M; i += blockDim.x) fac[i] = (i == k) ? 0.0 : aug[i * W + k] * inv;
        __syncthreads();
        int ncol = W - (k + 1);
        for (int idx = threadIdx.x; idx < M * ncol; idx += blockDim.x) {
            int i = idx / ncol, jj = k + 1 + (idx - i * ncol);
            if (i != k) aug[i * W + jj] -= fac[i] * aug[k * W + jj];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < M; i += blockDim.x) fac[i] = aug[i * W + M] / aug[i * W + i];  // v
    __syncthreads();
    for (int m = threadIdx.x; m <= M; m += blockDim.x) {
        double acc = 0;
        if (m < M) {  // Toeplitz scatter of Rbar = -v a^T onto r[0..M-1]
            for (int i = 0; i + m < M; ++i) {
                acc -= fac[i] * a[i + m];
                if (m > 0) acc -= fac[i + m] * a[i];
            }
        }
        if (m >= 1) acc += sbar * a[m - 1] - fac[m - 1];  // pbar
        if (m == 0) acc += sbar;                          // r0 inside the gain
        gr[f * (M + 1) + m] = (T)acc;
    }
}

// Fused Frame -> Window -> autocorrelation -> Levinson (README.md:198-201 of the reference);
// one wave per frame (M <= 63), blockDim/64 frames per block.  dynamic LDS: L elements per wave.
template <typename T>
__global__ void frame_window_lpc_kernel(const T* __restrict__ x, long Tlen, long N, long F, int L, int P,
                                        int left, int mode, const T* __restrict__ w, int M, double eps,
                                        T* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    T* xs = reinterpret_cast<T*>(smem_raw) + (size_t)wave * L;
    long f = (long)blockIdx.x * nw + wave;
    const bool active = f < F;
    if (active) {
        long b = f / N, n = f - b * N;
        const T* xb = x + b * Tlen;
        for (int l = lane; l < L; l += 64) xs[l] = load_padded(xb, n * P + l - left, Tlen, mode) * (w ? w[l] : T(1));
    }
    __syncthreads();
    if (!active) return;
    double r_lane = 0.0;
    for (int m = 0; m <= M; ++m) {
        double acc = 0;
        for (int l = lane; l + m < L; l += 64) acc += (double)xs[l] * (double)xs[l + m];
        acc = wave_sum(acc);
        if (lane == m) r_lane = acc;
    }
    levinson_wave_reg<T>(r_lane, M, eps, out + f * (M + 1));
}

// ---------------------------------------------------------------------------------------------
// Tuned fused Frame -> Window -> autocorrelation -> Levinson for float32 input, lpc_order 24,
// frame_length <= 512 (README.md:198-201 of the reference; BASELINE configs[3]).
//   * one wave64 per workgroup owns up to 64 consecutive frames of one utterance;
//   * autocorrelation: 4 frames per pass (16 lanes each) out of one LDS-resident stretch of
//     3P + L samples (each sample read from HBM once); lane j of a frame owns a contiguous run of
//     samples, keeps 25 + 24 windowed samples in registers (as float64: products of float32 are
//     exact) and accumulates its 25 lag sums with 625 statically indexed float64 FMAs per 25-sample
//     block; the 16 partials are added by an xor butterfly;
//   * Levinson-Durbin: one frame per LANE for all 64 frames at once, float64, fully unrolled in
//     registers (no cross-lane traffic at all), then [K, a_1..a_24] is written out.
// dynamic LDS: in_buf[(3P + L) rounded] floats | wtab[L + 64] floats | rbuf[64][25] doubles
constexpr int kLpcM1 = 25;

// Sum over the 16 lanes of a DPP row, left in every lane: four rotate-and-add steps (row_ror 8, 4, 2, 1) on the
// vector ALU.  (`__shfl_xor` on a double is two ds_bpermute_b32: 200 LDS-crossbar instructions per pass for the 25
// lag sums, which kept the LDS pipe busier than the float64 FMAs kept the ALU.)  Addition is commutative, so every
// lane of the row ends with the bit-identical sum.
template <int SH>
__device__ __forceinline__ double row_ror_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + SH, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + SH, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_allsum(double v)
{
    v += row_ror_f64<8>(v);
    v += row_ror_f64<4>(v);
    v += row_ror_f64<2>(v);
    v += row_ror_f64<1>(v);
    return v;
}

template <int... Is>
__device__ __forceinline__ void lag_block(double (&acc)[kLpcM1], const double (&xw)[2 * kLpcM1 - 1],
                                          std::integer_sequence<int, Is...>)
{
    // acc[m] += xw[i] * xw[i + m] for i, m in 0..24, flattened (Is = 25 i + m)
    ((acc[Is % kLpcM1] = __builtin_fma(xw[Is / kLpcM1], xw[Is / kLpcM1 + Is % kLpcM1], acc[Is % kLpcM1])), ...);
}

__global__ __launch_bounds__(64, 2) void frame_window_lpc24_kernel(
    const float* __restrict__ x, long Tlen, long N, int L, int P, int left, int mode,
    const float* __restrict__ w, double eps, float* __restrict__ out, long total_sc, int sc_per_utt,
    int in_floats, int wtab_floats, unsigned* __restrict__ queue, int fpi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* in_buf = reinterpret_cast<float*>(smem_raw);
    float* wtab = in_buf + in_floats;
    double* rbuf = reinterpret_cast<double*>(wtab + wtab_floats);
    const int lane = threadIdx.x;
    const int j = lane & 15, fl = lane >> 4;
    const int C = (L + 15) >> 4;                       // samples per lane
    const int nblk = (C + kLpcM1 - 1) / kLpcM1;        // 25-sample blocks per lane
    for (int l = lane; l < wtab_floats; l += 64) wtab[l] = l < L ? (w ? w[l] : 1.f) : 0.f;   // w == NULL: a window of ones

    // Work items are chunks of fpi <= 64 consecutive frames of one utterance (the launcher sizes them so that the
    // utterances split evenly and the item count is close to a whole number of rounds: 200 frames = 4 x 52 instead
    // of 64 + 64 + 64 + 8), handed out by a ticket counter.
    for (;;) {
        unsigned ticket = 0;
        if (lane == 0) ticket = atomicAdd(queue, 1u);
        const long tk = (long)__builtin_amdgcn_readfirstlane(ticket);
        if (tk >= total_sc) {
            // Every wave draws exactly one ticket past the end, so the counter stops at total_sc + (number of waves): the wave that
            // drew the last of those hands it back zeroed (DSA_LPC_SCRATCH_IS_CLEAN: no fill launch before the next call) -- no
            // second counter.  (A separate "waves done" counter put 3072 more atomics on one address at the very end: + 0.03 ms.)
            if (lane == 0 && tk == total_sc + (long)(gridDim.x * (blockDim.x >> 6)) - 1)
                __hip_atomic_store(queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        const long b = tk / sc_per_utt;
        const long ci = tk - b * sc_per_utt;
        const long fbase = ci * fpi;
        const int nfr = (int)((N - fbase) < fpi ? (N - fbase) : fpi);
        const float* xb = x + b * Tlen;
        const int npass = (nfr + 3) >> 2;
        for (int p = 0; p < npass; ++p) {
            const long frame0 = fbase + 4 * p;
            const int nvalid = (int)((N - frame0) < 4 ? (N - frame0) : 4);
            __syncthreads();
            {   // stage the shared stretch; zero-fill what the lanes may read beyond it
                const long g0 = frame0 * P - left;
                const int need = (nvalid - 1) * P + L;
                const bool interior = g0 >= 0 && g0 + need <= Tlen;
                if (interior && (((size_t)(xb + g0)) & 15) == 0 && (need & 3) == 0) {
                    const float4* src4 = reinterpret_cast<const float4*>(xb + g0);
                    float4* dst4 = reinterpret_cast<float4*>(in_buf);
                    for (int sidx = lane; sidx < (need >> 2); sidx += 64) dst4[sidx] = src4[sidx];
                } else {
                    for (int sidx = lane; sidx < need; sidx += 64) in_buf[sidx] = load_padded(xb, g0 + sidx, Tlen, mode);
                }
                for (int sidx = need + lane; sidx < in_floats; sidx += 64) in_buf[sidx] = 0.f;
            }
            __syncthreads();
            double acc[kLpcM1];
#pragma unroll
            for (int m = 0; m < kLpcM1; ++m) acc[m] = 0.0;
            const float* fsrc = in_buf + fl * P;
            for (int blk = 0; blk < nblk; ++blk) {
                const int t0 = j * C + blk * kLpcM1;  // this lane's block start within the frame
                const int own = C - blk * kLpcM1;     // samples of the block that belong to the lane
                double xw[2 * kLpcM1 - 1];
#pragma unroll
                for (int i = 0; i < 2 * kLpcM1 - 1; ++i) {
                    const int t = t0 + i;
                    // window.py:190 in float32 (as the reference), then exact promotion; zero past the
                    // frame, and zero for "own" positions that belong to the next lane (i >= own)
                    const float v = (t < L) ? fsrc[t] * wtab[t] : 0.f;
                    xw[i] = (double)v;
                }
                if (own >= kLpcM1) {
                    lag_block(acc, xw, std::make_integer_sequence<int, kLpcM1 * kLpcM1>{});
                } else {
                    // partial last block: the leading factor stops at the end of the lane's own run
#pragma unroll
                    for (int i = 0; i < kLpcM1; ++i) {
                        const double li = i < own ? xw[i] : 0.0;
#pragma unroll
                        for (int m = 0; m < kLpcM1; ++m) acc[m] = __builtin_fma(li, xw[i + m], acc[m]);
                    }
                }
            }
            // add the 16 lanes of the frame (xor butterfly inside the 16-lane group)
#pragma unroll
            for (int m = 0; m < kLpcM1; ++m) {
                acc[m] = row16_allsum(acc[m]);
            }
            if (fl < nvalid) {
                double* rrow = rbuf + (size_t)(4 * p + fl) * kLpcM1;
#pragma unroll
                for (int m = 0; m < kLpcM1; ++m)
                    if ((m & 15) == j) rrow[m] = acc[m];  // lane j stores lags j and j + 16
            }
        }
        __syncthreads();
        // ---- Levinson-Durbin, one frame per lane (levdur.py:113-127 as a recursion) ----
        if (lane < nfr) {
            double r[kLpcM1], a[kLpcM1];
#pragma unroll
            for (int m = 0; m < kLpcM1; ++m) {
                r[m] = rbuf[(size_t)lane * kLpcM1 + m];
                a[m] = 0.0;
            }
            double Ecur = r[0] + eps;
#pragma unroll
            for (int m = 1; m < kLpcM1; ++m) {
                double s = r[m];
#pragma unroll
                for (int q = 1; q < m; ++q) s = __builtin_fma(a[q], r[m - q], s);
                const double kk = -s / Ecur;
#pragma unroll
                for (int q = 1; 2 * q <= m; ++q) {
                    const double aq = a[q], amq = a[m - q];
                    a[q] = __builtin_fma(kk, amq, aq);
                    if (q != m - q) a[m - q] = __builtin_fma(kk, aq, amq);
                }
                a[m] = kk;
                Ecur *= (1.0 - kk * kk);
            }
            double gsum = r[0];  // un-regularised r0 (levdur.py:124)
#pragma unroll
            for (int m = 1; m < kLpcM1; ++m) gsum = __builtin_fma(r[m], a[m], gsum);
            float* o = out + ((b * N + fbase + lane) * (long)kLpcM1);
            o[0] = (float)sqrt(gsum);
#pragma unroll
            for (int m = 1; m < kLpcM1; ++m) o[m] = (float)a[m];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Round 4: the same fused Frame -> Window -> autocorrelation -> Levinson with the lag sums on the float32 MATRIX instruction.
// The float64 vector unit was the whole kernel above (625 v_fma_f64 per 25 samples and lane: 0.24 ms per 204 800 frames).  Here a
// wave takes one frame at a time and forms its lag sums as a banded Gram product: with u_b = the b-th block of 16 windowed samples,
//     C1 = sum_b u_b u_b^T,   C2 = sum_b u_b u_{b+1}^T,   C3 = sum_b u_b u_{b+2}^T        (16 x 16 each)
// hold every product xw[l] xw[l + m], m <= 24, exactly once: entry (i, j) of C_s is lag 16 (s - 1) + j - i summed over the
// positions l = i (mod 16).  Operands need no staging at all: lane = i + 16 k of v_mfma_f32_16x16x4_f32 is sample 64 e + lane of
// the frame -- a coalesced load times the window -- and the B operand of C2 / C3 is the same load 16 / 32 samples further on
// (L1 hits).  The products run as 3-term binary16 splits on the matrix pipe (9 instructions per frame, see the loop); the 16 entries of a lag are added in FLOAT64 (scattered through LDS so that a lane reads its lag's
// 16 slots as four 16-byte reads), then Levinson-Durbin in float64, one frame per lane, as above.
// Accuracy: the lag sums are ~1e-7 r[0] from the exact ones (what the reference's own float32 FFT route has); the exact kernel
// stays selectable (DSA_LPC_LAGSUMS=f64) and is what float64 input runs.
// x = hi + lo in binary16 (round to nearest), two values at a time: one packed conversion, then lo = binary16(x - float(hi)) as ONE
// v_fma_mixlo_f16 / v_fma_mixhi_f16 per value (binary16 operand from its half register, float32 multiply-add, result rounded into
// the low / high half).  The wait state after each half-register write is what gfx950 wants before the register is touched again
// (same helper as csrc/mcep_mfma_f16.h:split2).
typedef _Float16 lp_h2 __attribute__((ext_vector_type(2)));
typedef float lp_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lp_split2(float x0, float x1, lp_h2& hi, lp_h2& lo)
{
    hi = __builtin_convertvector(lp_f2{x0, x1}, lp_h2);
    const unsigned hb = __builtin_bit_cast(unsigned, hi);
    unsigned lb;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\ts_nop 0\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0"
        : "=&v"(lb) : "v"(hb), "v"(x0), "v"(x1));
    lo = __builtin_bit_cast(lp_h2, lb);
}

// The scatter area of a frame: lag rows kLsDS floats apart + a dump row.  kLsDS = 23: with lag = 16 s + j - (4 g + r) and slot
// i = 4 g + r the 64 lanes of a store hit bank (23 j + 8 g + const) mod 32 -- two lanes per bank, the floor for 256 bytes -- where
// the 16-byte aligned stride 20 of round 4 put (j + g) mod 8 classes, i.e. eight lanes, on one bank; entries whose lag is outside
// [0, 24] go to a per-lane dump slot instead of all onto row 25's four addresses (sixteen lanes on one address).  Rows are read
// back as 16-byte accesses at 4-byte alignment.  (round-4 review: lds_conflict_frac 0.49 in this kernel)
constexpr int kLsDS = 23;
constexpr int kLsArea = ((26 * kLsDS + 3) & ~3) + 64;   // floats per frame of a round: 26 rows (25 = never written, read by idle lanes) + dump
typedef float lp_f4a4 __attribute__((ext_vector_type(4), aligned(4)));
#ifndef LPC_ABL
#define LPC_ABL 0   // measurement builds only (tools/gpu_abl_lpc.sh): 1 no recursion, 2 no scatter / sums, 4 no products, 8 no maximum / scale
#endif
template <int NE, int LC>   // NE = 8: operand elements per lane (512 samples); LC: the frame length at compile time (0: run time)
__global__ __launch_bounds__(256, 3) void frame_window_lpc24_mfma_kernel(
    const float* __restrict__ x, long Tlen, long N, int L_rt, int P, int left, int mode, const float* __restrict__ w, double eps,
    float* __restrict__ out, long total_sc, int sc_per_utt, unsigned* __restrict__ queue, int fpi)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef _Float16 lp_h8 __attribute__((ext_vector_type(8)));
    constexpr int DS = kLsDS;                      // floats per lag row of the scatter area (see kLsDS)
    // dynamic LDS, per wave: rbuf[fpi][25] doubles | dm[2][26][DS] floats (a scatter area per frame of a round)  (fpi = 40 at the bench
    // geometry: three workgroups per CU)
    extern __shared__ __attribute__((aligned(16))) unsigned char lpc_smem[];
    const int L = LC ? LC : L_rt;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wave_bytes = fpi * kLpcM1 * (int)sizeof(double) + 2 * kLsArea * (int)sizeof(float);
    double* rbuf = reinterpret_cast<double*>(lpc_smem + (size_t)wave * wave_bytes);
    float* dm = reinterpret_cast<float*>(rbuf + fpi * kLpcM1);
    const int j = lane & 15, g = lane >> 4;
    // Operand element e of lane (j, g) is sample 16 (8 g + e) + j: the lane's eight elements are eight CONSECUTIVE blocks of 16
    // samples (k-slot (g, e) <-> block 8 g + e; any bijection serves, A and B use the same one).  The operands of C2 / C3 -- the same
    // samples one / two blocks on -- are then the lane's own elements one / two places on, i.e. (packed two to a register) one
    // v_alignbit per register resp. the next register, plus element 0 of the lane 16 further on for the last place(s): one
    // ds_bpermute per frame and half.  (With k-slot <-> block 4 e + g the three operand sets were three loads, window products,
    // scalings and splits of the same samples: 0.107 of the kernel's 0.145 ms were operand preparation.)
    // window values (w == NULL: ones); samples past the frame are selected away
    float wa[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int la = 128 * g + 16 * e + j;
        wa[e] = la < L ? (w ? w[la] : 1.f) : 0.f;
    }
    // scatter addresses of this lane's 12 matrix entries: entry (tile s, register r) = C_s[4 g + r][j], lag = 16 s + j - (4 g + r);
    // lags outside [0, 24] go to the spare row 25
    int addr[3][4];
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * g + r, lag = 16 * s_ + j - i;
            addr[s_][r] = (lag >= 0 && lag < kLpcM1) ? lag * DS + i : kLsArea - 64 + lane;
        }
    // Items are dealt out statically, item = wave + k (number of waves), unless a counter is given (queue != NULL: tickets).  With
    // the ticket counter every item and every wave's exit was an atomic on ONE address -- 5120 + 3072 of them per launch at the
    // bench size, about one per 16 ns: the counter's throughput, not the arithmetic, set the launch time (0.105 of 0.135 ms were
    // left with everything but the loads removed, tools/gpu_abl_lpc.sh).
    const long nwaves_g = (long)gridDim.x * (blockDim.x >> 6);
    long next_static = (long)blockIdx.x * (blockDim.x >> 6) + wave;
    for (;;) {
        long tk;
        if (queue) {
            unsigned ticket = 0;
            if (lane == 0) ticket = atomicAdd(queue, 1u);
            tk = (long)__builtin_amdgcn_readfirstlane(ticket);
        } else {
            tk = next_static;
            next_static += nwaves_g;
        }
        if (tk >= total_sc) {
            // (tickets: every wave draws exactly one past the end, so the counter stops at total_sc + the number of waves; the wave
            // that drew the last of those hands it back zeroed -- DSA_LPC_SCRATCH_IS_CLEAN: no fill launch before the next call)
            if (queue && lane == 0 && tk == total_sc + nwaves_g - 1) __hip_atomic_store(queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        const long b = tk / sc_per_utt;
        const long ci = tk - b * sc_per_utt;
        const long fbase = ci * fpi;
        const int nfr = (int)((N - fbase) < fpi ? (N - fbase) : fpi);
        const float* xb = x + b * Tlen;
        // TWO frames per round (frame fi + u uses scatter area u): loads -> maximum -> split -> products -> scatter -> sums is one
        // dependent chain per frame, and at three waves per SIMD a wave that walks its frames one at a time leaves the vector unit
        // idle half the time (46 % busy, 0.150 ms); the two chains of a round share no data and interleave.  The samples of the
        // next round are requested while this one is in the matrix pipeline.
        constexpr int U = 2;
        float a[U][8];
        const int soff = 128 * g + j;   // this lane's first sample inside a frame
        auto fetch = [&](int u, int fi) __attribute__((always_inline)) {
            const long start = (fbase + fi) * P - left;
            if (start >= 0 && start + 512 <= Tlen) {   // uniform: every sample any lane touches exists
                const float* src = xb + start + soff;
#pragma unroll
                for (int e = 0; e < 8; ++e) a[u][e] = src[16 * e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) a[u][e] = soff + 16 * e < L ? load_padded(xb, start + soff + 16 * e, Tlen, mode) : 0.f;
            }
        };
        // (an odd count: the last round's second frame is the first one again, computed and not stored)
        // (Tried twice: the samples of two rounds in flight, two register sets and the round loop unrolled by two -- with the ticket counter
        // 0.135 -> 0.145 ms, without it 0.087 -> 0.093 ms.)
        fetch(0, 0);
        fetch(1, 1 < nfr ? 1 : 0);
        for (int fi = 0; fi < nfr; fi += U) {
            float va[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e)   // window.py:190 in float32 (as the reference); samples past the frame are selected away, never multiplied
                    va[u][e] = soff + 16 * e < L ? a[u][e] * wa[e] : 0.f;
            if (fi + U < nfr) {
                fetch(0, fi + U);
                fetch(1, fi + U + 1 < nfr ? fi + U + 1 : fi + U);
            }
            // The Gram products on the BINARY16 matrix pipe (separate from the float32 datapath the float64 sums and the recursion
            // need).  Every value is split hi + lo into two binary16 numbers after scaling the frame by a power of two that puts its
            // largest sample in [2^13, 2^14) (lo stays a normal number down to 2^-17 of the maximum); a product is three
            // instructions, hi hi + hi lo + lo hi (the dropped lo lo is 2^-22 of the product), exact binary16 products accumulated
            // in float32 over the whole frame at once (K = 32 blocks of 16 samples cover 512).  9 matrix instructions per frame
            // instead of 19 float32 ones (round 4, first version: 608 cycles of the float32 datapath per frame).
            float fmx[U];
            int sh[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                fmx[u] = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) fmx[u] = __builtin_fmaxf(fmx[u], __builtin_fabsf(va[u][e]));
            }
#define DSA_LPC_MAX(CTRL, RM)                                                                                              \
    _Pragma("unroll") for (int u = 0; u < U; ++u) fmx[u] =                                                                 \
        __builtin_fmaxf(fmx[u], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(fmx[u]), CTRL, RM, 0xf, false)))
            DSA_LPC_MAX(0x111, 0xf); DSA_LPC_MAX(0x112, 0xf); DSA_LPC_MAX(0x114, 0xf); DSA_LPC_MAX(0x118, 0xf);   // row_shr 1, 2, 4, 8
            DSA_LPC_MAX(0x142, 0xa); DSA_LPC_MAX(0x143, 0xc);                                                     // row_bcast 15, 31
#undef DSA_LPC_MAX
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float fmax_all = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fmx[u]), 63));
                int fe = __builtin_amdgcn_frexp_expf(fmax_all);            // fmax_all = m 2^fe, m in [0.5, 1)
                fe = fe < -100 ? -100 : (fe > 100 ? 100 : fe);             // silent frames / denormals: any scale will do
                sh[u] = (LPC_ABL & 8) ? 3 : 14 - fe;                       // scaled maximum in [2^13, 2^14)
            }
            lp_h8 ah[U], al[U], bh[U], bl[U], ch[U], cl[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                typedef unsigned lp_u4 __attribute__((ext_vector_type(4)));
                lp_u4 hr, lr;   // the packed halves: register k = elements (2 k, 2 k + 1)
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    lp_h2 h, l;
                    lp_split2(__builtin_ldexpf(va[u][e], sh[u]), __builtin_ldexpf(va[u][e + 1], sh[u]), h, l);
                    hr[e >> 1] = __builtin_bit_cast(unsigned, h);
                    lr[e >> 1] = __builtin_bit_cast(unsigned, l);
                }
                // elements (0, 1) of the lane 16 further on (blocks 8 (g + 1), 8 (g + 1) + 1); past the last row: zeros
                unsigned nh = (unsigned)__builtin_amdgcn_ds_bpermute(4 * ((lane + 16) & 63), (int)hr[0]);
                unsigned nl = (unsigned)__builtin_amdgcn_ds_bpermute(4 * ((lane + 16) & 63), (int)lr[0]);
                nh = g == 3 ? 0u : nh;
                nl = g == 3 ? 0u : nl;
                const lp_u4 bhr = {__builtin_amdgcn_alignbit(hr[1], hr[0], 16), __builtin_amdgcn_alignbit(hr[2], hr[1], 16),
                                   __builtin_amdgcn_alignbit(hr[3], hr[2], 16), __builtin_amdgcn_alignbit(nh, hr[3], 16)};
                const lp_u4 blr = {__builtin_amdgcn_alignbit(lr[1], lr[0], 16), __builtin_amdgcn_alignbit(lr[2], lr[1], 16),
                                   __builtin_amdgcn_alignbit(lr[3], lr[2], 16), __builtin_amdgcn_alignbit(nl, lr[3], 16)};
                const lp_u4 chr = {hr[1], hr[2], hr[3], nh}, clr = {lr[1], lr[2], lr[3], nl};
                ah[u] = __builtin_bit_cast(lp_h8, hr);
                al[u] = __builtin_bit_cast(lp_h8, lr);
                bh[u] = __builtin_bit_cast(lp_h8, bhr);
                bl[u] = __builtin_bit_cast(lp_h8, blr);
                ch[u] = __builtin_bit_cast(lp_h8, chr);
                cl[u] = __builtin_bit_cast(lp_h8, clr);
            }
            f4 c1[U], c2[U], c3[U];
#pragma unroll
            for (int u = 0; u < U; ++u) c1[u] = c2[u] = c3[u] = f4{0.f, 0.f, 0.f, 0.f};
            if (!(LPC_ABL & 4)) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c1[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[u], ah[u], c1[u], 0, 0, 0);
                c2[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[u], bh[u], c2[u], 0, 0, 0);
                c3[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[u], ch[u], c3[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c1[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], al[u], c1[u], 0, 0, 0);
                c2[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], bl[u], c2[u], 0, 0, 0);
                c3[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], cl[u], c3[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c1[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], ah[u], c1[u], 0, 0, 0);
                c2[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], bh[u], c2[u], 0, 0, 0);
                c3[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], ch[u], c3[u], 0, 0, 0);
            }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) { c1[u][0] = va[u][0]; c2[u][0] = va[u][1]; c3[u][0] = va[u][2]; }
            }
            if (LPC_ABL & 2) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (lane < kLpcM1 && fi + u < nfr) rbuf[(fi + u) * kLpcM1 + lane] = c1[u][0] + c2[u][1] + c3[u][2] + (lane == 0 ? 1.0 : 0.0);
                continue;
            }
            __builtin_amdgcn_wave_barrier();   // the previous round's row reads are done (LDS operations of a wave run in order)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dm[u * kLsArea + addr[0][r]] = c1[u][r];
                    dm[u * kLsArea + addr[1][r]] = c2[u][r];
                    dm[u * kLsArea + addr[2][r]] = c3[u][r];
                }
            __builtin_amdgcn_wave_barrier();
            {
                // lane (m, h) = (lane & 31, lane >> 5) adds slots 8 h .. 8 h + 7 of lag m in float64; the halves meet through one
                // cross-half exchange (rows 25 .. 31 read the spare row: finite or not, their sums are never stored)
                const int m_ = lane & 31, h_ = lane >> 5;
                double sm[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const lp_f4a4* row = reinterpret_cast<const lp_f4a4*>(dm + u * kLsArea + (m_ < kLpcM1 ? m_ : kLpcM1) * DS + 8 * h_);
                    const f4 q0 = row[0], q1 = row[1];
                    sm[u] = ((double)q0[0] + (double)q0[1]) + ((double)q0[2] + (double)q0[3]);
                    sm[u] += ((double)q1[0] + (double)q1[1]) + ((double)q1[2] + (double)q1[3]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int lo = __builtin_amdgcn_ds_bpermute(4 * (lane ^ 32), __double2loint(sm[u]));
                    const int hi = __builtin_amdgcn_ds_bpermute(4 * (lane ^ 32), __double2hiint(sm[u]));
                    const double other = __hiloint2double(hi, lo);
                    // the same association on both halves: (slots 0..7) + (slots 8..15)
                    const double tot = h_ == 0 ? sm[u] + other : other + sm[u];
                    if (lane < kLpcM1 && fi + u < nfr) rbuf[(fi + u) * kLpcM1 + lane] = ldexp(tot, -2 * sh[u]);   // the frame's scale, undone exactly
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- Levinson-Durbin, one frame per lane (levdur.py:113-127 as a recursion), as in frame_window_lpc24_kernel ----
        if ((LPC_ABL & 1) && lane < nfr) {
            float* o = out + ((b * N + fbase + lane) * (long)kLpcM1);
#pragma unroll
            for (int m = 0; m < kLpcM1; ++m) o[m] = (float)rbuf[(size_t)lane * kLpcM1 + m];
        } else if (lane < nfr) {
            double r[kLpcM1], al[kLpcM1];
#pragma unroll
            for (int m = 0; m < kLpcM1; ++m) {
                r[m] = rbuf[(size_t)lane * kLpcM1 + m];
                al[m] = 0.0;
            }
            double Ecur = r[0] + eps;
#pragma unroll
            for (int m = 1; m < kLpcM1; ++m) {
                double s_ = r[m];
#pragma unroll
                for (int q = 1; q < m; ++q) s_ = __builtin_fma(al[q], r[m - q], s_);
                const double kk = -s_ / Ecur;
#pragma unroll
                for (int q = 1; 2 * q <= m; ++q) {
                    const double aq = al[q], amq = al[m - q];
                    al[q] = __builtin_fma(kk, amq, aq);
                    if (q != m - q) al[m - q] = __builtin_fma(kk, aq, amq);
                }
                al[m] = kk;
                Ecur *= (1.0 - kk * kk);
            }
            double gsum = r[0];  // un-regularised r0 (levdur.py:124)
#pragma unroll
            for (int m = 1; m < kLpcM1; ++m) gsum = __builtin_fma(r[m], al[m], gsum);
            float* o = out + ((b * N + fbase + lane) * (long)kLpcM1);
            o[0] = (float)sqrt(gsum);
#pragma unroll
            for (int m = 1; m < kLpcM1; ++m) o[m] = (float)al[m];
        }
        __builtin_amdgcn_wave_barrier();   // rbuf is rewritten by the next chunk
    }
}

// ---------------------------------------------------------------------------------------------
// Tuned backward of LinearPredictiveCodingAnalysis for float32 frames, lpc_order 24, 25 <= L <= 512
// (the adjoint of acorr.py:110-120 + levdur.py:113-127 in one launch).  One wave64 per workgroup
// owns 64 consecutive frames:
//   A. lag sums r[0..24] recomputed 4 frames per pass (16 lanes each) exactly as in the forward;
//   B. one frame per LANE, float64, fully unrolled: the Levinson recursion is re-run and carries a
//      general right-hand side along (u = (R + eps I)^{-1} abar, by the order-update with the
//      reversed predictor), then with sbar = Kbar / (2K), v = u - sbar a (because R a = -r_1):
//        rbar[m] = -sum_{|i-j|=m} v_i a_j  (+ sbar a_m - v_m for m >= 1;  + sbar for m = 0);
//   C. xbar[l] = sum_m rbar[m] (x[l+m] + x[l-m]), 4 frames per pass, each lane a contiguous run of
//      samples with the 49-sample window in registers (2 x 625 float64 FMAs per 25 samples), the
//      result staged through LDS and written back as whole rows.
// dynamic LDS: in_buf[4][S] floats (24 zeros | frame | zeros) | out_buf[4][16 C] floats |
//              rbuf[64][25] doubles (r, then rbar) | gbuf[64 * 25] floats (cotangent rows)
template <int... Is>
__device__ __forceinline__ void corr_block_fwd(double (&acc)[kLpcM1], const double (&g)[kLpcM1],
                                               const double (&xw)[2 * kLpcM1 - 1], std::integer_sequence<int, Is...>)
{
    // acc[i] += g[m] * xw[i + m]   (Is = 25 i + m)
    ((acc[Is / kLpcM1] = __builtin_fma(g[Is % kLpcM1], xw[Is / kLpcM1 + Is % kLpcM1], acc[Is / kLpcM1])), ...);
}

template <int... Is>
__device__ __forceinline__ void corr_block_rev(double (&acc)[kLpcM1], const double (&g)[kLpcM1],
                                               const double (&xw)[2 * kLpcM1 - 1], std::integer_sequence<int, Is...>)
{
    // acc[i] += g[m] * xw[24 + i - m]   (xw[24 + i] is the lane's sample i)
    ((acc[Is / kLpcM1] =
          __builtin_fma(g[Is % kLpcM1], xw[kLpcM1 - 1 + Is / kLpcM1 - Is % kLpcM1], acc[Is / kLpcM1])),
     ...);
}

__global__ __launch_bounds__(64, 1) void lpc24_bwd_kernel(const float* __restrict__ gout,
                                                          const float* __restrict__ x, long F, int L, double eps,
                                                          float* __restrict__ gx, long total_sc, int S, int So, int fpi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* in_buf = reinterpret_cast<float*>(smem_raw);
    float* out_buf = in_buf + 4 * S;
    double* rbuf = reinterpret_cast<double*>(out_buf + 4 * So);
    // the cotangent rows are dead before the output staging starts: they share its buffer when it is large enough
    float* gbuf = 4 * So >= 64 * kLpcM1 ? out_buf : reinterpret_cast<float*>(rbuf + 64 * kLpcM1);
    const int lane = threadIdx.x;
    const int j = lane & 15, fl = lane >> 4;
    const int C = (L + 15) >> 4;
    const int nblk = (C + kLpcM1 - 1) / kLpcM1;
    const bool vec4 = (L & 3) == 0 && ((((size_t)x) | ((size_t)gx)) & 15) == 0;
    for (int l = lane; l < 4 * S; l += 64) in_buf[l] = 0.f;  // the pads stay zero for the whole kernel

    auto stage = [&](long frame0, int nvalid) {
        __syncthreads();
        for (int fr = 0; fr < nvalid; ++fr) {
            const float* src = x + (frame0 + fr) * (long)L;
            float* dst = in_buf + fr * S + (kLpcM1 - 1);
            if (vec4) {
                for (int q = lane; q < (L >> 2); q += 64)
                    reinterpret_cast<float4*>(dst)[q] = reinterpret_cast<const float4*>(src)[q];
            } else {
                for (int t = lane; t < L; t += 64) dst[t] = src[t];
            }
        }
        __syncthreads();
    };

    for (long sc = blockIdx.x; sc < total_sc; sc += gridDim.x) {
        const long fbase = sc * fpi;   // fpi <= 64 frames per work item (see the launcher)
        const int nfr = (int)((F - fbase) < fpi ? (F - fbase) : fpi);
        const int npass = (nfr + 3) >> 2;
        __syncthreads();
        for (int q = lane; q < nfr * kLpcM1; q += 64) gbuf[q] = gout[fbase * kLpcM1 + q];
        // ---- A: lag sums ----
        for (int p = 0; p < npass; ++p) {
            const int nvalid = (nfr - 4 * p) < 4 ? (nfr - 4 * p) : 4;
            stage(fbase + 4 * p, nvalid);
            double acc[kLpcM1];
#pragma unroll
            for (int m = 0; m < kLpcM1; ++m) acc[m] = 0.0;
            const float* fsrc = in_buf + fl * S + (kLpcM1 - 1);
            for (int blk = 0; blk < nblk; ++blk) {
                const int t0 = j * C + blk * kLpcM1;
                const int own = C - blk * kLpcM1;
                double xw[2 * kLpcM1 - 1];
#pragma unroll
                for (int i = 0; i < 2 * kLpcM1 - 1; ++i) xw[i] = (double)fsrc[t0 + i];
                if (own >= kLpcM1) {
                    lag_block(acc, xw, std::make_integer_sequence<int, kLpcM1 * kLpcM1>{});
                } else {
#pragma unroll
                    for (int i = 0; i < kLpcM1; ++i) {
                        const double li = i < own ? xw[i] : 0.0;
#pragma unroll
                        for (int m = 0; m < kLpcM1; ++m) acc[m] = __builtin_fma(li, xw[i + m], acc[m]);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < kLpcM1; ++m) {
                acc[m] = row16_allsum(acc[m]);
            }
            if (fl < nvalid) {
                double* rrow = rbuf + (size_t)(4 * p + fl) * kLpcM1;
#pragma unroll
                for (int m = 0; m < kLpcM1; ++m)
                    if ((m & 15) == j) rrow[m] = acc[m];
            }
        }
        __syncthreads();
        // ---- B: adjoint of the Yule-Walker solve, one frame per lane ----
        if (lane < nfr) {
            double r[kLpcM1], a[kLpcM1], u[kLpcM1];
            double* row = rbuf + (size_t)lane * kLpcM1;
            const float* go = gbuf + lane * kLpcM1;
#pragma unroll
            for (int m = 0; m < kLpcM1; ++m) {
                r[m] = row[m];
                a[m] = 0.0;
                u[m] = 0.0;
            }
            double Ecur = r[0] + eps;
#pragma unroll
            for (int m = 1; m < kLpcM1; ++m) {
                // order update of the general solve with the order-(m-1) predictor and its error
                double d = (double)go[m];
#pragma unroll
                for (int q = 1; q < m; ++q) d = __builtin_fma(-r[m - q], u[q], d);
                const double mu = d / Ecur;
#pragma unroll
                for (int q = 1; q < m; ++q) u[q] = __builtin_fma(mu, a[m - q], u[q]);
                u[m] = mu;
                // Levinson step (as in the forward kernel)
                double s = r[m];
#pragma unroll
                for (int q = 1; q < m; ++q) s = __builtin_fma(a[q], r[m - q], s);
                const double kk = -s / Ecur;
#pragma unroll
                for (int q = 1; 2 * q <= m; ++q) {
                    const double aq = a[q], amq = a[m - q];
                    a[q] = __builtin_fma(kk, amq, aq);
                    if (q != m - q) a[m - q] = __builtin_fma(kk, aq, amq);
                }
                a[m] = kk;
                Ecur *= (1.0 - kk * kk);
            }
            double gsum = r[0];
#pragma unroll
            for (int m = 1; m < kLpcM1; ++m) gsum = __builtin_fma(r[m], a[m], gsum);
            const double sbar = (double)go[0] / (2.0 * sqrt(gsum));
#pragma unroll
            for (int m = 1; m < kLpcM1; ++m) u[m] = __builtin_fma(-sbar, a[m], u[m]);  // v
#pragma unroll
            for (int m = 0; m < kLpcM1; ++m) {
                double acc = (m == 0) ? sbar : __builtin_fma(sbar, a[m], -u[m]);
#pragma unroll
                for (int i = 1; i + m < kLpcM1; ++i) {
                    if (m < kLpcM1 - 1) {
                        acc = __builtin_fma(-u[i], a[i + m], acc);
                        if (m > 0) acc = __builtin_fma(-u[i + m], a[i], acc);
                    }
                }
                row[m] = acc;
            }
        }
        __syncthreads();
        // ---- C: adjoint of the lag sums ----
        for (int p = 0; p < npass; ++p) {
            const int nvalid = (nfr - 4 * p) < 4 ? (nfr - 4 * p) : 4;
            stage(fbase + 4 * p, nvalid);
            double g[kLpcM1];
            {
                const double* grow = rbuf + (size_t)(4 * p + (fl < nvalid ? fl : 0)) * kLpcM1;
#pragma unroll
                for (int m = 0; m < kLpcM1; ++m) g[m] = grow[m];
            }
            const float* fsrc = in_buf + fl * S + (kLpcM1 - 1);
            float* odst = out_buf + fl * So;
            for (int blk = 0; blk < nblk; ++blk) {
                const int t0 = j * C + blk * kLpcM1;
                const int own = C - blk * kLpcM1;
                double acc[kLpcM1];
#pragma unroll
                for (int i = 0; i < kLpcM1; ++i) acc[i] = 0.0;
                {
                    double xw[2 * kLpcM1 - 1];
#pragma unroll
                    for (int i = 0; i < 2 * kLpcM1 - 1; ++i) xw[i] = (double)fsrc[t0 + i];
                    corr_block_fwd(acc, g, xw, std::make_integer_sequence<int, kLpcM1 * kLpcM1>{});
                }
                {
                    double xw[2 * kLpcM1 - 1];
#pragma unroll
                    for (int i = 0; i < 2 * kLpcM1 - 1; ++i) xw[i] = (double)fsrc[t0 - (kLpcM1 - 1) + i];
                    corr_block_rev(acc, g, xw, std::make_integer_sequence<int, kLpcM1 * kLpcM1>{});
                }
                if (own >= kLpcM1) {
#pragma unroll
                    for (int i = 0; i < kLpcM1; ++i) odst[t0 + i] = (float)acc[i];
                } else {
#pragma unroll
                    for (int i = 0; i < kLpcM1; ++i)
                        if (i < own) odst[t0 + i] = (float)acc[i];
                }
            }
            __syncthreads();
            for (int fr = 0; fr < nvalid; ++fr) {
                float* dst = gx + (fbase + 4 * p + fr) * (long)L;
                const float* src = out_buf + fr * So;
                if (vec4) {
                    for (int q = lane; q < (L >> 2); q += 64)
                        reinterpret_cast<float4*>(dst)[q] = reinterpret_cast<const float4*>(src)[q];
                } else {
                    for (int t = lane; t < L; t += 64) dst[t] = src[t];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Round 5: backward of the fused Frame -> Window -> LPC launch in ONE launch (the adjoint of frame.py:120-141, window.py:185-193,
// acorr.py:110-120 and levdur.py:113-127 composed; README.md:198-201 of the reference with a gradient).  The module chain ran it as
// lpc24_bwd (0.71 ms per 204 800 frames: 1875 v_fma_f64 per four frames at one wave per SIMD behind synchronous stagings) + window
// and frame backward over two materialised (F, 400) tensors.  Here a wave owns a run of Hc hops of one utterance -- the output
// samples [h0 P, (h0 + Hc) P) -- and every frame that touches them (Hc + 4..5 frames at L = 400, P = 80: the halo is recomputed,
// nothing is exchanged between waves, and every sample's sum runs over its frames in increasing order whatever the partition:
// bit-identical for every Hc):
//   A. lag sums of every frame as the forward kernel forms them (banded Gram product of binary16 splits on the matrix pipe, the
//      16 entries of a lag added in float64) -> rbuf;
//   B. one frame per LANE, float64, fully unrolled: the Levinson recursion re-run with a general right-hand side riding along
//      (as lpc24_bwd_kernel) -> the cotangent of the lag sums rbar[0..24], left in the frame's row as float32 scaled by a power of
//      two (its largest entry in [2^13, 2^14));
//   C. the adjoint of the lag sums  xwbar[l] = sum_m rbar[m] (xw[l + m] + xw[l - m])  -- a 49-tap symmetric filter per frame -- as a
//      banded matrix product on the binary16 matrix pipe: with u_c the c-th block of 16 windowed samples and E[q] = ext[q - 32]
//      (ext[m] = rbar[|m|], doubled at 0), block c of xwbar is  sum_{s=-2..2} T_s u_{c+s},  (T_s)[i][j] = E[16 s + 32 + j - i]:
//      A operand (16 x 96) = the Toeplitz band, row i = 8 consecutive taps from E[k - i] (a 16-byte LDS read at a 2-byte
//      aligned address: gfx950 takes unaligned DS accesses), B operand (96 x 16) = column c = samples 16 c - 32 + k (aligned),
//      3 k-steps x 2 column tiles x 3 split terms = 18 v_mfma_f32_16x16x32_f16 per frame instead of 1250 v_fma_f64 per 25
//      samples and lane.  The result layout gives lane (c, g) the four consecutive samples 16 c + 4 g .. + 3: times the window,
//      scale undone, added into the wave's overlap-add stretch in LDS (one 16-byte read-modify-write per tile and lane; frames in
//      increasing order); the own range of the stretch goes to gx as whole rows at the end.
// dynamic LDS per workgroup (one wave): rbuf[nfr_max][26] doubles | stretch[(nfr_max - 1) P + 512] floats |
//   area shared by phase A (scatter rows dm[2][26][20] floats) and phase C (XH[640] XL[640] EH[128] EL[128] binary16)
typedef _Float16 lp_h8u __attribute__((ext_vector_type(8), aligned(2)));
typedef float lp_f4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int kLbRow = 26;          // doubles per frame row of rbuf: 25 lag sums (then: 25 scaled taps as floats, their shift) | the samples' shift
constexpr int kLbRing = 1024;       // floats of the overlap-add ring (frame_length + frame_period <= 1024)
constexpr int kLbOps = 2 * 640 + 2 * 128;   // binary16 values of one operand set of phase C: XH | XL | EH | EL
constexpr int kLbAreaBytes = 2 * kLbOps * 2 > 2 * kLsArea * 4 ? 2 * kLbOps * 2 : 2 * kLsArea * 4;
__host__ __device__ inline long lb_floordiv(long a, long b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

template <int LC>
__global__ __launch_bounds__(64, 2) void frame_window_lpc24_bwd_mfma_kernel(
    const float* __restrict__ gout, const float* __restrict__ x, long Tlen, long N, int L_rt, int P, int left,
    const float* __restrict__ w, double eps, float* __restrict__ gx, long total_items, int items_per_utt, int Hc, int nfr_max)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef _Float16 lp_h8 __attribute__((ext_vector_type(8)));
    typedef unsigned lp_u4 __attribute__((ext_vector_type(4)));
    constexpr int DS = kLsDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lb_smem[];
    const int L = LC ? LC : L_rt;
    const int lane = threadIdx.x;
    double* rbuf = reinterpret_cast<double*>(lb_smem);
    float* ring = reinterpret_cast<float*>(rbuf + nfr_max * kLbRow);
    float* dm = ring + kLbRing;
    // phase C operand arrays, one set per frame of a round (u), over phase A's scatter rows: per set XH | XL [32 zeros | 512 samples |
    // 96 zeros] and EH | EL (E[q] at q + 16, q in [-16, 112)), set stride kLbOps halves
    _Float16* XH = reinterpret_cast<_Float16*>(dm);
    _Float16* XL = XH + 640;
    _Float16* EH = XL + 640;
    _Float16* EL = EH + 128;
    const int j = lane & 15, g = lane >> 4;
    // ---- per-lane constants ----
    float wa[8];    // phase A: element e of lane (j, g) is sample 128 g + 16 e + j
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int la = 128 * g + 16 * e + j;
        wa[e] = la < L ? (w ? w[la] : 1.f) : 0.f;
    }
    float wc[8];    // phase C input: samples 8 lane .. 8 lane + 7
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int la = 8 * lane + e;
        wc[e] = la < L ? (w ? w[la] : 1.f) : 0.f;
    }
    float wo[2][4]; // phase C output: samples 16 (16 nt + j) + 4 g + r
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int la = 16 * (16 * nt + j) + 4 * g + r;
            wo[nt][r] = la < L ? (w ? w[la] : 1.f) : 0.f;
        }
    int addr[3][4];
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * g + r, lag = 16 * s_ + j - i;
            addr[s_][r] = (lag >= 0 && lag < kLpcM1) ? lag * DS + i : kLsArea - 64 + lane;
        }

    for (int q = lane; q < kLbRing / 4; q += 64) reinterpret_cast<f4*>(ring)[q] = f4{0.f, 0.f, 0.f, 0.f};   // (every flush leaves its slots zero)
    for (long item = blockIdx.x; item < total_items; item += gridDim.x) {
        const long b = item / items_per_utt;
        const long h0 = (item - b * items_per_utt) * Hc;
        const long s0 = h0 * P;
        long s1 = s0 + (long)Hc * P;
        if (s1 > Tlen) s1 = Tlen;
        if (s0 >= s1) continue;
        long n_lo = lb_floordiv(s0 + left - L, P) + 1;
        long n_hi = lb_floordiv(s1 - 1 + left, P);
        if (n_lo < 0) n_lo = 0;
        if (n_hi > N - 1) n_hi = N - 1;
        const int nfr = (int)(n_hi - n_lo + 1);       // <= nfr_max (host); may be <= 0 when no frame reaches the range (L < P)
        const float* xb = x + b * Tlen;
        __builtin_amdgcn_wave_barrier();
#ifndef LPB_ABL
#define LPB_ABL 0   // measurement builds only: 1 no lag sums, 2 no recursion, 4 no filter / overlap-add, 8 taps read at an aligned address,
                    // 16 no ring update / flush, 32 no matrix products
#endif
        // ================= A: lag sums (the forward kernel's rounds of two frames) =================
        if (nfr > 0 && !(LPB_ABL & 1)) {
            constexpr int U = 2;
            float a[U][8];
            const int soff = 128 * g + j;
            auto fetch = [&](int u, int fi) __attribute__((always_inline)) {
                const long start = (n_lo + fi) * P - left;
                if (start >= 0 && start + 512 <= Tlen) {
                    const float* src = xb + start + soff;
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[u][e] = src[16 * e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const long si = start + soff + 16 * e;
                        a[u][e] = (soff + 16 * e < L && si >= 0 && si < Tlen) ? xb[si] : 0.f;
                    }
                }
            };
            fetch(0, 0);
            fetch(1, 1 < nfr ? 1 : 0);
            for (int fi = 0; fi < nfr; fi += U) {
                float va[U][8];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int e = 0; e < 8; ++e) va[u][e] = soff + 16 * e < L ? a[u][e] * wa[e] : 0.f;
                if (fi + U < nfr) {
                    fetch(0, fi + U);
                    fetch(1, fi + U + 1 < nfr ? fi + U + 1 : fi + U);
                }
                float fmx[U];
                int sh[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    fmx[u] = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) fmx[u] = __builtin_fmaxf(fmx[u], __builtin_fabsf(va[u][e]));
                }
#define DSA_LPC_MAX(CTRL, RM)                                                                                              \
    _Pragma("unroll") for (int u = 0; u < U; ++u) fmx[u] =                                                                 \
        __builtin_fmaxf(fmx[u], __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(fmx[u]), CTRL, RM, 0xf, false)))
                DSA_LPC_MAX(0x111, 0xf); DSA_LPC_MAX(0x112, 0xf); DSA_LPC_MAX(0x114, 0xf); DSA_LPC_MAX(0x118, 0xf);
                DSA_LPC_MAX(0x142, 0xa); DSA_LPC_MAX(0x143, 0xc);
#undef DSA_LPC_MAX
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float fmax_all = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fmx[u]), 63));
                    int fe = __builtin_amdgcn_frexp_expf(fmax_all);
                    fe = fe < -100 ? -100 : (fe > 100 ? 100 : fe);
                    sh[u] = 14 - fe;
                }
                lp_h8 ah[U], al[U], bh[U], bl[U], ch[U], cl[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    lp_u4 hr, lr;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        lp_h2 h, l;
                        lp_split2(__builtin_ldexpf(va[u][e], sh[u]), __builtin_ldexpf(va[u][e + 1], sh[u]), h, l);
                        hr[e >> 1] = __builtin_bit_cast(unsigned, h);
                        lr[e >> 1] = __builtin_bit_cast(unsigned, l);
                    }
                    unsigned nh = (unsigned)__builtin_amdgcn_ds_bpermute(4 * ((lane + 16) & 63), (int)hr[0]);
                    unsigned nl = (unsigned)__builtin_amdgcn_ds_bpermute(4 * ((lane + 16) & 63), (int)lr[0]);
                    nh = g == 3 ? 0u : nh;
                    nl = g == 3 ? 0u : nl;
                    const lp_u4 bhr = {__builtin_amdgcn_alignbit(hr[1], hr[0], 16), __builtin_amdgcn_alignbit(hr[2], hr[1], 16),
                                       __builtin_amdgcn_alignbit(hr[3], hr[2], 16), __builtin_amdgcn_alignbit(nh, hr[3], 16)};
                    const lp_u4 blr = {__builtin_amdgcn_alignbit(lr[1], lr[0], 16), __builtin_amdgcn_alignbit(lr[2], lr[1], 16),
                                       __builtin_amdgcn_alignbit(lr[3], lr[2], 16), __builtin_amdgcn_alignbit(nl, lr[3], 16)};
                    const lp_u4 chr = {hr[1], hr[2], hr[3], nh}, clr = {lr[1], lr[2], lr[3], nl};
                    ah[u] = __builtin_bit_cast(lp_h8, hr);
                    al[u] = __builtin_bit_cast(lp_h8, lr);
                    bh[u] = __builtin_bit_cast(lp_h8, bhr);
                    bl[u] = __builtin_bit_cast(lp_h8, blr);
                    ch[u] = __builtin_bit_cast(lp_h8, chr);
                    cl[u] = __builtin_bit_cast(lp_h8, clr);
                }
                f4 c1[U], c2[U], c3[U];
#pragma unroll
                for (int u = 0; u < U; ++u) c1[u] = c2[u] = c3[u] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    c1[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[u], ah[u], c1[u], 0, 0, 0);
                    c2[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[u], bh[u], c2[u], 0, 0, 0);
                    c3[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[u], ch[u], c3[u], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    c1[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], al[u], c1[u], 0, 0, 0);
                    c2[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], bl[u], c2[u], 0, 0, 0);
                    c3[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], cl[u], c3[u], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    c1[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], ah[u], c1[u], 0, 0, 0);
                    c2[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], bh[u], c2[u], 0, 0, 0);
                    c3[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], ch[u], c3[u], 0, 0, 0);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dm[u * kLsArea + addr[0][r]] = c1[u][r];
                        dm[u * kLsArea + addr[1][r]] = c2[u][r];
                        dm[u * kLsArea + addr[2][r]] = c3[u][r];
                    }
                __builtin_amdgcn_wave_barrier();
                {
                    const int m_ = lane & 31, h_ = lane >> 5;
                    double sm[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const lp_f4a4* row = reinterpret_cast<const lp_f4a4*>(dm + u * kLsArea + (m_ < kLpcM1 ? m_ : kLpcM1) * DS + 8 * h_);
                        const f4 q0 = row[0], q1 = row[1];
                        sm[u] = ((double)q0[0] + (double)q0[1]) + ((double)q0[2] + (double)q0[3]);
                        sm[u] += ((double)q1[0] + (double)q1[1]) + ((double)q1[2] + (double)q1[3]);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int lo = __builtin_amdgcn_ds_bpermute(4 * (lane ^ 32), __double2loint(sm[u]));
                        const int hi = __builtin_amdgcn_ds_bpermute(4 * (lane ^ 32), __double2hiint(sm[u]));
                        const double other = __hiloint2double(hi, lo);
                        const double tot = h_ == 0 ? sm[u] + other : other + sm[u];
                        if (lane < kLpcM1 && fi + u < nfr) rbuf[(fi + u) * kLbRow + lane] = ldexp(tot, -2 * sh[u]);
                        if (lane == kLpcM1 && fi + u < nfr) rbuf[(fi + u) * kLbRow + kLpcM1] = (double)sh[u];   // the samples' shift, for phase C
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ================= B: adjoint of the Yule-Walker solve, one frame per lane (as lpc24_bwd_kernel) =================
        // (the lag sums stay in the frame's LDS row and are read where the recursion wants them: a, u in registers are 100 of them
        //  instead of 150, and three waves share a SIMD)
        if (lane < nfr && !(LPB_ABL & 2)) {
            double a[kLpcM1], u[kLpcM1];
            double* row = rbuf + (size_t)lane * kLbRow;
            const float* go = gout + (b * N + n_lo + lane) * (long)kLpcM1;
#pragma unroll
            for (int m = 0; m < kLpcM1; ++m) {
                a[m] = 0.0;
                u[m] = 0.0;
            }
            const double r0 = row[0];
            double Ecur = r0 + eps;
#pragma unroll
            for (int m = 1; m < kLpcM1; ++m) {
                double d = (double)go[m];
                double s = row[m];
#pragma unroll
                for (int q = 1; q < m; ++q) {
                    const double rv = row[m - q];
                    d = __builtin_fma(-rv, u[q], d);
                    s = __builtin_fma(a[q], rv, s);
                }
                const double mu = d / Ecur;
#pragma unroll
                for (int q = 1; q < m; ++q) u[q] = __builtin_fma(mu, a[m - q], u[q]);
                u[m] = mu;
                const double kk = -s / Ecur;
#pragma unroll
                for (int q = 1; 2 * q <= m; ++q) {
                    const double aq = a[q], amq = a[m - q];
                    a[q] = __builtin_fma(kk, amq, aq);
                    if (q != m - q) a[m - q] = __builtin_fma(kk, aq, amq);
                }
                a[m] = kk;
                Ecur *= (1.0 - kk * kk);
            }
            double gsum = r0;
#pragma unroll
            for (int m = 1; m < kLpcM1; ++m) gsum = __builtin_fma(row[m], a[m], gsum);
            const double sbar = (double)go[0] / (2.0 * sqrt(gsum));
#pragma unroll
            for (int m = 1; m < kLpcM1; ++m) u[m] = __builtin_fma(-sbar, a[m], u[m]);  // v
            // the taps ext[m] = rbar[m] (doubled at 0: the lag-0 sum's adjoint is 2 rbar[0] xw[l]) over the dead lag sums, as float64
            double mx = 0.0;
#pragma unroll
            for (int m = 0; m < kLpcM1; ++m) {
                double acc = (m == 0) ? sbar : __builtin_fma(sbar, a[m], -u[m]);
#pragma unroll
                for (int i = 1; i + m < kLpcM1; ++i) {
                    if (m < kLpcM1 - 1) {
                        acc = __builtin_fma(-u[i], a[i + m], acc);
                        if (m > 0) acc = __builtin_fma(-u[i + m], a[i], acc);
                    }
                }
                if (m == 0) acc += acc;
                row[m] = acc;
                mx = fmax(mx, fabs(acc));
            }
            // ... then in place as float32 scaled by 2^she (largest in [2^13, 2^14)): float m lies inside doubles <= m / 2, all read by
            // then.  A non-finite cotangent keeps the shift 0 and poisons its own frame only.
            int fe = 0;
            const bool okmx = mx > 0.0 && mx < 1e300;
            if (okmx) (void)frexp(mx, &fe);
            const int she = okmx ? 14 - fe : 0;
            float* frow = reinterpret_cast<float*>(row);
#pragma unroll
            for (int m = 0; m < kLpcM1; ++m) {
                const double tv = row[m];
                asm volatile("" ::: "memory");
                frow[m] = (float)ldexp(tv, she);
            }
            reinterpret_cast<int*>(frow)[kLpcM1] = she;
        }
        __builtin_amdgcn_wave_barrier();
        // ================= C: the 49-tap filter of every frame on the matrix pipe, overlap-add in a ring in LDS =================
        // Two frames per round (their chains share nothing and interleave).  The overlap-add runs in a ring of kLbRing floats:
        // position q of the item's stretch (sample n_lo P - left + q) lives at q mod kLbRing; after frame fi's contribution the
        // P positions [fi P, (fi + 1) P) are final -- no later frame reaches them -- and go to gx (own range only), their slots back to zero.
        for (int q = lane; q < 2 * kLbOps * 2 / 16; q += 64) reinterpret_cast<f4*>(dm)[q] = f4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_wave_barrier();
        {
            const long base = n_lo * P - left;        // sample of stretch position 0
            float* gxb = gx + b * Tlen;
            const bool vec = (P & 3) == 0;
            auto flush = [&](long q0, int cnt) __attribute__((always_inline)) {   // positions [q0, q0 + cnt) -> gx, slots zeroed
                if (vec && (((size_t)(gxb + base + q0)) & 15) == 0 && (cnt & 3) == 0) {
                    for (int t = 4 * lane; t < cnt; t += 256) {
                        f4* slot = reinterpret_cast<f4*>(ring + ((q0 + t) & (kLbRing - 1)));
                        const f4 v = *slot;
                        *slot = f4{0.f, 0.f, 0.f, 0.f};
                        const long sidx = base + q0 + t;
                        if (sidx >= s0 && sidx + 3 < s1) *reinterpret_cast<f4*>(gxb + sidx) = v;
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (sidx + r >= s0 && sidx + r < s1) gxb[sidx + r] = v[r];
                        }
                    }
                } else {
                    for (int t = lane; t < cnt; t += 64) {
                        float* slot = ring + ((q0 + t) & (kLbRing - 1));
                        const float v = *slot;
                        *slot = 0.f;
                        const long sidx = base + q0 + t;
                        if (sidx >= s0 && sidx < s1) gxb[sidx] = v;
                    }
                }
            };
            constexpr int U = 2;
            float c8[U][8];
            // this lane's band offset into the taps (lane-constant), as the dword that holds its first tap + a funnel shift
            typedef unsigned lp_u4a4 __attribute__((ext_vector_type(4), aligned(4)));
            const int eoff = 16 + 8 * g - j;                 // >= 1
            const unsigned* etap = reinterpret_cast<const unsigned*>(EH) + (eoff >> 1);
            const unsigned esh = (eoff & 1) ? 16u : 0u;
            auto fetchc = [&](int u, int fi) __attribute__((always_inline)) {
                const long start = (n_lo + fi) * P - left;
                if (start >= 0 && start + 512 <= Tlen) {
                    const lp_f4u* src = reinterpret_cast<const lp_f4u*>(xb + start + 8 * lane);
                    const lp_f4u q0 = src[0], q1 = src[1];
                    c8[u][0] = q0[0]; c8[u][1] = q0[1]; c8[u][2] = q0[2]; c8[u][3] = q0[3];
                    c8[u][4] = q1[0]; c8[u][5] = q1[1]; c8[u][6] = q1[2]; c8[u][7] = q1[3];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const long si = start + 8 * lane + e;
                        c8[u][e] = (8 * lane + e < L && si >= 0 && si < Tlen) ? xb[si] : 0.f;
                    }
                }
            };
            const int nfr_c = (LPB_ABL & 4) ? 0 : nfr;
            if (nfr_c > 0) {
                fetchc(0, 0);
                fetchc(1, 1 < nfr_c ? 1 : 0);
            }
            for (int fi = 0; fi < nfr_c; fi += U) {
                int back[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int fu = fi + u < nfr_c ? fi + u : fi;    // (an odd count: the round's second frame is its first again, never added)
                    const double* row = rbuf + (size_t)fu * kLbRow;
                    const float* frow = reinterpret_cast<const float*>(row);
                    const int shx = (int)row[kLpcM1];
                    const int she = reinterpret_cast<const int*>(frow)[kLpcM1];
                    back[u] = -(shx + she);
                    // windowed samples 8 lane .. 8 lane + 7, scaled and split, one 16-byte store per half
                    lp_u4 hr, lr;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const float v0 = 8 * lane + e < L ? c8[u][e] * wc[e] : 0.f, v1 = 8 * lane + e + 1 < L ? c8[u][e + 1] * wc[e + 1] : 0.f;
                        lp_h2 h, l;
                        lp_split2(__builtin_ldexpf(v0, shx), __builtin_ldexpf(v1, shx), h, l);
                        hr[e >> 1] = __builtin_bit_cast(unsigned, h);
                        lr[e >> 1] = __builtin_bit_cast(unsigned, l);
                    }
                    *reinterpret_cast<lp_u4*>(XH + u * kLbOps + 32 + 8 * lane) = hr;
                    *reinterpret_cast<lp_u4*>(XL + u * kLbOps + 32 + 8 * lane) = lr;
                    // taps: E[q], q = 8 + lane (lanes 0 .. 48) = ext[lane - 24] = the row's entry |lane - 24|
                    if (lane < 2 * kLpcM1 - 1) {
                        const int m = lane < kLpcM1 - 1 ? kLpcM1 - 1 - lane : lane - (kLpcM1 - 1);
                        const float tv = frow[m];
                        const _Float16 th = (_Float16)tv;
                        EH[u * kLbOps + 16 + 8 + lane] = th;
                        EL[u * kLbOps + 16 + 8 + lane] = (_Float16)(tv - (float)th);
                    }
                }
                if (fi + U < nfr_c) {
                    fetchc(0, fi + U);
                    fetchc(1, fi + U + 1 < nfr_c ? fi + U + 1 : fi + U);
                }
                __builtin_amdgcn_wave_barrier();
                lp_h8 eh[U][3], el[U][3];
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        // eight taps from band offset 32 t + 8 g - j: read as FIVE aligned dwords from the dword that holds the first one
                        // and funnel-shifted by 0 / 16 bits (lanes j, j + 1 then read the same dwords -- a broadcast -- where the
                        // 2-byte aligned 16-byte read of the first version cost the launch 0.07 of 0.34 ms: ablations LPB_ABL 8 / 64 / 128)
                        const unsigned* eph = etap + (u * kLbOps + 32 * t) / 2;
                        const unsigned* epl = eph + 64;
                        const lp_u4a4 hq = *reinterpret_cast<const lp_u4a4*>(eph), lq = *reinterpret_cast<const lp_u4a4*>(epl);
                        const unsigned h4 = eph[4], l4 = epl[4];
                        const lp_u4 hs = {__builtin_amdgcn_alignbit(hq[1], hq[0], esh), __builtin_amdgcn_alignbit(hq[2], hq[1], esh),
                                          __builtin_amdgcn_alignbit(hq[3], hq[2], esh), __builtin_amdgcn_alignbit(h4, hq[3], esh)};
                        const lp_u4 ls = {__builtin_amdgcn_alignbit(lq[1], lq[0], esh), __builtin_amdgcn_alignbit(lq[2], lq[1], esh),
                                          __builtin_amdgcn_alignbit(lq[3], lq[2], esh), __builtin_amdgcn_alignbit(l4, lq[3], esh)};
                        eh[u][t] = __builtin_bit_cast(lp_h8, hs);
                        el[u][t] = __builtin_bit_cast(lp_h8, ls);
                    }
                f4 acc[U][2];
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u][0] = acc[u][1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    lp_h8 xh[U][2], xl[U][2];
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const int off = u * kLbOps + 16 * (16 * nt + j) + 32 * t + 8 * g;   // (+ 32 of padding, - 32 of the band's reach)
                            xh[u][nt] = *reinterpret_cast<const lp_h8*>(XH + off);
                            xl[u][nt] = *reinterpret_cast<const lp_h8*>(XL + off);
                        }
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) acc[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(el[u][t], xh[u][nt], acc[u][nt], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) acc[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh[u][t], xl[u][nt], acc[u][nt], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) acc[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eh[u][t], xh[u][nt], acc[u][nt], 0, 0, 0);
                }
                // lane (c, g): samples 16 c + 4 g + r of the frame; times the window (zero past the frame), scale undone; frames in order
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (fi + u < nfr_c && !(LPB_ABL & 16)) {
                        const long q0 = (long)(fi + u) * P;
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const int l0 = 16 * (16 * nt + j) + 4 * g;
                            if (vec) {
                                f4* slot = reinterpret_cast<f4*>(ring + ((q0 + l0) & (kLbRing - 1)));
                                f4 cur = *slot;
#pragma unroll
                                for (int r = 0; r < 4; ++r)   // (selected, not multiplied, past the frame: a non-finite frame stays inside its own samples)
                                    cur[r] += wo[nt][r] != 0.f ? __builtin_ldexpf(acc[u][nt][r], back[u]) * wo[nt][r] : 0.f;
                                *slot = cur;
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    ring[(q0 + l0 + r) & (kLbRing - 1)] += wo[nt][r] != 0.f ? __builtin_ldexpf(acc[u][nt][r], back[u]) * wo[nt][r] : 0.f;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        flush(q0, P);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            // what the last frame left beyond its first P positions (all zeros when there was no frame)
            if (nfr_c > 0) flush((long)nfr_c * P, 512 > P ? 512 - P : 0);
            // samples of the own range that no frame reaches (frame_period > frame_length, or no frame at all): zero gradient
            {
                const long covered_lo = nfr_c > 0 ? base : s1, covered_hi = nfr_c > 0 ? base + (long)(nfr_c - 1) * P + 512 : s1;
                for (long sidx = s0 + lane; sidx < s1; sidx += 64)
                    if (sidx < covered_lo || sidx >= covered_hi) gxb[sidx] = 0.f;
            }
        }
    }
}

template <typename T>
static int acorr_fwd_impl(const void* x, int64_t F, int L, int M, int fmt, void* r, hipStream_t st)
{
    size_t lds = sizeof(T) * (size_t)L;
    if (lds > 60 * 1024) return fail(DSA_ERR_UNSUPPORTED, "acorr: frame too long for LDS%s");
    hipLaunchKernelGGL((acorr_fwd_kernel<T>), dim3((unsigned)F), dim3(256), lds, st, (const T*)x, (long)F, L,
                       M, fmt, (T*)r);
    return check_launch("acorr_fwd");
}

template <typename T>
static int levdur_fwd_impl(const void* r, int64_t F, int M, double eps, void* out, hipStream_t st)
{
    if (M <= 63) {
        hipLaunchKernelGGL((levdur_fwd_kernel<T>), dim3((unsigned)((F + 3) / 4)), dim3(256), 0, st, (const T*)r,
                           (long)F, M, eps, (T*)out);
        return check_launch("levdur_fwd");
    }
    size_t lds = sizeof(double) * 3 * (size_t)(M + 1);
    if (lds > 60 * 1024) return fail(DSA_ERR_UNSUPPORTED, "levdur: order too large for LDS%s");
    hipLaunchKernelGGL((levdur_fwd_lds_kernel<T>), dim3((unsigned)F), dim3(64), lds, st, (const T*)r, (long)F, M,
                       eps, (T*)out);
    return check_launch("levdur_fwd_lds");
}

template <typename T>
static int acorr_bwd_impl(const void* gr, const void* x, int64_t F, int L, int M, int fmt, void* gx,
                          hipStream_t st)
{
    size_t lds = (((size_t)L * sizeof(T) + 7) & ~(size_t)7) + sizeof(double) * (size_t)(M + 1);
    if (lds > 60 * 1024) return fail(DSA_ERR_UNSUPPORTED, "acorr_bwd: frame too long for LDS%s");
    hipLaunchKernelGGL((acorr_bwd_kernel<T>), dim3((unsigned)F), dim3(256), lds, st, (const T*)gr,
                       (const T*)x, (long)F, L, M, fmt, (T*)gx);
    return check_launch("acorr_bwd");
}

template <typename T>
static int levdur_bwd_impl(const void* gout, const void* r, const void* out, int64_t F, int M, double eps,
                           void* gr, hipStream_t st)
{
    size_t lds = sizeof(double) * ((size_t)(M + 1) + 3 * (size_t)M + (size_t)M * (M + 1));
    if (lds > 60 * 1024) return fail(DSA_ERR_UNSUPPORTED, "levdur_bwd: order too large for LDS%s");
    hipLaunchKernelGGL((levdur_bwd_kernel<T>), dim3((unsigned)F), dim3(64), lds, st, (const T*)gout,
                       (const T*)r, (const T*)out, (long)F, M, eps, (T*)gr);
    return check_launch("levdur_bwd");
}

}  // namespace dsa

using namespace dsa;

DSA_EXPORT int dsa_acorr_fwd(const void* x, int64_t F, int32_t L, int32_t M, int32_t out_format, int32_t dtype,
                             void* r, void* stream)
{
    DSA_REQUIRE(L > 0 && M >= 0 && M < L && F >= 0, "acorr: acr_order must be less than frame_length");
    DSA_REQUIRE(out_format >= 0 && out_format <= 3, "acorr: unknown out_format");
    if (F == 0) return DSA_OK;
    if (dtype == DSA_F32) return acorr_fwd_impl<float>(x, F, L, M, out_format, r, (hipStream_t)stream);
    if (dtype == DSA_F64) return acorr_fwd_impl<double>(x, F, L, M, out_format, r, (hipStream_t)stream);
    return fail(DSA_ERR_UNSUPPORTED, "acorr: unsupported dtype%s");
}

DSA_EXPORT int dsa_acorr_bwd(const void* gr, const void* x, int64_t F, int32_t L, int32_t M, int32_t out_format,
                             int32_t dtype, void* gx, void* stream)
{
    DSA_REQUIRE(L > 0 && M >= 0 && M < L && F >= 0, "acorr_bwd: acr_order must be less than frame_length");
    if (F == 0) return DSA_OK;
    if (dtype == DSA_F32) return acorr_bwd_impl<float>(gr, x, F, L, M, out_format, gx, (hipStream_t)stream);
    if (dtype == DSA_F64) return acorr_bwd_impl<double>(gr, x, F, L, M, out_format, gx, (hipStream_t)stream);
    return fail(DSA_ERR_UNSUPPORTED, "acorr_bwd: unsupported dtype%s");
}

DSA_EXPORT int dsa_levdur_fwd(const void* r, int64_t F, int32_t M, double eps, int32_t dtype, void* out,
                              void* stream)
{
    DSA_REQUIRE(M >= 0 && F >= 0 && eps >= 0, "levdur: lpc_order and eps must be non-negative");
    if (F == 0) return DSA_OK;
    if (dtype == DSA_F32) return levdur_fwd_impl<float>(r, F, M, eps, out, (hipStream_t)stream);
    if (dtype == DSA_F64) return levdur_fwd_impl<double>(r, F, M, eps, out, (hipStream_t)stream);
    return fail(DSA_ERR_UNSUPPORTED, "levdur: unsupported dtype%s");
}

DSA_EXPORT int dsa_levdur_bwd(const void* gout, const void* r, const void* out, int64_t F, int32_t M, double eps,
                              int32_t dtype, void* gr, void* stream)
{
    DSA_REQUIRE(M >= 0 && F >= 0, "levdur_bwd: lpc_order must be non-negative");
    if (F == 0) return DSA_OK;
    if (dtype == DSA_F32) return levdur_bwd_impl<float>(gout, r, out, F, M, eps, gr, (hipStream_t)stream);
    if (dtype == DSA_F64) return levdur_bwd_impl<double>(gout, r, out, F, M, eps, gr, (hipStream_t)stream);
    return fail(DSA_ERR_UNSUPPORTED, "levdur_bwd: unsupported dtype%s");
}

template <typename T>
static int lpc_fwd_impl(const void* x, int64_t F, int L, int M, double eps, void* out, hipStream_t st)
{
    // a frame is its own one-frame "utterance": T = L, P = L, no padding, unit window not needed
    T* r = nullptr;
    if (hipMallocAsync((void**)&r, sizeof(T) * (size_t)F * (M + 1), st) != hipSuccess)
        return fail(DSA_ERR_LAUNCH, "lpc: workspace allocation failed%s");
    int rc = acorr_fwd_impl<T>(x, F, L, M, DSA_ACORR_NAIVE, r, st);
    if (rc == DSA_OK) rc = levdur_fwd_impl<T>(r, F, M, eps, out, st);
    hipFreeAsync(r, st);
    return rc;
}

template <typename T>
static int lpc_bwd_impl(const void* gout, const void* x, const void* out, int64_t F, int L, int M, double eps,
                        void* gx, hipStream_t st)
{
    T *r = nullptr, *gr = nullptr;
    size_t bytes = sizeof(T) * (size_t)F * (M + 1);
    if (hipMallocAsync((void**)&r, bytes, st) != hipSuccess || hipMallocAsync((void**)&gr, bytes, st) != hipSuccess)
        return fail(DSA_ERR_LAUNCH, "lpc_bwd: workspace allocation failed%s");
    int rc = acorr_fwd_impl<T>(x, F, L, M, DSA_ACORR_NAIVE, r, st);
    if (rc == DSA_OK) rc = levdur_bwd_impl<T>(gout, r, out, F, M, eps, gr, st);
    if (rc == DSA_OK) rc = acorr_bwd_impl<T>(gr, x, F, L, M, DSA_ACORR_NAIVE, gx, st);
    hipFreeAsync(r, st);
    hipFreeAsync(gr, st);
    return rc;
}

DSA_EXPORT int dsa_frame_window_lpc_fwd(const void* x, int64_t B, int64_t T, int32_t L, int32_t P, const void* w,
                                        int32_t center, int32_t pad_mode, int32_t M, double eps, int32_t dtype,
                                        void* scratch, void* out, void* stream);

DSA_EXPORT int dsa_lpc_fwd(const void* x, int64_t F, int32_t L, int32_t M, double eps, int32_t dtype, void* scratch,
                           void* out, void* stream)
{
    DSA_REQUIRE(L > 0 && M >= 0 && M < L && F >= 0 && eps >= 0, "lpc: lpc_order must be less than frame_length");
    if (F == 0) return DSA_OK;
    // float32, order 24, 25 <= L <= 512: the already-framed (and windowed) rows ARE a waveform of F * L samples
    // framed with period L, no centring and a window of ones -- the fused tuned kernel (frame_window_lpc24_kernel)
    // takes it from there: 0.35 ms instead of 1.9 ms per 204 800 frames for the module chain
    // LPC(Window(Frame(x))) of the reference's README.md:198-201.
    // (w = NULL: the window of ones; scratch = NULL: no ticket counter available, the generic kernel runs)
    if (dtype == DSA_F32 && M == 24 && L >= 25 && L <= 512 && scratch)
        return dsa_frame_window_lpc_fwd(x, 1, F * (int64_t)L, L, L, nullptr, 0, DSA_PAD_CONSTANT, M, eps, dtype, scratch, out, stream);
    if (dtype == DSA_F32) return lpc_fwd_impl<float>(x, F, L, M, eps, out, (hipStream_t)stream);
    if (dtype == DSA_F64) return lpc_fwd_impl<double>(x, F, L, M, eps, out, (hipStream_t)stream);
    return fail(DSA_ERR_UNSUPPORTED, "lpc: unsupported dtype%s");
}

DSA_EXPORT int dsa_lpc_bwd(const void* gout, const void* x, const void* out, int64_t F, int32_t L, int32_t M,
                           double eps, int32_t dtype, void* gx, void* stream)
{
    DSA_REQUIRE(L > 0 && M >= 0 && M < L && F >= 0, "lpc_bwd: lpc_order must be less than frame_length");
    if (F == 0) return DSA_OK;
    if (dtype == DSA_F32 && M == 24 && L >= 25 && L <= 512) {
        const int C = (L + 15) >> 4, nblk = (C + kLpcM1 - 1) / kLpcM1;
        const int S = (2 * (kLpcM1 - 1) + 15 * C + kLpcM1 * nblk + 4 + 3) & ~3;  // 24 | reach of the last lane | 24
        const int So = (16 * C + 3) & ~3;
        size_t lds_t = (size_t)(4 * S + 4 * So) * 4 + 64 * kLpcM1 * sizeof(double) +
                       (4 * So >= 64 * kLpcM1 ? 0 : 64 * kLpcM1 * sizeof(float));
        long grid = 256L * 4;   // 280 registers: one wave per SIMD (a 256-register build spills and is slower)
        // frames per work item: 64 fills phase B's lanes, but 204 800 frames / 64 = 3200 items are 3.125 per wave --
        // four rounds with the last one an eighth full.  Pick the multiple of 4 (<= 64) that makes the item count a
        // whole number of rounds (here 52 frames: 3939 items, 3.85 rounds of 13 passes instead of 4 of 16).
        int fpi = 64;
        if (F > grid * 64) {
            const long rounds = (F + grid * 64 - 1) / (grid * 64);
            long f = (F + rounds * grid - 1) / (rounds * grid);
            f = (f + 3) & ~3L;
            if (f >= 16 && f <= 64) fpi = (int)f;
        }
        long total_sc = (long)((F + fpi - 1) / fpi);
        if (grid > total_sc) grid = total_sc;
        hipLaunchKernelGGL(lpc24_bwd_kernel, dim3((unsigned)grid), dim3(64), lds_t, (hipStream_t)stream,
                           (const float*)gout, (const float*)x, (long)F, L, eps, (float*)gx, total_sc, S, So, fpi);
        return check_launch("lpc24_bwd");
    }
    if (dtype == DSA_F32) return lpc_bwd_impl<float>(gout, x, out, F, L, M, eps, gx, (hipStream_t)stream);
    if (dtype == DSA_F64) return lpc_bwd_impl<double>(gout, x, out, F, L, M, eps, gx, (hipStream_t)stream);
    return fail(DSA_ERR_UNSUPPORTED, "lpc_bwd: unsupported dtype%s");
}

DSA_EXPORT int dsa_frame_window_lpc_fwd(const void* x, int64_t B, int64_t T, int32_t L, int32_t P, const void* w,
                                        int32_t center, int32_t pad_mode, int32_t M, double eps, int32_t dtype,
                                        void* scratch, void* out, void* stream)
{
    DSA_REQUIRE(L > 0 && P > 0 && T > 0 && B >= 0 && M >= 0 && M < L && eps >= 0, "frame_window_lpc: invalid sizes");
    const bool scratch_clean = (pad_mode & DSA_LPC_SCRATCH_IS_CLEAN) != 0;   // the caller's scratch is zero and private to this stream
    pad_mode &= ~DSA_LPC_SCRATCH_IS_CLEAN;
    const bool exact_flag = (pad_mode & DSA_LPC_EXACT_LAGSUMS) != 0;   // exact float64 lag sums asked for by the caller
    pad_mode &= ~DSA_LPC_EXACT_LAGSUMS;
    DSA_REQUIRE(pad_mode >= 0 && pad_mode <= 3, "frame_window_lpc: unknown pad mode");
    int64_t N = dsa_num_frames(T, P), F = B * N;
    if (F == 0) return DSA_OK;
    int left = center ? L / 2 : 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSA_F32 && M == 24 && L <= 512 && L >= 25 && scratch) {   // the tuned kernel draws work items from a counter in `scratch`
        int in_floats = ((3 * P + L + 64 + kLpcM1 * 2) + 3) & ~3;
        int wtab_floats = (L + 64 + 3) & ~3;
        size_t lds_t = (size_t)(in_floats + wtab_floats) * 4 + 64 * kLpcM1 * sizeof(double);
        if (lds_t <= 60 * 1024) {
            int waves_per_cu = (int)(144 * 1024 / lds_t);
            if (waves_per_cu > 8) waves_per_cu = 8;
            long grid = 256L * waves_per_cu;
            // frames per work item: close to a whole number of items per wave, utterances split evenly
            int fpi = 64;
            if (F > grid * 64) {
                const long k = (F + grid * 64 - 1) / (grid * 64);
                long f = (F + k * grid - 1) / (k * grid);
                f = (f + 3) & ~3L;
                if (f >= 16 && f <= 64) fpi = (int)f;
            }
            {
                const long cpu = (N + fpi - 1) / fpi;            // chunks per utterance
                long f = ((N + cpu - 1) / cpu + 3) & ~3L;        // ... of equal size
                if (f >= 4 && f <= 64) fpi = (int)f;
            }
            int sc_per_utt = (int)((N + fpi - 1) / fpi);
            long total_sc = (long)B * sc_per_utt;
            if (grid > total_sc) grid = total_sc;
            // lag sums: float32 matrix instruction (default) or the float64 vector unit (DSA_LPC_LAGSUMS=f64: exact sums)
            static const bool exact_env = [] { const char* e = getenv("DSA_LPC_LAGSUMS"); return e && e[0] == 'f' && e[1] == '6'; }();
            const bool exact = exact_env || exact_flag;
            static const bool tickets = [] { const char* e = getenv("DSA_LPC_TICKETS"); return e && e[0] == '1'; }();   // A/B: the ticket counter
            // ticket counter (the float64 kernel; the default kernel deals its items out statically and needs none): the first word of
            // the caller's scratch, zeroed in stream order before the launch unless the caller says it is
            unsigned* queue = (unsigned*)scratch;
            if ((exact || tickets) && !scratch_clean && hipMemsetAsync(queue, 0, sizeof(unsigned), st) != hipSuccess)
                return fail(DSA_ERR_LAUNCH, "frame_window_lpc: cannot reset the ticket counter%s");
            if (!exact && L <= 512) {
                // three workgroups per CU need 4 (200 fpi + 5312) <= 53 KB: at most 41 frames per item, utterances split evenly.
                // (Tried: frames per item chosen so that every one of the 3072 resident waves gets the same number of items -- 34
                // frames, 6144 items at the bench size instead of 40 / 5120: 0.135 -> 0.140 ms; more recursion phases on fewer lanes.)
                if (fpi > 41) {
                    const long cpu = (N + 40) / 41;
                    long f = ((N + cpu - 1) / cpu + 3) & ~3L;
                    if (f > 41) f = 40;
                    fpi = (int)f;
                    sc_per_utt = (int)((N + fpi - 1) / fpi);
                    total_sc = (long)B * sc_per_utt;
                }
                const int lds_m = 4 * (fpi * kLpcM1 * (int)sizeof(double) + 2 * kLsArea * (int)sizeof(float));
                long wgs = (total_sc + 3) / 4;
                const long wg_cap = 256L * (lds_m <= 53 * 1024 ? 3 : 2);   // workgroups of four waves per CU
                if (wgs > wg_cap) wgs = wg_cap;
#define DSA_LPC_MFMA(NEV, LCV)                                                                                                     \
    do {                                                                                                                           \
        static std::atomic<uint64_t> attr_m{0};                                                                                    \
        if (lds_m > 48 * 1024 && !ensure_dynamic_lds((const void*)frame_window_lpc24_mfma_kernel<NEV, LCV>, lds_m, attr_m))        \
            return fail(DSA_ERR_LAUNCH, "frame_window_lpc: cannot reserve LDS%s");                                                 \
        hipLaunchKernelGGL((frame_window_lpc24_mfma_kernel<NEV, LCV>), dim3((unsigned)wgs), dim3(256), lds_m, st, (const float*)x, \
                           (long)T, (long)N, L, P, left, pad_mode, (const float*)w, eps, (float*)out, total_sc, sc_per_utt, tickets ? queue : nullptr, fpi); \
    } while (0)
                if (L == 400) DSA_LPC_MFMA(8, 400);   // the 25 ms window at 16 kHz
                else DSA_LPC_MFMA(8, 0);
#undef DSA_LPC_MFMA
                return check_launch("frame_window_lpc24_mfma_fwd");
            }
            hipLaunchKernelGGL(frame_window_lpc24_kernel, dim3((unsigned)grid), dim3(64), lds_t, st, (const float*)x,
                               (long)T, (long)N, L, P, left, pad_mode, (const float*)w, eps, (float*)out, total_sc,
                               sc_per_utt, in_floats, wtab_floats, queue, fpi);
            return check_launch("frame_window_lpc24_fwd");
        }
    }
    const int nw = 4;
    size_t esz = dtype == DSA_F32 ? 4 : 8;
    size_t lds = (size_t)L * esz * nw;
    if (M > 63) return fail(DSA_ERR_UNSUPPORTED, "frame_window_lpc: fused kernel supports lpc_order <= 63%s");
    if (lds > 60 * 1024) return fail(DSA_ERR_UNSUPPORTED, "frame_window_lpc: frame too long for LDS%s");
    unsigned grid = (unsigned)((F + nw - 1) / nw);
    if (dtype == DSA_F32)
        hipLaunchKernelGGL((frame_window_lpc_kernel<float>), dim3(grid), dim3(64 * nw), lds, st, (const float*)x,
                           (long)T, (long)N, (long)F, L, P, left, pad_mode, (const float*)w, M, eps, (float*)out);
    else if (dtype == DSA_F64)
        hipLaunchKernelGGL((frame_window_lpc_kernel<double>), dim3(grid), dim3(64 * nw), lds, st, (const double*)x,
                           (long)T, (long)N, (long)F, L, P, left, pad_mode, (const double*)w, M, eps, (double*)out);
    else
        return fail(DSA_ERR_UNSUPPORTED, "frame_window_lpc: unsupported dtype%s");
    return check_launch("frame_window_lpc_fwd");
}

DSA_EXPORT int dsa_frame_window_lpc_bwd(const void* gout, const void* x, int64_t B, int64_t T, int32_t L, int32_t P, const void* w,
                                        int32_t center, int32_t pad_mode, int32_t M, double eps, int32_t dtype, void* gx, void* stream)
{
    DSA_REQUIRE(L > 0 && P > 0 && T > 0 && B >= 0 && M >= 0 && M < L && eps >= 0, "frame_window_lpc_bwd: invalid sizes");
    const int64_t N = dsa_num_frames(T, P);
    if (B * N == 0) return DSA_OK;
    DSA_REQUIRE(gout && x && gx, "frame_window_lpc_bwd: null pointer");
    if (!(dtype == DSA_F32 && M == 24 && L >= 25 && L <= 512 && pad_mode == DSA_PAD_CONSTANT))
        return fail(DSA_ERR_UNSUPPORTED, "frame_window_lpc_bwd: the one-launch backward covers float32, lpc_order 24, 25 <= frame_length <= 512, constant padding%s");
    const int left = center ? L / 2 : 0;
    // frames that touch a run of Hc hops: Hc + halo, each a lane of phase B; the overlap-add ring holds frame_length + frame_period
    if (L + P > kLbRing) return fail(DSA_ERR_UNSUPPORTED, "frame_window_lpc_bwd: frame_length + frame_period above 1024%s");
    const long halo = lb_floordiv(left - 1, P) - lb_floordiv(left - L, P);
    long Hc = 64 - halo;
    if (Hc > N) Hc = N;
    if (Hc < 1 || halo > 48)
        return fail(DSA_ERR_UNSUPPORTED, "frame_window_lpc_bwd: this frame_length / frame_period pair does not fit the one-launch backward%s");
    // runs of about 40 hops: LDS for nine waves per CU (the lag-sum rows are 208 bytes per frame) against 5 halo frames per run
    if (Hc > 40 && halo <= 8) Hc = 40;
    long ipu = (N + Hc - 1) / Hc;
    Hc = (N + ipu - 1) / ipu;                      // equal runs per utterance
    const int nfr_max = (int)(Hc + halo);
    const size_t lds = (size_t)nfr_max * kLbRow * sizeof(double) + (size_t)kLbRing * 4 + kLbAreaBytes;
    const long total = (long)B * ipu;
    long per_cu = (long)(160 * 1024 / lds);
    if (per_cu > 8) per_cu = 8;                    // two waves per SIMD (256 registers)
    long grid = 256L * per_cu;
    if (grid > total) grid = total;
    hipStream_t st = (hipStream_t)stream;
#define DSA_LPCB(LCV)                                                                                                                   \
    do {                                                                                                                                \
        static std::atomic<uint64_t> attr_b{0};                                                                                         \
        if (lds > 48 * 1024 && !ensure_dynamic_lds((const void*)frame_window_lpc24_bwd_mfma_kernel<LCV>, (int)lds, attr_b))             \
            return fail(DSA_ERR_LAUNCH, "frame_window_lpc_bwd: cannot reserve LDS%s");                                                  \
        hipLaunchKernelGGL((frame_window_lpc24_bwd_mfma_kernel<LCV>), dim3((unsigned)grid), dim3(64), lds, st, (const float*)gout,      \
                           (const float*)x, (long)T, (long)N, L, P, left, (const float*)w, eps, (float*)gx, total, (int)ipu, (int)Hc,   \
                           nfr_max);                                                                                    \
    } while (0)
    if (L == 400) DSA_LPCB(400);
    else DSA_LPCB(0);
#undef DSA_LPCB
    return check_launch("frame_window_lpc24_bwd_mfma");
}
