// Cepstral analysis by the improved cepstral method:
// CepstralAnalysis._forward, diffsptk/modules/fftcep.py:116-136.
//
// MI355X-first restatement.  Every transform of the reference acts on a real, even sequence, so all of them
// are ONE (H x H) cosine matrix A[k][n] = c_k cos(2 pi k n / L)  (H = L/2 + 1, c = 1 at k = 0 and H - 1, else 2):
//     irfft(log x)[:H]     = (log x) A / L          (fftcep.py:122)
//     hfft(e)[:H]          = e A                    (fftcep.py:127; the output is even, its first H values suffice)
//     ihfft(y).real        = y A / L                (fftcep.py:129)
// so one frame is:  ehat = log(x) A / L;  v = ehat[:N];  e = ehat with [:N] zeroed;  n_iter times
//     y = e A with negatives set to 0;  e2 = y A / L;  t = (1 + accel) e2[:N];  v += t;  e = e2 - pad(t);
// out = v with the end point(s) halved (fftcep.py:134-135).  With n_iter = 0 (the default) only N columns of A
// are touched.  Generic kernel pair (float32 / float64, any L, M): one workgroup per frame, the frame's vectors
// in LDS, thread n owns column n of every product (A is row-major: the reads of a row are coalesced and L2
// resident).  The backward runs the adjoint chain with the clamp masks the forward saved (bit sets), using
// A[n][k] = (c_n / c_k) A[k][n] so that both directions stream A the same way.
#include "common.h"

#include <cstdlib>

namespace dsa {

template <typename T>
__device__ __forceinline__ T fc_weight(int k, int H) { return (k == 0 || k == H - 1) ? T(1) : T(2); }

// dst[n] = scale * sum_{k < klim} src[k] A[k][n] for n < nlim  (all threads; src in LDS; result returned per thread
// for n = threadIdx.x + 256 r through the callback)
template <typename T, typename Fn>
__device__ __forceinline__ void fc_product(const T* __restrict__ A, int H, const T* src, int klim, int nlim, Fn&& sink)
{
    for (int n = threadIdx.x; n < nlim; n += blockDim.x) {
        T s0 = 0, s1 = 0;
        int k = 0;
        for (; k + 1 < klim; k += 2) {
            s0 += src[k] * A[(long)k * H + n];
            s1 += src[k + 1] * A[(long)(k + 1) * H + n];
        }
        if (k < klim) s0 += src[k] * A[(long)k * H + n];
        sink(n, s0 + s1);
    }
}

// The same product through a radix-2 FFT in LDS when L = 2 (H - 1) is a power of two (n_iter > 0: every product is a full
// H x H one): sum_k c_k src[k] cos(2 pi k n / L) is the real part of the L-point transform of the even extension of src.
// 9 butterfly passes instead of 257 rows of A streamed from L2 per product and frame (4.7 -> 0.3 ms per 51 200 frames at
// n_iter = 3).  The twiddles come from row 1 of A (2 cos(2 pi n / L)): no new argument.  Bit-reversed results.
template <typename T>
__device__ __forceinline__ void fc_fft_twiddles(const T* __restrict__ A, int H, T* tw)
{
    const int L = 2 * (H - 1);
    for (int m = threadIdx.x; m < L / 2; m += blockDim.x) {
        const int q = m - L / 4;
        tw[2 * m] = T(0.5) * A[H + m];                        // cos(2 pi m / L)
        tw[2 * m + 1] = T(-0.5) * A[H + (q < 0 ? -q : q)];     // -sin = -cos(2 pi (m - L/4) / L)
    }
}
__device__ __forceinline__ int fc_brev(int k, int lg) { return (int)(__brev((unsigned)k) >> (32 - lg)); }
// Round 3: the even extension of src is REAL, so the L-point transform runs as a complex transform of HALF the length on the
// packed sequence z[n] = x[2n] + i x[2n+1], followed by the split's real part
//   Re X[k] = (Zr[k] + Zr[L/2 - k]) / 2 + cos(2 pi k / L) (Zi[k] + Zi[L/2 - k]) / 2 - sin(2 pi k / L) (Zr[k] - Zr[L/2 - k]) / 2
// (2.25 x fewer butterflies than the full-length transform with a zero imaginary part that this replaced).  Results in
// natural order in fout[0 .. H - 1].
template <typename T>
__device__ __forceinline__ void fc_fft_product(const T* src, int klim, int H, T* fre, T* fim, T* fout, const T* tw)
{
    const int L = 2 * (H - 1), n = L >> 1, lg = 31 - __clz(n);
    for (int m = threadIdx.x; m < n; m += blockDim.x) {
        const int j0 = 2 * m, j1 = 2 * m + 1;
        const int k0 = j0 < H ? j0 : L - j0, k1 = j1 < H ? j1 : L - j1;
        fre[m] = k0 < klim ? src[k0] : T(0);
        fim[m] = k1 < klim ? src[k1] : T(0);
    }
    __syncthreads();
    // lds_fft_pow2 of stft.hip, restated here for this translation unit (forward sign, natural in, bit-reversed out): two
    // radix-2 stages per pass through registers (the four points i0, i0 + h/2, i0 + h, i0 + h + h/2 are closed under
    // stages s and s - 1): half the LDS traffic and barriers; the second pair's twiddle is -i times the first's, stage
    // s - 1's is its square.  The twiddle table is the L-point one: W_n^(j n / 2h) = W_L^(j L / 2h).
    int sft = lg - 1;
    for (; sft >= 1; sft -= 2) {
        const int h = 1 << sft, h2 = h >> 1, tstep = L >> (sft + 1);
        for (int t = threadIdx.x; t < (n >> 2); t += blockDim.x) {
            const int j = t & (h2 - 1);
            const int i0 = ((t >> (sft - 1)) << (sft + 1)) | j;
            const T a0r = fre[i0], a0i = fim[i0], a1r = fre[i0 + h2], a1i = fim[i0 + h2];
            const T a2r = fre[i0 + h], a2i = fim[i0 + h], a3r = fre[i0 + h + h2], a3i = fim[i0 + h + h2];
            const T c1 = tw[2 * (j * tstep)], s1 = tw[2 * (j * tstep) + 1];
            const T c2 = s1, s2 = -c1;
            const T c3 = c1 * c1 - s1 * s1, s3 = T(2) * c1 * s1;
            const T u0r = a0r + a2r, u0i = a0i + a2i, d0r = a0r - a2r, d0i = a0i - a2i;
            const T u1r = a1r + a3r, u1i = a1i + a3i, d1r = a1r - a3r, d1i = a1i - a3i;
            const T v0r = d0r * c1 - d0i * s1, v0i = d0r * s1 + d0i * c1;
            const T v1r = d1r * c2 - d1i * s2, v1i = d1r * s2 + d1i * c2;
            fre[i0] = u0r + u1r;
            fim[i0] = u0i + u1i;
            const T e0r = u0r - u1r, e0i = u0i - u1i;
            fre[i0 + h2] = e0r * c3 - e0i * s3;
            fim[i0 + h2] = e0r * s3 + e0i * c3;
            fre[i0 + h] = v0r + v1r;
            fim[i0 + h] = v0i + v1i;
            const T e1r = v0r - v1r, e1i = v0i - v1i;
            fre[i0 + h + h2] = e1r * c3 - e1i * s3;
            fim[i0 + h + h2] = e1r * s3 + e1i * c3;
        }
        __syncthreads();
    }
    if (sft == 0) {   // odd number of stages: the last one on its own (half = 1, twiddle 1)
        for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
            const int i = t << 1;
            const T ar = fre[i], ai = fim[i], br = fre[i + 1], bi = fim[i + 1];
            fre[i] = ar + br;
            fim[i] = ai + bi;
            fre[i + 1] = ar - br;
            fim[i + 1] = ai - bi;
        }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        T v;
        if (k == 0) {
            v = fre[0] + fim[0];
        } else if (k == n) {
            v = fre[0] - fim[0];
        } else {
            const int pa = fc_brev(k, lg), pb = fc_brev(n - k, lg);
            const T ar = fre[pa], ai = fim[pa], br = fre[pb], bi = -fim[pb];
            const T dr = T(0.5) * (ar - br), di = T(0.5) * (ai - bi);
            v = T(0.5) * (ar + br) + tw[2 * k] * di + tw[2 * k + 1] * dr;   // tw = (cos, -sin)(2 pi k / L)
        }
        fout[k] = v;
    }
    __syncthreads();
}

template <typename T, bool FFT = false>
__global__ __launch_bounds__(256) void fftcep_fwd_kernel(const T* __restrict__ x, long F, int H, int N,
                                                         const T* __restrict__ A, T accel, int n_iter,
                                                         T* __restrict__ out, unsigned long long* __restrict__ masks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem[];
    T* e = reinterpret_cast<T*>(fc_smem);   // [H]
    T* y = e + H;                           // [H]
    T* v = y + H;                           // [N]
    T* fre = v + N;                         // FFT: [L/2] [L/2], the real results [H], the twiddles [L]
    T* fim = fre + (H - 1);
    T* fout = fim + (H - 1);
    T* tw = fout + H;
    const long f = blockIdx.x;
    const T invL = T(1) / T(2 * (H - 1));
    const int W64 = (H + 63) / 64;
    if (FFT) fc_fft_twiddles<T>(A, H, tw);
    for (int k = threadIdx.x; k < H; k += blockDim.x) e[k] = dsa_log(x[f * H + k]);   // fftcep.py:122
    __syncthreads();
    if (FFT) {
        fc_fft_product<T>(e, H, H, fre, fim, fout, tw);
        for (int n = threadIdx.x; n < H; n += blockDim.x) y[n] = fout[n] * invL;
    } else {
        fc_product<T>(A, H, e, H, n_iter > 0 ? H : N, [&](int n, T s) { y[n] = s * invL; });
    }
    __syncthreads();
    for (int n = threadIdx.x; n < H; n += blockDim.x) {   // fftcep.py:123-124
        if (n < N) v[n] = y[n];
        e[n] = n < N ? T(0) : y[n];
    }
    __syncthreads();
    for (int it = 0; it < n_iter; ++it) {
        // y = hfft(e) with negatives cleared (fftcep.py:127-128); the clamp pattern is kept for the backward
        if (FFT) fc_fft_product<T>(e, H, H, fre, fim, fout, tw);
        for (int n0 = 0; n0 < H; n0 += blockDim.x) {
            const int n = n0 + threadIdx.x;
            T s = 0;
            if (FFT) {
                if (n < H) s = fout[n];
            } else if (n < H) {
                T s0 = 0, s1 = 0;
                int k = 0;
                for (; k + 1 < H; k += 2) {
                    s0 += e[k] * A[(long)k * H + n];
                    s1 += e[k + 1] * A[(long)(k + 1) * H + n];
                }
                if (k < H) s0 += e[k] * A[(long)k * H + n];
                s = s0 + s1;
            }
            const bool keep = n < H && !(s < T(0));
            if (n < H) y[n] = s < T(0) ? T(0) : s;
            const unsigned long long bits = __ballot(keep);
            if (masks && (threadIdx.x & 63) == 0 && n0 + (int)threadIdx.x < H)
                masks[(f * n_iter + it) * W64 + ((n0 + threadIdx.x) >> 6)] = bits;
        }
        __syncthreads();
        T r[9];   // e2 = ihfft(y).real (fftcep.py:129): ceil(H / blockDim.x) <= 9 columns per thread (H <= 513, >= 64 threads)
#pragma unroll
        for (int i_ = 0; i_ < 9; ++i_) r[i_] = T(0);
        if (FFT) fc_fft_product<T>(y, H, H, fre, fim, fout, tw);
#pragma unroll
        for (int cnt = 0; cnt < 9; ++cnt) {   // (static register indices: column cnt of this thread)
            const int n = threadIdx.x + cnt * (int)blockDim.x;
            if (n >= H) break;
            T s0 = 0, s1 = 0;
            if (FFT) {
                s0 = fout[n];
            } else {
                int k = 0;
                for (; k + 1 < H; k += 2) {
                    s0 += y[k] * A[(long)k * H + n];
                    s1 += y[k + 1] * A[(long)(k + 1) * H + n];
                }
                if (k < H) s0 += y[k] * A[(long)k * H + n];
            }
            r[cnt] = (s0 + s1) * invL;
        }
        __syncthreads();
#pragma unroll
        for (int cnt = 0; cnt < 9; ++cnt) {   // fftcep.py:130-132
            const int n = threadIdx.x + cnt * (int)blockDim.x;
            if (n >= H) break;
            const T e2 = r[cnt];
            if (n < N) {
                const T t = e2 * (T(1) + accel);
                v[n] += t;
                e[n] = e2 - t;
            } else {
                e[n] = e2;
            }
        }
        __syncthreads();
    }
    for (int n = threadIdx.x; n < N; n += blockDim.x) {   // fftcep.py:134-135
        const bool half = n == 0 || (H == N && n == N - 1);
        out[f * N + n] = half ? T(0.5) * v[n] : v[n];
    }
}

template <typename T, bool FFT = false>
__global__ __launch_bounds__(256) void fftcep_bwd_kernel(const T* __restrict__ gout, const T* __restrict__ x, long F, int H,
                                                         int N, const T* __restrict__ A, T accel, int n_iter,
                                                         const unsigned long long* __restrict__ masks, T* __restrict__ gx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem[];
    T* ge = reinterpret_cast<T*>(fc_smem);   // [H]  cotangent of e, pre-divided by c_k when used as a product input
    T* gy = ge + H;                          // [H]
    T* gv = gy + H;                          // [N]
    T* fre = gv + N;                         // FFT: [L/2] [L/2], the real results [H], the twiddles [L]
    T* fim = fre + (H - 1);
    T* fout = fim + (H - 1);
    T* tw = fout + H;
    const long f = blockIdx.x;
    const T invL = T(1) / T(2 * (H - 1));
    const int W64 = (H + 63) / 64;
    if (FFT) fc_fft_twiddles<T>(A, H, tw);
    for (int n = threadIdx.x; n < H; n += blockDim.x) {
        if (n < N) {
            const bool half = n == 0 || (H == N && n == N - 1);
            gv[n] = (half ? T(0.5) : T(1)) * gout[f * N + n];
        }
        ge[n] = T(0);
    }
    __syncthreads();
    for (int it = n_iter - 1; it >= 0; --it) {
        // through e = e2 - pad(t), v += t, t = (1 + accel) e2[:N]:  ge2[:N] = (1 + accel) gv - accel ge[:N]
        for (int k = threadIdx.x; k < H; k += blockDim.x) {
            const T g2 = k < N ? (T(1) + accel) * gv[k] - accel * ge[k] : ge[k];
            ge[k] = g2 / fc_weight<T>(k, H);
        }
        __syncthreads();
        // e2 = y A / L  =>  gy[n] = sum_k ge2[k] A[n][k] / L = (c_n / L) sum_k (ge2[k] / c_k) A[k][n];  then the clamp
        if (FFT) fc_fft_product<T>(ge, H, H, fre, fim, fout, tw);
        for (int n0 = 0; n0 < H; n0 += blockDim.x) {
            const int n = n0 + threadIdx.x;
            if (n < H) {
                T s0 = 0, s1 = 0;
                if (FFT) {
                    s0 = fout[n];
                } else {
                    int k = 0;
                    for (; k + 1 < H; k += 2) {
                        s0 += ge[k] * A[(long)k * H + n];
                        s1 += ge[k + 1] * A[(long)(k + 1) * H + n];
                    }
                    if (k < H) s0 += ge[k] * A[(long)k * H + n];
                }
                const unsigned long long bits = masks[(f * n_iter + it) * W64 + (n >> 6)];
                const bool keep = (bits >> (n & 63)) & 1ull;
                // stored pre-divided by c_n for the next product: (c_n / L) s / c_n
                gy[n] = keep ? (s0 + s1) * invL : T(0);
            }
        }
        __syncthreads();
        // y = clamp(e A)  =>  ge[n] = sum_k gz[k] A[n][k] = c_n sum_k (gz[k] / c_k) A[k][n]
        T r[9];
#pragma unroll
        for (int i_ = 0; i_ < 9; ++i_) r[i_] = T(0);
        if (FFT) fc_fft_product<T>(gy, H, H, fre, fim, fout, tw);
#pragma unroll
        for (int cnt = 0; cnt < 9; ++cnt) {
            const int n = threadIdx.x + cnt * (int)blockDim.x;
            if (n >= H) break;
            T s0 = 0, s1 = 0;
            if (FFT) {
                s0 = fout[n];
            } else {
                int k = 0;
                for (; k + 1 < H; k += 2) {
                    s0 += gy[k] * A[(long)k * H + n];
                    s1 += gy[k + 1] * A[(long)(k + 1) * H + n];
                }
                if (k < H) s0 += gy[k] * A[(long)k * H + n];
            }
            r[cnt] = (s0 + s1) * fc_weight<T>(n, H);
        }
        __syncthreads();
#pragma unroll
        for (int cnt = 0; cnt < 9; ++cnt) {
            const int n = threadIdx.x + cnt * (int)blockDim.x;
            if (n < H) ge[n] = r[cnt];
        }
        __syncthreads();
    }
    // ehat = log(x) A / L feeds v (first N) and the initial e (the rest)
    for (int k = threadIdx.x; k < H; k += blockDim.x) gy[k] = (k < N ? gv[k] : ge[k]) / fc_weight<T>(k, H);
    __syncthreads();
    const int klim = n_iter > 0 ? H : N;
    if (FFT) fc_fft_product<T>(gy, klim, H, fre, fim, fout, tw);
    for (int n = threadIdx.x; n < H; n += blockDim.x) {
        T s0 = 0, s1 = 0;
        if (FFT) {
            s0 = fout[n];
        } else {
            int k = 0;
            for (; k + 1 < klim; k += 2) {
                s0 += gy[k] * A[(long)k * H + n];
                s1 += gy[k + 1] * A[(long)(k + 1) * H + n];
            }
            if (k < klim) s0 += gy[k] * A[(long)k * H + n];
        }
        gx[f * H + n] = (s0 + s1) * fc_weight<T>(n, H) * invL / x[f * H + n];
    }
}

template <typename T>
static int fftcep_launch(bool bwd, const void* gout, const void* x, int64_t F, int H, int N, const void* A, double accel,
                         int n_iter, void* out, void* masks, void* gx, hipStream_t st)
{
    if (F == 0) return DSA_OK;
    const size_t lds = sizeof(T) * (2 * (size_t)H + N);
    if (H > 512 || lds > 60 * 1024) return fail(DSA_ERR_UNSUPPORTED, "fftcep: fft_length above 1022 is not supported%s");
    const int L = 2 * (H - 1);
    static const bool direct_only = [] {
        const char* e = getenv("DSA_FFTCEP_DIRECT");
        return e && atoi(e) != 0;
    }();
    if (n_iter > 0 && L >= 32 && (L & (L - 1)) == 0 && !direct_only) {   // full H x H products: FFT in LDS
        const size_t lds_fft = lds + sizeof(T) * (2 * (size_t)L + H);   // transform halves, real results, twiddles
        // one wave per frame (DSA_FFTCEP_BLOCK overrides for A/B): the barriers between the butterfly passes become single-wave
        // barriers and a compute unit holds four times as many frames
        static const int forced = [] { const char* e = getenv("DSA_FFTCEP_BLOCK"); return e ? atoi(e) : 0; }();
        const int block = forced > 0 ? forced : 64;
        if (!bwd)
            hipLaunchKernelGGL((fftcep_fwd_kernel<T, true>), dim3((unsigned)F), dim3(block), lds_fft, st, (const T*)x, (long)F, H, N,
                               (const T*)A, (T)accel, n_iter, (T*)out, (unsigned long long*)masks);
        else
            hipLaunchKernelGGL((fftcep_bwd_kernel<T, true>), dim3((unsigned)F), dim3(block), lds_fft, st, (const T*)gout, (const T*)x,
                               (long)F, H, N, (const T*)A, (T)accel, n_iter, (const unsigned long long*)masks, (T*)gx);
        return check_launch(bwd ? "fftcep_fft_bwd" : "fftcep_fft_fwd");
    }
    if (!bwd)
        hipLaunchKernelGGL((fftcep_fwd_kernel<T>), dim3((unsigned)F), dim3(256), lds, st, (const T*)x, (long)F, H, N,
                           (const T*)A, (T)accel, n_iter, (T*)out, (unsigned long long*)masks);
    else
        hipLaunchKernelGGL((fftcep_bwd_kernel<T>), dim3((unsigned)F), dim3(256), lds, st, (const T*)gout, (const T*)x,
                           (long)F, H, N, (const T*)A, (T)accel, n_iter, (const unsigned long long*)masks, (T*)gx);
    return check_launch(bwd ? "fftcep_bwd" : "fftcep_fwd");
}

}  // namespace dsa

using namespace dsa;

DSA_EXPORT int dsa_fftcep_fwd(const void* x, int64_t F, int32_t fft_length, int32_t cep_order, const void* A, double accel,
                              int32_t n_iter, int32_t dtype, void* out, void* masks, void* stream)
{
    DSA_REQUIRE(fft_length > 1 && fft_length % 2 == 0 && cep_order >= 0 && fft_length >= 2 * cep_order && F >= 0,
                "fftcep: cep_order must be less than or equal to fft_length // 2");
    DSA_REQUIRE(accel >= 0 && n_iter >= 0, "fftcep: accel and n_iter must be non-negative");
    const int H = fft_length / 2 + 1, N = cep_order + 1;
    hipStream_t st = (hipStream_t)stream;
    // n_iter = 0: out = log(x) A[:, :N] / L with the first coefficient halved -- the matrix-core front-end kernel
    if (dtype == DSA_F32 && n_iter == 0 && N <= 48 && N < H && H <= 320 && F > 0) {
        const char* g = getenv("DSA_FFTCEP_GENERIC");
        if (!(g && g[0] && g[0] != '0'))
            return fbank_mfma_launch_ex(x, F, H, A, N, H, 1.0, 0.0, 2, 1, 1.0 / fft_length, out, nullptr, st, "fftcep_mfma_fwd");
    }
    if (dtype == DSA_F32) return fftcep_launch<float>(false, nullptr, x, F, H, N, A, accel, n_iter, out, masks, nullptr, st);
    if (dtype == DSA_F64) return fftcep_launch<double>(false, nullptr, x, F, H, N, A, accel, n_iter, out, masks, nullptr, st);
    return fail(DSA_ERR_UNSUPPORTED, "fftcep: unsupported dtype%s");
}

DSA_EXPORT int dsa_fftcep_bwd(const void* gout, const void* x, int64_t F, int32_t fft_length, int32_t cep_order, const void* A,
                              double accel, int32_t n_iter, const void* masks, int32_t dtype, void* gx, void* stream)
{
    DSA_REQUIRE(fft_length > 1 && fft_length % 2 == 0 && cep_order >= 0 && fft_length >= 2 * cep_order && F >= 0,
                "fftcep_bwd: cep_order must be less than or equal to fft_length // 2");
    DSA_REQUIRE(F == 0 || n_iter == 0 || masks, "fftcep_bwd: the clamp masks of the forward are required when n_iter > 0");
    const int H = fft_length / 2 + 1, N = cep_order + 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSA_F32) return fftcep_launch<float>(true, gout, x, F, H, N, A, accel, n_iter, nullptr, (void*)masks, gx, st);
    if (dtype == DSA_F64) return fftcep_launch<double>(true, gout, x, F, H, N, A, accel, n_iter, nullptr, (void*)masks, gx, st);
    return fail(DSA_ERR_UNSUPPORTED, "fftcep_bwd: unsupported dtype%s");
}
