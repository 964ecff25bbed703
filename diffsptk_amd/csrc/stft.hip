// STFT family for gfx950: Frame (a1), Window (a2), fftr (a3), Spectrum (a4), fused STFT (a5).
//
// Two kernel families:
//  * generic  -- any frame length / period / even fft length, float32 and float64, every
//                option of the reference.  One workgroup per frame, direct DFT against a
//                host-built twiddle table.  Correctness path for odd configurations and for
//                float64 (gradcheck); never the fast path.
//  * tuned    -- nfft = 512, float32: the BASELINE configuration.  One workgroup handles 16
//                consecutive frames of one utterance: the waveform stretch they share is read
//                from HBM once into LDS (frames overlap there, not in HBM), each frame is
//                transformed by 16 lanes (4 frames per wave64) as a 256-point complex FFT =
//                radix-16 in registers -> twiddle -> 16x16 transpose through LDS -> radix-16,
//                and the real-FFT split + |.|^2 + eps + formatting is fused into the
//                coalesced write of the (frames x 257) tile.
//                Algorithmic HBM traffic: P*4 B read + 257*4 B written per frame.
//
// Reference semantics: diffsptk/modules/{frame,window,fftr,spec,stft}.py (cited per kernel).
#include "common.h"

namespace dsa {

// =========================================================================== generic kernels

// Frame._forward frame.py:120-141.  grid = F frames, any block size.
template <typename T>
__global__ void frame_fwd_kernel(const T* __restrict__ x, long Tlen, long N, int L, int P, int left,
                                 int zmean, int mode, T* __restrict__ y)
{
    __shared__ T scratch[16];
    long f = blockIdx.x;
    long b = f / N, n = f - b * N;
    const T* xb = x + b * Tlen;
    T* row = y + f * L;
    T acc = 0;
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        T v = load_padded(xb, n * P + l - left, Tlen, mode);
        row[l] = v;
        acc += v;
    }
    if (zmean) {  // frame.py:139-140
        T mean = block_sum(acc, scratch) / T(L);
        for (int l = threadIdx.x; l < L; l += blockDim.x) row[l] -= mean;
    }
}

// Frame without zmean, float32, L % 4 == 0: four samples per thread and a 16-byte store (the rows of y are 16-byte
// aligned then); the four samples come as one 16-byte load when the source run is inside the waveform and aligned
// (P, left multiples of 4), else one by one through the padding rule.  Persistent grid: the one-workgroup-per-frame
// kernel above launches B * N tiny workgroups (0.17 ms per 204 800 frames against 0.07 ms here).
__global__ __launch_bounds__(256) void frame_fwd_vec4_kernel(const float* __restrict__ x, long Tlen, long N, long F, int L, int P,
                                                             int left, int mode, int src_aligned, float* __restrict__ y)
{
    const int L4 = L >> 2;
    const long total = F * L4;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const long f = q / L4;
        const int l = (int)(q - f * L4) << 2;
        const long b = f / N, n = f - b * N;
        const float* xb = x + b * Tlen;
        const long s0 = n * P + l - left;
        float4 v;
        if (src_aligned && s0 >= 0 && s0 + 4 <= Tlen) {
            v = *reinterpret_cast<const float4*>(xb + s0);
        } else {
            v.x = load_padded(xb, s0, Tlen, mode);
            v.y = load_padded(xb, s0 + 1, Tlen, mode);
            v.z = load_padded(xb, s0 + 2, Tlen, mode);
            v.w = load_padded(xb, s0 + 3, Tlen, mode);
        }
        *reinterpret_cast<float4*>(y + f * (long)L + l) = v;
    }
}

// adjoint of Frame: gx[b,t] = sum over (n,l) whose source index is t of g'[b,n,l], where
// g' = gy - mean_l(gy) if zmean.  Gather formulation (deterministic, no atomics) for constant
// padding; the non-constant modes fold several padded positions onto one sample and use a
// per-utterance serial-over-frames scatter within one block (deterministic as well).
template <typename T>
__global__ void frame_bwd_const_kernel(const T* __restrict__ gy, const T* __restrict__ gmean,
                                       long Tlen, long N, int L, int P, int left,
                                       T* __restrict__ gx)
{
    const long tb = (Tlen + blockDim.x - 1) / blockDim.x;   // blocks per utterance: (utterance, block) folded into grid.x
    const long b = blockIdx.x / tb;
    long t = ((long)blockIdx.x - b * tb) * blockDim.x + threadIdx.x;
    if (t >= Tlen) return;
    // frames n with 0 <= t + left - n*P < L
    long p = t + left;
    long n_hi = p / P;
    if (n_hi > N - 1) n_hi = N - 1;
    long n_lo = p - L + 1 <= 0 ? 0 : (p - L + P) / P;  // ceil((p-L+1)/P)
    T acc = 0;
    for (long n = n_lo; n <= n_hi; ++n) {
        long l = p - n * P;
        T g = gy[(b * N + n) * L + l];
        if (gmean) g -= gmean[b * N + n];
        acc += g;
    }
    gx[b * Tlen + t] = acc;
}

template <typename T>
__global__ void row_mean_kernel(const T* __restrict__ g, int L, T* __restrict__ m)
{
    __shared__ T scratch[16];
    long f = blockIdx.x;
    T acc = 0;
    for (int l = threadIdx.x; l < L; l += blockDim.x) acc += g[f * L + l];
    T s = block_sum(acc, scratch);
    if (threadIdx.x == 0) m[f] = s / T(L);
}

// general-mode adjoint: one block per utterance, frames visited in order, each frame's L
// contributions added by distinct threads (a frame never maps two l onto the same t unless
// the padding folds, in which case the fold is resolved by a second serial pass) -- simple and
// deterministic; only used for reflect/replicate/circular padding.
template <typename T>
__global__ void frame_bwd_general_kernel(const T* __restrict__ gy, const T* __restrict__ gmean,
                                         long Tlen, long N, int L, int P, int left, int mode,
                                         T* __restrict__ gx)
{
    long b = blockIdx.x;
    T* gxb = gx + b * Tlen;
    for (long t = threadIdx.x; t < Tlen; t += blockDim.x) gxb[t] = 0;
    __syncthreads();
    // interior (un-folded) part: gather
    for (long t = threadIdx.x; t < Tlen; t += blockDim.x) {
        long p = t + left;
        long n_hi = p / P;
        if (n_hi > N - 1) n_hi = N - 1;
        long n_lo = p - L + 1 <= 0 ? 0 : (p - L + P) / P;
        T acc = 0;
        for (long n = n_lo; n <= n_hi; ++n) {
            T g = gy[(b * N + n) * L + (p - n * P)];
            if (gmean) g -= gmean[b * N + n];
            acc += g;
        }
        gxb[t] = acc;
    }
    __syncthreads();
    // folded part: padded positions i < 0 or i >= T, visited serially by thread 0
    if (threadIdx.x == 0) {
        for (long n = 0; n < N; ++n)
            for (int l = 0; l < L; ++l) {
                long i = n * P + l - left;
                if (i >= 0 && i < Tlen) continue;
                long j = pad_src_index(i, Tlen, mode);
                if (j < 0) continue;
                T g = gy[(b * N + n) * L + l];
                if (gmean) g -= gmean[b * N + n];
                gxb[j] += g;
            }
    }
}

// Window._forward window.py:185-193
template <typename T>
__global__ void window_fwd_kernel(const T* __restrict__ x, long F, int L, const T* __restrict__ w,
                                  int L2, T* __restrict__ y)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = F * L2;
    for (; i < total; i += (long)gridDim.x * blockDim.x) {
        long f = i / L2;
        int l = (int)(i - f * L2);
        y[i] = l < L ? x[f * L + l] * w[l] : T(0);
    }
}

// y = x * w row-wise for float32 rows whose length is a multiple of 4 and needs no padding / cropping (the forward and
// the backward of Window are the same product then): 16-byte accesses and a 32-bit remainder per float4 instead of
// a 64-bit division per element (0.16 -> 0.11 ms per 204 800 frames of 400 samples).
__global__ __launch_bounds__(256) void window_vec4_kernel(const float4* __restrict__ x, unsigned n4, unsigned L4,
                                                          const float4* __restrict__ w, float4* __restrict__ y)
{
    for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += gridDim.x * blockDim.x) {
        const float4 a = x[q], b = w[q % L4];
        y[q] = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
    }
}

static bool window_vec4_ok(const void* a, const void* b, const void* w, int64_t F, int L, int L2)
{
    return L == L2 && (L & 3) == 0 && F * (int64_t)L >= 4096 && F * (int64_t)(L >> 2) < (1LL << 31) &&
           ((((size_t)a) | ((size_t)b) | ((size_t)w)) & 15) == 0;
}

template <typename T>
__global__ void window_bwd_kernel(const T* __restrict__ gy, long F, int L, const T* __restrict__ w,
                                  int L2, T* __restrict__ gx)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = F * L;
    for (; i < total; i += (long)gridDim.x * blockDim.x) {
        long f = i / L;
        int l = (int)(i - f * L);
        gx[i] = l < L2 ? gy[f * L2 + l] * w[l] : T(0);
    }
}

// gw[l] = sum_f gy[f,l] * x[f,l]; one block per l, fixed summation order (deterministic)
template <typename T>
__global__ void window_gw_kernel(const T* __restrict__ gy, const T* __restrict__ x, long F, int L,
                                 int L2, T* __restrict__ gw)
{
    __shared__ T scratch[16];
    int l = blockIdx.x;
    T acc = 0;
    if (l < L2)
        for (long f = threadIdx.x; f < F; f += blockDim.x) acc += gy[f * L2 + l] * x[f * L + l];
    T s = block_sum(acc, scratch);
    if (threadIdx.x == 0) gw[l] = s;
}

// Generic fused row transform: (optional framing) -> (optional zmean) -> (optional window) ->
// direct DFT of length nfft -> formatter.  Covers fftr (fftr.py:136-151), the b-only branch of
// Spectrum (spec.py:165-178) and STFT (stft.py:237-241) for any configuration.
//   out_kind 0: fftr formats (DSA_FFTR_*), 1: spectrum formats (DSA_SPEC_*).
// twiddle: (nfft, 2) = (cos, -sin)(2 pi m / nfft).
// dynamic LDS: Lrow elements of T.
// In-place radix-2 decimation-in-frequency FFT of n = 2^lg complex points held in LDS, run by the whole workgroup
// (forward sign; tw = (cos, -sin)(2 pi m / n)).  Natural-order input, BIT-REVERSED output: X[k] sits at fft_brev(k, lg).
// The generic row transforms use it whenever fft_length is a power of two (FFT = true): the direct sum they fall
// back to costs n^2 / 2 multiply-adds per row -- 8.4 M at the 4096 points the MLSA filter's impulse responses use.
template <typename T>
__device__ __forceinline__ void lds_fft_pow2(T* re, T* im, int n, int lg, const T* __restrict__ tw, int tmul = 1)
{   // tmul: the table is that of length n * tmul (a half-size transform reads every second entry of the full table)
    // Two radix-2 stages at a time (s and s - 1): the four points i0, i0 + h/2, i0 + h, i0 + h + h/2 (bits s and s - 1 of
    // i0 clear) are closed under both, so they pass through registers once -- half the LDS traffic and barriers of
    // stage-by-stage radix 2, a sixth of its twiddle reads (from memory: one per group; the second pair's stage-s twiddle is
    // -i times the first's -- a quarter turn further -- and stage s - 1's is its square), same bit-reversed output.
    int s = lg - 1;
    for (; s >= 1; s -= 2) {
        const int h = 1 << s, h2 = h >> 1;
        const int tstep = (n >> (s + 1)) * tmul;
        for (int t = threadIdx.x; t < (n >> 2); t += blockDim.x) {
            const int j = t & (h2 - 1);
            const int i0 = ((t >> (s - 1)) << (s + 1)) | j;
            const T a0r = re[i0], a0i = im[i0], a1r = re[i0 + h2], a1i = im[i0 + h2];
            const T a2r = re[i0 + h], a2i = im[i0 + h], a3r = re[i0 + h + h2], a3i = im[i0 + h + h2];
            const T c1 = tw[2 * (j * tstep)], s1 = tw[2 * (j * tstep) + 1];   // W^(j tstep): the group's one table read
            const T c2 = s1, s2 = -c1;                                          // W^((j + h/2) tstep) = -i W^(j tstep)
            const T c3 = c1 * c1 - s1 * s1, s3 = T(2) * c1 * s1;                // W^(2 j tstep), the twiddle of stage s - 1
            // stage s
            const T u0r = a0r + a2r, u0i = a0i + a2i, d0r = a0r - a2r, d0i = a0i - a2i;
            const T u1r = a1r + a3r, u1i = a1i + a3i, d1r = a1r - a3r, d1i = a1i - a3i;
            const T v0r = d0r * c1 - d0i * s1, v0i = d0r * s1 + d0i * c1;
            const T v1r = d1r * c2 - d1i * s2, v1i = d1r * s2 + d1i * c2;
            // stage s - 1
            re[i0] = u0r + u1r;
            im[i0] = u0i + u1i;
            const T e0r = u0r - u1r, e0i = u0i - u1i;
            re[i0 + h2] = e0r * c3 - e0i * s3;
            im[i0 + h2] = e0r * s3 + e0i * c3;
            re[i0 + h] = v0r + v1r;
            im[i0 + h] = v0i + v1i;
            const T e1r = v0r - v1r, e1i = v0i - v1i;
            re[i0 + h + h2] = e1r * c3 - e1i * s3;
            im[i0 + h + h2] = e1r * s3 + e1i * c3;
        }
        __syncthreads();
    }
    if (s == 0) {   // odd number of stages: the last one on its own (half = 1, twiddle 1)
        for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
            const int i = t << 1;
            const T ar = re[i], ai = im[i], br = re[i + 1], bi = im[i + 1];
            const T c = tw[0], sn = tw[1];
            re[i] = ar + br;
            im[i] = ai + bi;
            const T dr = ar - br, di = ai - bi;
            re[i + 1] = dr * c - di * sn;
            im[i + 1] = dr * sn + di * c;
        }
        __syncthreads();
    }
}
__device__ __forceinline__ int fft_brev(int k, int lg) { return (int)(__brev((unsigned)k) >> (32 - lg)); }

// dynamic LDS: L elements of T (FFT: + 2 nfft).
template <typename T, bool FFT = false>
__global__ void row_dft_kernel(const T* __restrict__ x, long Tlen, long N, int L, int P, int left,
                               int mode, int zmean, const T* __restrict__ w, int nfft,
                               const T* __restrict__ twiddle, int out_kind, int fmt, T eps,
                               int use_floor, T floor_lin, T* __restrict__ y)
{
    extern __shared__ unsigned char smem_raw[];
    T* xs = reinterpret_cast<T*>(smem_raw);
    T* fre = xs + L;        // FFT only
    T* fim = fre + nfft;
    __shared__ T scratch[16];
    long f = blockIdx.x;
    long b = f / N, n = f - b * N;
    const T* xb = x + b * Tlen;
    T acc = 0;
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        T v = load_padded(xb, n * P + l - left, Tlen, mode);
        xs[l] = v;
        acc += v;
    }
    T mean = 0;
    if (zmean) mean = block_sum(acc, scratch) / T(L);
    __syncthreads();
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        T v = xs[l] - mean;
        xs[l] = w ? v * w[l] : v;
    }
    __syncthreads();
    const int K = nfft / 2 + 1;
    const int Lc = L < nfft ? L : nfft;  // rfft(x, n) crops when the row is longer than n
    const bool inverse_adj = out_kind == 1 && fmt == DSA_SPEC_COMPLEX_INV;   // complex output times c_k / nfft
    const bool complex_out = (out_kind == 0 && fmt == DSA_FFTR_COMPLEX) ||
                             (out_kind == 1 && fmt == DSA_SPEC_COMPLEX) || inverse_adj;
    const int lg = 31 - __clz(nfft);
    if (FFT) {
        for (int l = threadIdx.x; l < nfft; l += blockDim.x) {
            fre[l] = l < Lc ? xs[l] : T(0);
            fim[l] = T(0);
        }
        __syncthreads();
        lds_fft_pow2(fre, fim, nfft, lg, twiddle);
    }
    T smax = 0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        T re = 0, im = 0;
        if (FFT) {
            const int q = fft_brev(k, lg);
            re = fre[q], im = fim[q];
        } else {
            int idx = 0;
            for (int l = 0; l < Lc; ++l) {
                T c = twiddle[2 * idx], s = twiddle[2 * idx + 1];
                re += xs[l] * c;
                im += xs[l] * s;
                idx += k;
                if (idx >= nfft) idx -= nfft;
            }
        }
        if (complex_out) {
            const T sc = inverse_adj ? ((k == 0 || k == K - 1) ? T(1) : T(2)) / T(nfft) : T(1);
            y[(f * K + k) * 2] = re * sc;
            y[(f * K + k) * 2 + 1] = im * sc;
        } else if (out_kind == 0) {
            T v;
            switch (fmt) {
            case DSA_FFTR_REAL: v = re; break;
            case DSA_FFTR_IMAG: v = im; break;
            case DSA_FFTR_AMPLITUDE: v = dsa_sqrt(re * re + im * im); break;
            default: {
                T a = dsa_sqrt(re * re + im * im);  // abs() then square(), fftr.py:119
                v = a * a;
            }
            }
            y[f * K + k] = v;
        } else {
            T a = dsa_sqrt(re * re + im * im);  // fftr amplitude (spec.py:139), then spec.py:173
            T s = a * a + eps;
            if (use_floor) {
                y[f * K + k] = s;  // formatted after the row maximum is known
                smax = s > smax ? s : smax;
            } else {
                y[f * K + k] = spec_format(s, fmt);
            }
        }
    }
    if (out_kind == 1 && use_floor && !complex_out) {  // spec.py:174-176
        T m = block_max(smax, scratch);
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            T s = y[f * K + k];
            T fl = m * floor_lin;
            y[f * K + k] = spec_format(s > fl ? s : fl, fmt);
        }
    }
}

// Backward of row_dft_kernel.  Recomputes X (nothing but x is saved by the forward), forms the
// complex cotangent C[k] = dL/dRe X + i dL/dIm X for the requested format, applies the adjoint
// of the half-spectrum DFT  gxw[l] = sum_k Re(C[k] exp(+i theta k l)), then the adjoints of the
// window multiply and of zmean.  Output: gframe (F, L) = cotangent of the framed samples (the
// overlap-add into the waveform is done by frame_bwd); gwpart (F, L) = per-frame contribution
// to the window gradient (NULL unless the window is learnable).
// dynamic LDS: (L + 3K) elements of T (FFT: + 2 nfft).
template <typename T, bool FFT = false>
__global__ void row_dft_bwd_kernel(const T* __restrict__ x, long Tlen, long N, int L, int P, int left,
                                   int mode, int zmean, const T* __restrict__ w, int nfft,
                                   const T* __restrict__ twiddle, int out_kind, int fmt, T eps,
                                   int use_floor, T floor_lin, const T* __restrict__ gy,
                                   T* __restrict__ gframe, T* __restrict__ gwpart)
{
    extern __shared__ unsigned char smem_raw[];
    T* xc = reinterpret_cast<T*>(smem_raw);
    __shared__ T scratch[16];
    const int K = nfft / 2 + 1;
    const int Lc = L < nfft ? L : nfft;
    T* Cre = xc + L;
    T* Cim = Cre + K;
    T* fre = Cim + 2 * K;   // FFT only (behind the gs array)
    T* fim = fre + nfft;
    const int lg = 31 - __clz(nfft);
    long f = blockIdx.x;
    long b = f / N, n = f - b * N;
    const T* xb = x + b * Tlen;
    T acc = 0;
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        T v = load_padded(xb, n * P + l - left, Tlen, mode);
        xc[l] = v;
        acc += v;
    }
    T mean = 0;
    if (zmean) mean = block_sum(acc, scratch) / T(L);
    __syncthreads();
    for (int l = threadIdx.x; l < L; l += blockDim.x) xc[l] -= mean;
    __syncthreads();
    const bool inverse_cot = out_kind == 1 && fmt == DSA_SPEC_COMPLEX_INV;
    const bool complex_out = (out_kind == 0 && fmt == DSA_FFTR_COMPLEX) ||
                             (out_kind == 1 && fmt == DSA_SPEC_COMPLEX) || inverse_cot;
    // a complex cotangent (format "complex", the inverse transforms) does not depend on the spectrum: no forward transform
    if (FFT && !complex_out) {
        for (int l = threadIdx.x; l < nfft; l += blockDim.x) {
            fre[l] = l < Lc ? (w ? xc[l] * w[l] : xc[l]) : T(0);
            fim[l] = T(0);
        }
        __syncthreads();
        lds_fft_pow2(fre, fim, nfft, lg, twiddle);
    }
    T smax = 0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        T re = 0, im = 0;
        if (complex_out) {
        } else if (FFT) {
            const int q = fft_brev(k, lg);
            re = fre[q], im = fim[q];
        } else {
            int idx = 0;
            for (int l = 0; l < Lc; ++l) {
                T xv = w ? xc[l] * w[l] : xc[l];
                re += xv * twiddle[2 * idx];
                im += xv * twiddle[2 * idx + 1];
                idx += k;
                if (idx >= nfft) idx -= nfft;
            }
        }
        T cr, ci;
        if (complex_out) {
            cr = gy[(f * K + k) * 2];
            ci = gy[(f * K + k) * 2 + 1];
            if (inverse_cot) {   // irfft weights c_k / nfft
                const T ck = ((k == 0 || k == K - 1) ? T(1) : T(2)) / T(nfft);
                cr *= ck;
                ci *= ck;
            }
        } else if (out_kind == 0) {
            T g = gy[f * K + k];
            switch (fmt) {
            case DSA_FFTR_REAL: cr = g; ci = 0; break;
            case DSA_FFTR_IMAG: cr = 0; ci = g; break;
            case DSA_FFTR_AMPLITUDE: {
                T a = dsa_sqrt(re * re + im * im);
                T sc = a > T(0) ? g / a : T(0);
                cr = sc * re; ci = sc * im;
                break;
            }
            default: cr = T(2) * g * re; ci = T(2) * g * im;
            }
        } else {
            // keep (re, im) for now; the cotangent of s needs the row maximum when floored
            cr = re; ci = im;
            T sv = re * re + im * im + eps;
            smax = sv > smax ? sv : smax;
        }
        Cre[k] = cr;
        Cim[k] = ci;
    }
    if (out_kind == 1 && !complex_out) {
        // cotangent of s = |X|^2 + eps through the formatter and the relative floor
        // s' = max(s, m * floor), m = amax(s) (spec.py:173-177): floored bins pass their
        // cotangent (times floor) to the arg-max bin.
        T* gsarr = Cim + K;
        T m = use_floor ? block_max(smax, scratch) : T(0);
        T fl = m * floor_lin;
        __syncthreads();
        T lost = 0;
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            T re = Cre[k], im = Cim[k];
            T sv = re * re + im * im + eps;
            bool floored = use_floor && sv < fl;
            T se = floored ? fl : sv;
            T g = gy[f * K + k];
            T gs;
            switch (fmt) {
            case DSA_SPEC_DB: gs = g * T(4.342944819032518) / se; break;  // 10 / ln 10
            case DSA_SPEC_LOGMAG: gs = g * T(0.5) / se; break;
            case DSA_SPEC_MAG: gs = g * T(0.5) / dsa_sqrt(se); break;
            default: gs = g;
            }
            if (floored) {
                lost += gs;
                gs = 0;
            }
            gsarr[k] = gs;
        }
        T tot = use_floor ? block_sum(lost, scratch) * floor_lin : T(0);
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            T re = Cre[k], im = Cim[k];
            T gs = gsarr[k];
            if (use_floor && (re * re + im * im + eps) == m) gs += tot;
            Cre[k] = T(2) * gs * re;
            Cim[k] = T(2) * gs * im;
        }
    }
    __syncthreads();
    if (FFT) {
        // sum_k Re(C[k] e^{+i theta k l}) = Re FFT(conj(C), zero-extended to nfft points)[l]
        for (int k = threadIdx.x; k < nfft; k += blockDim.x) {
            fre[k] = k < K ? Cre[k] : T(0);
            fim[k] = k < K ? -Cim[k] : T(0);
        }
        __syncthreads();
        lds_fft_pow2(fre, fim, nfft, lg, twiddle);
    }
    T gsum = 0;
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        T g = 0;
        if (FFT) {
            if (l < Lc) g = fre[fft_brev(l, lg)];
        } else if (l < Lc) {
            int idx = 0;
            for (int k = 0; k < K; ++k) {
                g += Cre[k] * twiddle[2 * idx] + Cim[k] * twiddle[2 * idx + 1];
                idx += l;
                if (idx >= nfft) idx -= nfft;
            }
        }
        if (gwpart) gwpart[f * L + l] = g * xc[l];
        T gf = w ? g * w[l] : g;
        gsum += gf;
        gframe[f * L + l] = gf;
    }
    if (zmean) {
        T gm = block_sum(gsum, scratch) / T(L);
        for (int l = threadIdx.x; l < L; l += blockDim.x) gframe[f * L + l] -= gm;
    }
}

// out[l] = sum_f part[f, l] in a fixed order (deterministic window gradient)
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ part, long F, int L, T* __restrict__ out)
{
    __shared__ T scratch[16];
    int l = blockIdx.x;
    T acc = 0;
    for (long f = threadIdx.x; f < F; f += blockDim.x) acc += part[f * L + l];
    T s = block_sum(acc, scratch);
    if (threadIdx.x == 0) out[l] = s;
}

// Spectrum with a denominator (spec.py:160-171): combines |B| and |A| amplitude rows.
// ab:(F,K) or NULL, aa:(F,K) or NULL (at least aa here), gain:(F) = a[:,0].
template <typename T>
__global__ void spec_ratio_kernel(const T* __restrict__ ab, const T* __restrict__ aa,
                                  const T* __restrict__ a, int la, int K, T eps, int use_floor,
                                  T floor_lin, int fmt, T* __restrict__ y)
{
    __shared__ T scratch[16];
    long f = blockIdx.x;
    T gain = a[f * la];
    T smax = 0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        T X = ab ? gain * (ab[f * K + k] / aa[f * K + k]) : gain / aa[f * K + k];
        T s = X * X + eps;
        smax = s > smax ? s : smax;
        y[f * K + k] = use_floor ? s : spec_format(s, fmt);
    }
    if (use_floor) {
        T m = block_max(smax, scratch);
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            T s = y[f * K + k];
            T fl = m * floor_lin;
            y[f * K + k] = spec_format(s > fl ? s : fl, fmt);
        }
    }
}

// Backward of Spectrum with a denominator (spec.py:160-177): X = K |B| / |A| (or K / |A| when b is
// absent), s = X^2 + eps -> floor -> format.  One block per row; B(w), A(w) recomputed by direct DFT.
//   Xbar = 2 X sbar;  |B|bar = Xbar K / |A|;  |A|bar = -Xbar X / |A|;  Kbar = sum_k Xbar X / K
//   bbar[l] = Re sum_k (|B|bar B/|B|) e^{+i theta k l}   (same for a[1:], a[0] = K gets Kbar)
// dynamic LDS: (lb + la + 5K) elements of T.
template <typename T>
__global__ void spec_ratio_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ b, int lb,
                                      const T* __restrict__ a, int la, int nfft, const T* __restrict__ twiddle,
                                      T eps, int use_floor, T floor_lin, int fmt, T* __restrict__ gb,
                                      T* __restrict__ ga)
{
    extern __shared__ unsigned char smem_raw[];
    __shared__ T scratch[16];
    const int K = nfft / 2 + 1;
    T* bs = reinterpret_cast<T*>(smem_raw);
    T* as = bs + lb;            // a1 = [1, a[1:]]
    T* Bre = as + la;
    T* Bim = Bre + K;
    T* Are = Bim + K;
    T* Aim = Are + K;
    T* gsv = Aim + K;           // cotangent of s per bin, later Xbar
    const long f = blockIdx.x;
    const int Lb = lb < nfft ? lb : nfft, La = la < nfft ? la : nfft;
    for (int l = threadIdx.x; l < lb; l += blockDim.x) bs[l] = b ? b[f * lb + l] : T(0);
    for (int l = threadIdx.x; l < la; l += blockDim.x) as[l] = l == 0 ? T(1) : a[f * la + l];
    __syncthreads();
    const T gain = a[f * la];
    T smax = 0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        T br = 0, bi = 0, ar = 0, ai = 0;
        int idx = 0;
        for (int l = 0; l < (Lb > La ? Lb : La); ++l) {
            const T c = twiddle[2 * idx], sn = twiddle[2 * idx + 1];
            if (b && l < Lb) { br += bs[l] * c; bi += bs[l] * sn; }
            if (l < La) { ar += as[l] * c; ai += as[l] * sn; }
            idx += k;
            if (idx >= nfft) idx -= nfft;
        }
        Bre[k] = br; Bim[k] = bi; Are[k] = ar; Aim[k] = ai;
        const T ab = b ? dsa_sqrt(br * br + bi * bi) : T(1), aa = dsa_sqrt(ar * ar + ai * ai);
        const T X = gain * ab / aa;
        const T sv = X * X + eps;
        smax = sv > smax ? sv : smax;
    }
    const T m = use_floor ? block_max(smax, scratch) : T(0);
    const T fl = m * floor_lin;
    __syncthreads();
    T lost = 0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const T ab = b ? dsa_sqrt(Bre[k] * Bre[k] + Bim[k] * Bim[k]) : T(1);
        const T aa = dsa_sqrt(Are[k] * Are[k] + Aim[k] * Aim[k]);
        const T X = gain * ab / aa;
        const T sv = X * X + eps;
        const bool floored = use_floor && sv < fl;
        const T se = floored ? fl : sv;
        T g = gy[f * K + k];
        switch (fmt) {
        case DSA_SPEC_DB: g *= T(4.342944819032518) / se; break;
        case DSA_SPEC_LOGMAG: g *= T(0.5) / se; break;
        case DSA_SPEC_MAG: g *= T(0.5) / dsa_sqrt(se); break;
        default: break;
        }
        if (floored) { lost += g; g = 0; }
        gsv[k] = g;
    }
    const T tot = use_floor ? block_sum(lost, scratch) * floor_lin : T(0);
    __syncthreads();
    T kacc = 0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const T ab = b ? dsa_sqrt(Bre[k] * Bre[k] + Bim[k] * Bim[k]) : T(1);
        const T aa = dsa_sqrt(Are[k] * Are[k] + Aim[k] * Aim[k]);
        const T X = gain * ab / aa;
        T gs = gsv[k];
        if (use_floor && (X * X + eps) == m) gs += tot;
        const T Xbar = T(2) * X * gs;
        kacc += Xbar * ab / aa;                       // dX/dK = |B|/|A|
        const T abar_b = Xbar * gain / aa;            // d/d|B|
        const T abar_a = -Xbar * X / aa;              // d/d|A|
        // complex cotangents of B and A through the amplitude (0 at an exact zero, like torch.abs)
        const T sb = (b && ab > T(0)) ? abar_b / ab : T(0), sa = aa > T(0) ? abar_a / aa : T(0);
        Bre[k] *= sb; Bim[k] *= sb; Are[k] *= sa; Aim[k] *= sa;
    }
    const T kbar = block_sum(kacc, scratch);
    __syncthreads();
    for (int l = threadIdx.x; l < (lb > la ? lb : la); l += blockDim.x) {
        T accb = 0, acca = 0;
        if (l < nfft) {
            int idx = 0;
            for (int k = 0; k < K; ++k) {
                const T c = twiddle[2 * idx], sn = twiddle[2 * idx + 1];
                accb += Bre[k] * c + Bim[k] * sn;
                acca += Are[k] * c + Aim[k] * sn;
                idx += l;
                if (idx >= nfft) idx -= nfft;
            }
        }
        if (gb && l < lb) gb[f * lb + l] = accb;
        if (l < la) ga[f * la + l] = l == 0 ? kbar : acca;
    }
}

// remove_gain (utils/private.py:200-209): a1 = [1, a[1:]]
template <typename T>
__global__ void remove_gain_kernel(const T* __restrict__ a, long F, int la, T* __restrict__ a1)
{
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * la) return;
    a1[i] = (i % la == 0) ? T(1) : a[i];
}

// =========================================================================== tuned rFFT-512 path

struct alignas(8) cf {   // 8-byte aligned: LDS accesses of a complex value become one ds_read_b64 / ds_write_b64 (not read2_b32 pairs)
    float re, im;
};
__device__ __forceinline__ cf operator+(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf operator-(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf cmul(cf a, cf b)
{
    return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

// 4-point DFT in place, forward (W4 = -i); INV conjugates the kernel.
template <bool INV>
__device__ __forceinline__ void dft4(cf& a0, cf& a1, cf& a2, cf& a3)
{
    cf s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13;
    a2 = s02 - s13;
    if (!INV) {
        a1 = {d02.re + d13.im, d02.im - d13.re};  // d02 - i d13
        a3 = {d02.re - d13.im, d02.im + d13.re};  // d02 + i d13
    } else {
        a1 = {d02.re - d13.im, d02.im + d13.re};
        a3 = {d02.re + d13.im, d02.im - d13.re};
    }
}

// 16-point DFT in registers (radix 4 x 4).  Output order: X[k] sits in v[4*(k&3) + (k>>2)].
template <bool INV>
__device__ __forceinline__ void fft16(cf (&v)[16])
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    constexpr float sg = INV ? 1.f : -1.f;  // sign of the imaginary part of W16^e
#pragma unroll
    for (int n0 = 0; n0 < 4; ++n0) dft4<INV>(v[n0], v[n0 + 4], v[n0 + 8], v[n0 + 12]);
    // after the first pass v[n0 + 4q] = B[n0][q]; twiddle by W16^(n0*q)
    v[1 + 4 * 1] = cmul(v[1 + 4 * 1], cf{C1, sg * S1});   // e = 1
    v[1 + 4 * 2] = cmul(v[1 + 4 * 2], cf{R2, sg * R2});   // e = 2
    v[1 + 4 * 3] = cmul(v[1 + 4 * 3], cf{S1, sg * C1});   // e = 3
    v[2 + 4 * 1] = cmul(v[2 + 4 * 1], cf{R2, sg * R2});   // e = 2
    v[2 + 4 * 2] = cf{-sg * v[2 + 4 * 2].im, sg * v[2 + 4 * 2].re};   // e = 4: (0, sg) * v
    v[2 + 4 * 3] = cmul(v[2 + 4 * 3], cf{-R2, sg * R2});  // e = 6
    v[3 + 4 * 1] = cmul(v[3 + 4 * 1], cf{S1, sg * C1});   // e = 3
    v[3 + 4 * 2] = cmul(v[3 + 4 * 2], cf{-R2, sg * R2});  // e = 6
    v[3 + 4 * 3] = cmul(v[3 + 4 * 3], cf{-C1, -sg * S1}); // e = 9
#pragma unroll
    for (int q = 0; q < 4; ++q) dft4<INV>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}
#define FFT16_OUT(k) (4 * ((k)&3) + ((k) >> 2))

constexpr int kFPW = 4;         // frames per wave (16 lanes each)
constexpr int kTile = kFPW * 257;  // floats in one output tile (4 rows)
// Per-frame stride of the forward kernel's complex tile: ODD, so that the 8-byte transposed reads of the two
// frames a 32-lane LDS service group covers land on opposite bank parities (stride 256: 2-way conflict on
// every one of the 16 reads; PMC: 39 % of the kernel's LDS cycles were conflict cycles).
constexpr int kZS = 272;   // = 16 x 17: the padded transpose tile of a frame (see the forward kernel)

#ifdef DSA_STFT_TIMING
__device__ unsigned long long g_stft_stamps[16];
#define STFT_STAMP(i)                                                                 \
    do {                                                                              \
        if (wid == 0 && lane == 0 && c == nw)                                         \
            g_stft_stamps[i] = __builtin_readcyclecounter();                          \
    } while (0)
#else
#define STFT_STAMP(i)
#endif

// ShortTimeFourierTransform._forward stft.py:237-241 for nfft = 512, float32.
// One wave64 per workgroup, autonomous (no inter-wave barriers): it owns kFPW = 4 consecutive
// frames of one utterance per pass -- the 3P + L samples they share are read from HBM once into
// LDS -- and writes their 4 x 257 output rows as one contiguous, 16-byte aligned run of float4.
// LDS is kept to ~10 KB per wave so that 12+ waves fit a CU (the pass is a long dependent chain;
// throughput comes from waves in flight):
//   zbuf[kFPW][256] cf : (a) first the input stretch (3P + L floats), (b) then the 16 x 16
//                        transpose tiles (row stride 17: element (k1, j) at k1*17 + j),
//                        (c) then the spectra Z in natural order, (d) finally the staged 4 x 257
//                        output tile -- each use is dead before the next begins;
//   t256[16][16] cf    : W256^(j k1), shared by the 4 frames;   fmax[kFPW].
// ABL: 0 production | 1 no output stores | 2 no FFT butterflies | 3 no input staging
//      (ablation knob for tools/bench_stft.cpp)
// The workgroup IS one wave: LDS operations of a wave execute in order, so the phases only need a compiler
// fence.  (__syncthreads() = s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: its vmcnt(0) made every pass wait for the
// previous pass's output stores before touching LDS.)
#define DSA_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
// PLAIN: power format, no relative floor, constant padding, fixed at compile time (the bench path): the format
// branches and the per-frame maxima leave the register allocation.
// LC: frame length fixed at compile time (0 = runtime).  With LC = 400 the selects that cut a lane's 32 samples at
// the frame end fold away for 15 of the 16 sample pairs, and the three pairs past the frame are constant zeros
// that the compiler propagates through the first FFT stage.
// The PLAIN instantiations need < 128 registers (stft.hip is built without packed-float32 selection), so four
// waves fit a SIMD; LDS is what limits them then, so two waves share a workgroup and with it the 2 KB twiddle table
// (8 workgroups x 19.5 KB per CU).  The waves stay autonomous: each fills the whole table itself (identical values)
// before its first use, and no barrier is ever needed.
template <int ABL, bool ZMEAN, bool PLAIN = false, int LC = 0>
__global__ __launch_bounds__(128, 4) void stft512_fwd_kernel(
    const float* __restrict__ x, long Tlen, long N, int L, int P, int left, int mode_arg,
    const float* __restrict__ w, const float* __restrict__ twiddle, float eps, int use_floor_arg,
    float floor_lin, int fmt_arg, float* __restrict__ y, long total_chunks, int chunks_per_utt,
    int io_floats)
{
    const int use_floor = PLAIN ? 0 : use_floor_arg;
    const int fmt = PLAIN ? (int)DSA_SPEC_POWER : fmt_arg;
    const int mode = PLAIN ? (int)DSA_PAD_CONSTANT : mode_arg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int WPB = 2;   // waves per workgroup
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    cf* zbuf = reinterpret_cast<cf*>(smem_raw) + wv * kFPW * kZS;
    float* io_buf = reinterpret_cast<float*>(zbuf);  // aliases zbuf (see above)
    cf* t256 = reinterpret_cast<cf*>(smem_raw) + WPB * kFPW * kZS;
    float* fmax = reinterpret_cast<float*>(t256 + 256) + wv * kFPW;
    (void)io_floats;
    const long wid = (long)blockIdx.x * WPB + wv, nw = (long)gridDim.x * WPB;   // this wave, all waves

    const int lane = threadIdx.x & 63;
    const int j = lane & 15;   // lane within the frame group
    const int fl = lane >> 4;  // frame slot within the pass (0..3)

    // per-wave constants: per-lane window, W256^(j*k1) table, split twiddles of this lane's bins
    float wreg[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        int l = 2 * j + 32 * (r >> 1) + (r & 1);
        wreg[r] = l < L ? w[l] : 0.f;
    }
    for (int i = lane; i < 256; i += 64) {
        int m = 2 * (i & 15) * (i >> 4);  // entry [k1 = i >> 4][j = i & 15]: W256^(j k1) = W512^(2 j k1)
        // HALVED: the factor 1/2 of the real-FFT split X[k] = (S + ...) / 2 rides on the twiddle (exact: a power of two)
        t256[i] = cf{0.5f * twiddle[2 * m], 0.5f * twiddle[2 * m + 1]};
    }
    const cf twA = cf{twiddle[2 * (lane + 1)], twiddle[2 * (lane + 1) + 1]};    // W512^(lane+1)
    const cf twB = cf{twiddle[2 * (lane + 65)], twiddle[2 * (lane + 65) + 1]};  // W512^(lane+65)
    const float inv_L = 1.f / (float)L;
    const int K = 257;
    const bool complex_out = fmt == DSA_SPEC_COMPLEX || fmt == DSA_SPEC_COMPLEX_INV;
    // DSA_SPEC_COMPLEX_INV as a FORWARD format: the complex spectrum times c_k / 512 (the adjoint of the inverse
    // transform's weights: the backward of dsa_istft_fwd)
    const float osc = fmt == DSA_SPEC_COMPLEX_INV ? 2.f / 512.f : 1.f, osc_edge = fmt == DSA_SPEC_COMPLEX_INV ? 1.f / 512.f : 1.f;
    cf* zf = zbuf + fl * kZS;

    // (utterance, chunk) of pass c advance incrementally: one 64-bit division per wave instead of one per pass
    long b = wid / chunks_per_utt;
    int ci = (int)(wid - b * chunks_per_utt);
    const long b_step = nw / chunks_per_utt;
    const int ci_step = (int)(nw - b_step * chunks_per_utt);
    // PLAIN: the NEXT pass's stretch is fetched into registers (3 x float4 per lane) while this pass computes, so
    // the HBM round trip leaves the dependent chain of a pass; possible because this instantiation does not spill.
    float4 pre0 = make_float4(0.f, 0.f, 0.f, 0.f), pre1 = pre0, pre2 = pre0;
    bool pre_ok = false;
    for (long c = wid; c < total_chunks; c += nw, b += b_step, ci += ci_step) {
        if (ci >= chunks_per_utt) {
            ci -= chunks_per_utt;
            ++b;
        }
        const long frame0 = (long)ci * kFPW;
        const int nvalid = (int)((N - frame0) < kFPW ? (N - frame0) : kFPW);
        const float* xb = x + b * Tlen;
        DSA_WAVE_SYNC();  // previous pass is done with the LDS tile (single-wave workgroup)
        STFT_STAMP(0);
        // ---- stage the shared waveform stretch (each sample read from HBM once) ----
        if (ABL != 3) {
            const long g0 = frame0 * P - left;
            const int need = (nvalid - 1) * P + L;  // samples the valid frames touch
            const bool interior = g0 >= 0 && g0 + need <= Tlen;
            if (PLAIN && pre_ok) {
                float4* dst4 = reinterpret_cast<float4*>(io_buf);
                const int n4 = need >> 2;
                if (lane < n4) dst4[lane] = pre0;
                if (lane + 64 < n4) dst4[lane + 64] = pre1;
                if (lane + 128 < n4) dst4[lane + 128] = pre2;
            } else if (interior && (((size_t)(xb + g0)) & 15) == 0) {
                const float4* src4 = reinterpret_cast<const float4*>(xb + g0);
                float4* dst4 = reinterpret_cast<float4*>(io_buf);
                const int n4 = need >> 2;
                for (int s = lane; s < n4; s += 64) dst4[s] = src4[s];
                for (int s = (n4 << 2) + lane; s < need; s += 64) io_buf[s] = xb[g0 + s];
            } else {
                for (int s = lane; s < need; s += 64) io_buf[s] = load_padded(xb, g0 + s, Tlen, mode);
            }
        }
        DSA_WAVE_SYNC();
        STFT_STAMP(1);
        if (PLAIN && ABL != 3) {   // issue the next pass's loads (no wait here)
            long b2 = b + b_step;
            int ci2 = ci + ci_step;
            if (ci2 >= chunks_per_utt) {
                ci2 -= chunks_per_utt;
                ++b2;
            }
            pre_ok = false;
            if (c + nw < total_chunks) {
                const long fr2 = (long)ci2 * kFPW;
                const int nv2 = (int)((N - fr2) < kFPW ? (N - fr2) : kFPW);
                const long g2 = fr2 * P - left;
                const int need2 = (nv2 - 1) * P + L;
                const float* xb2 = x + b2 * Tlen;
                if (g2 >= 0 && g2 + need2 <= Tlen && (((size_t)(xb2 + g2)) & 15) == 0 && (need2 & 3) == 0 && need2 <= 768) {
                    const float4* src4 = reinterpret_cast<const float4*>(xb2 + g2);
                    const int n4 = need2 >> 2;
                    pre0 = src4[lane < n4 ? lane : n4 - 1];
                    pre1 = src4[lane + 64 < n4 ? lane + 64 : n4 - 1];
                    pre2 = src4[lane + 128 < n4 ? lane + 128 : n4 - 1];
                    pre_ok = true;
                }
            }
        }
        // ---- per frame: window, 256-point complex FFT (16 lanes x 16 points) ----
        cf v[16];
        {
            const float* src = io_buf + fl * P + 2 * j;
            int lim = (LC ? LC : L) - 2 * j;  // element (m1, c) belongs to the frame iff 32 m1 + c < lim
            // recomputed per pass on purpose: hoisted out of the pass loop, the 32 lane masks of the selects below
            // occupy 64 scalar registers for the whole kernel and push the loop's scalars into spills
            if (!LC) asm volatile("" : "+v"(lim));
            float sum = 0.f;
            // all 16 LDS reads are issued back to back (reading past the frame stays inside the tile);
            // samples past the frame are then selected away, never multiplied: zero padding is exact
            // and non-finite neighbours stay out of frames that do not contain them
            float2 raw[16];
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) raw[m1] = make_float2(src[32 * m1], src[32 * m1 + 1]);
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) {
                const float a0 = 32 * m1 < lim ? raw[m1].x : 0.f;
                const float a1 = 32 * m1 + 1 < lim ? raw[m1].y : 0.f;
                v[m1] = cf{a0, a1};
                if (ZMEAN) sum += a0 + a1;
            }
            float mean = 0.f;
            if (ZMEAN) {  // frame.py:139-140: mean over the L samples of the frame
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
                mean = sum * inv_L;
            }
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) {
                float a0 = v[m1].re, a1 = v[m1].im;
                if (ZMEAN) {
                    a0 = 32 * m1 < lim ? a0 - mean : 0.f;
                    a1 = 32 * m1 + 1 < lim ? a1 - mean : 0.f;
                }
                v[m1] = cf{a0 * wreg[2 * m1], a1 * wreg[2 * m1 + 1]};  // window.py:190 (wreg = 0 past L)
            }
        }
        DSA_WAVE_SYNC();  // every lane has its samples: the stretch may be overwritten
        STFT_STAMP(2);
        if (ABL != 2) fft16<false>(v);
        STFT_STAMP(3);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1)  // twiddle, then transposed store: (k1, j) -> k1*17 + j (row stride 17: every
            zf[k1 * 17 + j] = cmul(v[FFT16_OUT(k1)], t256[k1 * 16 + j]);   // address is lane base + immediate, no XOR math)
        DSA_WAVE_SYNC();
        STFT_STAMP(4);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = zf[j * 17 + i];  // lane k1 = j reads A[i][k1]: 34 j floats apart, 16 distinct bank pairs
        DSA_WAVE_SYNC();
        STFT_STAMP(5);
        if (ABL != 2) fft16<false>(v);
        STFT_STAMP(6);
#pragma unroll
        for (int k0 = 0; k0 < 16; ++k0) zf[j + 16 * k0] = v[FFT16_OUT(k0)];  // Z[k1 + 16 k0], natural order
        DSA_WAVE_SYNC();
        STFT_STAMP(7);
        // ---- real-FFT split, two bins (k, 256-k) per lane from one pair (Z[k], Z[256-k]) ----
        //   S = a + conj(b), Dd = a - conj(b), Pp = W Dd:
        //   2 X[k] = (S.re + Pp.im, S.im - Pp.re),  2 X[256-k] = (S.re - Pp.im, -S.im - Pp.re)
        // All pairs are read before anything is written: the staged tile reuses the same LDS.
        const long row0 = b * N + frame0;
        const long out0 = row0 * K;
        float* stage = io_buf;
        float2* y2 = reinterpret_cast<float2*>(y);
        // pairs (k, 256 - k) for k = 1..128: part 0 -> k = lane + 1, part 1 -> k = lane + 65 (lane 63
        // gets the self-pair k = 128); bins 0 and 256 come from Z[0] alone: X[0] = re + im, X[256] = re - im
        cf pa[kFPW][2], pb[kFPW][2], z0[kFPW];
#pragma unroll
        for (int f = 0; f < kFPW; ++f) {
            const cf* z = zbuf + f * kZS;
            pa[f][0] = z[lane + 1];
            pb[f][0] = z[255 - lane];
            pa[f][1] = z[lane + 65];
            pb[f][1] = z[191 - lane];
            z0[f] = z[0];
        }
        DSA_WAVE_SYNC();
        float fm[kFPW] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < kFPW; ++f) {
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const int k = part == 0 ? lane + 1 : lane + 65;
                const cf W = part == 0 ? twA : twB;
                const cf a = pa[f][part], bq = pb[f][part];
                const cf S = {a.re + bq.re, a.im - bq.im};
                const cf Dd = {a.re - bq.re, a.im + bq.im};
                const cf Pp = cmul(W, Dd);
                const cf X1 = {S.re + Pp.im, S.im - Pp.re};     // Z arrives halved (see t256)
                const cf X2 = {S.re - Pp.im, -S.im - Pp.re};
                if (complex_out) {
                    if (f < nvalid && ABL != 1) {
                        y2[out0 + f * K + k] = make_float2(X1.re * osc, X1.im * osc);
                        y2[out0 + f * K + 256 - k] = make_float2(X2.re * osc, X2.im * osc);
                    }
                } else {
                    const float s1 = X1.re * X1.re + X1.im * X1.im + eps;  // spec.py:173
                    const float s2 = X2.re * X2.re + X2.im * X2.im + eps;
                    stage[f * K + k] = s1;
                    stage[f * K + 256 - k] = s2;
                    if (use_floor) {
                        const float mx = s1 > s2 ? s1 : s2;
                        fm[f] = mx > fm[f] ? mx : fm[f];
                    }
                }
            }
            // the two real-valued end bins
            const float x0 = 2.f * (z0[f].re + z0[f].im), x256 = 2.f * (z0[f].re - z0[f].im);   // these two take Z[0] whole
            if (complex_out) {
                if (f < nvalid && ABL != 1 && lane == 0) {
                    y2[out0 + f * K] = make_float2(x0 * osc_edge, 0.f);
                    y2[out0 + f * K + 256] = make_float2(x256 * osc_edge, 0.f);
                }
            } else {
                const float s0 = x0 * x0 + eps, s256 = x256 * x256 + eps;
                if (lane == 0) {
                    stage[f * K] = s0;
                    stage[f * K + 256] = s256;
                }
                if (use_floor) {
                    const float mx = s0 > s256 ? s0 : s256;
                    fm[f] = mx > fm[f] ? mx : fm[f];
                }
            }
        }
        if (complex_out) continue;
        if (use_floor) {  // per-frame maximum for the relative floor (spec.py:174-176)
#pragma unroll
            for (int f = 0; f < kFPW; ++f) {
                float m = wave_max(fm[f]);
                if (lane == 0) fmax[f] = m;
            }
        }
        DSA_WAVE_SYNC();
        STFT_STAMP(8);
        // ---- formatter + coalesced write of the staged tile ----
        const bool plain = !use_floor && fmt == DSA_SPEC_POWER;
        if (nvalid == kFPW && (row0 & 3) == 0) {
            // 4 rows x 257 floats = 257 float4, 16-byte aligned because row0 % 4 == 0
            float4* y4 = reinterpret_cast<float4*>(y + out0);
            const float4* s4 = reinterpret_cast<const float4*>(stage);
#pragma unroll
            for (int jj = 0; jj < 5; ++jj) {
                const int t = lane + 64 * jj;
                if (t < K) {
                    float4 q = s4[t];
                    if (!plain) {
                        float o4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            const int idx = 4 * t + cc;
                            float sv = o4[cc];
                            if (use_floor) {
                                const int f = idx >= 3 * K ? 3 : (idx >= 2 * K ? 2 : (idx >= K ? 1 : 0));
                                const float flv = fmax[f] * floor_lin;
                                sv = sv > flv ? sv : flv;
                            }
                            o4[cc] = spec_format(sv, fmt);
                        }
                        q = make_float4(o4[0], o4[1], o4[2], o4[3]);
                    }
                    if (ABL != 1 || q.x == 123.456f) y4[t] = q;
                }
            }
        } else {
            const int total = nvalid * K;
            for (int idx = lane; idx < total; idx += 64) {
                float sv = stage[idx];
                if (use_floor) {
                    const int f = idx >= 3 * K ? 3 : (idx >= 2 * K ? 2 : (idx >= K ? 1 : 0));
                    const float flv = fmax[f] * floor_lin;
                    sv = sv > flv ? sv : flv;
                }
                y[out0 + idx] = spec_format(sv, fmt);
            }
        }
        STFT_STAMP(9);
    }
}

}  // namespace dsa
#include "stft_pk.h"
#include "stft_bwd_pk.h"
#include "stft_pk_big.h"
#include "stft_bwd_pk_big.h"
namespace dsa {

// ------------------------------------------------------------------ host-side dispatch helpers
template <typename T>
static int launch_row_dft(const void* x, int64_t B, int64_t Tlen, int64_t N, int L, int P, int left,
                          int mode, int zmean, const void* w, int nfft, const void* twiddle,
                          int out_kind, int fmt, double eps, int use_floor, double floor_db,
                          void* y, hipStream_t st)
{
    int64_t F = B * N;
    if (F == 0) return DSA_OK;
    T floor_lin = use_floor ? (T)pow(10.0, floor_db / 10.0) : T(0);
    size_t lds = sizeof(T) * (size_t)L;
    int threads = nfft / 2 + 1 >= 192 ? 256 : (nfft / 2 + 1 >= 96 ? 128 : 64);
    // power-of-two lengths: radix-2 FFT in LDS (DSA_ROWDFT_DIRECT=1 keeps the direct sum, for A/B runs and tests)
    static const bool direct_only = [] {
        const char* e = getenv("DSA_ROWDFT_DIRECT");
        return e && atoi(e) != 0;
    }();
    const size_t lds_fft = lds + sizeof(T) * 2 * (size_t)nfft;
    if (!direct_only && nfft >= 32 && (nfft & (nfft - 1)) == 0 && lds_fft <= 150 * 1024) {
        static std::atomic<uint64_t> lds_set{0};
        if (lds_fft > 48 * 1024 &&
            !ensure_dynamic_lds(reinterpret_cast<const void*>(&row_dft_kernel<T, true>), 150 * 1024, lds_set))
            return fail(DSA_ERR_LAUNCH, "row_fft: cannot raise the dynamic LDS limit%s");
        hipLaunchKernelGGL((row_dft_kernel<T, true>), dim3((unsigned)F), dim3(nfft >= 512 ? 256 : threads), lds_fft, st,
                           (const T*)x, (long)Tlen, (long)N, L, P, left, mode, zmean, (const T*)w, nfft, (const T*)twiddle,
                           out_kind, fmt, (T)eps, use_floor, floor_lin, (T*)y);
        return check_launch("row_fft_generic");
    }
    if (lds > 60 * 1024) return fail(DSA_ERR_UNSUPPORTED, "row_dft: frame too long for LDS%s");
    hipLaunchKernelGGL((row_dft_kernel<T>), dim3((unsigned)F), dim3(threads), lds, st, (const T*)x,
                       (long)Tlen, (long)N, L, P, left, mode, zmean, (const T*)w, nfft,
                       (const T*)twiddle, out_kind, fmt, (T)eps, use_floor, floor_lin, (T*)y);
    return check_launch("row_dft_generic");
}

// ---------------------------------------------------------------------------------------------
// Generalized cepstral transformation in ONE launch (GeneralizedCepstrumToGeneralizedCepstrum._forward, mgc2mgc.py:333-361):
//   c01 = (0, c1[1:]) -> C1 = fft(c01, n) -> s = (1 + g1 C1)^(1/g1) (g1 = 0: exp C1) -> C2 = (|s|^g2 cos(g2 angle(s)) - 1) / g2
//   (g2 = 0: log |s|) -> c02 = ifft(C2).real[: M2 + 1] -> c2 = (c1[0], 2 c02[1:]).
// One workgroup per row, the n complex points in LDS; both transforms are the radix-2 LDS transform above: c01 is real, so C1 is
// Hermitian and C2 is real and even -- its inverse transform IS its forward transform / n, and only the real parts leave.
// As separate launches (row transform -> five element-wise operators -> adjoint row transform, modules/mgc2mgc.py) the
// 4096-point spectra of the MLSA filter's impulse responses went through memory seven times (profiles/r02_mlsa_single_stage_trace.txt:
// 7.2 of the 8.3 ms of the single-stage mode).  Forward only (the module composes the differentiable operators when a gradient is
// wanted).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gc2gc_fused_kernel(const T* __restrict__ c1, int n_in, int out_order, T g1, T g2, int nfft,
                                                         const T* __restrict__ tw, int flags, T* __restrict__ c2)
{
    // Both transforms act on REAL data (c01, and the real even C2), so each runs as a complex transform of HALF the length on
    // the packed sequence z[n] = x[2n] + i x[2n+1], followed by the split  X[k] = (Z[k] + conj Z[H-k]) / 2 - i W^k (Z[k] - conj Z[H-k]) / 2
    // (H = n / 2): half the butterflies and half the LDS traffic of the full-length version (2.59 -> see DESIGN ms per 51 200 rows
    // of 4096 points).  LDS: re[H] | im[H] | cb[H + 1] (the mapped half spectrum, natural order).
    extern __shared__ unsigned char smem_raw[];
    const int H = nfft >> 1;
    T* re = reinterpret_cast<T*>(smem_raw);
    T* im = re + H;
    T* cb = im + H;
    const long f = blockIdx.x;
    const T* row = c1 + f * n_in;
    const int lgh = 30 - __clz(nfft);   // log2(H)
    // flags: the per-row scalar steps mgc2mgc.py:217-300 wraps around the transformation, folded in (each was a pass over the
    // row in memory): 1 gnorm(in_gamma) before, 2 ignorm(out_gamma) after, 4 tail times out_gamma, 8 zeroth coefficient * out_gamma + 1
    T k0 = row[0], tin = T(1);
    if (flags & 1) {   // gnorm.py:99-109
        if (g1 == T(0)) k0 = dsa_exp(row[0]);
        else {
            const T z = T(1) + g1 * row[0];
            k0 = dsa_pow(z, T(1) / g1);
            tin = T(1) / z;
        }
    }
    for (int n = threadIdx.x; n < H; n += blockDim.x) {   // fft(c01, n): longer rows are cropped, c01[0] = 0
        const int i0 = 2 * n, i1 = 2 * n + 1;
        re[n] = (i0 >= 1 && i0 < n_in) ? row[i0] * tin : T(0);
        im[n] = i1 < n_in ? row[i1] * tin : T(0);
    }
    __syncthreads();
    lds_fft_pow2(re, im, H, lgh, tw, 2);   // Z[k] at position brev(k)
    constexpr T kPi = T(3.14159265358979323846);
    auto gmap = [&](T cr, T ci) -> T {
        T lmag, ang;   // log |s|, angle(s) (wrapped to (-pi, pi] as .angle() of the reference's polar(r, theta) is)
        if (g1 == T(0)) {
            lmag = cr;
            ang = ci;
        } else {
            const T zr = T(1) + g1 * cr, zi = g1 * ci;
            lmag = T(0.5) * dsa_log(zr * zr + zi * zi) / g1;
            ang = atan2(zi, zr) / g1;
        }
        if (g2 == T(0)) return lmag;
        ang -= T(2) * kPi * rint(ang / (T(2) * kPi));
        return (dsa_exp(g2 * lmag) * cos(ang * g2) - T(1)) / g2;
    };
    // split into X[k], X[H - k] and map both (C2 is real and even: cb[k], k = 0 .. H, carries it all)
    for (int k = threadIdx.x; k <= (H >> 1); k += blockDim.x) {
        if (k == 0) {
            const T zr = re[0], zi = im[0];
            cb[0] = gmap(zr + zi, T(0));
            cb[H] = gmap(zr - zi, T(0));
        } else {
            const int pa = fft_brev(k, lgh), pb = fft_brev(H - k, lgh);
            const T ar = re[pa], ai = im[pa], br = re[pb], bi = -im[pb];          // A = Z[k], B = conj Z[H - k]
            const T sr = T(0.5) * (ar + br), si = T(0.5) * (ai + bi), dr = T(0.5) * (ar - br), di = T(0.5) * (ai - bi);
            const T wr = tw[2 * k], wi = tw[2 * k + 1];                            // W_n^k = (cos, -sin)(2 pi k / n)
            const T pr = wr * dr - wi * di, pi_ = wr * di + wi * dr;              // W D
            cb[k] = gmap(sr + pi_, si - pr);                                       // X[k]     = S - i W D
            cb[H - k] = gmap(sr - pi_, -si - pr);                                  // X[H - k] = conj(S + i W D)
        }
    }
    __syncthreads();
    for (int m = threadIdx.x; m < H; m += blockDim.x) {   // pack the even sequence C2[0 .. n - 1]: C2[j] = cb[j <= H ? j : n - j]
        const int j0 = 2 * m, j1 = 2 * m + 1;
        re[m] = cb[j0 <= H ? j0 : nfft - j0];
        im[m] = cb[j1 <= H ? j1 : nfft - j1];
    }
    __syncthreads();
    lds_fft_pow2(re, im, H, lgh, tw, 2);
    T* out = c2 + f * (long)(out_order + 1);
    T sc = T(2) / T(nfft), o0 = k0;
    if (flags & 2) {   // ignorm.py:99-109
        if (g2 == T(0)) o0 = dsa_log(k0);
        else {
            const T zz = dsa_pow(k0, g2);
            o0 = (zz - T(1)) / g2;
            sc *= zz;
        }
    }
    if (flags & 4) sc *= g2;
    if (flags & 8) o0 = o0 * g2 + T(1);
    for (int m = threadIdx.x; m <= out_order; m += blockDim.x) {
        T v;
        if (m == 0) {
            v = o0;
        } else {
            const int n = m <= H ? m : nfft - m;   // the inverse transform of a real even spectrum is even
            T y;                                   // Re of the length-n transform of C2 at index n
            if (n == H) {
                y = re[0] - im[0];
            } else {
                const int pa = fft_brev(n, lgh), pb = fft_brev(H - n, lgh);
                const T ar = re[pa], ai = im[pa], br = re[pb], bi = -im[pb];
                const T dr = T(0.5) * (ar - br), di = T(0.5) * (ai - bi);
                y = T(0.5) * (ar + br) + tw[2 * n] * di + tw[2 * n + 1] * dr;
            }
            v = sc * y;
        }
        out[m] = v;
    }
}

// Backward of gc2gc_fused_kernel (flags = 0) in ONE launch per row: gc1 from the row c1 and the cotangent g2 of c2.
//   c2[0] = c1[0];  c2[m] = 2 c02[m],  c02 = Re ifft(C2),  C2[k] = f(X[k]) for the half spectrum k = 0 .. H of X = fft(c01).
// Three half-length transforms in LDS: X is recomputed from c01; the cotangent of the even spectrum is a cosine transform of
// g2, gcb[k] = (2 / n) w_k Re fft(g)[k] (w = 1 at k = 0, H, else 2: cb[k] is read for j = k and j = n - k); the element-wise
// chain rule gives (gXr, gXi)[k]; and gc01[m] = sum_{k=0}^{H} gXr[k] cos(2 pi k m / n) - gXi[k] sin(2 pi k m / n) is the
// unnormalised inverse real transform of the Hermitian spectrum Y (Y[0] = gXr[0], Y[H] = gXr[H], Y[k] = (gXr + i gXi)[k] / 2),
// run as the conjugate of a forward half-length transform of Z[k] = E[k] + i O[k], E = (Y[k] + conj Y[H-k]) / 2,
// O = conj(W)^k (Y[k] - conj Y[H-k]) / 2.  LDS: re[H] | im[H] | xr[H+1] | xi[H+1] | gcb[H+1].
template <typename T>
__global__ __launch_bounds__(256) void gc2gc_fused_bwd_kernel(const T* __restrict__ c1, const T* __restrict__ g2row, int n_in,
                                                             int out_order, T g1, T g2, int nfft, const T* __restrict__ tw,
                                                             T* __restrict__ gc1)
{
    extern __shared__ unsigned char smem_raw[];
    const int H = nfft >> 1;
    T* re = reinterpret_cast<T*>(smem_raw);
    T* im = re + H;
    T* xr = im + H;
    T* xi = xr + (H + 1);
    T* gcb = xi + (H + 1);
    const long f = blockIdx.x;
    const T* row = c1 + f * n_in;
    const T* grow = g2row + f * (long)(out_order + 1);
    const int lgh = 30 - __clz(nfft);
    constexpr T kPi = T(3.14159265358979323846);
    // half-length transform of a packed real sequence, then the split into the half spectrum (dr, di)[0 .. H]
    auto split_to = [&](T* dr, T* di) {
        for (int k = threadIdx.x; k <= (H >> 1); k += blockDim.x) {
            if (k == 0) {
                const T zr = re[0], zi = im[0];
                dr[0] = zr + zi;
                dr[H] = zr - zi;
                if (di) {
                    di[0] = T(0);
                    di[H] = T(0);
                }
            } else {
                const int pa = fft_brev(k, lgh), pb = fft_brev(H - k, lgh);
                const T ar = re[pa], ai = im[pa], br = re[pb], bi = -im[pb];
                const T sr = T(0.5) * (ar + br), si = T(0.5) * (ai + bi), dr_ = T(0.5) * (ar - br), di_ = T(0.5) * (ai - bi);
                const T wr = tw[2 * k], wi = tw[2 * k + 1];
                const T pr = wr * dr_ - wi * di_, pi_ = wr * di_ + wi * dr_;
                dr[k] = sr + pi_;
                dr[H - k] = sr - pi_;
                if (di) {
                    di[k] = si - pr;
                    di[H - k] = -si - pr;
                }
            }
        }
    };
    // ---- X = fft(c01) ----
    for (int n = threadIdx.x; n < H; n += blockDim.x) {
        const int i0 = 2 * n, i1 = 2 * n + 1;
        re[n] = (i0 >= 1 && i0 < n_in) ? row[i0] : T(0);
        im[n] = i1 < n_in ? row[i1] : T(0);
    }
    __syncthreads();
    lds_fft_pow2(re, im, H, lgh, tw, 2);
    split_to(xr, xi);
    __syncthreads();
    // ---- cosine transform of the cotangent: g[0] = 0, g[m] = g2[m] ----
    for (int n = threadIdx.x; n < H; n += blockDim.x) {
        const int i0 = 2 * n, i1 = 2 * n + 1;
        re[n] = (i0 >= 1 && i0 <= out_order) ? grow[i0] : T(0);
        im[n] = i1 <= out_order ? grow[i1] : T(0);
    }
    __syncthreads();
    lds_fft_pow2(re, im, H, lgh, tw, 2);
    split_to(gcb, static_cast<T*>(nullptr));
    __syncthreads();
    // ---- element-wise chain rule: (xr, xi)[k] <- (gXr, gXi)[k] ----
    for (int k = threadIdx.x; k <= H; k += blockDim.x) {
        const T cr = xr[k], ci = xi[k];
        const T gc = gcb[k] * ((k == 0 || k == H) ? T(2) : T(4)) / T(nfft);
        T lmag, ang, l_r, l_i, a_r, a_i;   // log |s|, angle(s) and their partial derivatives with respect to (cr, ci)
        if (g1 == T(0)) {
            lmag = cr; ang = ci;
            l_r = T(1); l_i = T(0); a_r = T(0); a_i = T(1);
        } else {
            const T zr = T(1) + g1 * cr, zi = g1 * ci, r2 = zr * zr + zi * zi;
            lmag = T(0.5) * dsa_log(r2) / g1;
            ang = atan2(zi, zr) / g1;
            l_r = zr / r2; l_i = zi / r2; a_r = -zi / r2; a_i = zr / r2;
        }
        T f_l, f_a;
        if (g2 == T(0)) {
            f_l = T(1); f_a = T(0);
        } else {
            ang -= T(2) * kPi * rint(ang / (T(2) * kPi));
            const T e = dsa_exp(g2 * lmag);
            f_l = e * cos(ang * g2);
            f_a = -e * sin(ang * g2);
        }
        xr[k] = gc * (f_l * l_r + f_a * a_r);
        xi[k] = gc * (f_l * l_i + f_a * a_i);
    }
    __syncthreads();
    // ---- gc01 = 2 * conj(fft_H(conj Z)) unpacked ----
    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        T yr, yi, br, bi;                      // Y[k], conj Y[H - k]
        if (k == 0) {
            yr = xr[0]; yi = T(0);
            br = xr[H]; bi = T(0);
        } else {
            yr = T(0.5) * xr[k]; yi = T(0.5) * xi[k];
            br = T(0.5) * xr[H - k]; bi = -T(0.5) * xi[H - k];
        }
        const T er = T(0.5) * (yr + br), ei = T(0.5) * (yi + bi), dr = T(0.5) * (yr - br), di = T(0.5) * (yi - bi);
        const T wr = tw[2 * k], wi = -tw[2 * k + 1];            // conj(W)^k = (cos, +sin)(2 pi k / n)
        const T or_ = wr * dr - wi * di, oi = wr * di + wi * dr;  // O[k]
        // Z = E + i O = (er - oi) + i (ei + or); the transform runs on conj Z
        re[k] = er - oi;
        im[k] = -(ei + or_);
    }
    __syncthreads();
    lds_fft_pow2(re, im, H, lgh, tw, 2);
    T* out = gc1 + f * n_in;
    for (int m = threadIdx.x; m < n_in; m += blockDim.x) {
        T v;
        if (m == 0) {
            v = grow[0];
        } else if (m >= nfft) {
            v = T(0);                          // (rows longer than the transform are cropped by the forward)
        } else {
            const int pos = fft_brev(m >> 1, lgh);
            v = T(2) * ((m & 1) ? -im[pos] : re[pos]);
        }
        out[m] = v;
    }
}

template <typename T>
static int gc2gc_launch(const void* c1, int64_t F, int n_in, int out_order, double g1, double g2, int nfft, const void* tw, int flags,
                        void* c2, hipStream_t st)
{
    const size_t lds = sizeof(T) * (3 * (size_t)(nfft / 2) + 1);
    static std::atomic<uint64_t> lds_set{0};
    if (lds > 48 * 1024 && !ensure_dynamic_lds(reinterpret_cast<const void*>(&gc2gc_fused_kernel<T>), 150 * 1024, lds_set))
        return fail(DSA_ERR_LAUNCH, "gc2gc: cannot raise the dynamic LDS limit%s");
    // short transforms: one wave per row (two butterflies per lane and pass, the passes' barriers are single-wave barriers);
    // DSA_GC2GC_BLOCK overrides for A/B runs
    static const int forced = [] { const char* e = getenv("DSA_GC2GC_BLOCK"); return e ? atoi(e) : 0; }();
    const int block = forced > 0 ? forced : (nfft <= 1024 ? 64 : 256);
    hipLaunchKernelGGL((gc2gc_fused_kernel<T>), dim3((unsigned)F), dim3(block), lds, st, (const T*)c1, n_in, out_order, (T)g1, (T)g2, nfft,
                       (const T*)tw, flags, (T*)c2);
    return check_launch("gc2gc_fused");
}

// Backward of stft512_fwd_kernel (autograd of stft.py:237-241), same
// wave-per-pass structure and LDS tile.  Per pass of 4 frames:
//   recompute Z (stage, window, FFT-256) -> split into X[k] -> cotangent S[k] of the half spectrum
//   (power formats: gs[k] X[k] with gs = gy * format'(s);  complex: gy / 2) -> Hermitian-pack into a
//   256-point complex spectrum Zin[k] = (a + b) + i conj(W^k) (a - b), a = S[k], b = conj(S[256-k]) ->
//   inverse FFT-256 (same radix-16 x 16 code, conjugated twiddles) = cotangent of the windowed
//   frame -> times window (-> zmean adjoint) -> overlap-add of the 4 frames inside the pass ->
//   one contiguous partial span of 3P + L samples per pass, written to `part` (pass-major).
// A second kernel (stft_span_gather_kernel) adds the <= ceil((3P+L)/(4P)) partial spans that cover
// each waveform sample in a fixed order: deterministic, no atomics.
#ifndef DSA_STFT_BWD_WAVES
#define DSA_STFT_BWD_WAVES 3
#endif
// CPLX: the cotangent is complex (format "complex" or an inverse transform): X is not needed, so the
// input stretch is not staged and the forward FFT is skipped.
// PLAIN: power format with constant padding, fixed at compile time (the training path of the bench
// configuration): the format switch and the padding modes leave the register allocation.
template <bool ZMEAN, bool CPLX = false, bool PLAIN = false>
__global__ __launch_bounds__((PLAIN || CPLX) ? 128 : 64, (PLAIN || CPLX) ? 4 : DSA_STFT_BWD_WAVES) void stft512_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gy, long Tlen, long N, int L, int P, int left,
    int mode_arg, const float* __restrict__ w, const float* __restrict__ twiddle, float eps, int fmt_arg,
    float* __restrict__ part, long total_chunks, int chunks_per_utt, int span)
{
    const int fmt = PLAIN ? (int)DSA_SPEC_POWER : fmt_arg;
    const int mode = PLAIN ? (int)DSA_PAD_CONSTANT : mode_arg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int WPB = (PLAIN || CPLX) ? 2 : 1;   // waves per workgroup (they share the twiddle table only, as in the forward)
    const int wv = WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    cf* zbuf = reinterpret_cast<cf*>(smem_raw) + wv * kFPW * kZS;
    float* io_buf = reinterpret_cast<float*>(zbuf);
    cf* t256 = WPB > 1 ? reinterpret_cast<cf*>(smem_raw) + WPB * kFPW * kZS : zbuf + kFPW * 256;
    const long wid = (long)blockIdx.x * WPB + wv, nw = (long)gridDim.x * WPB;

    const int lane = threadIdx.x & 63;
    const int j = lane & 15, fl = lane >> 4;
    float wreg[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        int l = 2 * j + 32 * (r >> 1) + (r & 1);
        wreg[r] = l < L ? w[l] : 0.f;
    }
    for (int i = lane; i < 256; i += 64) {
        int m = 2 * (i & 15) * (i >> 4);
        t256[i] = cf{twiddle[2 * m], twiddle[2 * m + 1]};
    }
    const cf twA = cf{twiddle[2 * lane], twiddle[2 * lane + 1]};
    const cf twB = cf{twiddle[2 * (lane + 64)], twiddle[2 * (lane + 64) + 1]};
    const float inv_L = 1.f / (float)L;
    const int K = 257;
    const bool complex_out = fmt == DSA_SPEC_COMPLEX || fmt == DSA_SPEC_COMPLEX_INV;
    // complex cotangent: the adjoint takes g / 2 (g at the two real-valued bins); the inverse transform's
    // weights c_k / 512 on top of that make it g / 512 everywhere
    const float cot_scale = fmt == DSA_SPEC_COMPLEX_INV ? 1.f / 512.f : 0.5f;
    const float cot_edge = fmt == DSA_SPEC_COMPLEX_INV ? 1.f : 2.f;
    cf* zf = zbuf + fl * 256;
    const float2* gy2 = reinterpret_cast<const float2*>(gy);

    // (utterance, chunk) of pass c advance incrementally: one 64-bit division per wave instead of one per pass
    long b = wid / chunks_per_utt;
    int ci = (int)(wid - b * chunks_per_utt);
    const long b_step = nw / chunks_per_utt;
    const int ci_step = (int)(nw - b_step * chunks_per_utt);
    for (long c = wid; c < total_chunks; c += nw, b += b_step, ci += ci_step) {
        if (ci >= chunks_per_utt) {
            ci -= chunks_per_utt;
            ++b;
        }
        const long frame0 = (long)ci * kFPW;
        const int nvalid = (int)((N - frame0) < kFPW ? (N - frame0) : kFPW);
        const float* xb = x + b * Tlen;
        DSA_WAVE_SYNC();
        cf v[16];
        int lim = L - 2 * j;
        asm volatile("" : "+v"(lim));   // per pass on purpose: hoisted, the lane masks of the selects fill the scalar registers
        if constexpr (!CPLX) {
        {   // stage the input stretch
            const long g0 = frame0 * P - left;
            const int need = (nvalid - 1) * P + L;
            const bool interior = g0 >= 0 && g0 + need <= Tlen;
            if (interior && (((size_t)(xb + g0)) & 15) == 0) {
                const float4* src4 = reinterpret_cast<const float4*>(xb + g0);
                float4* dst4 = reinterpret_cast<float4*>(io_buf);
                const int n4 = need >> 2;
                for (int s = lane; s < n4; s += 64) dst4[s] = src4[s];
                for (int s = (n4 << 2) + lane; s < need; s += 64) io_buf[s] = xb[g0 + s];
            } else {
                for (int s = lane; s < need; s += 64) io_buf[s] = load_padded(xb, g0 + s, Tlen, mode);
            }
        }
        DSA_WAVE_SYNC();
        {
            const float* src = io_buf + fl * P + 2 * j;
            float sum = 0.f;
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) {
                float a0 = 0.f, a1 = 0.f;
                if (32 * m1 + 32 <= L) {
                    a0 = src[32 * m1];
                    a1 = src[32 * m1 + 1];
                } else if (32 * m1 < L) {
                    a0 = 32 * m1 < lim ? src[32 * m1] : 0.f;
                    a1 = 32 * m1 + 1 < lim ? src[32 * m1 + 1] : 0.f;
                }
                v[m1] = cf{a0, a1};
                if (ZMEAN) sum += a0 + a1;
            }
            float mean = 0.f;
            if (ZMEAN) {
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
                mean = sum * inv_L;
            }
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) {
                float a0 = v[m1].re, a1 = v[m1].im;
                if (ZMEAN) {
                    a0 = 32 * m1 < lim ? a0 - mean : 0.f;
                    a1 = 32 * m1 + 1 < lim ? a1 - mean : 0.f;
                }
                v[m1] = cf{a0 * wreg[2 * m1], a1 * wreg[2 * m1 + 1]};
            }
        }
        DSA_WAVE_SYNC();
        fft16<false>(v);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) zf[k1 * 16 + (j ^ k1)] = cmul(v[FFT16_OUT(k1)], t256[k1 * 16 + j]);
        DSA_WAVE_SYNC();
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = zf[j * 16 + (i ^ j)];
        DSA_WAVE_SYNC();
        fft16<false>(v);
#pragma unroll
        for (int k0 = 0; k0 < 16; ++k0) zf[j + 16 * k0] = v[FFT16_OUT(k0)];
        DSA_WAVE_SYNC();
        }
        // ---- split, cotangent, Hermitian packing (pairs read first, then written in place) ----
        const long out0 = (b * N + frame0) * K;
        cf pa[kFPW][3], pb[kFPW][3];
#pragma unroll
        for (int f = 0; f < kFPW; ++f) {
            const cf* z = zbuf + f * 256;
            if constexpr (CPLX) {
                pa[f][0] = pb[f][0] = pa[f][1] = pb[f][1] = pa[f][2] = pb[f][2] = cf{0.f, 0.f};
            } else {
                pa[f][0] = z[lane];
                pb[f][0] = z[(256 - lane) & 255];
                pa[f][1] = z[lane + 64];
                pb[f][1] = z[192 - lane];
                pa[f][2] = z[128];
                pb[f][2] = pa[f][2];
            }
        }
        DSA_WAVE_SYNC();
#pragma unroll
        for (int f = 0; f < kFPW; ++f) {
            cf* z = zbuf + f * 256;
            const bool fv = f < nvalid;
#pragma unroll
            for (int part_i = 0; part_i < 3; ++part_i) {
                const int k = part_i == 0 ? lane : (part_i == 1 ? lane + 64 : 128);
                const cf W = part_i == 0 ? twA : (part_i == 1 ? twB : cf{0.f, -1.f});
                const cf a = pa[f][part_i], bq = pb[f][part_i];
                const cf S = {a.re + bq.re, a.im - bq.im};
                const cf Dd = {a.re - bq.re, a.im + bq.im};
                const cf Pp = cmul(W, Dd);
                const cf X1 = {0.5f * (S.re + Pp.im), 0.5f * (S.im - Pp.re)};     // X[k]
                const cf X2 = {0.5f * (S.re - Pp.im), 0.5f * (-S.im - Pp.re)};    // X[256-k]
                // half-spectrum cotangents S1 = S[k], S2 = S[256-k]
                cf S1, S2;
                if (complex_out) {
                    const float2 g1 = fv ? gy2[out0 + f * K + k] : make_float2(0.f, 0.f);
                    const float2 g2 = fv ? gy2[out0 + f * K + 256 - k] : make_float2(0.f, 0.f);
                    S1 = cf{cot_scale * g1.x, cot_scale * g1.y};
                    S2 = cf{cot_scale * g2.x, cot_scale * g2.y};
                } else {
                    const float s1 = X1.re * X1.re + X1.im * X1.im + eps;
                    const float s2 = X2.re * X2.re + X2.im * X2.im + eps;
                    float g1 = fv ? gy[out0 + f * K + k] : 0.f;
                    float g2 = fv ? gy[out0 + f * K + 256 - k] : 0.f;
                    switch (fmt) {  // d format(s) / d s  (spec.py:123-132)
                    case DSA_SPEC_DB: g1 *= 4.342944819032518f / s1; g2 *= 4.342944819032518f / s2; break;
                    case DSA_SPEC_LOGMAG: g1 *= 0.5f / s1; g2 *= 0.5f / s2; break;
                    case DSA_SPEC_MAG: g1 *= 0.5f / sqrtf(s1); g2 *= 0.5f / sqrtf(s2); break;
                    default: break;
                    }
                    S1 = cf{g1 * X1.re, g1 * X1.im};
                    S2 = cf{g2 * X2.re, g2 * X2.im};
                }
                if (part_i == 0) {
                    // k = 0 pairs with 256: both real-valued bins carry the full (not half) weight
                    const float e0 = complex_out ? cot_edge : 2.f;
                    const float s0r = lane == 0 ? e0 * S1.re : S1.re, s0i = lane == 0 ? 0.f : S1.im;
                    const float s6r = lane == 0 ? e0 * S2.re : S2.re, s6i = lane == 0 ? 0.f : S2.im;
                    S1 = cf{s0r, s0i};
                    S2 = cf{s6r, s6i};
                }
                // Zin[k] = (a + b) + i Q, Zin[256-k] = conj(a + b) + i conj(Q), a = S1, b = conj(S2),
                // Q = conj(W) (a - b)
                const cf ab = {S1.re + S2.re, S1.im - S2.im};
                const cf amb = {S1.re - S2.re, S1.im + S2.im};
                const cf Q = cmul(cf{W.re, -W.im}, amb);
                if (part_i == 2) {
                    if (lane == 0) z[128] = cf{2.f * S1.re, -2.f * S1.im};  // 2 conj(S[128])
                } else {
                    z[k] = cf{ab.re - Q.im, ab.im + Q.re};
                    if (!(part_i == 0 && lane == 0)) z[256 - k] = cf{ab.re + Q.im, -ab.im + Q.re};
                }
            }
        }
        DSA_WAVE_SYNC();
        // ---- inverse FFT-256 (unnormalised, conjugated twiddles), same data movement ----
#pragma unroll
        for (int m1 = 0; m1 < 16; ++m1) v[m1] = zf[j + 16 * m1];
        DSA_WAVE_SYNC();
        fft16<true>(v);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            const cf t = t256[k1 * 16 + j];
            zf[k1 * 16 + (j ^ k1)] = cmul(v[FFT16_OUT(k1)], cf{t.re, -t.im});
        }
        DSA_WAVE_SYNC();
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = zf[j * 16 + (i ^ j)];
        DSA_WAVE_SYNC();
        fft16<true>(v);
        // lane j now holds time points m = j + 16 k0: samples l = 2m, 2m+1 -- the forward's own
        // register <-> sample map, so the window (and zmean adjoint) reuse wreg / the 16-lane sum
        {
            float gsum = 0.f;
#pragma unroll
            for (int k0 = 0; k0 < 16; ++k0) {
                cf o = v[FFT16_OUT(k0)];
                o = cf{o.re * wreg[2 * k0], o.im * wreg[2 * k0 + 1]};
                v[FFT16_OUT(k0)] = o;
                if (ZMEAN) gsum += o.re + o.im;
            }
            float gm = 0.f;
            if (ZMEAN) {
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) gsum += __shfl_xor(gsum, o, 16);
                gm = gsum * inv_L;
            }
#pragma unroll
            for (int k0 = 0; k0 < 16; ++k0) {
                cf o = v[FFT16_OUT(k0)];
                if (ZMEAN) {
                    o.re = 32 * k0 < lim ? o.re - gm : 0.f;
                    o.im = 32 * k0 + 1 < lim ? o.im - gm : 0.f;
                }
                zf[j + 16 * k0] = o;  // gframe[l] as floats: l = 2 (j + 16 k0) + {0, 1}
            }
        }
        DSA_WAVE_SYNC();
        // ---- overlap-add of the pass's frames; one contiguous partial span per pass ----
        float* dst = part + c * (long)span;
        for (int sidx = lane; sidx < span; sidx += 64) {
            float acc = 0.f;
#pragma unroll
            for (int f = 0; f < kFPW; ++f) {
                const int l = sidx - f * P;
                if (f < nvalid && l >= 0 && l < L) acc += reinterpret_cast<const float*>(zbuf + f * 256)[l];
            }
            dst[sidx] = acc;
        }
    }
}

// gx[b][t] = sum over the passes whose span covers t (adjoint of the on-the-fly padding: positions
// outside [0, T) are dropped for constant padding -- other modes use the generic backward).
// div != nullptr (inverse STFT): the sum is divided by div[t] + div_eps, the overlap-added squared window
// (unframe.py:203-205), so Unframe's division costs no pass of its own.
__global__ void stft_span_gather_kernel(const float* __restrict__ part, long B, long Tlen, int P, int left, int span,
                                        int chunks_per_utt, float* __restrict__ gx, const float* __restrict__ div,
                                        float div_eps)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tlen) return;
    const long p = t + left;          // position in the padded signal
    const long stride = (long)kFPW * P;  // pass c starts at padded position c * stride
    long c_hi = p / stride;
    if (c_hi > chunks_per_utt - 1) c_hi = chunks_per_utt - 1;
    long c_lo = p - span + 1 <= 0 ? 0 : (p - span + stride) / stride;
    for (long b = blockIdx.y; b < B; b += gridDim.y) {   // grid.y is capped at 65535 utterances
        float acc = 0.f;
        for (long c = c_lo; c <= c_hi; ++c) acc += part[(b * chunks_per_utt + c) * (long)span + (p - c * stride)];
        gx[b * Tlen + t] = div ? acc / (div[t] + div_eps) : acc;
    }
}

static int stft512_lds_bytes(int L, int P, int* io_floats)
{
    // the input stretch and the staged output tile both live inside the 4 x 256 complex tile
    int span = (kFPW - 1) * P + L;
    *io_floats = (span + 3) & ~3;
    if (*io_floats > kFPW * 512) return 1 << 30;  // stretch does not fit: use the generic kernel
    return kFPW * kZS * 8 + 256 * 8 + 16;   // (the backward kernel keeps stride 256 inside the same allocation)
}

static int stft512_lds_bytes2() { return 2 * kFPW * kZS * 8 + 256 * 8 + 2 * kFPW * 4; }   // two waves + the shared table

template <int ABL>
static void stft512_launch(bool zmean, dim3 grid, int lds, hipStream_t st, const float* x, long T, long N, int L,
                           int P, int left, int mode, const float* w, const float* tw, float eps, int use_floor,
                           float floor_lin, int fmt, float* y, long total_chunks, int chunks_per_utt, int io_floats)
{
    // `grid` counts waves; they are paired into 128-thread workgroups that share the twiddle table
    (void)lds;
    const bool plain = !use_floor && fmt == DSA_SPEC_POWER && mode == DSA_PAD_CONSTANT;
    const dim3 g2((grid.x + 1) / 2);
    const int lds2 = stft512_lds_bytes2();
    // DSA_STFT_PK (A/B knob): 2 = packed-float32 kernel with register-direct stores (stft_pk.h) where it applies
    // (default), 1 = the same with the staged output tile, 0 = scalar-float32 kernel
    static const int use_pk = [] {
        const char* e = getenv("DSA_STFT_PK");
        return e ? atoi(e) : 2;
    }();
    // (round 6: every pad mode -- the mode only changes what the passes that reach over an utterance's end read -- and, as an
    //  instantiation of its own, zmean and the relative floor)
    if (use_pk && ABL == 0 && fmt == DSA_SPEC_POWER && L == 400 && (P & 1) == 0 && 3 * P + 512 <= kFPW * kZS * 2) {
        if (zmean || use_floor) {
            hipLaunchKernelGGL((stft512_fwd_pk_kernel<0, 400, true, 0, false, 0, true>), g2, dim3(128), lds2, st, x, T, N, L, P, left, w, tw, eps, y,
                               total_chunks, chunks_per_utt, (const float*)nullptr, 0.f, 0.f, 0, 0, mode, (int)zmean, use_floor ? floor_lin : -1.f);
            return;
        }
        // a wave walks a RUN of consecutive passes (their shared samples come from its CU's cache, not from memory twice);
        // DSA_STFT_RUN=0: the round-robin order of rounds 1-4 (A/B)
        static const int run_env = [] { const char* e = getenv("DSA_STFT_RUN"); return e ? atoi(e) : 0; }();
        const int run_len = run_env;
        if (use_pk == 1)   // staged 16-byte stores (A/B)
            hipLaunchKernelGGL((stft512_fwd_pk_kernel<0, 400, false>), g2, dim3(128), lds2, st, x, T, N, L, P, left, w, tw, eps, y,
                               total_chunks, chunks_per_utt, (const float*)nullptr, 0.f, 0.f, 0, run_len, mode);
#define DSA_PK_XCD(ABLV)                                                                                              \
    hipLaunchKernelGGL((stft512_fwd_pk_kernel<ABLV, 400, true>), g2, dim3(128), lds2, st, x, T, N, L, P, left, w, tw, eps, y, \
                       total_chunks, chunks_per_utt, (const float*)nullptr, 0.f, 0.f, 0, 0, mode)
        else if (use_pk == 3) DSA_PK_XCD(1024);   // experiments (A/B): XCD-chunked workgroup order, C = 4 / 8 / 2, XCD-contiguous
        else if (use_pk == 4) DSA_PK_XCD(4096);
        else if (use_pk == 5) DSA_PK_XCD(2048);
        else if (use_pk == 6) DSA_PK_XCD(256);
        else if (use_pk == 8)   // four-wave workgroups: four adjacent passes per workgroup
            hipLaunchKernelGGL((stft512_fwd_pk_kernel<0, 400, true, 0, false, 4>), dim3((grid.x + 3) / 4), dim3(256),
                               4 * kFPW * kZS * 8 + 256 * 8 + 64, st, x, T, N, L, P, left, w, tw, eps, y, total_chunks,
                               chunks_per_utt, (const float*)nullptr, 0.f, 0.f, 0, run_len, mode);
        else if (use_pk == 9) {   // eight-wave workgroups
            static std::atomic<uint64_t> a9{0};
            ensure_dynamic_lds(reinterpret_cast<const void*>(&stft512_fwd_pk_kernel<0, 400, true, 0, false, 8>), 8 * kFPW * kZS * 8 + 256 * 8 + 64, a9);
            hipLaunchKernelGGL((stft512_fwd_pk_kernel<0, 400, true, 0, false, 8>), dim3((grid.x + 7) / 8), dim3(512),
                               8 * kFPW * kZS * 8 + 256 * 8 + 64, st, x, T, N, L, P, left, w, tw, eps, y, total_chunks,
                               chunks_per_utt, (const float*)nullptr, 0.f, 0.f, 0, run_len, mode);
        } else if (use_pk == 10) {   // sixteen-wave workgroups: a CU's whole complement of waves on sixteen adjacent passes
            static std::atomic<uint64_t> a10{0};
            ensure_dynamic_lds(reinterpret_cast<const void*>(&stft512_fwd_pk_kernel<0, 400, true, 0, false, 16>), 16 * kFPW * kZS * 8 + 256 * 8 + 64, a10);
            hipLaunchKernelGGL((stft512_fwd_pk_kernel<0, 400, true, 0, false, 16>), dim3((grid.x + 15) / 16), dim3(1024),
                               16 * kFPW * kZS * 8 + 256 * 8 + 64, st, x, T, N, L, P, left, w, tw, eps, y, total_chunks,
                               chunks_per_utt, (const float*)nullptr, 0.f, 0.f, 0, run_len, mode);
        } else if (use_pk == 7)   // the stretch fetched two passes ahead (two register sets, window table in LDS, four-wave workgroups)
            hipLaunchKernelGGL((stft512_fwd_pk_kernel<0, 400, true, 0, true>), dim3((grid.x + 3) / 4), dim3(256),
                               4 * kFPW * kZS * 8 + 256 * 8 + 16 * 13 * 8 + 64, st, x, T, N, L, P, left, w, tw, eps, y, total_chunks,
                               chunks_per_utt, (const float*)nullptr, 0.f, 0.f, 0, run_len, mode);
#undef DSA_PK_XCD
        else               // 8-byte stores straight from the split's registers (default)
            hipLaunchKernelGGL((stft512_fwd_pk_kernel<0, 400, true>), g2, dim3(128), lds2, st, x, T, N, L, P, left, w, tw, eps, y,
                               total_chunks, chunks_per_utt, (const float*)nullptr, 0.f, 0.f, 0, run_len, mode);
        return;
    }
#define DSA_STFT_FWD_LAUNCH(ZM, PL, LCV)                                                                                 \
    hipLaunchKernelGGL((stft512_fwd_kernel<ABL, ZM, PL, LCV>), g2, dim3(128), lds2, st, x, T, N, L, P, left, mode, w, tw, \
                       eps, use_floor, floor_lin, fmt, y, total_chunks, chunks_per_utt, io_floats)
    if (zmean) DSA_STFT_FWD_LAUNCH(true, false, 0);
    else if (plain && L == 400) DSA_STFT_FWD_LAUNCH(false, true, 400);
    else if (plain) DSA_STFT_FWD_LAUNCH(false, true, 0);
    else DSA_STFT_FWD_LAUNCH(false, false, 0);
#undef DSA_STFT_FWD_LAUNCH
}

// ---- inverse path helpers ----
// irfft(Y)[n] = sum_k c_k / N Re(Y_k e^{+2 pi i k n / N}), c = 1 at DC / Nyquist, 2 in between (ifftr.py:138):
// that is the ADJOINT of rfft (the backward kernels of this file) applied to G_k = c_k / N Y_k.
template <typename T>
__global__ void irfft_scale_kernel(const T* __restrict__ y, long total, int K, int nfft, T* __restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // complex element index
    if (i >= total) return;
    const int k = (int)(i % K);
    const T c = ((k == 0 || k == nfft / 2) ? T(1) : T(2)) / T(nfft);
    out[2 * i] = y[2 * i] * c;
    out[2 * i + 1] = y[2 * i + 1] * c;
}
// Unframe._forward unframe.py:203-205: x / (sum of squared windows + 1e-16), the divisor shared by all rows
template <typename T>
__global__ void div_rows_kernel(const T* __restrict__ x, long B, long Tlen, const T* __restrict__ d, T eps,
                                T* __restrict__ out)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Tlen) return;
    const T r = T(1) / (d[t] + eps);
    for (long b = blockIdx.y; b < B; b += gridDim.y) out[b * Tlen + t] = x[b * Tlen + t] * r;
}

// One Griffin-Lim phase update (griffin.py:263-282), element-wise over (B, N, K) complex bins:
//   t' = t (first) or (1 - gamma) d_prev + gamma t;   diff = t' - t_prev;   c = t' + alpha diff;   d = t' + beta diff;
//   z = sqrt(y + 1e-16) * c / (|c| + eps);   t_prev <- t',  d_prev <- d.
// t:(B, Nt, K) is the STFT of the previous estimate (Nt >= N frames; the extra ones are dropped, griffin.py:270);
// t == nullptr initialises: z = sqrt(y + 1e-16) * exp(i phase) (phase == nullptr: zeros).
template <typename T>
__global__ void griffin_update_kernel(const T* __restrict__ t, long B, long Nt, long N, int K, const T* __restrict__ y,
                                      const T* __restrict__ phase, T* __restrict__ t_prev, T* __restrict__ d_prev, int first,
                                      T alpha, T beta, T gamma, T eps, T* __restrict__ z)
{
    const long NK = N * K, total = B * NK;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const T s = dsa_sqrt(y[i] + T(1e-16));   // griffin.py:263-264
        T cr, ci;
        if (!t) {
            const T ph = phase ? phase[i] : T(0);
            z[2 * i] = s * dsa_cos(ph);
            z[2 * i + 1] = s * dsa_sin(ph);
            continue;
        }
        const long b = i / NK;
        const long it = b * Nt * K + (i - b * NK);
        T tr = t[2 * it], ti = t[2 * it + 1], dr, di;
        if (first) {
            cr = dr = tr;
            ci = di = ti;
        } else {
            tr = (T(1) - gamma) * d_prev[2 * i] + gamma * tr;
            ti = (T(1) - gamma) * d_prev[2 * i + 1] + gamma * ti;
            const T fr = tr - t_prev[2 * i], fi = ti - t_prev[2 * i + 1];
            cr = tr + alpha * fr;
            ci = ti + alpha * fi;
            dr = tr + beta * fr;
            di = ti + beta * fi;
        }
        t_prev[2 * i] = tr;
        t_prev[2 * i + 1] = ti;
        d_prev[2 * i] = dr;
        d_prev[2 * i + 1] = di;
        const T r = s / (dsa_sqrt(cr * cr + ci * ci) + eps);   // griffin.py:281
        z[2 * i] = cr * r;
        z[2 * i + 1] = ci * r;
    }
}

}  // namespace dsa

using namespace dsa;

// =========================================================================== C-ABI

DSA_EXPORT int dsa_version(void) { return DSA_VERSION; }
DSA_EXPORT const char* dsa_last_error(void) { return err_buf(); }
DSA_EXPORT const char* dsa_last_kernel(void) { return kernel_name(); }
DSA_EXPORT int dsa_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(DSA_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}
DSA_EXPORT int64_t dsa_num_frames(int64_t T, int32_t P) { return (T <= 0 || P <= 0) ? 0 : (T - 1) / P + 1; }

DSA_EXPORT int dsa_frame_fwd(const void* x, int64_t B, int64_t T, int32_t L, int32_t P, int32_t center,
                             int32_t zmean, int32_t pad_mode, int32_t dtype, void* y, void* stream)
{
    DSA_REQUIRE(L > 0 && P > 0 && T > 0 && B >= 0, "frame: sizes must be positive");
    DSA_REQUIRE(pad_mode >= 0 && pad_mode <= 3, "frame: unknown pad mode");
    // F.pad(mode="reflect") needs every pad amount -- (L//2, (L-1)//2) centred, (0, L-1) otherwise, frame.py:130-137 --
    // below the signal length
    DSA_REQUIRE(pad_mode != DSA_PAD_REFLECT || (center ? L / 2 : L - 1) < T || L == 1,
                "frame: reflect padding needs pad < input length");
    int64_t N = dsa_num_frames(T, P), F = B * N;
    if (F == 0) return DSA_OK;
    int left = center ? L / 2 : 0;
    int threads = L >= 192 ? 256 : (L >= 96 ? 128 : 64);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSA_F32 && !zmean && (L & 3) == 0 && (((size_t)y) & 15) == 0 && F * (int64_t)L >= 4096) {
        const int src_aligned = (P & 3) == 0 && (left & 3) == 0 && (T & 3) == 0 && (((size_t)x) & 15) == 0;
        long blocks = (long)((F * (int64_t)(L >> 2) + 255) / 256);
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(frame_fwd_vec4_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (long)T, (long)N,
                           (long)F, L, P, left, pad_mode, src_aligned, (float*)y);
        return check_launch("frame_fwd_vec4");
    }
    if (dtype == DSA_F32)
        hipLaunchKernelGGL((frame_fwd_kernel<float>), dim3((unsigned)F), dim3(threads), 0, st,
                           (const float*)x, (long)T, (long)N, L, P, left, zmean, pad_mode, (float*)y);
    else if (dtype == DSA_F64)
        hipLaunchKernelGGL((frame_fwd_kernel<double>), dim3((unsigned)F), dim3(threads), 0, st,
                           (const double*)x, (long)T, (long)N, L, P, left, zmean, pad_mode, (double*)y);
    else
        return fail(DSA_ERR_UNSUPPORTED, "frame: unsupported dtype%s");
    return check_launch("frame_fwd");
}

template <typename T>
static int frame_bwd_impl(const void* gy, int64_t B, int64_t Tlen, int L, int P, int center,
                          int zmean, int pad_mode, void* gx, hipStream_t st)
{
    int64_t N = dsa_num_frames(Tlen, P), F = B * N;
    int left = center ? L / 2 : 0;
    T* gmean = nullptr;
    if (zmean) {
        // d/dx of (y - mean(y)) = g - mean(g): per-frame mean of the cotangent
        if (hipMallocAsync((void**)&gmean, sizeof(T) * (size_t)F, st) != hipSuccess)
            return fail(DSA_ERR_LAUNCH, "frame_bwd: workspace allocation failed%s");
        hipLaunchKernelGGL((row_mean_kernel<T>), dim3((unsigned)F), dim3(64), 0, st, (const T*)gy, L, gmean);
    }
    if (pad_mode == DSA_PAD_CONSTANT) {
        const int64_t nblk = ((Tlen + 255) / 256) * B;
        if (nblk > 0x7fffffffLL) return fail(DSA_ERR_UNSUPPORTED, "frame_bwd: batch too large for one launch%s");
        dim3 grid((unsigned)nblk);
        hipLaunchKernelGGL((frame_bwd_const_kernel<T>), grid, dim3(256), 0, st, (const T*)gy, gmean,
                           (long)Tlen, (long)N, L, P, left, (T*)gx);
    } else {
        hipLaunchKernelGGL((frame_bwd_general_kernel<T>), dim3((unsigned)B), dim3(256), 0, st,
                           (const T*)gy, gmean, (long)Tlen, (long)N, L, P, left, pad_mode, (T*)gx);
    }
    int rc = check_launch("frame_bwd");
    if (gmean) hipFreeAsync(gmean, st);
    return rc;
}

DSA_EXPORT int dsa_frame_bwd(const void* gy, int64_t B, int64_t T, int32_t L, int32_t P, int32_t center,
                             int32_t zmean, int32_t pad_mode, int32_t dtype, void* gx, void* stream)
{
    DSA_REQUIRE(L > 0 && P > 0 && T > 0 && B >= 0, "frame_bwd: sizes must be positive");
    if (B == 0) return DSA_OK;
    if (dtype == DSA_F32) return frame_bwd_impl<float>(gy, B, T, L, P, center, zmean, pad_mode, gx, (hipStream_t)stream);
    if (dtype == DSA_F64) return frame_bwd_impl<double>(gy, B, T, L, P, center, zmean, pad_mode, gx, (hipStream_t)stream);
    return fail(DSA_ERR_UNSUPPORTED, "frame_bwd: unsupported dtype%s");
}

DSA_EXPORT int dsa_window_fwd(const void* x, int64_t F, int32_t L, const void* w, int32_t L2, int32_t dtype,
                              void* y, void* stream)
{
    DSA_REQUIRE(L > 0 && L2 > 0 && F >= 0, "window: sizes must be positive");
    if (F == 0) return DSA_OK;
    int64_t total = F * L2;
    unsigned grid = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSA_F32) {
        if (window_vec4_ok(x, y, w, F, L, L2)) {
            const unsigned n4 = (unsigned)(F * (int64_t)(L >> 2));
            hipLaunchKernelGGL(window_vec4_kernel, dim3((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256), dim3(256), 0, st,
                               (const float4*)x, n4, (unsigned)(L >> 2), (const float4*)w, (float4*)y);
            return check_launch("window_vec4");
        }
        hipLaunchKernelGGL((window_fwd_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)x,
                           (long)F, L, (const float*)w, L2, (float*)y);
    } else if (dtype == DSA_F64)
        hipLaunchKernelGGL((window_fwd_kernel<double>), dim3(grid), dim3(256), 0, st, (const double*)x,
                           (long)F, L, (const double*)w, L2, (double*)y);
    else
        return fail(DSA_ERR_UNSUPPORTED, "window: unsupported dtype%s");
    return check_launch("window_fwd");
}

DSA_EXPORT int dsa_window_bwd(const void* gy, const void* x, int64_t F, int32_t L, const void* w, int32_t L2,
                              int32_t dtype, void* gx, void* gw, void* stream)
{
    DSA_REQUIRE(L > 0 && L2 > 0 && F >= 0, "window_bwd: sizes must be positive");
    if (F == 0) return DSA_OK;
    int64_t total = F * L;
    unsigned grid = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSA_F32) {
        if (window_vec4_ok(gy, gx, w, F, L, L2)) {
            const unsigned n4 = (unsigned)(F * (int64_t)(L >> 2));
            hipLaunchKernelGGL(window_vec4_kernel, dim3((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256), dim3(256), 0, st,
                               (const float4*)gy, n4, (unsigned)(L >> 2), (const float4*)w, (float4*)gx);
        } else
            hipLaunchKernelGGL((window_bwd_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)gy,
                               (long)F, L, (const float*)w, L2, (float*)gx);
        if (gw)
            hipLaunchKernelGGL((window_gw_kernel<float>), dim3(L), dim3(256), 0, st, (const float*)gy,
                               (const float*)x, (long)F, L, L2, (float*)gw);
    } else if (dtype == DSA_F64) {
        hipLaunchKernelGGL((window_bwd_kernel<double>), dim3(grid), dim3(256), 0, st, (const double*)gy,
                           (long)F, L, (const double*)w, L2, (double*)gx);
        if (gw)
            hipLaunchKernelGGL((window_gw_kernel<double>), dim3(L), dim3(256), 0, st, (const double*)gy,
                               (const double*)x, (long)F, L, L2, (double*)gw);
    } else
        return fail(DSA_ERR_UNSUPPORTED, "window_bwd: unsupported dtype%s");
    return check_launch("window_bwd");
}

DSA_EXPORT int dsa_gc2gc_fwd(const void* c1, int64_t F, int32_t n_in, int32_t out_order, double in_gamma, double out_gamma,
                             int32_t nfft, const void* twiddle, int32_t flags, int32_t dtype, void* c2, void* stream)
{
    DSA_REQUIRE(F >= 0 && n_in >= 1 && out_order >= 0, "gc2gc: sizes must be positive");
    DSA_REQUIRE(nfft >= 4 && (nfft & (nfft - 1)) == 0, "gc2gc: n_fft must be a power of two");
    DSA_REQUIRE(flags >= 0 && flags < 16, "gc2gc: unknown flags");
    if (out_order + 1 > nfft || F > 0x7fffffffLL) return fail(DSA_ERR_UNSUPPORTED, "gc2gc: out_order + 1 must not exceed n_fft%s");
    if (F == 0) return DSA_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSA_F32 && (size_t)nfft * 8 <= 150 * 1024) return gc2gc_launch<float>(c1, F, n_in, out_order, in_gamma, out_gamma, nfft, twiddle, flags, c2, st);
    if (dtype == DSA_F64 && (size_t)nfft * 16 <= 150 * 1024) return gc2gc_launch<double>(c1, F, n_in, out_order, in_gamma, out_gamma, nfft, twiddle, flags, c2, st);
    return fail(DSA_ERR_UNSUPPORTED, "gc2gc: unsupported dtype or n_fft too long for LDS%s");
}

DSA_EXPORT int dsa_gc2gc_bwd(const void* c1, const void* g2, int64_t F, int32_t n_in, int32_t out_order, double in_gamma,
                             double out_gamma, int32_t nfft, const void* twiddle, int32_t dtype, void* gc1, void* stream)
{
    DSA_REQUIRE(F >= 0 && n_in >= 1 && out_order >= 0, "gc2gc_bwd: sizes must be positive");
    DSA_REQUIRE(nfft >= 4 && (nfft & (nfft - 1)) == 0, "gc2gc_bwd: n_fft must be a power of two (>= 4)");
    DSA_REQUIRE(out_order + 1 <= nfft, "gc2gc_bwd: out_order + 1 must not exceed n_fft");
    if (F == 0) return DSA_OK;
    hipStream_t st = (hipStream_t)stream;
    const int block = nfft <= 1024 ? 64 : 256;
    if (dtype == DSA_F32 && (size_t)nfft * 10 + 64 <= 150 * 1024) {
        const size_t lds = sizeof(float) * (5 * (size_t)(nfft / 2) + 3);
        static std::atomic<uint64_t> lds_set{0};
        if (lds > 48 * 1024 && !ensure_dynamic_lds(reinterpret_cast<const void*>(&gc2gc_fused_bwd_kernel<float>), 150 * 1024, lds_set))
            return fail(DSA_ERR_LAUNCH, "gc2gc_bwd: cannot raise the dynamic LDS limit%s");
        hipLaunchKernelGGL((gc2gc_fused_bwd_kernel<float>), dim3((unsigned)F), dim3(block), lds, st, (const float*)c1, (const float*)g2,
                           n_in, out_order, (float)in_gamma, (float)out_gamma, nfft, (const float*)twiddle, (float*)gc1);
        return check_launch("gc2gc_fused_bwd");
    }
    if (dtype == DSA_F64 && (size_t)nfft * 20 + 64 <= 150 * 1024) {
        const size_t lds = sizeof(double) * (5 * (size_t)(nfft / 2) + 3);
        static std::atomic<uint64_t> lds_set{0};
        if (lds > 48 * 1024 && !ensure_dynamic_lds(reinterpret_cast<const void*>(&gc2gc_fused_bwd_kernel<double>), 150 * 1024, lds_set))
            return fail(DSA_ERR_LAUNCH, "gc2gc_bwd: cannot raise the dynamic LDS limit%s");
        hipLaunchKernelGGL((gc2gc_fused_bwd_kernel<double>), dim3((unsigned)F), dim3(block), lds, st, (const double*)c1, (const double*)g2,
                           n_in, out_order, in_gamma, out_gamma, nfft, (const double*)twiddle, (double*)gc1);
        return check_launch("gc2gc_fused_bwd");
    }
    return fail(DSA_ERR_UNSUPPORTED, "gc2gc_bwd: unsupported dtype or n_fft too long for LDS%s");
}

DSA_EXPORT int dsa_fftr_fwd(const void* x, int64_t F, int32_t len_in, int32_t nfft, int32_t out_format,
                            const void* twiddle, int32_t dtype, void* y, void* stream)
{
    DSA_REQUIRE(len_in > 0 && nfft > 0 && nfft % 2 == 0, "fftr: fft_length must be positive even");
    DSA_REQUIRE(out_format >= 0 && out_format <= 4, "fftr: unknown out_format");
    hipStream_t st = (hipStream_t)stream;
    // rows are "utterances" of len_in samples holding exactly one frame each
    if (dtype == DSA_F32)
        return launch_row_dft<float>(x, F, len_in, 1, len_in, len_in, 0, 0, 0, nullptr, nfft, twiddle, 0,
                                     out_format, 0.0, 0, 0.0, y, st);
    if (dtype == DSA_F64)
        return launch_row_dft<double>(x, F, len_in, 1, len_in, len_in, 0, 0, 0, nullptr, nfft, twiddle, 0,
                                      out_format, 0.0, 0, 0.0, y, st);
    return fail(DSA_ERR_UNSUPPORTED, "fftr: unsupported dtype%s");
}

template <typename T>
static int spec_fwd_impl(const void* b, int lb, const void* a, int la, int64_t F, int nfft, double eps,
                         int use_floor, double floor_db, int fmt, const void* twiddle, void* y,
                         hipStream_t st)
{
    if (!a)
        return launch_row_dft<T>(b, F, lb, 1, lb, lb, 0, 0, 0, nullptr, nfft, twiddle, 1, fmt, eps,
                                 use_floor, floor_db, y, st);
    // denominator present: amplitude rows of b and of remove_gain(a), then the ratio kernel
    const int K = nfft / 2 + 1;
    T *amp_b = nullptr, *amp_a = nullptr, *a1 = nullptr;
    size_t rows = sizeof(T) * (size_t)F * K;
    if (hipMallocAsync((void**)&amp_a, rows, st) != hipSuccess ||
        hipMallocAsync((void**)&a1, sizeof(T) * (size_t)F * la, st) != hipSuccess ||
        (b && hipMallocAsync((void**)&amp_b, rows, st) != hipSuccess))
        return fail(DSA_ERR_LAUNCH, "spec: workspace allocation failed%s");
    int64_t tot = F * la;
    hipLaunchKernelGGL((remove_gain_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st,
                       (const T*)a, (long)F, la, a1);
    int rc = launch_row_dft<T>(a1, F, la, 1, la, la, 0, 0, 0, nullptr, nfft, twiddle, 0, DSA_FFTR_AMPLITUDE,
                               0.0, 0, 0.0, amp_a, st);
    if (rc == DSA_OK && b)
        rc = launch_row_dft<T>(b, F, lb, 1, lb, lb, 0, 0, 0, nullptr, nfft, twiddle, 0, DSA_FFTR_AMPLITUDE,
                               0.0, 0, 0.0, amp_b, st);
    if (rc == DSA_OK) {
        T floor_lin = use_floor ? (T)pow(10.0, floor_db / 10.0) : T(0);
        hipLaunchKernelGGL((spec_ratio_kernel<T>), dim3((unsigned)F), dim3(64), 0, st, (const T*)amp_b,
                           (const T*)amp_a, (const T*)a, la, K, (T)eps, use_floor, floor_lin, fmt, (T*)y);
        rc = check_launch("spec_ratio");
    }
    hipFreeAsync(amp_a, st);
    hipFreeAsync(a1, st);
    if (amp_b) hipFreeAsync(amp_b, st);
    return rc;
}

DSA_EXPORT int dsa_spec_fwd(const void* b, int32_t lb, const void* a, int32_t la, int64_t F, int32_t nfft,
                            double eps, int32_t use_floor, double relative_floor_db, int32_t out_format,
                            const void* twiddle, int32_t dtype, void* y, void* stream)
{
    DSA_REQUIRE(F == 0 || b || a, "spec: either b or a must be specified");
    DSA_REQUIRE(nfft > 1 && nfft % 2 == 0, "spec: fft_length must be positive even");
    DSA_REQUIRE(out_format >= 0 && out_format <= 3, "spec: unknown out_format");
    if (F == 0) return DSA_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSA_F32)
        return spec_fwd_impl<float>(b, lb, a, la, F, nfft, eps, use_floor, relative_floor_db, out_format, twiddle, y, st);
    if (dtype == DSA_F64)
        return spec_fwd_impl<double>(b, lb, a, la, F, nfft, eps, use_floor, relative_floor_db, out_format, twiddle, y, st);
    return fail(DSA_ERR_UNSUPPORTED, "spec: unsupported dtype%s");
}

DSA_EXPORT int dsa_stft_fwd(const void* x, int64_t B, int64_t T, int32_t L, int32_t P, int32_t nfft,
                            const void* w, const void* twiddle, int32_t center, int32_t zmean,
                            int32_t pad_mode, double eps, int32_t use_floor, double relative_floor_db,
                            int32_t out_format, int32_t dtype, int32_t algo, void* y, void* stream)
{
    DSA_REQUIRE(L > 0 && P > 0 && T > 0 && B >= 0, "stft: sizes must be positive");
    DSA_REQUIRE(nfft > 1 && nfft % 2 == 0, "stft: fft_length must be positive even");
    DSA_REQUIRE(pad_mode >= 0 && pad_mode <= 3, "stft: unknown pad mode");
    DSA_REQUIRE(out_format >= 0 && out_format <= 5, "stft: unknown out_format");
    // F.pad(mode="reflect") needs every pad amount below the signal length (frame.py:130-137: (L//2, (L-1)//2) when
    // centred, (0, L-1) otherwise) and rejects a one-sample signal
    DSA_REQUIRE(pad_mode != DSA_PAD_REFLECT || (center ? L / 2 : L - 1) < T || L == 1,
                "stft: reflect padding needs pad < input length");
    hipStream_t st = (hipStream_t)stream;
    int64_t N = dsa_num_frames(T, P);
    if (B * N == 0) return DSA_OK;
    int left = center ? L / 2 : 0;
    int in_floats = 0;
    int lds = stft512_lds_bytes(L, P, &in_floats);
    bool tuned_ok = dtype == DSA_F32 && nfft == 512 && L <= 512 && lds <= 64 * 1024;
    if (algo == DSA_ALGO_TUNED && !tuned_ok)
        return fail(DSA_ERR_UNSUPPORTED, "stft: tuned kernel needs float32, fft_length 512, frame_length <= 512%s");
    if (tuned_ok && algo != DSA_ALGO_GENERIC) {
        int chunks_per_utt = (int)((N + kFPW - 1) / kFPW);
        long total_chunks = (long)B * chunks_per_utt;
        // persistent single-wave workgroups; every one of them must be resident from the start, so
        // leave headroom under the 160 KB of LDS (12 x 13.3 KB does not always fit: measured slower)
        int waves_per_cu = 144 * 1024 / lds;
        if (waves_per_cu > 12) waves_per_cu = 12;
        if (waves_per_cu < 1) waves_per_cu = 1;
        waves_per_cu = 16;   // four waves per SIMD (two-wave workgroups, 19.5 KB of LDS each)
        long grid = 256L * waves_per_cu;  // persistent waves
        if (grid > total_chunks) grid = total_chunks;
        float floor_lin = use_floor ? (float)pow(10.0, relative_floor_db / 10.0) : 0.f;
        stft512_launch<0>(zmean != 0, dim3((unsigned)grid), lds, st, (const float*)x, (long)T, (long)N, L, P, left,
                          pad_mode, (const float*)w, (const float*)twiddle, (float)eps, use_floor, floor_lin,
                          out_format, (float*)y, total_chunks, chunks_per_utt, in_floats);
        return check_launch("stft512_fwd");
    }
    // fft_length 1024 / 2048 (the 44.1 / 48 kHz set-ups of utils/public.py:61-104), power format, constant padding, no zmean, no
    // relative floor: the packed kernel of stft_pk_big.h (round 6; DSA_STFT_BIG=0: the generic kernel, for A/B runs)
    static const bool big_on = [] { const char* e = getenv("DSA_STFT_BIG"); return !(e && e[0] == '0'); }();
    if (big_on && dtype == DSA_F32 && algo != DSA_ALGO_GENERIC && (nfft == 1024 || nfft == 2048) && !zmean && !use_floor &&
        out_format == DSA_SPEC_POWER && pad_mode == DSA_PAD_CONSTANT && L <= nfft && (L & 1) == 0 && (P & 1) == 0 && (left & 1) == 0 &&
        (T & 1) == 0 && (((size_t)x) & 7) == 0 && B * N < (int64_t(1) << 31)) {
        const int S = nfft / 512, FPP = 4 / S;
        const int need = (L + 32 * S - 1) / (32 * S);   // sample pairs per lane
        const int chunks_per_utt = (int)((N + FPP - 1) / FPP);
        const long total_chunks = (long)B * chunks_per_utt;
        const int lds_big = 4 * 4 * kZS * 8 + 256 * 8;
        long wgs = (total_chunks + 3) / 4;
        if (wgs > 256L * 3) wgs = 256L * 3;   // persistent: three four-wave workgroups per CU (126 .. 167 registers: 3 .. 4 waves per SIMD)
#define DSA_BIG_LAUNCH(SV, NRV)                                                                                                   \
    hipLaunchKernelGGL((stft_big_fwd_pk_kernel<SV, NRV>), dim3((unsigned)wgs), dim3(256), lds_big, st, (const float*)x, (long)T, (long)N, \
                       L, P, left, (const float*)w, (const float*)twiddle, (float)eps, (float*)y, total_chunks, chunks_per_utt)
        if (S == 2) {
            if (need <= 10) DSA_BIG_LAUNCH(2, 10);
            else if (need <= 13) DSA_BIG_LAUNCH(2, 13);
            else DSA_BIG_LAUNCH(2, 16);
        } else {
            if (need <= 10) DSA_BIG_LAUNCH(4, 10);
            else if (need <= 13) DSA_BIG_LAUNCH(4, 13);
            else DSA_BIG_LAUNCH(4, 16);
        }
#undef DSA_BIG_LAUNCH
        return check_launch(S == 2 ? "stft1024_fwd" : "stft2048_fwd");
    }
    if (dtype == DSA_F32)
        return launch_row_dft<float>(x, B, T, N, L, P, left, pad_mode, zmean, w, nfft, twiddle, 1, out_format,
                                     eps, use_floor, relative_floor_db, y, st);
    if (dtype == DSA_F64)
        return launch_row_dft<double>(x, B, T, N, L, P, left, pad_mode, zmean, w, nfft, twiddle, 1, out_format,
                                      eps, use_floor, relative_floor_db, y, st);
    return fail(DSA_ERR_UNSUPPORTED, "stft: unsupported dtype%s");
}

// --------------------------------------------------------------------------- fused STFT -> mel filter bank
// Host side of the filter-bank epilogue of stft512_fwd_pk_kernel<.., FB = true> (stft_pk.h).
// dsa_fbank_scan_plan turns H (host, float64, 257 x C) into the (64, 32) float32 per-lane table; it is the C
// statement of diffsptk_amd/utils/tables.py:fbank_scan_plan / fbank_scan_table (tests compare the two bit for bit).
DSA_EXPORT int dsa_fbank_scan_plan(const double* H, int32_t K, int32_t C, float* table)
{
    DSA_REQUIRE(H && table, "fbank_scan_plan: null pointer");
    if (K != 257 || C < 1 || C > 126) return fail(DSA_ERR_UNSUPPORTED, "fbank_scan_plan: needs 257 bins and at most 126 channels%s");
    int jk[257] = {0};
    double wd[257] = {0.0}, wu[257] = {0.0};
    int prev = 0;
    for (int k = 1; k < K - 1; ++k) {
        int nz[3], n = 0;
        for (int c = 0; c < C; ++c) {
            const double h = H[(size_t)k * C + c];
            if (!std::isfinite(h)) return fail(DSA_ERR_UNSUPPORTED, "fbank_scan_plan: non-finite weight%s");
            if (h != 0.0) {
                if (n < 3) nz[n] = c;
                ++n;
            }
        }
        int j;
        if (n == 0) {
            j = prev;
        } else if (n == 1) {
            const int c = nz[0];
            if (c >= prev) j = c, wu[k] = H[(size_t)k * C + c];
            else if (c + 1 >= prev) j = c + 1, wd[k] = H[(size_t)k * C + c];
            else return fail(DSA_ERR_UNSUPPORTED, "fbank_scan_plan: channels are not ordered along the bins%s");
        } else if (n == 2 && nz[1] == nz[0] + 1 && nz[1] >= prev) {
            j = nz[1], wd[k] = H[(size_t)k * C + nz[0]], wu[k] = H[(size_t)k * C + nz[1]];
        } else {
            return fail(DSA_ERR_UNSUPPORTED, "fbank_scan_plan: a bin feeds more than two adjacent channels%s");
        }
        jk[k] = prev = j;
    }
    for (int c = 0; c < C; ++c)
        if (!std::isfinite(H[c]) || !std::isfinite(H[(size_t)(K - 1) * C + c]))
            return fail(DSA_ERR_UNSUPPORTED, "fbank_scan_plan: non-finite weight%s");
    memset(table, 0, sizeof(float) * 64 * 32);
    int32_t* ti = reinterpret_cast<int32_t*>(table);
    bool valid[2][128] = {{false}};
    for (int h = 0; h < 2; ++h) {
        int j0[64], j1[64], run[64];
        for (int l = 0; l < 64; ++l) {
            const int b0 = h == 0 ? 2 * l + 1 : 255 - 2 * l, b1 = h == 0 ? 2 * l + 2 : 254 - 2 * l;
            j0[l] = jk[b0], j1[l] = jk[b1];
            float* t = table + l * 32;
            t[0 + h] = (float)wd[b0], t[2 + h] = (float)wu[b0];
            t[4 + h] = (float)wd[b1], t[6 + h] = (float)wu[b1];
            if (h == 1 && l == 63) t[4 + h] = t[6 + h] = 0.f;   // bin 128 belongs to the lower half
            t[8 + h] = j0[l] != j1[l] ? 0.f : 1.f;
        }
        run[0] = 0;
        for (int l = 1; l < 64; ++l) run[l] = (j0[l] == j1[l] && j1[l - 1] == j0[l]) ? run[l - 1] + 1 : 0;
        for (int l = 0; l < 64; ++l) {
            float* t = table + l * 32;
            float* m = t + 10 + 6 * h;
            m[0] = run[l] >= 1, m[1] = run[l] >= 2, m[2] = run[l] >= 4, m[3] = run[l] >= 8;
            m[4] = ((l / 16) % 2 == 1) && run[l] >= l % 16 + 1;
            m[5] = l >= 32 && run[l] >= l - 31;
            t[22 + h] = (l > 0 && j1[l - 1] == j0[l]) ? 1.f : 0.f;
            const bool isE = l == 63 || j0[l + 1] != j1[l], isM = j0[l] != j1[l];
            ti[l * 32 + 24] |= (j1[l] << (8 * h)) | (j0[l] << (16 + 8 * h));
            ti[l * 32 + 25] |= ((int)isE << h) | ((int)isM << (2 + h));
            if (isE) valid[h][j1[l]] = true;
            if (isM) valid[h][j0[l]] = true;
        }
    }
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 2; ++r) {
            const int c = l + 64 * r;
            if (c >= C) continue;
            table[l * 32 + 26 + 2 * r] = (float)H[c];
            table[l * 32 + 27 + 2 * r] = (float)H[(size_t)(K - 1) * C + c];
            // channel c reads the up-slope sums of interval c and the down-slope sums of interval c + 1
            ti[l * 32 + 30] |= ((int)valid[0][c] | ((int)valid[1][c] << 1) | ((int)valid[0][c + 1] << 2) | ((int)valid[1][c + 1] << 3)) << (4 * r);
        }
    bool has_ends = false;
    for (int c = 0; c < C; ++c) has_ends = has_ends || H[c] != 0.0 || H[(size_t)(K - 1) * C + c] != 0.0;
    if (has_ends)
        for (int l = 0; l < 64; ++l) ti[l * 32 + 30] |= 256;   // bit 8 (every lane): bins 0 / 256 carry weight
    return DSA_OK;
}

DSA_EXPORT int dsa_stft_fbank_fwd(const void* x, int64_t B, int64_t T, int32_t L, int32_t P, int32_t nfft, const void* w,
                                  const void* twiddle, int32_t center, double eps, const void* plan, int32_t C, double floor,
                                  double gamma, int32_t use_power, int32_t dtype, void* y, void* stream)
{
    DSA_REQUIRE(L > 0 && P > 0 && T > 0 && B >= 0, "stft_fbank: sizes must be positive");
    DSA_REQUIRE(B == 0 || (x && w && twiddle && plan && y), "stft_fbank: null pointer");
    DSA_REQUIRE(floor > 0, "stft_fbank: floor must be positive");
    if (!(dtype == DSA_F32 && nfft == 512 && L == 400 && (P & 1) == 0 && 3 * P + 512 <= kFPW * kZS * 2 && C >= 1 && C <= 126))
        return fail(DSA_ERR_UNSUPPORTED,
                    "stft_fbank: the fused kernel needs float32, fft_length 512, frame_length 400, an even frame period and at most "
                    "126 channels (use dsa_stft_fwd + dsa_fbank_fwd)%s");
    hipStream_t st = (hipStream_t)stream;
    const int64_t N = dsa_num_frames(T, P);
    if (B * N == 0) return DSA_OK;
    const int left = center ? L / 2 : 0;
    const int chunks_per_utt = (int)((N + kFPW - 1) / kFPW);
    const long total_chunks = (long)B * chunks_per_utt;
    long waves = 256L * 16;   // four waves per SIMD, four-wave workgroups
    if (waves > total_chunks) waves = total_chunks;
    const int lds = 4 * kFPW * kZS * 8 + 256 * 8 + 16 * 13 * 8 + 128 * 8;
#define DSA_FB_LAUNCH(MODE)                                                                                                  \
    hipLaunchKernelGGL((stft512_fwd_pk_kernel<0, 400, true, MODE>), dim3((unsigned)((waves + 3) / 4)), dim3(256), lds, st,     \
                       (const float*)x, (long)T, (long)N, L, P, left, (const float*)w, (const float*)twiddle, (float)eps,      \
                       (float*)y, total_chunks, chunks_per_utt, (const float*)plan, (float)floor, (float)gamma, C, 0, (int)DSA_PAD_CONSTANT)
    if (use_power) DSA_FB_LAUNCH(1);
    else DSA_FB_LAUNCH(2);
#undef DSA_FB_LAUNCH
    return check_launch("stft512_fbank_fwd");
}

// --------------------------------------------------------------------------- backward entries
namespace dsa {

template <typename T>
static int launch_row_dft_bwd(const void* x, int64_t B, int64_t Tlen, int64_t N, int L, int P, int left,
                              int mode, int zmean, const void* w, int nfft, const void* twiddle,
                              int out_kind, int fmt, double eps, int use_floor, double floor_db,
                              const void* gy, void* gframe, void* gwpart, hipStream_t st)
{
    int64_t F = B * N;
    if (F == 0) return DSA_OK;
    T floor_lin = use_floor ? (T)pow(10.0, floor_db / 10.0) : T(0);
    const int K = nfft / 2 + 1;
    size_t lds = sizeof(T) * ((size_t)L + 3 * (size_t)K);
    static const bool direct_only = [] {
        const char* e = getenv("DSA_ROWDFT_DIRECT");
        return e && atoi(e) != 0;
    }();
    const size_t lds_fft = lds + sizeof(T) * 2 * (size_t)nfft;
    if (!direct_only && nfft >= 32 && (nfft & (nfft - 1)) == 0 && lds_fft <= 150 * 1024) {
        static std::atomic<uint64_t> lds_set{0};
        if (lds_fft > 48 * 1024 &&
            !ensure_dynamic_lds(reinterpret_cast<const void*>(&row_dft_bwd_kernel<T, true>), 150 * 1024, lds_set))
            return fail(DSA_ERR_LAUNCH, "row_fft_bwd: cannot raise the dynamic LDS limit%s");
        hipLaunchKernelGGL((row_dft_bwd_kernel<T, true>), dim3((unsigned)F), dim3(256), lds_fft, st, (const T*)x, (long)Tlen,
                           (long)N, L, P, left, mode, zmean, (const T*)w, nfft, (const T*)twiddle, out_kind, fmt,
                           (T)eps, use_floor, floor_lin, (const T*)gy, (T*)gframe, (T*)gwpart);
        return check_launch("row_fft_bwd_generic");
    }
    if (lds > 60 * 1024) return fail(DSA_ERR_UNSUPPORTED, "row_dft_bwd: frame too long for LDS%s");
    hipLaunchKernelGGL((row_dft_bwd_kernel<T>), dim3((unsigned)F), dim3(256), lds, st, (const T*)x, (long)Tlen,
                       (long)N, L, P, left, mode, zmean, (const T*)w, nfft, (const T*)twiddle, out_kind, fmt,
                       (T)eps, use_floor, floor_lin, (const T*)gy, (T*)gframe, (T*)gwpart);
    return check_launch("row_dft_bwd_generic");
}

template <typename T>
static int stft_bwd_generic(const void* gy, const void* x, int64_t B, int64_t Tlen, int L, int P, int nfft,
                            const void* w, const void* twiddle, int center, int zmean, int pad_mode,
                            double eps, int use_floor, double floor_db, int fmt, void* gx, void* gw,
                            hipStream_t st)
{
    int64_t N = dsa_num_frames(Tlen, P), F = B * N;
    int left = center ? L / 2 : 0;
    T *gframe = nullptr, *gwpart = nullptr;
    size_t bytes = sizeof(T) * (size_t)F * L;
    if (hipMallocAsync((void**)&gframe, bytes, st) != hipSuccess ||
        (gw && hipMallocAsync((void**)&gwpart, bytes, st) != hipSuccess))
        return fail(DSA_ERR_LAUNCH, "stft_bwd: workspace allocation failed%s");
    int rc = launch_row_dft_bwd<T>(x, B, Tlen, N, L, P, left, pad_mode, zmean, w, nfft, twiddle, 1, fmt, eps,
                                   use_floor, floor_db, gy, gframe, gwpart, st);
    // overlap-add (zmean already folded into gframe)
    if (rc == DSA_OK) rc = frame_bwd_impl<T>(gframe, B, Tlen, L, P, center, 0, pad_mode, gx, st);
    if (rc == DSA_OK && gw) {
        hipLaunchKernelGGL((colsum_kernel<T>), dim3(L), dim3(256), 0, st, (const T*)gwpart, (long)F, L, (T*)gw);
        rc = check_launch("window_grad_colsum");
    }
    (void)hipFreeAsync(gframe, st);
    if (gwpart) (void)hipFreeAsync(gwpart, st);
    return rc;
}

}  // namespace dsa

DSA_EXPORT int dsa_fftr_bwd(const void* gy, const void* x, int64_t F, int32_t len_in, int32_t nfft,
                            int32_t out_format, const void* twiddle, int32_t dtype, void* gx, void* stream)
{
    DSA_REQUIRE(len_in > 0 && nfft > 0 && nfft % 2 == 0, "fftr_bwd: fft_length must be positive even");
    DSA_REQUIRE(out_format >= 0 && out_format <= 4, "fftr_bwd: unknown out_format");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSA_F32)
        return launch_row_dft_bwd<float>(x, F, len_in, 1, len_in, len_in, 0, 0, 0, nullptr, nfft, twiddle, 0,
                                         out_format, 0.0, 0, 0.0, gy, gx, nullptr, st);
    if (dtype == DSA_F64)
        return launch_row_dft_bwd<double>(x, F, len_in, 1, len_in, len_in, 0, 0, 0, nullptr, nfft, twiddle, 0,
                                          out_format, 0.0, 0, 0.0, gy, gx, nullptr, st);
    return fail(DSA_ERR_UNSUPPORTED, "fftr_bwd: unsupported dtype%s");
}

DSA_EXPORT int dsa_spec_bwd(const void* gy, const void* b, int32_t lb, const void* a, int32_t la, int64_t F,
                            int32_t nfft, double eps, int32_t use_floor, double relative_floor_db,
                            int32_t out_format, const void* twiddle, int32_t dtype, void* gb, void* ga,
                            void* stream)
{
    DSA_REQUIRE(F == 0 || b || a, "spec_bwd: either b or a must be specified");
    DSA_REQUIRE(nfft > 1 && nfft % 2 == 0, "spec_bwd: fft_length must be positive even");
    hipStream_t st = (hipStream_t)stream;
    if (a) {
        DSA_REQUIRE(F == 0 || ga != nullptr, "spec_bwd: ga is required when a is given");
        if (F == 0) return DSA_OK;
        const int K = nfft / 2 + 1;
        const size_t esz = dtype == DSA_F32 ? 4 : 8;
        const size_t lds = esz * ((size_t)lb + la + 5 * (size_t)K);
        if (lds > 60 * 1024) return fail(DSA_ERR_UNSUPPORTED, "spec_bwd: rows too long for LDS%s");
        const double fl = use_floor ? pow(10.0, relative_floor_db / 10.0) : 0.0;
        if (dtype == DSA_F32)
            hipLaunchKernelGGL((spec_ratio_bwd_kernel<float>), dim3((unsigned)F), dim3(128), lds, st, (const float*)gy,
                               (const float*)b, b ? lb : 0, (const float*)a, la, nfft, (const float*)twiddle, (float)eps,
                               use_floor, (float)fl, out_format, (float*)gb, (float*)ga);
        else if (dtype == DSA_F64)
            hipLaunchKernelGGL((spec_ratio_bwd_kernel<double>), dim3((unsigned)F), dim3(128), lds, st, (const double*)gy,
                               (const double*)b, b ? lb : 0, (const double*)a, la, nfft, (const double*)twiddle, eps,
                               use_floor, fl, out_format, (double*)gb, (double*)ga);
        else
            return fail(DSA_ERR_UNSUPPORTED, "spec_bwd: unsupported dtype%s");
        return check_launch("spec_ratio_bwd");
    }
    if (dtype == DSA_F32)
        return launch_row_dft_bwd<float>(b, F, lb, 1, lb, lb, 0, 0, 0, nullptr, nfft, twiddle, 1, out_format, eps,
                                         use_floor, relative_floor_db, gy, gb, nullptr, st);
    if (dtype == DSA_F64)
        return launch_row_dft_bwd<double>(b, F, lb, 1, lb, lb, 0, 0, 0, nullptr, nfft, twiddle, 1, out_format, eps,
                                          use_floor, relative_floor_db, gy, gb, nullptr, st);
    return fail(DSA_ERR_UNSUPPORTED, "spec_bwd: unsupported dtype%s");
}

// dsa_stft_bwd and dsa_istft_fwd: div / div_eps only with out_format DSA_SPEC_COMPLEX_INV (the result is divided by
// div[t] + div_eps); x may be NULL then (a complex cotangent needs no X; the generic kernels get zeros).
static int stft_bwd_impl(const void* gy, const void* x, int64_t B, int64_t T, int32_t L, int32_t P,
                         int32_t nfft, const void* w, const void* twiddle, int32_t center, int32_t zmean,
                         int32_t pad_mode, double eps, int32_t use_floor, double relative_floor_db,
                         int32_t out_format, int32_t dtype, int32_t algo, void* gx, void* gw, void* stream,
                         const void* div, double div_eps)
{
    DSA_REQUIRE(L > 0 && P > 0 && T > 0 && B >= 0, "stft_bwd: sizes must be positive");
    DSA_REQUIRE(nfft > 1 && nfft % 2 == 0, "stft_bwd: fft_length must be positive even");
    DSA_REQUIRE(pad_mode >= 0 && pad_mode <= 3, "stft_bwd: unknown pad mode");
    DSA_REQUIRE(out_format >= 0 && out_format <= 5, "stft_bwd: unknown out_format");
    if (B == 0) return DSA_OK;
    hipStream_t st = (hipStream_t)stream;
    {
        // fft_length 1024 / 2048 (the 44.1 / 48 kHz set-ups), power format, constant padding, no zmean, no relative floor, fixed window:
        // the packed kernel of stft_bwd_pk_big.h (round 6; DSA_STFT_BIG_BWD=0: the generic backward, for A/B runs)
        const char* bigb_e = getenv("DSA_STFT_BIG_BWD");   // (read per call: the tests switch it in-process)
        const bool bigb_on = !(bigb_e && bigb_e[0] == '0');
        const int64_t Nb = dsa_num_frames(T, P);
        const int leftb = center ? L / 2 : 0;
        if (bigb_on && dtype == DSA_F32 && algo != DSA_ALGO_GENERIC && (nfft == 1024 || nfft == 2048) && !zmean && !use_floor && !gw && !div &&
            out_format == DSA_SPEC_POWER && pad_mode == DSA_PAD_CONSTANT && L <= nfft && (L & 1) == 0 && (P & 1) == 0 && (leftb & 1) == 0 &&
            (T & 1) == 0 && (((size_t)x) & 7) == 0 && (((size_t)gx) & 7) == 0 && B * Nb < (int64_t(1) << 31) && x && gy && gx) {
            const int S = nfft / 512, FPP = 4 / S;
            const int need = (L + 32 * S - 1) / (32 * S);   // sample pairs per lane
            const int ppu = (int)((Nb + FPP - 1) / FPP);    // passes per utterance
            const int warm = ((L + P - 1) / P - 1 + FPP - 1) / FPP;   // passes whose tails a run inherits
            const long waves = 256L * 2 * 4;                 // two four-wave workgroups per CU
            long want = (waves + B - 1) / B;                 // runs per utterance that fill the chip ...
            const long longest = ppu / (4 * (warm > 0 ? warm : 1)) > 0 ? ppu / (4 * (warm > 0 ? warm : 1)) : 1;   // ... but no run shorter than four warm-ups
            const int runs = (int)(want < longest ? want : longest);
            const long items = (long)B * runs;
            const long wv = items < waves ? items : waves;
            const dim3 g2((unsigned)((wv + 3) / 4));
            const int lds_bb = 4 * 4 * kZS * 8 + 256 * 8 + 4 * (2 * 256 * S) * 4;
#define DSA_BIGB_LAUNCH(SV, NRV)                                                                                                        \
    do {                                                                                                                                \
        static std::atomic<uint64_t> abb{0};                                                                                            \
        if (!ensure_dynamic_lds(reinterpret_cast<const void*>(&stft_big_bwd_pk_kernel<SV, NRV>), lds_bb, abb))                          \
            return fail(DSA_ERR_LAUNCH, "stft_bwd: cannot reserve LDS%s");                                                              \
        hipLaunchKernelGGL((stft_big_bwd_pk_kernel<SV, NRV>), g2, dim3(256), lds_bb, st, (const float*)x, (const float*)gy, (long)T,    \
                           (long)Nb, L, P, leftb, (const float*)w, (const float*)twiddle, (float*)gx, items, runs, ppu, warm);          \
    } while (0)
            if (S == 2) {
                if (need <= 10) DSA_BIGB_LAUNCH(2, 10);
                else if (need <= 13) DSA_BIGB_LAUNCH(2, 13);
                else DSA_BIGB_LAUNCH(2, 16);
            } else {
                if (need <= 10) DSA_BIGB_LAUNCH(4, 10);
                else if (need <= 13) DSA_BIGB_LAUNCH(4, 13);
                else DSA_BIGB_LAUNCH(4, 16);
            }
#undef DSA_BIGB_LAUNCH
            return check_launch(S == 2 ? "stft1024_bwd" : "stft2048_bwd");
        }
    }
    {
        int io_floats = 0;
        int lds = stft512_lds_bytes(L, P, &io_floats);
        bool tuned_ok = dtype == DSA_F32 && nfft == 512 && L <= 512 && lds <= 64 * 1024 && !use_floor && !gw &&
                        pad_mode == DSA_PAD_CONSTANT;
        if (algo == DSA_ALGO_TUNED && !tuned_ok)
            return fail(DSA_ERR_UNSUPPORTED,
                        "stft_bwd: tuned kernel needs float32, fft_length 512, constant padding, no floor, fixed window%s");
        if (tuned_ok && algo != DSA_ALGO_GENERIC) {
            int64_t N = dsa_num_frames(T, P);
            int chunks_per_utt = (int)((N + kFPW - 1) / kFPW);
            long total_chunks = (long)B * chunks_per_utt;
            int span = (kFPW - 1) * P + L;
            // the packed kernel with the overlap-add carried in registers (stft_bwd_pk.h): one launch, no workspace.
            // DSA_STFT_BWD_PK=0 keeps the two-kernel path (A/B)
            {
            const bool cplx = out_format == DSA_SPEC_COMPLEX || out_format == DSA_SPEC_COMPLEX_INV;
            const int left = center ? L / 2 : 0;
            static const int use_pk = [] {
                const char* e = getenv("DSA_STFT_BWD_PK");
                return e ? atoi(e) : 1;
            }();
            if (use_pk && !zmean && L == 400 && (P == 80 || P == 160) && (cplx || out_format == DSA_SPEC_POWER || out_format == DSA_SPEC_MAG)) {
                const int ppu = chunks_per_utt;                       // passes of four frames per utterance
                const long waves = 256L * 16;
                long want = (waves + B - 1) / B;                       // runs per utterance that fill the chip ...
                static const int min_run = [] {   // A/B knob: shortest run of passes a wave takes (each run adds one warm-up pass)
                    const char* e = getenv("DSA_STFT_BWD_MINRUN");
                    return e && atoi(e) > 0 ? atoi(e) : 4;
                }();
                const long longest = ppu / min_run > 0 ? ppu / min_run : 1;
                const int runs = (int)(want < longest ? want : longest);
                const long items = (long)B * runs;
                const long wv = items < waves ? items : waves;
                const dim3 g2((unsigned)((wv + 3) / 4));             // four-wave workgroups share the twiddle and window tables
                const int lds2 = 4 * kFPW * kZS * 8 + 256 * 8 + 16 * 13 * 8;
                const float cs = out_format == DSA_SPEC_COMPLEX_INV ? 1.f / 512.f : 0.5f;
                const float ce = out_format == DSA_SPEC_COMPLEX_INV ? 1.f : 2.f;
                // frame periods of 5 ms and 10 ms at 16 kHz (the 25 ms window): one instantiation each
#define DSA_STFT_BWD_PK_LAUNCH(PC, CP, MG)                                                                                   \
    hipLaunchKernelGGL((stft512_bwd_pk_kernel<400, PC, CP, MG>), g2, dim3(256), lds2, st, (const float*)x, (const float*)gy, \
                       (long)T, (long)N, left, (const float*)w, (const float*)twiddle, cs, ce, (float*)gx, items, runs, ppu,   \
                       (const float*)div, (float)div_eps, (float)eps)
                if (P == 80) {
                    if (cplx) DSA_STFT_BWD_PK_LAUNCH(80, true, false);
                    else if (out_format == DSA_SPEC_MAG) DSA_STFT_BWD_PK_LAUNCH(80, false, true);
                    else DSA_STFT_BWD_PK_LAUNCH(80, false, false);
                } else {
                    if (cplx) DSA_STFT_BWD_PK_LAUNCH(160, true, false);
                    else if (out_format == DSA_SPEC_MAG) DSA_STFT_BWD_PK_LAUNCH(160, false, true);
                    else DSA_STFT_BWD_PK_LAUNCH(160, false, false);
                }
#undef DSA_STFT_BWD_PK_LAUNCH
                return check_launch("stft512_bwd_pk");
            }
            }
            float* part = nullptr;
            if (hipMallocAsync((void**)&part, sizeof(float) * (size_t)total_chunks * span, st) != hipSuccess)
                return fail(DSA_ERR_LAUNCH, "stft_bwd: workspace allocation failed%s");
            int waves_per_cu = 144 * 1024 / lds;
            if (waves_per_cu > 4 * DSA_STFT_BWD_WAVES) waves_per_cu = 4 * DSA_STFT_BWD_WAVES;
            long grid = 256L * waves_per_cu;
            if (grid > total_chunks) grid = total_chunks;
            int left = center ? L / 2 : 0;
            const bool cplx = out_format == DSA_SPEC_COMPLEX || out_format == DSA_SPEC_COMPLEX_INV;
#define DSA_STFT_BWD_LAUNCH(ZM, CP, PL)                                                                                   \
    hipLaunchKernelGGL((stft512_bwd_kernel<ZM, CP, PL>), dim3((unsigned)grid), dim3(64), lds, st, (const float*)x,        \
                       (const float*)gy, (long)T, (long)N, L, P, left, pad_mode, (const float*)w, (const float*)twiddle, \
                       (float)eps, out_format, part, total_chunks, chunks_per_utt, span)
            if (cplx) {
                long waves = 256L * 16;
                if (waves > total_chunks) waves = total_chunks;
                const int lds2 = 2 * kFPW * kZS * 8 + 256 * 8 + 16;
                const dim3 g2((unsigned)((waves + 1) / 2));
                if (zmean)
                    hipLaunchKernelGGL((stft512_bwd_kernel<true, true, false>), g2, dim3(128), lds2, st, (const float*)x,
                                       (const float*)gy, (long)T, (long)N, L, P, left, pad_mode, (const float*)w,
                                       (const float*)twiddle, (float)eps, out_format, part, total_chunks, chunks_per_utt, span);
                else
                    hipLaunchKernelGGL((stft512_bwd_kernel<false, true, false>), g2, dim3(128), lds2, st, (const float*)x,
                                       (const float*)gy, (long)T, (long)N, L, P, left, pad_mode, (const float*)w,
                                       (const float*)twiddle, (float)eps, out_format, part, total_chunks, chunks_per_utt, span);
            }
            else if (zmean) DSA_STFT_BWD_LAUNCH(true, false, false);
            else if (out_format == DSA_SPEC_POWER && pad_mode == DSA_PAD_CONSTANT) {
                // the plain instantiation: four waves per SIMD in two-wave workgroups (as the forward)
                long waves = 256L * 16;
                if (waves > total_chunks) waves = total_chunks;
                const int lds2 = 2 * kFPW * kZS * 8 + 256 * 8 + 16;
                hipLaunchKernelGGL((stft512_bwd_kernel<false, false, true>), dim3((unsigned)((waves + 1) / 2)), dim3(128), lds2,
                                   st, (const float*)x, (const float*)gy, (long)T, (long)N, L, P, left, pad_mode,
                                   (const float*)w, (const float*)twiddle, (float)eps, out_format, part, total_chunks,
                                   chunks_per_utt, span);
            }
            else DSA_STFT_BWD_LAUNCH(false, false, false);
#undef DSA_STFT_BWD_LAUNCH
            int rc = check_launch("stft512_bwd");
            if (rc == DSA_OK) {
                dim3 g2((unsigned)((T + 255) / 256), (unsigned)(B < 65535 ? B : 65535));
                hipLaunchKernelGGL(stft_span_gather_kernel, g2, dim3(256), 0, st, (const float*)part, (long)B, (long)T, P, left,
                                   span, chunks_per_utt, (float*)gx, (const float*)div, (float)div_eps);
                rc = check_launch("stft512_bwd");
            }
            (void)hipFreeAsync(part, st);
            return rc;
        }
    }
    if (dtype != DSA_F32 && dtype != DSA_F64) return fail(DSA_ERR_UNSUPPORTED, "stft_bwd: unsupported dtype%s");
    const size_t esz = dtype == DSA_F32 ? 4 : 8;
    void* x0 = nullptr;
    if (!x) {   // inverse transform through the generic kernels: they read a waveform, give them zeros
        if (hipMallocAsync(&x0, esz * (size_t)B * T, st) != hipSuccess || hipMemsetAsync(x0, 0, esz * (size_t)B * T, st) != hipSuccess)
            return fail(DSA_ERR_LAUNCH, "istft: workspace allocation failed%s");
        x = x0;
    }
    int rc = dtype == DSA_F32
                 ? stft_bwd_generic<float>(gy, x, B, T, L, P, nfft, w, twiddle, center, zmean, pad_mode, eps, use_floor,
                                           relative_floor_db, out_format, gx, gw, st)
                 : stft_bwd_generic<double>(gy, x, B, T, L, P, nfft, w, twiddle, center, zmean, pad_mode, eps, use_floor,
                                            relative_floor_db, out_format, gx, gw, st);
    if (x0) (void)hipFreeAsync(x0, st);
    if (rc == DSA_OK && div) rc = dsa_div_rows(gx, B, T, div, div_eps, dtype, gx, stream);
    return rc;
}

DSA_EXPORT int dsa_stft_bwd(const void* gy, const void* x, int64_t B, int64_t T, int32_t L, int32_t P,
                            int32_t nfft, const void* w, const void* twiddle, int32_t center, int32_t zmean,
                            int32_t pad_mode, double eps, int32_t use_floor, double relative_floor_db,
                            int32_t out_format, int32_t dtype, int32_t algo, void* gx, void* gw, void* stream)
{
    DSA_REQUIRE(B == 0 || T == 0 || x != nullptr, "stft_bwd: the waveform is required");
    return stft_bwd_impl(gy, x, B, T, L, P, nfft, w, twiddle, center, zmean, pad_mode, eps, use_floor, relative_floor_db,
                         out_format, dtype, algo, gx, gw, stream, nullptr, 0.0);
}

// InverseShortTimeFourierTransform._forward istft.py:186-193 in one call: y:(B,N,nfft/2+1) complex pairs ->
// out:(B,T) = overlap-add(window * irfft(y)[:L]) / (d + d_eps), d:(T) the overlap-added squared window.
DSA_EXPORT int dsa_istft_fwd(const void* y, int64_t B, int64_t T, int32_t L, int32_t P, int32_t nfft, const void* w,
                             const void* twiddle, int32_t center, const void* d, double d_eps, int32_t dtype, int32_t algo,
                             void* out, void* stream)
{
    DSA_REQUIRE(d != nullptr, "istft: the window-square sum is required");
    return stft_bwd_impl(y, nullptr, B, T, L, P, nfft, w, twiddle, center, 0, DSA_PAD_CONSTANT, 0.0, 0, 0.0,
                         DSA_SPEC_COMPLEX_INV, dtype, algo, out, nullptr, stream, d, d_eps);
}

// --------------------------------------------------------------------------- inverse path (8(f) row 2)
DSA_EXPORT int dsa_irfft_scale(const void* y, int64_t F, int32_t nfft, int32_t dtype, void* out, void* stream)
{
    DSA_REQUIRE(nfft > 1 && nfft % 2 == 0, "irfft_scale: fft_length must be positive even");
    const int K = nfft / 2 + 1;
    const long total = (long)F * K;
    if (total == 0) return DSA_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (dtype == DSA_F32)
        hipLaunchKernelGGL((irfft_scale_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)y, total, K, nfft, (float*)out);
    else if (dtype == DSA_F64)
        hipLaunchKernelGGL((irfft_scale_kernel<double>), dim3(blocks), dim3(256), 0, st, (const double*)y, total, K, nfft, (double*)out);
    else
        return fail(DSA_ERR_UNSUPPORTED, "irfft_scale: unsupported dtype%s");
    return check_launch("irfft_scale");
}

DSA_EXPORT int dsa_div_rows(const void* x, int64_t B, int64_t T, const void* d, double eps, int32_t dtype, void* out,
                            void* stream)
{
    DSA_REQUIRE(B >= 0 && T >= 0, "div_rows: sizes must be non-negative");
    if (B * T == 0) return DSA_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((T + 255) / 256), (unsigned)(B < 1024 ? B : 1024));
    if (dtype == DSA_F32)
        hipLaunchKernelGGL((div_rows_kernel<float>), grid, dim3(256), 0, st, (const float*)x, (long)B, (long)T, (const float*)d, (float)eps, (float*)out);
    else if (dtype == DSA_F64)
        hipLaunchKernelGGL((div_rows_kernel<double>), grid, dim3(256), 0, st, (const double*)x, (long)B, (long)T, (const double*)d, eps, (double*)out);
    else
        return fail(DSA_ERR_UNSUPPORTED, "div_rows: unsupported dtype%s");
    return check_launch("div_rows");
}

DSA_EXPORT int dsa_griffin_update(const void* t, int64_t B, int64_t Nt, int64_t N, int32_t K, const void* y, const void* phase,
                                  void* t_prev, void* d_prev, int32_t first, double alpha, double beta, double gamma,
                                  double eps, int32_t dtype, void* z, void* stream)
{
    DSA_REQUIRE(B >= 0 && N >= 0 && K > 0 && Nt >= N, "griffin_update: the transform must cover the spectrogram's frames");
    DSA_REQUIRE(!t || (t_prev && d_prev), "griffin_update: the momentum buffers are required after the initial step");
    const long total = (long)B * N * K;
    if (total == 0) return DSA_OK;
    hipStream_t st = (hipStream_t)stream;
    long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (dtype == DSA_F32)
        hipLaunchKernelGGL((griffin_update_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)t, (long)B,
                           (long)Nt, (long)N, K, (const float*)y, (const float*)phase, (float*)t_prev, (float*)d_prev, first,
                           (float)alpha, (float)beta, (float)gamma, (float)eps, (float*)z);
    else if (dtype == DSA_F64)
        hipLaunchKernelGGL((griffin_update_kernel<double>), dim3((unsigned)blocks), dim3(256), 0, st, (const double*)t, (long)B,
                           (long)Nt, (long)N, K, (const double*)y, (const double*)phase, (double*)t_prev, (double*)d_prev,
                           first, alpha, beta, gamma, eps, (double*)z);
    else
        return fail(DSA_ERR_UNSUPPORTED, "griffin_update: unsupported dtype%s");
    return check_launch("griffin_update");
}
