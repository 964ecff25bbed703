// Mel-cepstral analysis (a6-a10) for gfx950: FrequencyTransform matmul and the
// MelCepstralAnalysis Newton iteration, forward and backward.
//
// Reference: diffsptk/modules/mcep.py:189-224, freqt.py:141-143, utils/private.py:291-302.
//
// MI355X-first restatement.  The reference runs, per Newton step, ifreqt (matmul) -> rfft ->
// exp -> irfft -> rfreqt (matmul) -> build (F,25,25) Toeplitz + Hankel -> LU solve, each as a
// separate ATen pass over HBM.  Every map except exp and the solve is LINEAR, so the host
// composes them once per configuration in float64 (diffsptk_amd/utils/tables.py):
//     G = irfft(.)[:H+1] with halved ends, then freqt          (H+1, M+1)
//     D = ifreqt, then Re rfft(., nfft)                        (M+1, H+1)
//     E = irfft(.)[:H+1], then rfreqt                          (H+1, 2M+1)
// One step is then  d = mc D ; e = exp(log X - 2 d) ; rt = e E ;
//                   mc += (T(rt[:M+1]) + H(rt))^{-1} (rt[:M+1] - alpha_vec)
// with D/E/G of the SAME shapes as the reference's warping matrices -- the three FFTs per step
// disappear and nothing but X (in) and mc (out) touches HBM.  T + H = 2 sum_w e(w) c(w) c(w)^T
// is symmetric positive definite, so elimination needs no pivoting.
//
// Kernel families:
//  * generic : one workgroup per frame, any (nfft, M), float32/float64, forward and backward.
//  * tuned   : see mcep_mfma.hip (float32, f32 MFMA, 16 frames per wave).
#include "common.h"

namespace dsa {

// out(F,L2) = c(F,L1) @ A(L1,L2): one thread per output element, A streamed from L2.
template <typename T>
__global__ void matmul_rows_kernel(const T* __restrict__ c, long F, int L1, const T* __restrict__ A,
                                   int L2, T* __restrict__ out)
{
    extern __shared__ unsigned char smem_raw[];
    T* cs = reinterpret_cast<T*>(smem_raw);
    long f = blockIdx.x;
    for (int j = threadIdx.x; j < L1; j += blockDim.x) cs[j] = c[f * L1 + j];
    __syncthreads();
    for (int i = threadIdx.x; i < L2; i += blockDim.x) {
        T s = 0;
        for (int j = 0; j < L1; ++j) s += cs[j] * A[(long)j * L2 + i];
        out[f * L2 + i] = s;
    }
}

// gc(F,L1) = gout(F,L2) @ A^T
template <typename T>
__global__ void matmul_rows_t_kernel(const T* __restrict__ g, long F, int L1, const T* __restrict__ A,
                                     int L2, T* __restrict__ gc)
{
    extern __shared__ unsigned char smem_raw[];
    T* gs = reinterpret_cast<T*>(smem_raw);
    long f = blockIdx.x;
    for (int i = threadIdx.x; i < L2; i += blockDim.x) gs[i] = g[f * L2 + i];
    __syncthreads();
    for (int j = threadIdx.x; j < L1; j += blockDim.x) {
        T s = 0;
        for (int i = 0; i < L2; ++i) s += gs[i] * A[(long)j * L2 + i];
        gc[f * L1 + j] = s;
    }
}

// out(F,Lout) = c(F,Lin) @ M for a SMALL matrix (M and a 64-row tile of c fit in LDS): M = A (L1 x L2,
// TRANS = false) or A^T (TRANS = true).  Persistent workgroups of 256 threads; per tile of 64 rows the
// rows are copied to LDS as one contiguous stretch (row stride Lin + 1: conflict-free), and a thread owns
// output column i for 4 consecutive rows (one read of M feeds 4 FMAs; the rows' reads are broadcasts).
// The one-workgroup-per-row kernels above launch F tiny workgroups: 0.10 ms for the 40 x 13 DCT of 204 800
// frames against 0.02 ms here.
constexpr int kMrRows = 64;

template <typename T, bool TRANS>
__global__ __launch_bounds__(256) void matmul_rows_lds_kernel(const T* __restrict__ c, long F, int Lin,
                                                              const T* __restrict__ A, int L1, int L2, int Lout,
                                                              T* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* Ms = reinterpret_cast<T*>(smem_raw);   // [Lin][Lout]
    T* cs = Ms + (size_t)Lin * Lout;          // [64][Lin + 1]
    const int S = Lin + 1;
    for (int q = threadIdx.x; q < Lin * Lout; q += 256) {
        const int j = q / Lout, i = q - j * Lout;
        Ms[q] = TRANS ? A[(long)i * L2 + j] : A[(long)j * L2 + i];
    }
    const long ntiles = (F + kMrRows - 1) / kMrRows;
    const int items = Lout * (kMrRows / 4);
    for (long tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const long f0 = tl * kMrRows;
        const int rows = (int)(F - f0 < kMrRows ? F - f0 : kMrRows);
        __syncthreads();
        for (int q = threadIdx.x; q < kMrRows * Lin; q += 256) {
            const int r = q / Lin, j = q - r * Lin;
            cs[r * S + j] = r < rows ? c[f0 * Lin + q] : T(0);
        }
        __syncthreads();
        for (int it = threadIdx.x; it < items; it += 256) {
            const int rg = it / Lout, i = it - rg * Lout;
            const T* c0 = cs + (4 * rg) * S;
            T acc[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll 4
            for (int j = 0; j < Lin; ++j) {
                const T a = Ms[j * Lout + i];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += c0[q * S + j] * a;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * rg + q < rows) out[(f0 + 4 * rg + q) * Lout + i] = acc[q];
        }
    }
}

template <typename T, bool TRANS>
static bool matmul_rows_lds_launch(const void* c, int64_t F, int Lin, const void* A, int L1, int L2, int Lout, void* out,
                                   hipStream_t st)
{
    const size_t lds = sizeof(T) * ((size_t)Lin * Lout + (size_t)kMrRows * (Lin + 1));
    if (lds > 48 * 1024) return false;   // (the geometry alone decides: a row's result must not depend on how many rows share the call)
    long blocks = (long)((F + kMrRows - 1) / kMrRows);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL((matmul_rows_lds_kernel<T, TRANS>), dim3((unsigned)blocks), dim3(256), lds, st, (const T*)c, (long)F,
                       Lin, (const T*)A, L1, L2, Lout, (T*)out);
    return true;
}

// Reduce the (M1 x M1) symmetric positive definite system held in LDS as an augmented
// row-major matrix Aug[M1][W] (W - M1 right-hand sides) to diagonal form by Gauss-Jordan
// elimination without pivoting; afterwards x_c[i] = Aug[i][M1 + c] / Aug[i][i].
// `fac` is M1 elements of scratch.  All threads of the block cooperate.
template <typename T>
__device__ void gauss_jordan(T* Aug, int M1, int W, T* fac)
{
    for (int k = 0; k < M1; ++k) {
        __syncthreads();  // previous step's updates (incl. the new pivot) are visible
        T inv = T(1) / Aug[k * W + k];
        // factors first (column k is read by every update of this step)
        for (int i = threadIdx.x; i < M1; i += blockDim.x)
            fac[i] = (i == k) ? T(0) : Aug[i * W + k] * inv;
        __syncthreads();
        int ncol = W - (k + 1);
        for (int idx = threadIdx.x; idx < M1 * ncol; idx += blockDim.x) {
            int i = idx / ncol;
            int jj = k + 1 + (idx - i * ncol);
            if (i != k) Aug[i * W + jj] -= fac[i] * Aug[k * W + jj];
        }
    }
    __syncthreads();
}

// LDS carve-up shared by the generic forward and backward kernels
template <typename T>
struct McepLds {
    T *logx, *e, *mc, *rt, *aug, *sol, *aux1, *aux2;
    __device__ McepLds(unsigned char* base, int K, int M1, int M2)
    {
        T* p = reinterpret_cast<T*>(base);
        logx = p; p += K;
        e = p; p += K;
        aux1 = p; p += K;   // backward: gradient wrt log X
        mc = p; p += M1;
        sol = p; p += M1;
        aux2 = p; p += M1;  // backward: running gradient wrt mc
        rt = p; p += M2;
        aug = p;            // M1 * (M1 + 2)
    }
    static size_t bytes(int K, int M1, int M2)
    {
        return sizeof(T) * ((size_t)3 * K + 3 * M1 + M2 + (size_t)M1 * (M1 + 2));
    }
};

// one Newton step's forward quantities from mc (in LDS): e, rt, augmented system
template <typename T>
__device__ void newton_forward_parts(McepLds<T>& s, int K, int M1, int M2, int W,
                                     const T* __restrict__ D, const T* __restrict__ E,
                                     const T* __restrict__ av, const T* extra_rhs)
{
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        T d = 0;
        for (int m = 0; m < M1; ++m) d += s.mc[m] * D[(long)m * K + k];  // mcep.py:210-211
        s.e[k] = dsa_exp(s.logx[k] - d - d);                            // mcep.py:212
    }
    __syncthreads();
    for (int j = threadIdx.x; j < M2; j += blockDim.x) {
        T r = 0;
        for (int k = 0; k < K; ++k) r += s.e[k] * E[(long)k * M2 + j];  // mcep.py:214-215
        s.rt[j] = r;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < M1 * W; idx += blockDim.x) {
        int i = idx / W, j = idx - i * W;
        // R + Q (mcep.py:219-221) | right-hand side r - alpha_vector (mcep.py:216-217) | extra
        s.aug[idx] = j < M1 ? s.rt[i > j ? i - j : j - i] + s.rt[i + j]
                            : (j == M1 ? s.rt[i] - av[i] : extra_rhs[i]);
    }
    __syncthreads();
}

template <typename T>
__global__ void mcep_generic_fwd_kernel(const T* __restrict__ X, long F, int K, int M1, int M2,
                                        int n_iter, const T* __restrict__ G, const T* __restrict__ D,
                                        const T* __restrict__ E, const T* __restrict__ av,
                                        T* __restrict__ mc_out, T* __restrict__ hist)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    McepLds<T> s(smem_raw, K, M1, M2);
    const long f = blockIdx.x;
    for (int k = threadIdx.x; k < K; k += blockDim.x) s.logx[k] = dsa_log(X[f * K + k]);  // :203
    __syncthreads();
    for (int m = threadIdx.x; m < M1; m += blockDim.x) {
        T v = 0;
        for (int k = 0; k < K; ++k) v += s.logx[k] * G[(long)k * M1 + m];  // :204-207
        s.mc[m] = v;
        if (hist) hist[f * M1 + m] = v;
    }
    __syncthreads();
    for (int it = 0; it < n_iter; ++it) {
        newton_forward_parts(s, K, M1, M2, M1 + 1, D, E, av, (const T*)nullptr);
        gauss_jordan(s.aug, M1, M1 + 1, s.sol);
        for (int m = threadIdx.x; m < M1; m += blockDim.x) {
            T v = s.mc[m] + s.aug[m * (M1 + 1) + M1] / s.aug[m * (M1 + 1) + m];  // :221-222
            s.mc[m] = v;
            if (hist) hist[((long)(it + 1) * F + f) * M1 + m] = v;
        }
        __syncthreads();
    }
    for (int m = threadIdx.x; m < M1; m += blockDim.x) mc_out[f * M1 + m] = s.mc[m];
}

// Backward of the unrolled iteration.  For step k with saved mc_k and
// g_k = mc_{k+1} - mc_k, cotangent mbar of mc_{k+1}:
//   u = A^{-1} mbar (A symmetric);  Abar = -u g^T;  rtbar[m] = sum_{i+j=m} Abar_ij
//   + [m<=M] (sum_{|i-j|=m} Abar_ij + u_m);  ebar = rtbar E^T;  zbar = ebar * e;
//   logxbar += zbar;  mbar_k = mbar - 2 zbar D^T.
// Finally logxbar += mbar_0 G^T and Xbar = logxbar / X.
template <typename T>
__global__ void mcep_generic_bwd_kernel(const T* __restrict__ gmc, const T* __restrict__ X,
                                        const T* __restrict__ hist, long F, int K, int M1, int M2,
                                        int n_iter, const T* __restrict__ G, const T* __restrict__ D,
                                        const T* __restrict__ E, const T* __restrict__ av,
                                        T* __restrict__ gX)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    McepLds<T> s(smem_raw, K, M1, M2);
    T* lbar = s.aux1;
    T* mbar = s.aux2;
    const long f = blockIdx.x;
    const int W = M1 + 2;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        s.logx[k] = dsa_log(X[f * K + k]);
        lbar[k] = 0;
    }
    for (int m = threadIdx.x; m < M1; m += blockDim.x) mbar[m] = gmc[f * M1 + m];
    __syncthreads();
    for (int it = n_iter - 1; it >= 0; --it) {
        for (int m = threadIdx.x; m < M1; m += blockDim.x) s.mc[m] = hist[((long)it * F + f) * M1 + m];
        __syncthreads();
        // two right-hand sides: b = r - alpha_vector (re-derives g = mc_{k+1} - mc_k without the
        // cancellation of subtracting saved float32 iterates) and mbar
        newton_forward_parts(s, K, M1, M2, W, D, E, av, (const T*)mbar);
        gauss_jordan(s.aug, M1, W, s.sol);
        for (int m = threadIdx.x; m < M1; m += blockDim.x) {
            T dinv = T(1) / s.aug[m * W + m];
            s.mc[m] = s.aug[m * W + M1] * dinv;       // g
            s.sol[m] = s.aug[m * W + M1 + 1] * dinv;  // u
        }
        __syncthreads();
        // rtbar (into s.rt)
        for (int m = threadIdx.x; m < M2; m += blockDim.x) {
            T acc = 0;
            // Hankel part: i + j = m
            int ilo = m - (M1 - 1) > 0 ? m - (M1 - 1) : 0;
            int ihi = m < M1 - 1 ? m : M1 - 1;
            for (int i = ilo; i <= ihi; ++i) acc -= s.sol[i] * s.mc[m - i];
            if (m < M1) {
                // Toeplitz part: |i - j| = m
                for (int i = 0; i + m < M1; ++i) {
                    acc -= s.sol[i] * s.mc[i + m];
                    if (m > 0) acc -= s.sol[i + m] * s.mc[i];
                }
                acc += s.sol[m];  // through the right-hand side r - alpha_vector
            }
            s.rt[m] = acc;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            T eb = 0;
            for (int j = 0; j < M2; ++j) eb += s.rt[j] * E[(long)k * M2 + j];
            T zb = eb * s.e[k];
            lbar[k] += zb;
            s.e[k] = T(-2) * zb;  // dbar
        }
        __syncthreads();
        for (int m = threadIdx.x; m < M1; m += blockDim.x) {
            T acc = mbar[m];
            for (int k = 0; k < K; ++k) acc += s.e[k] * D[(long)m * K + k];
            mbar[m] = acc;
        }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        T acc = lbar[k];
        for (int m = 0; m < M1; ++m) acc += mbar[m] * G[(long)k * M1 + m];
        gX[f * K + k] = acc / X[f * K + k];
    }
}

template <typename T>
static int mcep_generic_fwd(const void* X, int64_t F, int nfft, int M, int n_iter, const void* G,
                            const void* D, const void* E, const void* av, void* mc, void* hist,
                            hipStream_t st)
{
    int K = nfft / 2 + 1, M1 = M + 1, M2 = 2 * M + 1;
    size_t lds = McepLds<T>::bytes(K, M1, M2);
    if (lds > 160 * 1024) return fail(DSA_ERR_UNSUPPORTED, "mcep: configuration exceeds LDS%s");
    if (lds > 48 * 1024)   // (a growing size: set on every such launch, for the current device)
        hipFuncSetAttribute((const void*)mcep_generic_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mcep_generic_fwd_kernel<T>), dim3((unsigned)F), dim3(128), lds, st, (const T*)X,
                       (long)F, K, M1, M2, n_iter, (const T*)G, (const T*)D, (const T*)E, (const T*)av,
                       (T*)mc, (T*)hist);
    return check_launch("mcep_generic_fwd");
}

template <typename T>
static int mcep_generic_bwd(const void* gmc, const void* X, const void* hist, int64_t F, int nfft, int M,
                            int n_iter, const void* G, const void* D, const void* E, const void* av,
                            void* gX, hipStream_t st)
{
    int K = nfft / 2 + 1, M1 = M + 1, M2 = 2 * M + 1;
    size_t lds = McepLds<T>::bytes(K, M1, M2);
    if (lds > 160 * 1024) return fail(DSA_ERR_UNSUPPORTED, "mcep: configuration exceeds LDS%s");
    if (lds > 48 * 1024)
        hipFuncSetAttribute((const void*)mcep_generic_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mcep_generic_bwd_kernel<T>), dim3((unsigned)F), dim3(128), lds, st, (const T*)gmc,
                       (const T*)X, (const T*)hist, (long)F, K, M1, M2, n_iter, (const T*)G, (const T*)D,
                       (const T*)E, (const T*)av, (T*)gX);
    return check_launch("mcep_generic_bwd");
}

// tuned kernels (mcep_mfma.hip)
int mcep_mfma_supported(int nfft, int M, int dtype);
int64_t mcep_mfma_images_bytes();
int mcep_mfma_prepare(const void* G, const void* D, const void* E, void* images, hipStream_t st);
struct StftIn;
int mcep_mfma_fwd(const void* X, int64_t F, int n_iter, const void* G, const void* D, const void* E, const void* av,
                  const void* images, void* scratch, void* mc, void* hist, hipStream_t st, bool scratch_clean = false,
                  const StftIn* sti = nullptr, bool hist_has_rt = false, bool overlapped = false, int reserve_cus = 0);
int stft_mcep_fused_fwd(const void* x, int64_t B, int64_t T, int P, int center, const void* window, const void* twiddle, double eps,
                        int n_iter, const void* G, const void* D, const void* E, const void* av, const void* images, void* scratch,
                        void* mc, void* hist, void* X_out, hipStream_t st, bool scratch_clean, bool hist_has_rt, bool overlapped, int pad_mode,
                        int zmean, float floor_lin, int reserve_cus);
int mcep_mfma_bwd(const void* gmc, const void* X, const void* hist, int64_t F, int n_iter, const void* av,
                  const void* images, void* scratch, void* gX, hipStream_t st, bool has_workspace = false, bool hist_has_rt = false);

}  // namespace dsa

using namespace dsa;

DSA_EXPORT int dsa_freqt_fwd(const void* c, int64_t F, int32_t L1, const void* A, int32_t L2, int32_t dtype,
                             void* out, void* stream)
{
    DSA_REQUIRE(L1 > 0 && L2 > 0 && F >= 0, "freqt: sizes must be positive");
    if (F == 0) return DSA_OK;
    hipStream_t st = (hipStream_t)stream;
    // long float32 rows (the 257-bin spectra of mgcep.py:199-209 against 25 .. 49-column matrices): the float32 matrix-core
    // row product of fbank.hip, 48 output columns per launch (exact float32 products, float32 accumulation, as here)
    static const bool no_mfma = [] {
        const char* e = getenv("DSA_FREQT_GENERIC");
        return e && atoi(e) != 0;
    }();
    // (chosen from the geometry alone: a row's result must not depend on how many rows share the call)
    if (dtype == DSA_F32 && L1 > 48 && L1 <= 320 && L2 <= 192 && !no_mfma) {
        for (int c0 = 0; c0 < L2; c0 += 48) {
            const int cs = L2 - c0 < 48 ? L2 - c0 : 48;
            const int rc = fbank_mfma_launch_ex(c, F, L1, (const float*)A + c0, cs, L2, 1.0, 0.0, 1, 4, 1.0, (float*)out + c0, nullptr,
                                                st, "freqt_mfma_fwd", nullptr, 0, nullptr, L2);
            if (rc != DSA_OK) return rc;
        }
        return DSA_OK;
    }
    if ((dtype == DSA_F32 && matmul_rows_lds_launch<float, false>(c, F, L1, A, L1, L2, L2, out, st)) ||
        (dtype == DSA_F64 && matmul_rows_lds_launch<double, false>(c, F, L1, A, L1, L2, L2, out, st)))
        return check_launch("freqt_lds_fwd");
    int threads = L2 >= 192 ? 256 : (L2 >= 96 ? 128 : 64);
    if (dtype == DSA_F32)
        hipLaunchKernelGGL((matmul_rows_kernel<float>), dim3((unsigned)F), dim3(threads), sizeof(float) * L1, st,
                           (const float*)c, (long)F, L1, (const float*)A, L2, (float*)out);
    else if (dtype == DSA_F64)
        hipLaunchKernelGGL((matmul_rows_kernel<double>), dim3((unsigned)F), dim3(threads), sizeof(double) * L1, st,
                           (const double*)c, (long)F, L1, (const double*)A, L2, (double*)out);
    else
        return fail(DSA_ERR_UNSUPPORTED, "freqt: unsupported dtype%s");
    return check_launch("freqt_fwd");
}

DSA_EXPORT int dsa_freqt_bwd(const void* gout, int64_t F, int32_t L1, const void* A, int32_t L2, int32_t dtype,
                             void* gc, void* stream)
{
    DSA_REQUIRE(L1 > 0 && L2 > 0 && F >= 0, "freqt_bwd: sizes must be positive");
    if (F == 0) return DSA_OK;
    hipStream_t st = (hipStream_t)stream;
    if ((dtype == DSA_F32 && matmul_rows_lds_launch<float, true>(gout, F, L2, A, L1, L2, L1, gc, st)) ||
        (dtype == DSA_F64 && matmul_rows_lds_launch<double, true>(gout, F, L2, A, L1, L2, L1, gc, st)))
        return check_launch("freqt_lds_bwd");
    int threads = L1 >= 192 ? 256 : (L1 >= 96 ? 128 : 64);
    if (dtype == DSA_F32)
        hipLaunchKernelGGL((matmul_rows_t_kernel<float>), dim3((unsigned)F), dim3(threads), sizeof(float) * L2, st,
                           (const float*)gout, (long)F, L1, (const float*)A, L2, (float*)gc);
    else if (dtype == DSA_F64)
        hipLaunchKernelGGL((matmul_rows_t_kernel<double>), dim3((unsigned)F), dim3(threads), sizeof(double) * L2, st,
                           (const double*)gout, (long)F, L1, (const double*)A, L2, (double*)gc);
    else
        return fail(DSA_ERR_UNSUPPORTED, "freqt_bwd: unsupported dtype%s");
    return check_launch("freqt_bwd");
}

DSA_EXPORT int64_t dsa_mcep_images_bytes(int32_t nfft, int32_t M, int32_t dtype)
{
    return mcep_mfma_supported(nfft, M, dtype) ? mcep_mfma_images_bytes() : 0;
}

DSA_EXPORT int dsa_mcep_prepare(const void* G, const void* D, const void* E, int32_t nfft, int32_t M, int32_t dtype,
                                void* images, void* stream)
{
    if (!mcep_mfma_supported(nfft, M, dtype)) return DSA_OK;   // nothing to prepare: the generic kernels read G, D, E
    DSA_REQUIRE(G && D && E && images, "mcep_prepare: G, D, E and the images buffer are required");
    return mcep_mfma_prepare(G, D, E, images, (hipStream_t)stream);
}

DSA_EXPORT int dsa_mcep_fwd(const void* X, int64_t F, int32_t nfft, int32_t M, int32_t n_iter, const void* G,
                            const void* D, const void* E, const void* alpha_vec, int32_t dtype, int32_t algo,
                            const void* images, void* scratch, void* mc, void* mc_hist, void* stream)
{
    DSA_REQUIRE(nfft > 1 && nfft % 2 == 0, "mcep: fft_length must be positive even");
    DSA_REQUIRE(M >= 0 && 2 * M <= nfft, "mcep: cep_order must be in [0, fft_length/2]");
    DSA_REQUIRE(n_iter >= 0 && F >= 0, "mcep: n_iter must be non-negative");
    if (F == 0) return DSA_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool scratch_clean = (algo & DSA_ALGO_SCRATCH_IS_CLEAN) != 0;
    const bool hist_has_rt = (algo & DSA_ALGO_HIST_HAS_RT) != 0;
    const bool overlapped = (algo & DSA_ALGO_OVERLAPPED_LAUNCHES) != 0;
    const int reserve_cus = DSA_ALGO_RESERVED_CUS(algo);
    algo &= ~(DSA_ALGO_SCRATCH_IS_CLEAN | DSA_ALGO_HIST_HAS_RT | DSA_ALGO_OVERLAPPED_LAUNCHES | DSA_ALGO_RESERVE_CUS(63));
    bool tuned_ok = mcep_mfma_supported(nfft, M, dtype) != 0;
    if (algo == DSA_ALGO_TUNED && !tuned_ok)
        return fail(DSA_ERR_UNSUPPORTED, "mcep: tuned kernel needs float32, fft_length 512, cep_order 24%s");
    if (algo == DSA_ALGO_TUNED && !(images && scratch))
        return fail(DSA_ERR_INVALID_ARGUMENT, "mcep: the tuned kernel needs the prepared images (dsa_mcep_prepare) and a scratch buffer%s");
    if (tuned_ok && algo != DSA_ALGO_GENERIC && images && scratch)
        return mcep_mfma_fwd(X, F, n_iter, G, D, E, alpha_vec, images, scratch, mc, mc_hist, st, scratch_clean, nullptr, hist_has_rt, overlapped,
                             reserve_cus);
    if (dtype == DSA_F32) return mcep_generic_fwd<float>(X, F, nfft, M, n_iter, G, D, E, alpha_vec, mc, mc_hist, st);
    if (dtype == DSA_F64) return mcep_generic_fwd<double>(X, F, nfft, M, n_iter, G, D, E, alpha_vec, mc, mc_hist, st);
    return fail(DSA_ERR_UNSUPPORTED, "mcep: unsupported dtype%s");
}

// (0.2.0) dsa_stft_mcep_fwd with the framing / spectrum options of dsa_stft_fwd the one launch covers: zmean, the pad mode, the relative
// floor (stft.py:86-104; power format only)
DSA_EXPORT int dsa_stft_mcep_opts_fwd(const void* x, int64_t B, int64_t T, int32_t L, int32_t P, int32_t nfft, const void* window,
                                      const void* twiddle, int32_t center, int32_t zmean, int32_t pad_mode, double eps, int32_t use_floor,
                                      double relative_floor_db, int32_t M, int32_t n_iter, const void* G, const void* D, const void* E,
                                      const void* alpha_vec, int32_t dtype, int32_t algo, const void* images, void* scratch, void* mc,
                                      void* mc_hist, void* X_out, void* stream)
{
    DSA_REQUIRE(B >= 0 && T >= 0 && P >= 1 && L >= 1 && n_iter >= 0, "stft_mcep: invalid sizes");
    DSA_REQUIRE(pad_mode >= 0 && pad_mode <= 3, "stft_mcep: unknown pad mode");
    const bool scratch_clean = (algo & DSA_ALGO_SCRATCH_IS_CLEAN) != 0;
    const bool hist_has_rt = (algo & DSA_ALGO_HIST_HAS_RT) != 0;
    const bool overlapped = (algo & DSA_ALGO_OVERLAPPED_LAUNCHES) != 0;
    DSA_REQUIRE(pad_mode != DSA_PAD_REFLECT || (center ? L / 2 : L - 1) < T || L == 1, "stft_mcep: reflect padding needs pad < input length");
    const int64_t N = T <= 0 ? 0 : (T - 1) / P + 1;
    if (!(mcep_mfma_supported(nfft, M, dtype) && L == 400 && B * N < (int64_t(1) << 31) && T < (int64_t(1) << 31)))
        return fail(DSA_ERR_UNSUPPORTED, "stft_mcep: the fused launch covers float32, frame_length 400, fft_length 512, cep_order 24%s");
    if (B * N == 0) return DSA_OK;   // an empty batch is a no-op (its tensors have no storage: checked before the pointers)
    DSA_REQUIRE(images && scratch, "stft_mcep: the prepared images (dsa_mcep_prepare) and a scratch buffer are required");
    DSA_REQUIRE(x && window && twiddle && G && D && E && alpha_vec && mc, "stft_mcep: null pointer");
    const float floor_lin = use_floor ? (float)pow(10.0, relative_floor_db / 10.0) : -1.f;
    return stft_mcep_fused_fwd(x, B, T, P, center, window, twiddle, eps, n_iter, G, D, E, alpha_vec, images, scratch, mc, mc_hist,
                               X_out, (hipStream_t)stream, scratch_clean, hist_has_rt, overlapped, pad_mode, zmean != 0, floor_lin,
                               DSA_ALGO_RESERVED_CUS(algo));
}

DSA_EXPORT int dsa_stft_mcep_fwd(const void* x, int64_t B, int64_t T, int32_t L, int32_t P, int32_t nfft, const void* window,
                                 const void* twiddle, int32_t center, double eps, int32_t M, int32_t n_iter, const void* G,
                                 const void* D, const void* E, const void* alpha_vec, int32_t dtype, int32_t algo,
                                 const void* images, void* scratch, void* mc, void* mc_hist, void* X_out, void* stream)
{
    // (the plain configuration; DSA_ALGO_PAD_MODE(m) in `algo` selects the pad mode)
    return dsa_stft_mcep_opts_fwd(x, B, T, L, P, nfft, window, twiddle, center, 0, (algo >> 12) & 3, eps, 0, 0.0, M, n_iter, G, D, E, alpha_vec,
                                  dtype, algo & ~(3 << 12), images, scratch, mc, mc_hist, X_out, stream);
}

DSA_EXPORT int dsa_mcep_bwd(const void* gmc, const void* X, const void* mc_hist, int64_t F, int32_t nfft,
                            int32_t M, int32_t n_iter, const void* G, const void* D, const void* E,
                            const void* alpha_vec, int32_t dtype, int32_t algo, const void* images, void* scratch,
                            void* gX, void* stream)
{
    DSA_REQUIRE(nfft > 1 && nfft % 2 == 0, "mcep_bwd: fft_length must be positive even");
    DSA_REQUIRE(M >= 0 && 2 * M <= nfft && n_iter >= 0 && F >= 0, "mcep_bwd: invalid sizes");
    if (F == 0) return DSA_OK;
    DSA_REQUIRE(mc_hist != nullptr, "mcep_bwd: the forward history is required");
    hipStream_t st = (hipStream_t)stream;
    const bool has_workspace = (algo & DSA_ALGO_SCRATCH_HAS_WORKSPACE) != 0;
    const bool hist_has_rt = (algo & DSA_ALGO_HIST_HAS_RT) != 0;
    algo &= ~(DSA_ALGO_SCRATCH_HAS_WORKSPACE | DSA_ALGO_HIST_HAS_RT);
    bool tuned_ok = mcep_mfma_supported(nfft, M, dtype) != 0;
    if (algo == DSA_ALGO_TUNED && !tuned_ok)
        return fail(DSA_ERR_UNSUPPORTED, "mcep_bwd: tuned kernel needs float32, fft_length 512, cep_order 24%s");
    if (algo == DSA_ALGO_TUNED && !(images && scratch))
        return fail(DSA_ERR_INVALID_ARGUMENT, "mcep_bwd: the tuned kernel needs the prepared images (dsa_mcep_prepare) and a scratch buffer%s");
    if (tuned_ok && algo != DSA_ALGO_GENERIC && images && scratch)
        return mcep_mfma_bwd(gmc, X, mc_hist, F, n_iter, alpha_vec, images, scratch, gX, st, has_workspace, hist_has_rt);
    if (dtype == DSA_F32) return mcep_generic_bwd<float>(gmc, X, mc_hist, F, nfft, M, n_iter, G, D, E, alpha_vec, gX, st);
    if (dtype == DSA_F64) return mcep_generic_bwd<double>(gmc, X, mc_hist, F, nfft, M, n_iter, G, D, E, alpha_vec, gX, st);
    return fail(DSA_ERR_UNSUPPORTED, "mcep_bwd: unsupported dtype%s");
}
