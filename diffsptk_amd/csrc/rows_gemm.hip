// General row products on the float32 matrix instruction, for the geometries without a tuned kernel
// (the 48 kHz set-ups of utils/public.py:22-104 of the reference: fft_length 1024 / 2048, cepstral orders 34 .. 60).
//
//   out (F, N) = op_out( op_in(c) (F, K)  x  B (K, N) ),   B = A (K x N, row stride lda)  or  A^T (A: N x K, row stride lda)
//
// replaces, with the library's own code, what rounds 2-3 handed to the vendor GEMM and to stock element-wise operators:
//   * freqt.py:141-143 / mcep.py:286-288 row products with rows of 512 values and more (ops.MatmulRowsFn, both directions);
//   * the Newton step of MelCepstralAnalysis (mcep.py:203-215) for these geometries as fused launches:
//       PRO_LOG     c enters as log(c)                      (mcep.py:203 feeding :204-207)
//       EPI_EXPSUB  out = exp(aux - 2 acc)                  (mcep.py:210-212: e = exp(log X - 2 mc D))
// Arithmetic: v_mfma_f32_16x16x4_f32 -- exact float32 products, float32 accumulation (what the reference's float32 matmul
// does, in another summation order).  Bound: the float32 matrix rate (157 TFLOP/s dense); these products are a third of a
// step of the untuned analysis, the batched solve is the rest.
//
// Mapping: a workgroup of four waves owns 64 rows (frames) x up to 128 columns; a wave owns 16 rows and keeps 8 column tiles
// (16 x 16 accumulators).  K is walked in chunks of 32: the chunk of B is staged once per workgroup in LDS ([32][128] floats,
// double-buffered, coalesced global reads along the columns -- or along K for the transposed form), the wave's own 16 x 32
// values of c come straight from memory as two 16-byte loads per lane (k-slot g of step t <-> k = K0 + 8 g + t, so a lane's
// eight values are contiguous; op_in is applied once per value).
// (Tried and removed, round 4: 16 rows per workgroup with K split over its four waves for small batches -- 800 workgroups instead
// of 200 at 12 800 rows: 101 us against 82 us for the 1025 x 99 product; every wave then stages its own chunks of B.)
#include "common.h"

namespace dsa {

typedef float rg_f4 __attribute__((ext_vector_type(4)));
typedef float rg_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float rg_f2 __attribute__((ext_vector_type(2)));

constexpr int kRgKC = 32;     // K per chunk
constexpr int kRgNTMax = 8;   // column tiles of 16 per workgroup (template parameter NT <= 8)
constexpr int kRgLD = kRgNTMax * 16 + 2;   // row stride of the staged chunk (floats): 8 rows = 1040 = 16 (mod 64), so the four k-slots
                                        // (rows 8 g + t) of a read hit four distinct groups of 16 banks

enum { RG_PRO_LOG = 1, RG_EPI_EXPSUB = 2, RG_TRANS = 4 };

template <int FLAGS, int NT>
__global__ __launch_bounds__(256) void rows_gemm_mfma_kernel(const float* __restrict__ c, long F, int K, const float* __restrict__ A,
                                                             int lda, int N, const float* __restrict__ aux, int ldaux,
                                                             float* __restrict__ out, int ldo)
{
    constexpr bool PRO_LOG = (FLAGS & RG_PRO_LOG) != 0, EPI_EXPSUB = (FLAGS & RG_EPI_EXPSUB) != 0, TRANS = (FLAGS & RG_TRANS) != 0;
    __shared__ __attribute__((aligned(16))) float bs[2][kRgKC * kRgLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const long f0 = (long)blockIdx.x * 64 + wave * 16;
    const int col0 = blockIdx.y * (NT * 16);
    const int ncols = (N - col0) < NT * 16 ? (N - col0) : NT * 16;   // columns beyond are staged as zeros and never stored
    const long fr = f0 + n < F ? f0 + n : F - 1;           // clamped row: its results are not stored
    const float* crow = c + fr * (long)K;

    rg_f4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = rg_f4{0.f, 0.f, 0.f, 0.f};

    // stage chunk `kc` of B into bs[buf]: [kk][col] = B[K0 + kk][col0 + col], zero outside the matrix
    auto stage = [&](int kc, int buf) __attribute__((always_inline)) {
        const int K0 = kc * kRgKC;
        if (!TRANS) {
            // 32 x 128 floats, 256 threads: thread -> (kk = tid / 8 + 32 * 0, 16 consecutive columns as four 16-byte loads)
            const int kk = tid >> 3, cq = (tid & 7) * 16;
            const bool kok = K0 + kk < K;
            const float* src = A + (long)(K0 + kk) * lda + col0 + cq;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rg_f4 v = {0.f, 0.f, 0.f, 0.f};
                const int cb = cq + 4 * q;
                if (kok && cb + 3 < ncols) {
                    v = *reinterpret_cast<const rg_f4u*>(src + 4 * q);
                } else if (kok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (cb + e < ncols) v[e] = src[4 * q + e];
                }
                *reinterpret_cast<rg_f2*>(&bs[buf][kk * kRgLD + cb]) = rg_f2{v[0], v[1]};   // rows are 8-byte aligned (stride 130)
                *reinterpret_cast<rg_f2*>(&bs[buf][kk * kRgLD + cb + 2]) = rg_f2{v[2], v[3]};
            }
        } else {
            // B[k][col] = A[col0 + col][K0 + k]: thread -> (col = tid / 2, 16 consecutive k): reads along K, transposed stores
            const int col = tid >> 1, kq = (tid & 1) * 16;
            const bool cok = col < ncols;
            const float* src = A + (long)(col0 + col) * lda + K0 + kq;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rg_f4 v = {0.f, 0.f, 0.f, 0.f};
                const int kb = K0 + kq + 4 * q;
                if (cok && kb + 3 < K) {
                    v = *reinterpret_cast<const rg_f4u*>(src + 4 * q);
                } else if (cok) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kb + e < K) v[e] = src[4 * q + e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) bs[buf][(kq + 4 * q + e) * kRgLD + col] = v[e];
            }
        }
    };
    // this lane's eight values of c for chunk kc: k = K0 + 8 g + t, t = 0..7
    auto load_c = [&](int kc, float (&cv)[8]) __attribute__((always_inline)) {
        const int kb = kc * kRgKC + 8 * g;
        if (kb + 7 < K) {
            const rg_f4 a = *reinterpret_cast<const rg_f4u*>(crow + kb), b = *reinterpret_cast<const rg_f4u*>(crow + kb + 4);
            cv[0] = a[0], cv[1] = a[1], cv[2] = a[2], cv[3] = a[3], cv[4] = b[0], cv[5] = b[1], cv[6] = b[2], cv[7] = b[3];
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) cv[t] = kb + t < K ? crow[kb + t] : (PRO_LOG ? 1.f : 0.f);   // log(1) = 0
        }
    };

    const int nchunk = (K + kRgKC - 1) / kRgKC;
    float cv[8], cn[8];
    stage(0, 0);
    load_c(0, cv);
    __syncthreads();
    for (int kc = 0; kc < nchunk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nchunk) {
            stage(kc + 1, buf ^ 1);      // the other buffer: its readers passed the barrier at the end of chunk kc - 1
            load_c(kc + 1, cn);
        }
        if (PRO_LOG) {
#pragma unroll
            for (int t = 0; t < 8; ++t) cv[t] = logf(cv[t]);   // mcep.py:203
        }
        const float* bl = &bs[buf][(8 * g) * kRgLD + n];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[t], bl[t * kRgLD + 16 * ct], acc[ct], 0, 0, 0);
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 8; ++t) cv[t] = cn[t];
    }
    // D register r of lane (j = n, g) is D[4 g + r][j]: frame 4 g + r of the wave's 16, column j of the tile
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        const int col = col0 + 16 * ct + n;
        if (col >= N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long f = f0 + 4 * g + r;
            if (f >= F) continue;
            float v = acc[ct][r];
            if (EPI_EXPSUB) v = expf(aux[f * (long)ldaux + col] - 2.f * v);   // mcep.py:212
            out[f * (long)ldo + col] = v;
        }
    }
}

template <int FLAGS, int NT>
static void rows_gemm_launch_nt(const void* c, int64_t F, int K, const void* A, int lda, int N, const void* aux, int ldaux, void* out,
                                int ldo, hipStream_t st)
{
    const dim3 grid((unsigned)((F + 63) / 64), (unsigned)((N + NT * 16 - 1) / (NT * 16)));
    hipLaunchKernelGGL((rows_gemm_mfma_kernel<FLAGS, NT>), grid, dim3(256), 0, st, (const float*)c, (long)F, K, (const float*)A, lda, N,
                       (const float*)aux, ldaux, (float*)out, ldo);
}
template <int FLAGS>
static void rows_gemm_launch_t(const void* c, int64_t F, int K, const void* A, int lda, int N, const void* aux, int ldaux, void* out,
                               int ldo, hipStream_t st)
{
    // column tiles per workgroup: the whole width when it fits eight tiles (every value of c is then read once), else eight
    const int nt = (N + 15) / 16;
    if (nt <= 2) rows_gemm_launch_nt<FLAGS, 2>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st);
    else if (nt <= 4) rows_gemm_launch_nt<FLAGS, 4>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st);
    else if (nt <= 6) rows_gemm_launch_nt<FLAGS, 6>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st);
    else if (nt == 7) rows_gemm_launch_nt<FLAGS, 7>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st);
    else rows_gemm_launch_nt<FLAGS, 8>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st);
}

// float32 only; flags: RG_PRO_LOG | RG_EPI_EXPSUB | RG_TRANS (see the head of the file)
int rows_gemm_mfma(const void* c, int64_t F, int K, const void* A, int lda, int N, int flags, const void* aux, int ldaux, void* out,
                   int ldo, hipStream_t st)
{
    switch (flags) {
    case 0: rows_gemm_launch_t<0>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st); break;
    case RG_TRANS: rows_gemm_launch_t<RG_TRANS>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st); break;
    case RG_PRO_LOG: rows_gemm_launch_t<RG_PRO_LOG>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st); break;
    case RG_EPI_EXPSUB: rows_gemm_launch_t<RG_EPI_EXPSUB>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st); break;
    default: return fail(DSA_ERR_UNSUPPORTED, "rows_gemm: unsupported combination of flags%s");
    }
    return check_launch((flags & RG_TRANS) ? "rows_gemm_mfma_t" : "rows_gemm_mfma");
}

// ---- element-wise companions of the untuned analysis WITH a graph (the fused prologue / epilogue above have no saved operands) ----
//   op 0: y = log(x)                 (mcep.py:203)            backward: gx = gy / x
//   op 1: y = exp(a - 2 b)           (mcep.py:210-212)        backward: ga = gy y, gb = -2 gy y
template <int OP, bool BWD>
__global__ __launch_bounds__(256) void rows_ew_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ gy, long nel, float* __restrict__ o0,
                                                      float* __restrict__ o1)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nel) return;
    if (OP == 0) {
        if (!BWD) o0[i] = logf(a[i]);
        else o0[i] = gy[i] / a[i];
    } else {
        if (!BWD) {
            o0[i] = expf(a[i] - 2.f * b[i]);
        } else {   // a = the saved output y
            const float t = gy[i] * a[i];
            o0[i] = t;
            o1[i] = -2.f * t;
        }
    }
}

int rows_ew(int op, int bwd, const void* a, const void* b, const void* gy, int64_t nel, void* o0, void* o1, hipStream_t st)
{
    const dim3 grid((unsigned)((nel + 255) / 256));
#define DSA_EW(OPV, BV)                                                                                                      \
    hipLaunchKernelGGL((rows_ew_kernel<OPV, BV>), grid, dim3(256), 0, st, (const float*)a, (const float*)b, (const float*)gy, \
                       (long)nel, (float*)o0, (float*)o1)
    if (op == 0 && !bwd) DSA_EW(0, false);
    else if (op == 0) DSA_EW(0, true);
    else if (op == 1 && !bwd) DSA_EW(1, false);
    else if (op == 1) DSA_EW(1, true);
    else return fail(DSA_ERR_UNSUPPORTED, "rows_ew: unknown operation%s");
#undef DSA_EW
    return check_launch(op == 0 ? "rows_log" : "rows_expsub");
}

}  // namespace dsa

// C-ABI: include/diffsptk_amd.h
DSA_EXPORT int dsa_rows_gemm(const void* c, int64_t F, int32_t K, const void* A, int32_t lda, int32_t N, int32_t flags, const void* aux,
                             int32_t ldaux, int32_t dtype, void* out, int32_t ldo, void* stream)
{
    DSA_REQUIRE(F >= 0 && K > 0 && N > 0 && lda > 0 && ldo >= N, "rows_gemm: invalid sizes");
    DSA_REQUIRE(c && A && out, "rows_gemm: null pointer");
    DSA_REQUIRE(!(flags & dsa::RG_EPI_EXPSUB) || (aux && ldaux >= N), "rows_gemm: the exp-sub epilogue needs aux");
    if (dtype != DSA_F32) return dsa::fail(DSA_ERR_UNSUPPORTED, "rows_gemm: float32 only%s");
    if (F == 0) return DSA_OK;
    return dsa::rows_gemm_mfma(c, F, K, A, lda, N, flags, aux, ldaux, out, ldo, (hipStream_t)stream);
}

DSA_EXPORT int dsa_rows_ew(int32_t op, int32_t backward, const void* a, const void* b, const void* gy, int64_t n, int32_t dtype,
                           void* o0, void* o1, void* stream)
{
    DSA_REQUIRE(n >= 0 && a && o0, "rows_ew: invalid arguments");
    if (dtype != DSA_F32) return dsa::fail(DSA_ERR_UNSUPPORTED, "rows_ew: float32 only%s");
    if (n == 0) return DSA_OK;
    return dsa::rows_ew(op, backward, a, b, gy, n, o0, o1, (hipStream_t)stream);
}
