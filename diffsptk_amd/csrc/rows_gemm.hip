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
#include <cstdlib>
#include <type_traits>

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

// The four values [pos, pos + 4) of a row holding `len` valid floats (pos a multiple of 4), `fill` beyond the end, in two halves:
// row_raw4 issues the load -- always in bounds, no branch -- and row_fix4 repairs the group that straddles the end, so that the
// repair (and with it the wait for the load) can sit a whole chunk later.  A straddling group reads the row's LAST four values and
// shifts them (len & 3 is uniform).  SMALL (a row shorter than four values): clamped single loads instead.
// (As `if (inside) 16-byte load else four guarded loads` every group of the main loop was two exec-masked branches with their own waits.)
template <bool SMALL>
__device__ __forceinline__ rg_f4 row_raw4(const float* __restrict__ row, int pos, int len)
{
    if constexpr (!SMALL) {
        return *reinterpret_cast<const rg_f4u*>(row + (pos + 3 < len ? pos : len - 4));
    } else {
        rg_f4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = row[pos + e < len ? pos + e : len - 1];
        return r;
    }
}
template <bool SMALL>
__device__ __forceinline__ rg_f4 row_fix4(rg_f4 w, int pos, int len, bool rowok, float fill)
{
    rg_f4 r;
    if constexpr (!SMALL) {
        const bool full = rowok && pos + 3 < len, part = rowok && pos < len && !full;
        const int cnt = len & 3;
        rg_f4 sh;
        sh[0] = cnt == 1 ? w[3] : (cnt == 2 ? w[2] : w[1]);
        sh[1] = cnt == 1 ? fill : (cnt == 2 ? w[3] : w[2]);
        sh[2] = cnt == 3 ? w[3] : fill;
        sh[3] = fill;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = full ? w[e] : (part ? sh[e] : fill);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (rowok && pos + e < len) ? w[e] : fill;
    }
    return r;
}

template <int FLAGS, int NT, bool SMALL>
__global__ __launch_bounds__(256) void rows_gemm_mfma_kernel(const float* __restrict__ c, long F, int K, const float* __restrict__ A,
                                                             int lda, int N, const float* __restrict__ aux, int ldaux,
                                                             float* __restrict__ out, int ldo)
{
    constexpr bool PRO_LOG = (FLAGS & RG_PRO_LOG) != 0, EPI_EXPSUB = (FLAGS & RG_EPI_EXPSUB) != 0, TRANS = (FLAGS & RG_TRANS) != 0;
    __shared__ __attribute__((aligned(16))) float bs[2][kRgKC * kRgLD];   // 2 x 4160 floats; the epilogue's 4 x 16 x 130 tile fits exactly
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

    // chunk `kc` of B: fetch() brings this thread's 16 values into registers, put() writes a quarter of them to bs[kc & 1] as
    // [kk][col] = B[K0 + kk][col0 + col], zero outside the matrix.  The loads of chunk kc + 2 are issued during chunk kc and written
    // to LDS during chunk kc + 1 (as one step, the wait for the loads sat in front of the matrix instructions: with one wave per
    // SIMD nothing hid it).  Chunk indices past the end are clamped: a redundant load / a store nobody reads, but no branch.
    const int nchunk = (K + kRgKC - 1) / kRgKC;
    auto clampc = [&](int kc) { return kc < nchunk ? kc : nchunk - 1; };
    // An INTERIOR chunk needs no repair at all (uniform test, one scalar branch around the selects): every k of the chunk exists, and a
    // group of four columns that straddles the matrix's edge may simply read on -- into the row below, which exists (hence the extra
    // row asked for in the plain product) -- because what it brings are columns >= N of B, i.e. columns of the result nobody stores.
    auto plain_b = [&](int kc) { return !SMALL && clampc(kc) * kRgKC + kRgKC + (TRANS ? 0 : 1) <= K; };
    auto plain_c = [&](int kc) { return !SMALL && clampc(kc) * kRgKC + kRgKC <= K; };
    rg_f4 sv[4];
    auto fetch = [&](int kc) __attribute__((always_inline)) {
        const int K0 = clampc(kc) * kRgKC;
        const bool pl = plain_b(kc);
        if (!TRANS) {
            // 32 x 128 floats, 256 threads: thread -> (kk = tid / 8, 16 consecutive columns as four 16-byte loads)
            const int kk = tid >> 3, cq = (tid & 7) * 16;
            const float* src = A + (long)(K0 + kk < K ? K0 + kk : K - 1) * lda + col0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cb = cq + 4 * q;
                if (pl) sv[q] = *reinterpret_cast<const rg_f4u*>(src + (cb < ncols ? cb : 0));
                else sv[q] = row_raw4<SMALL>(src, cb, ncols);
            }
        } else {
            // B[k][col] = A[col0 + col][K0 + k]: thread -> (col = tid / 2, 16 consecutive k): reads along K
            const int col = tid >> 1, kq = (tid & 1) * 16;
            const float* src = A + (long)(col0 + (col < ncols ? col : ncols - 1)) * lda;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (pl) sv[q] = *reinterpret_cast<const rg_f4u*>(src + K0 + kq + 4 * q);
                else sv[q] = row_raw4<SMALL>(src, K0 + kq + 4 * q, K);
            }
        }
    };
    auto put = [&](int kc, int q) __attribute__((always_inline)) {
        const int K0 = clampc(kc) * kRgKC, buf = kc & 1;
        const bool pl = plain_b(kc);
        if (!TRANS) {
            const int kk = tid >> 3, cb = (tid & 7) * 16 + 4 * q;
            rg_f4 v = sv[q];
            if (!pl) v = row_fix4<SMALL>(sv[q], cb, ncols, K0 + kk < K, 0.f);
            *reinterpret_cast<rg_f2*>(&bs[buf][kk * kRgLD + cb]) = rg_f2{v[0], v[1]};   // rows are 8-byte aligned (stride 130)
            *reinterpret_cast<rg_f2*>(&bs[buf][kk * kRgLD + cb + 2]) = rg_f2{v[2], v[3]};
        } else {
            const int col = tid >> 1, kq = (tid & 1) * 16;   // transposed stores
            rg_f4 v = sv[q];
            if (!pl) v = row_fix4<SMALL>(sv[q], K0 + kq + 4 * q, K, col < ncols, 0.f);
#pragma unroll
            for (int e = 0; e < 4; ++e) bs[buf][(kq + 4 * q + e) * kRgLD + col] = v[e];
        }
    };
    // this lane's eight values of c for chunk kc: k = K0 + 8 g + t, t = 0..7 (beyond K: log(1) = 0 under the log prologue)
    rg_f4 cr[2];
    auto load_c = [&](int kc) __attribute__((always_inline)) {
        const int kb = clampc(kc) * kRgKC + 8 * g;
        if (plain_c(kc)) {
            cr[0] = *reinterpret_cast<const rg_f4u*>(crow + kb);
            cr[1] = *reinterpret_cast<const rg_f4u*>(crow + kb + 4);
        } else {
            cr[0] = row_raw4<SMALL>(crow, kb, K);
            cr[1] = row_raw4<SMALL>(crow, kb + 4, K);
        }
    };
    auto take_c = [&](int kc, float (&cv)[8]) __attribute__((always_inline)) {
        const int kb = kc * kRgKC + 8 * g;
        rg_f4 a = cr[0], b = cr[1];
        if (!plain_c(kc)) {
            a = row_fix4<SMALL>(cr[0], kb, K, true, PRO_LOG ? 1.f : 0.f);
            b = row_fix4<SMALL>(cr[1], kb + 4, K, true, PRO_LOG ? 1.f : 0.f);
        }
        cv[0] = a[0], cv[1] = a[1], cv[2] = a[2], cv[3] = a[3], cv[4] = b[0], cv[5] = b[1], cv[6] = b[2], cv[7] = b[3];
    };

    float cv[8];
    fetch(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) put(0, q);
    fetch(1);
    load_c(0);
    __syncthreads();
    for (int kc = 0; kc < nchunk; ++kc) {
        take_c(kc, cv);
        load_c(kc + 1);
        if (PRO_LOG) {
#pragma unroll
            for (int t = 0; t < 8; ++t) cv[t] = logf(cv[t]);   // mcep.py:203
        }
        // Step t multiplies while the operands of step t + 1 are read and a slice of the staging work runs: steps 0-3 write a
        // quarter of chunk kc + 1 to LDS, step 4 issues the loads of chunk kc + 2 -- no wait for memory or LDS on the chain.  (The
        // interleaving itself -- one matrix instruction, up to six others -- is neutral: float32 matrix instructions and vector
        // instructions share the multipliers, see mcep_resid_mfma_kernel.)
        const float* bl = &bs[kc & 1][(8 * g) * kRgLD + n];
        float bv[2][NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) bv[0][ct] = bl[16 * ct];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t + 1 < 8) {
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) bv[(t + 1) & 1][ct] = bl[(t + 1) * kRgLD + 16 * ct];
            }
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[t], bv[t & 1][ct], acc[ct], 0, 0, 0);
            if (t < 4) put(kc + 1, t);
            else if (t == 4) fetch(kc + 2);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one matrix instruction
                __builtin_amdgcn_sched_group_barrier(0x6b6, 6, 0);   // up to six of: vector / scalar ALU, memory, LDS, transcendental
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    // Results leave through LDS so that a store instruction covers 64 consecutive columns of ONE row (256 contiguous bytes; straight
    // from the accumulators a lane owns four rows of one column: 64-byte pieces of four rows per instruction, 62 us for the
    // 12 800 x 1025 exp-sub product whose bytes take 13).  D register r of lane (j = n, g) is D[4 g + r][j]: frame 4 g + r of the
    // wave's 16, column j of the tile.  The wave's tile takes the staging buffers' place (all reads of them are behind the barrier).
    float* tile = &bs[0][0] + wave * (16 * kRgLD);
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(4 * g + r) * kRgLD + 16 * ct + n] = acc[ct][r];
    }
    __syncthreads();
    constexpr int NCH = (NT * 16 + 63) / 64;   // 64-column pieces of a row
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 4) {       // four rows at a time: their aux loads fly together
        float av[4][NCH], tv[4][NCH];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long f = f0 + r0 + i, fc = f < F ? f : F - 1;
#pragma unroll
            for (int h = 0; h < NCH; ++h) {
                const int cc = lane + 64 * h, ccc = cc < ncols ? cc : ncols - 1;
                if (EPI_EXPSUB) av[i][h] = aux[fc * (long)ldaux + col0 + ccc];
                tv[i][h] = tile[(r0 + i) * kRgLD + ccc];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long f = f0 + r0 + i;
#pragma unroll
            for (int h = 0; h < NCH; ++h) {
                const int cc = lane + 64 * h;
                float v = tv[i][h];
                if (EPI_EXPSUB) v = expf(av[i][h] - 2.f * v);   // mcep.py:212
                if (f < F && cc < ncols) out[f * (long)ldo + col0 + cc] = v;
            }
        }
    }
}

template <int FLAGS, int NT, bool SMALL>
static void rows_gemm_launch_nt(const void* c, int64_t F, int K, const void* A, int lda, int N, const void* aux, int ldaux, void* out,
                                int ldo, hipStream_t st)
{
    const dim3 grid((unsigned)((F + 63) / 64), (unsigned)((N + NT * 16 - 1) / (NT * 16)));
    hipLaunchKernelGGL((rows_gemm_mfma_kernel<FLAGS, NT, SMALL>), grid, dim3(256), 0, st, (const float*)c, (long)F, K, (const float*)A, lda, N,
                       (const float*)aux, ldaux, (float*)out, ldo);
}
template <int FLAGS>
static void rows_gemm_launch_t(const void* c, int64_t F, int K, const void* A, int lda, int N, const void* aux, int ldaux, void* out,
                               int ldo, hipStream_t st)
{
    // rows of fewer than four values (of c, or of A): the single-load variant, one instantiation
    if (K < 4 || N < 4) return rows_gemm_launch_nt<FLAGS, 2, true>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st);
    // column tiles per workgroup: the whole width when it fits eight tiles (every value of c is then read once), else eight.
    // DSA_ROWS_GEMM_NT = 2 / 4 / 6 / 8 caps it (more, narrower workgroups for small batches: measurement knob).
    int nt = (N + 15) / 16;
    static const int cap = [] {
        const char* e = getenv("DSA_ROWS_GEMM_NT");
        return e ? atoi(e) : 0;
    }();
    if (cap >= 2 && nt > cap) nt = cap;
    if (nt <= 2) rows_gemm_launch_nt<FLAGS, 2, false>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st);
    else if (nt <= 4) rows_gemm_launch_nt<FLAGS, 4, false>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st);
    else if (nt <= 6) rows_gemm_launch_nt<FLAGS, 6, false>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st);
    else if (nt == 7) rows_gemm_launch_nt<FLAGS, 7, false>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st);
    else rows_gemm_launch_nt<FLAGS, 8, false>(c, F, K, A, lda, N, aux, ldaux, out, ldo, st);
}

// ---- the spectral half of a Newton step of MelCepstralAnalysis in ONE launch (mcep.py:210-215) ----
//   rt:(F, N) = exp(logx - 2 mc D) E,     logx:(F, K), mc:(F, M1), D:(M1 x K), E:(K x N), N = 2 M1 - 1 <= 16 NT
// As two launches (EPI_EXPSUB product, then the plain product) the (F, K) matrix e made a round trip through memory and the first
// product's workgroups were one short dependent chain of memory round trips each: 40 + 50 us per step at 12 800 x 1025.  Here e is
// produced chunk by chunk in the layout the second product consumes: per chunk of 32 bins a wave forms S = mc D[:, chunk]
// (16 x 32, 2 MT matrix instructions, mc held in MT registers per lane for the whole kernel), turns it from the result layout
// (lane (j, g): rows 4 g .. 4 g + 3 of column j) into the operand layout (lane (row, g): bins 8 g .. 8 g + 7) through a
// wave-private LDS tile, applies exp(logx - 2 S) (logx read in that layout, a chunk ahead), and multiplies by the chunk of E as
// rows_gemm_mfma_kernel does.  (Tried: S^T = D^T mc^T instead, whose result layout -- lane (frame, g): bins 16 tile + 4 g + r -- is an
// A operand of the second product as it stands, no LDS tile: 73.2 -> 76.2 us per workgroup; the exp then waits on the end of the
// S chain instead of on an LDS read that the first matrix instructions of the product cover.)  Same staging discipline (loads a chunk ahead in registers, operands of step t + 1 read during step t).

// exp(x) in five instructions, about 2 ulp for results in the normal range: x log2(e) = t + r with t the rounded product and r its
// exact remainder plus the low part of log2(e), exp2(t) on the transcendental unit (v_exp_f32, 1 ulp), times (1 + r ln 2).  No
// scaling for results below 2^-126 (flushed): the operand here is exp(log X - 2 log of the model spectrum), a ratio near 1.  The
// library routine's range handling tripled the count, and on this kernel every vector instruction is paid in full: the float32
// matrix instruction runs on the vector unit's multipliers, nothing overlaps with it (tools/gpu_abl_resid.sh).
__device__ __forceinline__ float exp_ratio(float x)
{
    const float t = x * 0x1.715476p+0f;
    float r = __builtin_fmaf(x, 0x1.715476p+0f, -t);
    r = __builtin_fmaf(x, 0x1.4ae0bep-26f, r);
    const float e = __builtin_amdgcn_exp2f(t);
    // e (1 + r ln 2) as a product: an overflowed e stays +inf (e + e c would be inf - inf = NaN for c < 0, where expf gives inf)
    // and an underflowed one stays 0 -- the same five instructions; 1 + r ln 2 rounds the correction to half an ulp (2 ulp in all)
    return e * __builtin_fmaf(r, 0x1.62e430p-1f, 1.0f);
}

#ifndef RG_ABL
#define RG_ABL 0   // measurement builds only (tools/gpu_abl_resid.sh): 1 no product, 2 no S / transposition, 4 no exp, 8 no staging, 16 no barrier
#endif
template <int MT, int NT>
__global__ __launch_bounds__(256) void mcep_resid_mfma_kernel(const float* __restrict__ logx, long F, int K, const float* __restrict__ mc,
                                                              int M1, const float* __restrict__ D, int ldd, const float* __restrict__ E,
                                                              int lde, int N, float* __restrict__ out, int ldo)
{
    constexpr int LDD = 48;   // row stride of the staged chunk of D: rows 4 j + g, g = 0..3, land on four distinct groups of 16 banks
    constexpr int LDT = 36;   // row stride of the transposition tile (16-byte aligned rows; the four row groups hit distinct banks)
    __shared__ __attribute__((aligned(16))) float bs[2][kRgKC * kRgLD];
    __shared__ __attribute__((aligned(16))) float dsm[2][MT * 4 * LDD];
    __shared__ __attribute__((aligned(16))) float ts[4][16 * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const long f0 = (long)blockIdx.x * 64 + wave * 16;
    const long fr = f0 + n < F ? f0 + n : F - 1;           // clamped row: its results are not stored
    const float* xrow = logx + fr * (long)K;
    float cmc[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int k = 4 * j + g;
        cmc[j] = k < M1 ? mc[fr * (long)M1 + k] : 0.f;
    }
    rg_f4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = rg_f4{0.f, 0.f, 0.f, 0.f};

    // Staging: thread -> E row tid / 8, 16 columns at 16 (tid & 7); D row tid / 8 (+ 32), four bins at 4 (tid & 7); logx in the operand
    // layout.  Loads sit in registers for an iteration before they are written to LDS (row_raw4 / row_fix4); chunk indices past the
    // end are clamped (a redundant load / a store nobody reads) so that the loop body has no branch.
    const int nchunk = (K + kRgKC - 1) / kRgKC;
    rg_f4 sv[4], dv[2], cr[2];
    const int skk = tid >> 3, scq = (tid & 7) * 16, dq = (tid & 7) * 4;
    auto clampc = [&](int kc) { return kc < nchunk ? kc : nchunk - 1; };
    // interior chunks need no repair (see rows_gemm_mfma_kernel): every bin of the chunk exists; a straddling column group of E reads
    // on into the row below (hence one row more for E); rows of D past the order are row M1 - 1 again, finite, and meet zeros of mc
    auto plain_e = [&](int kc) { return clampc(kc) * kRgKC + kRgKC + 1 <= K; };
    auto plain_k = [&](int kc) { return clampc(kc) * kRgKC + kRgKC <= K; };
    auto fetch_e = [&](int kc) __attribute__((always_inline)) {
        const int K0 = clampc(kc) * kRgKC;
        const float* src = E + (long)(K0 + skk < K ? K0 + skk : K - 1) * lde;
        const bool pl = plain_e(kc);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cb = scq + 4 * q;
            if (pl) sv[q] = *reinterpret_cast<const rg_f4u*>(src + (cb < N ? cb : 0));
            else sv[q] = row_raw4<false>(src, cb, N);
        }
    };
    auto put_e = [&](int kc, int q) __attribute__((always_inline)) {   // quarter q of the thread's 16 columns
        const int K0 = clampc(kc) * kRgKC, buf = kc & 1;
        const int cb = scq + 4 * q;
        rg_f4 v = sv[q];
        if (!plain_e(kc)) v = row_fix4<false>(sv[q], cb, N, K0 + skk < K, 0.f);
        *reinterpret_cast<rg_f2*>(&bs[buf][skk * kRgLD + cb]) = rg_f2{v[0], v[1]};
        *reinterpret_cast<rg_f2*>(&bs[buf][skk * kRgLD + cb + 2]) = rg_f2{v[2], v[3]};
    };
    auto fetch_d = [&](int kc) __attribute__((always_inline)) {
        const int K0 = clampc(kc) * kRgKC;
        const bool pl = plain_k(kc);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = skk + 32 * h, rc = row < M1 ? row : M1 - 1;
            if (32 * h < MT * 4) {
                if (pl) dv[h] = *reinterpret_cast<const rg_f4u*>(D + (long)rc * ldd + K0 + dq);
                else dv[h] = row_raw4<false>(D + (long)rc * ldd, K0 + dq, K);
            }
        }
    };
    auto put_d = [&](int kc, int h) __attribute__((always_inline)) {
        const int K0 = clampc(kc) * kRgKC, buf = kc & 1;
        const int row = skk + 32 * h;
        if (row < MT * 4) {
            rg_f4 v = dv[h];
            if (!plain_k(kc)) v = row_fix4<false>(dv[h], K0 + dq, K, row < M1, 0.f);
            *reinterpret_cast<rg_f4*>(&dsm[buf][row * LDD + dq]) = v;
        }
    };
    auto fetch_x = [&](int kc) __attribute__((always_inline)) {
        const int kb = clampc(kc) * kRgKC + 8 * g;
        if (plain_k(kc)) {
            cr[0] = *reinterpret_cast<const rg_f4u*>(xrow + kb);
            cr[1] = *reinterpret_cast<const rg_f4u*>(xrow + kb + 4);
        } else {
            cr[0] = row_raw4<false>(xrow, kb, K);
            cr[1] = row_raw4<false>(xrow, kb + 4, K);
        }
    };

    // Iteration i produces e for chunk i (PROD) and multiplies chunk i - 1 (CONS); everything that is not a matrix instruction is cut
    // into eight slices between the consumer's eight k-steps: the exp of one operand, a quarter of the LDS writes of E_i, half of
    // those of D_{i+1}, the loads of E_{i+1} and D_{i+2} (written to LDS an iteration after they were issued, so no wait for memory
    // sits on the chain).  What the slicing does NOT buy is overlap of vector work with the matrix instructions: the float32 matrix
    // instruction executes on the vector unit's own multipliers, and the ablations (tools/gpu_abl_resid.sh, one workgroup per CU:
    // 86 us = product 23 + S and transposition 19 + exp 9 + staging 11 + barrier 5 + loop skeleton) add up exactly.  The launch
    // costs what its instructions cost; it saves the 105 MB round trip of e and a launch, not time on a small batch.
    float* tw = ts[wave];
    float cv[8];
    auto step = [&](auto prod_c, auto cons_c, int i) __attribute__((always_inline)) {
        constexpr bool PROD = decltype(prod_c)::value, CONS = decltype(cons_c)::value;
        rg_f4 xa, xb, sa, sb;
        float cn[8];
        if constexpr (PROD) {
            const int kb = i * kRgKC + 8 * g;
            xa = cr[0];
            xb = cr[1];
            if (!plain_k(i)) {
                xa = row_fix4<false>(cr[0], kb, K, true, 0.f);
                xb = row_fix4<false>(cr[1], kb + 4, K, true, 0.f);
            }
            fetch_x(i + 1);
            // S = mc D[:, chunk i]: two column tiles, MT steps of four
            rg_f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
            const float* dl = &dsm[i & 1][g * LDD + n];
            if (!(RG_ABL & 2)) {
            float dop[MT][2];   // all operands first: read one pair at a time, every pair of matrix instructions waited out an LDS round trip
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                dop[j][0] = dl[4 * j * LDD];
                dop[j][1] = dl[4 * j * LDD + 16];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cmc[j], dop[j][0], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cmc[j], dop[j][1], s1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // result layout -> operand layout through the wave's own tile (LDS operations of a wave complete in order)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tw[(4 * g + r) * LDT + n] = s0[r];
                tw[(4 * g + r) * LDT + 16 + n] = s1[r];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            sa = *reinterpret_cast<const rg_f4*>(&tw[n * LDT + 8 * g]);
            sb = *reinterpret_cast<const rg_f4*>(&tw[n * LDT + 8 * g + 4]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            } else { sa = xa; sb = xb; }
        }
        const float* bl = &bs[(i + 1) & 1][(8 * g) * kRgLD + n];   // E_{i-1}
        float bv[2][NT];
        if constexpr (CONS) {
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) bv[0][ct] = bl[16 * ct];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if constexpr (CONS) {
                if (t + 1 < 8) {
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) bv[(t + 1) & 1][ct] = bl[(t + 1) * kRgLD + 16 * ct];
                }
                if (!(RG_ABL & 1)) {
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(cv[t], bv[t & 1][ct], acc[ct], 0, 0, 0);   // :214-215
                } else {
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) acc[ct][0] += cv[t] * bv[t & 1][ct];
                }
            }
            if constexpr (PROD) {
                if (!(RG_ABL & 4)) cn[t] = exp_ratio((t < 4 ? xa[t & 3] : xb[t & 3]) - 2.f * (t < 4 ? sa[t & 3] : sb[t & 3]));   // mcep.py:210-212
                else cn[t] = (t < 4 ? xa[t & 3] : xb[t & 3]) - 2.f * (t < 4 ? sa[t & 3] : sb[t & 3]);
                asm volatile("" : "+v"(cn[t]));   // in this step's slice: not sunk to its use at the end of the iteration
                if (!(RG_ABL & 8)) {
                    if (t < 2) put_d(i + 1, t);
                    else if (t < 6) put_e(i, t - 2);
                    else if (t == 6) fetch_e(i + 1);
                    else fetch_d(i + 2);
                }
            }
            if constexpr (CONS) {
                // one matrix instruction, then up to seven others, and so on (see rows_gemm_mfma_kernel)
#pragma unroll
                for (int ct = 0; ct < NT; ++ct) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x6b6, 7, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (PROD) {
#pragma unroll
            for (int t = 0; t < 8; ++t) cv[t] = cn[t];
        }
        if (!(RG_ABL & 16)) __syncthreads();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    fetch_d(0);
    put_d(0, 0);
    put_d(0, 1);
    fetch_d(1);
    fetch_e(0);
    fetch_x(0);
    __syncthreads();
    step(T_{}, F_{}, 0);
    for (int i = 1; i < nchunk; ++i) step(T_{}, T_{}, i);
    step(F_{}, T_{}, nchunk);
    // rows leave through LDS, 64 consecutive columns of one row per store instruction (see rows_gemm_mfma_kernel)
    float* tile = &bs[0][0] + wave * (16 * kRgLD);
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(4 * g + r) * kRgLD + 16 * ct + n] = acc[ct][r];
    }
    __syncthreads();
    constexpr int NCH = (NT * 16 + 63) / 64;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const long f = f0 + rr;
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
            const int cc = lane + 64 * h;
            if (f < F && cc < N) out[f * (long)ldo + cc] = tile[rr * kRgLD + cc];
        }
    }
}

// float32; 2 <= M1 <= 55, N <= 112, K >= 4, N >= 4
int mcep_resid_mfma(const void* logx, int64_t F, int K, const void* mc, int M1, const void* D, int ldd, const void* E, int lde, int N,
                    void* out, int ldo, hipStream_t st)
{
    const dim3 grid((unsigned)((F + 63) / 64));
#define DSA_RESID(MTV, NTV)                                                                                                       \
    hipLaunchKernelGGL((mcep_resid_mfma_kernel<MTV, NTV>), grid, dim3(256), 0, st, (const float*)logx, (long)F, K, (const float*)mc, \
                       M1, (const float*)D, ldd, (const float*)E, lde, N, (float*)out, ldo)
    const int nt = (N + 15) / 16;
    if (M1 <= 28 && nt <= 4) DSA_RESID(7, 4);
    else if (M1 <= 36 && nt <= 6) DSA_RESID(9, 6);
    else if (M1 <= 44 && nt <= 6) DSA_RESID(11, 6);
    else if (M1 <= 52 && nt <= 7) DSA_RESID(13, 7);
    else if (M1 <= 56 && nt <= 7) DSA_RESID(14, 7);
    else return fail(DSA_ERR_UNSUPPORTED, "mcep_resid: order above 55%s");
#undef DSA_RESID
    return check_launch("mcep_resid_mfma");
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
    DSA_REQUIRE(F == 0 || (c && A && out), "rows_gemm: null pointer");
    DSA_REQUIRE(F == 0 || !(flags & dsa::RG_EPI_EXPSUB) || (aux && ldaux >= N), "rows_gemm: the exp-sub epilogue needs aux");
    if (dtype != DSA_F32) return dsa::fail(DSA_ERR_UNSUPPORTED, "rows_gemm: float32 only%s");
    if (F == 0) return DSA_OK;
    return dsa::rows_gemm_mfma(c, F, K, A, lda, N, flags, aux, ldaux, out, ldo, (hipStream_t)stream);
}

DSA_EXPORT int dsa_rows_ew(int32_t op, int32_t backward, const void* a, const void* b, const void* gy, int64_t n, int32_t dtype,
                           void* o0, void* o1, void* stream)
{
    DSA_REQUIRE(n >= 0 && (n == 0 || (a && o0)), "rows_ew: invalid arguments");
    if (dtype != DSA_F32) return dsa::fail(DSA_ERR_UNSUPPORTED, "rows_ew: float32 only%s");
    if (n == 0) return DSA_OK;
    return dsa::rows_ew(op, backward, a, b, gy, n, o0, o1, (hipStream_t)stream);
}

DSA_EXPORT int dsa_mcep_newton_resid(const void* logx, int64_t F, int32_t K, const void* mc, int32_t n, const void* D, int32_t ldd,
                                     const void* E, int32_t lde, int32_t dtype, void* rt, void* stream)
{
    DSA_REQUIRE(F >= 0 && K >= 4 && n >= 3 && ldd >= K && lde >= 2 * n - 1, "mcep_newton_resid: invalid sizes");
    DSA_REQUIRE(F == 0 || (logx && mc && D && E && rt), "mcep_newton_resid: null pointer");
    if (dtype != DSA_F32 || n > 55) return dsa::fail(DSA_ERR_UNSUPPORTED, "mcep_newton_resid: float32, orders up to 54%s");
    if (F == 0) return DSA_OK;
    return dsa::mcep_resid_mfma(logx, F, K, mc, n, D, ldd, E, lde, 2 * n - 1, rt, 2 * n - 1, (hipStream_t)stream);
}

DSA_EXPORT int64_t dsa_mcep_resid_images_bytes(int32_t K, int32_t n)
{
    if (K < 4 || n < 3 || n > 55) return 0;
    return dsa::mcep_resid_h_images_bytes(K, n);
}

DSA_EXPORT int dsa_mcep_resid_prepare(const void* D, int32_t ldd, const void* E, int32_t lde, int32_t K, int32_t n, int32_t dtype, void* images,
                                      void* stream)
{
    DSA_REQUIRE(K >= 4 && n >= 3 && ldd >= K && lde >= 2 * n - 1 && D && E && images, "mcep_resid_prepare: invalid arguments");
    if (dtype != DSA_F32 || n > 55) return dsa::fail(DSA_ERR_UNSUPPORTED, "mcep_resid_prepare: float32, orders up to 54%s");
    return dsa::mcep_resid_h_prepare(D, ldd, E, lde, K, n, images, (hipStream_t)stream);
}

// (0.2.0) mcep.py:208-222: ALL n_iter Newton steps in one persistent launch (csrc/mcep_big_f16.h); DSA_ERR_UNSUPPORTED where no
// instantiation covers the order -- the caller then alternates dsa_mcep_newton_resid_h and dsa_mcep_newton_update
DSA_EXPORT int dsa_mcep_newton_steps(const void* logx, int64_t F, int32_t K, const void* mc_in, int32_t n, const void* images,
                                     const void* alpha_vec, int32_t n_iter, int32_t dtype, void* mc_out, void* stream)
{
    DSA_REQUIRE(F >= 0 && K >= 4 && n >= 3 && n_iter >= 0, "mcep_newton_steps: invalid sizes");
    DSA_REQUIRE(F == 0 || (logx && mc_in && images && alpha_vec && mc_out), "mcep_newton_steps: null pointer");
    if (dtype != DSA_F32) return dsa::fail(DSA_ERR_UNSUPPORTED, "mcep_newton_steps: float32 only%s");
    if (F == 0) return DSA_OK;
    const int rc = dsa::mcep_big_newton(logx, F, K, mc_in, n, images, alpha_vec, n_iter, mc_out, (hipStream_t)stream);
    if (rc == DSA_ERR_UNSUPPORTED) return dsa::fail(DSA_ERR_UNSUPPORTED, "mcep_newton_steps: no one-launch kernel for this order (32 .. 54)%s");
    return rc;
}

// (0.2.1) the adjoint of dsa_mcep_newton_resid_h (csrc/mcep_resid_bwd_f16.h): images of their own
DSA_EXPORT int64_t dsa_mcep_resid_bwd_images_bytes(int32_t K, int32_t n)
{
    if (K < 4 || n < 33 || n > 55) return 0;   // what dsa_mcep_newton_resid_h_bwd covers (0: keep the composed gradient)
    return dsa::mcep_resid_bwd_images_bytes(K, n);
}

DSA_EXPORT int dsa_mcep_resid_bwd_prepare(const void* D, int32_t ldd, const void* E, int32_t lde, int32_t K, int32_t n, int32_t dtype, void* images,
                                          void* stream)
{
    DSA_REQUIRE(K >= 4 && n >= 3 && ldd >= K && lde >= 2 * n - 1 && D && E && images, "mcep_resid_bwd_prepare: invalid arguments");
    if (dtype != DSA_F32 || n < 33 || n > 55) return dsa::fail(DSA_ERR_UNSUPPORTED, "mcep_resid_bwd_prepare: float32, orders 32 .. 54%s");
    return dsa::mcep_resid_bwd_prepare(D, ldd, E, lde, K, n, images, (hipStream_t)stream);
}

DSA_EXPORT int dsa_mcep_newton_resid_h_bwd(const void* logx, int64_t F, int32_t K, const void* mc, int32_t n, const void* grt, const void* images,
                                           int32_t dtype, void* glogx, void* gmc, void* stream)
{
    DSA_REQUIRE(F >= 0 && K >= 4 && n >= 3, "mcep_newton_resid_h_bwd: invalid sizes");
    DSA_REQUIRE(F == 0 || (logx && mc && grt && images && gmc), "mcep_newton_resid_h_bwd: null pointer");   // (glogx may be NULL since 0.2.2: not accumulated)
    if (dtype != DSA_F32) return dsa::fail(DSA_ERR_UNSUPPORTED, "mcep_newton_resid_h_bwd: float32 only%s");
    if (F == 0) return DSA_OK;
    const int rc = dsa::mcep_resid_bwd_h(logx, F, K, mc, n, grt, images, glogx, gmc, (hipStream_t)stream);
    if (rc == DSA_ERR_UNSUPPORTED) return dsa::fail(DSA_ERR_UNSUPPORTED, "mcep_newton_resid_h_bwd: orders 32 .. 54%s");
    return rc;
}

// (0.2.2) glogx of the whole analysis in one pass over the bins (csrc/mcep_glogx_f16.h), after a sweep run with glogx = NULL
DSA_EXPORT int dsa_mcep_newton_glogx_h(const void* logx, int64_t F, int32_t K, const void* mcs, int32_t n, const void* grts, int32_t n_iter,
                                       const void* images, int32_t dtype, void* glogx, void* stream)
{
    DSA_REQUIRE(F >= 0 && K >= 4 && n >= 3 && n_iter >= 1, "mcep_newton_glogx_h: invalid sizes");
    DSA_REQUIRE(F == 0 || (logx && mcs && grts && images && glogx), "mcep_newton_glogx_h: null pointer");
    if (dtype != DSA_F32) return dsa::fail(DSA_ERR_UNSUPPORTED, "mcep_newton_glogx_h: float32 only%s");
    if (F == 0) return DSA_OK;
    return dsa::mcep_glogx_h(logx, F, K, mcs, n, grts, n_iter, images, glogx, (hipStream_t)stream);   // DSA_ERR_UNSUPPORTED: no error text
}

DSA_EXPORT int dsa_mcep_newton_resid_h(const void* logx, int64_t F, int32_t K, const void* mc, int32_t n, const void* images, int32_t dtype,
                                       void* rt, void* stream)
{
    DSA_REQUIRE(F >= 0 && K >= 4 && n >= 3, "mcep_newton_resid_h: invalid sizes");
    DSA_REQUIRE(F == 0 || (logx && mc && images && rt), "mcep_newton_resid_h: null pointer");
    if (dtype != DSA_F32 || n > 55) return dsa::fail(DSA_ERR_UNSUPPORTED, "mcep_newton_resid_h: float32, orders up to 54%s");
    if (F == 0) return DSA_OK;
    return dsa::mcep_resid_h_fwd(logx, F, K, mc, n, images, rt, 2 * n - 1, (hipStream_t)stream);
}
