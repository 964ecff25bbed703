// Mel filter-bank analysis: MelFilterBankAnalysis._forward,
// diffsptk/modules/fbank.py:306-321 -- the consumer of the STFT power spectrum:
//     y = glog(max(s @ H, floor)),  s = x (use_power) or sqrt(x);   E = log((2 sum_inner x + x_0 + x_last) / (2 (K-1)))
// Generic kernel pair (float32 / float64, any K, C): one wave64 owns 4 consecutive frames per pass, keeps
// their spectra interleaved in LDS ([K][4]: one 16-byte broadcast read feeds 4 FMAs) and lane c owns
// channel c.  H (K x C, triangular filters, 41 KB at K = 257, C = 40) is copied to LDS once per
// workgroup; while copying, every channel's range of non-zero bins (and, for the backward, every bin's
// range of non-zero channels) is recorded, so the loops touch only the ~2 K / C bins a triangular
// filter covers -- discovered from the data, not assumed: a dense matrix simply gets full ranges.
// HBM: 4 K + 4 C bytes per frame (1188 B at K = 257, C = 40).  MFCC (mfcc.py:244-256) = this kernel + the DCT-II/lifter matrix product on the
// frequency-transform kernel (dsa_freqt_fwd).
#include "common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

namespace dsa {

static bool env_flag(const char* name)
{
    const char* v = std::getenv(name);
    return v && v[0] && v[0] != '0';
}


constexpr int kFbFrames = 4;   // frames per wave and pass
constexpr int kFbMaxWaves = 16;  // waves per workgroup (as many as the LDS tiles next to the H copy allow)

template <typename T>
__device__ __forceinline__ T glog_fwd(T y, T gamma)
{
    return gamma == T(0) ? dsa_log(y) : (dsa_pow(y, gamma) - T(1)) / gamma;   // fbank.py:318
}
template <typename T>
__device__ __forceinline__ T glog_bwd(T y, T gamma)
{
    return gamma == T(0) ? T(1) / y : dsa_pow(y, gamma - T(1));
}

// stage 4 frames' spectra into tile[k][fi] (s = x or sqrt x) and return this lane's share of the 4 energy sums
template <typename T>
__device__ __forceinline__ void fbank_stage(const T* __restrict__ x, long f0, long F, int K, int use_power, T* tile,
                                            T (&esum)[kFbFrames], int lane)
{
#pragma unroll
    for (int fi = 0; fi < kFbFrames; ++fi) {
        esum[fi] = T(0);
        const long f = f0 + fi;
        for (int k = lane; k < K; k += 64) {
            const T v = f < F ? x[f * K + k] : T(1);
            tile[k * kFbFrames + fi] = use_power ? v : dsa_sqrt(v);          // fbank.py:315
            esum[fi] += (k == 0 || k == K - 1) ? v : T(2) * v;                // fbank.py:319
        }
    }
#pragma unroll
    for (int fi = 0; fi < kFbFrames; ++fi) esum[fi] = wave_sum(esum[fi]);
}

// H -> LDS and the non-zero ranges: lo_c / hi_c over bins for channel c (forward), lo_k / hi_k over channels
// for bin k (backward).  HLDS = false (matrix too large for LDS): H stays in global memory, full ranges.
template <typename T, bool HLDS>
__device__ __forceinline__ const T* fbank_prepare(const T* __restrict__ H, int K, int C, T* Hs, int* rng_c, int* rng_k)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    if (HLDS)
        for (int i = tid; i < K * C; i += nt) Hs[i] = H[i];
    __syncthreads();
    const T* Hm = HLDS ? Hs : H;
    for (int c = tid; c < C; c += nt) {
        int lo = K, hi = 0;
        if (HLDS) {
            for (int k = 0; k < K; ++k)
                if (Hm[(long)k * C + c] != T(0)) { lo = k < lo ? k : lo; hi = k + 1; }
        } else { lo = 0; hi = K; }
        rng_c[2 * c] = lo < hi ? lo : 0;
        rng_c[2 * c + 1] = hi;
    }
    if (rng_k)
        for (int k = tid; k < K; k += nt) {
            int lo = C, hi = 0;
            if (HLDS) {
                for (int c = 0; c < C; ++c)
                    if (Hm[(long)k * C + c] != T(0)) { lo = c < lo ? c : lo; hi = c + 1; }
            } else { lo = 0; hi = C; }
            rng_k[2 * k] = lo < hi ? lo : 0;
            rng_k[2 * k + 1] = hi;
        }
    __syncthreads();
    return Hm;
}

template <typename T, bool HLDS>
__global__ __launch_bounds__(kFbMaxWaves * 64) void fbank_fwd_kernel(const T* __restrict__ x, long F, int K,
                                                                  const T* __restrict__ H, int C, T floor, T gamma,
                                                                  int use_power, T* __restrict__ y, T* __restrict__ E)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fb_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T* tiles = reinterpret_cast<T*>(fb_smem);
    const int nwaves = blockDim.x >> 6;
    T* Hs = tiles + (size_t)nwaves * K * kFbFrames;
    int* rng_c = reinterpret_cast<int*>(Hs + (HLDS ? (size_t)K * C : 0));
    const T* Hm = fbank_prepare<T, HLDS>(H, K, C, Hs, rng_c, nullptr);
    T* tile = tiles + (size_t)wave * K * kFbFrames;
    const long groups = (F + kFbFrames - 1) / kFbFrames;
    for (long grp = (long)blockIdx.x * nwaves + wave; grp < groups; grp += (long)gridDim.x * nwaves) {
        const long f0 = grp * kFbFrames;
        T esum[kFbFrames];
        __builtin_amdgcn_wave_barrier();
        fbank_stage(x, f0, F, K, use_power, tile, esum, lane);
        __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < C; c += 64) {
            T acc[kFbFrames] = {T(0), T(0), T(0), T(0)};
            const int k1 = rng_c[2 * c + 1];
            for (int k = rng_c[2 * c]; k < k1; ++k) {
                const T h = Hm[(long)k * C + c];
#pragma unroll
                for (int fi = 0; fi < kFbFrames; ++fi) acc[fi] += tile[k * kFbFrames + fi] * h;   // fbank.py:316
            }
#pragma unroll
            for (int fi = 0; fi < kFbFrames; ++fi)
                if (f0 + fi < F) {
                    const T v = acc[fi] < floor ? floor : acc[fi];                                   // fbank.py:317 (torch.clip: NaN stays NaN)
                    y[(f0 + fi) * C + c] = glog_fwd(v, gamma);
                }
        }
        if (E && lane < kFbFrames && f0 + lane < F) {
            T s = esum[0];
#pragma unroll
            for (int fi = 1; fi < kFbFrames; ++fi) s = lane == fi ? esum[fi] : s;
            E[f0 + lane] = dsa_log(s / T(2 * (K - 1)));                                               // fbank.py:320
        }
    }
}

// gx = (gy * glog'(max(s H, floor)) * [s H >= floor]) H^T * ds/dx  +  gE * w / S
template <typename T, bool HLDS>
__global__ __launch_bounds__(kFbMaxWaves * 64) void fbank_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ gE,
                                                                  const T* __restrict__ x, long F, int K,
                                                                  const T* __restrict__ H, int C, T floor, T gamma,
                                                                  int use_power, T* __restrict__ gx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fb_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T* tiles = reinterpret_cast<T*>(fb_smem);
    const int nwaves = blockDim.x >> 6;
    T* Hs = tiles + (size_t)nwaves * (K + C) * kFbFrames;
    int* rng_c = reinterpret_cast<int*>(Hs + (HLDS ? (size_t)K * C : 0));
    int* rng_k = rng_c + 2 * C;
    const T* Hm = fbank_prepare<T, HLDS>(H, K, C, Hs, rng_c, rng_k);
    T* tile = tiles + (size_t)wave * (K + C) * kFbFrames;
    T* tch = tile + (size_t)K * kFbFrames;   // [C][4]: channel cotangents
    const long groups = (F + kFbFrames - 1) / kFbFrames;
    for (long grp = (long)blockIdx.x * nwaves + wave; grp < groups; grp += (long)gridDim.x * nwaves) {
        const long f0 = grp * kFbFrames;
        T esum[kFbFrames];
        __builtin_amdgcn_wave_barrier();
        fbank_stage(x, f0, F, K, use_power, tile, esum, lane);
        __builtin_amdgcn_wave_barrier();
        for (int c = lane; c < C; c += 64) {
            T acc[kFbFrames] = {T(0), T(0), T(0), T(0)};
            const int k1 = rng_c[2 * c + 1];
            for (int k = rng_c[2 * c]; k < k1; ++k) {
                const T h = Hm[(long)k * C + c];
#pragma unroll
                for (int fi = 0; fi < kFbFrames; ++fi) acc[fi] += tile[k * kFbFrames + fi] * h;
            }
#pragma unroll
            for (int fi = 0; fi < kFbFrames; ++fi) {
                T t = T(0);
                if (f0 + fi < F && acc[fi] >= floor) t = gy[(f0 + fi) * C + c] * glog_bwd(acc[fi], gamma);   // clamp passes at equality
                tch[c * kFbFrames + fi] = t;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < K; k += 64) {
            T acc[kFbFrames] = {T(0), T(0), T(0), T(0)};
            const T* hrow = Hm + (long)k * C;
            const int c1 = rng_k[2 * k + 1];
            for (int c = rng_k[2 * k]; c < c1; ++c) {
                const T h = hrow[c];
#pragma unroll
                for (int fi = 0; fi < kFbFrames; ++fi) acc[fi] += tch[c * kFbFrames + fi] * h;
            }
            const T wk = (k == 0 || k == K - 1) ? T(1) : T(2);
#pragma unroll
            for (int fi = 0; fi < kFbFrames; ++fi)
                if (f0 + fi < F) {
                    T g = acc[fi];
                    if (!use_power) g *= T(0.5) / tile[k * kFbFrames + fi];    // d sqrt(x) / dx
                    if (gE) g += gE[f0 + fi] * wk / esum[fi];
                    gx[(f0 + fi) * K + k] = g;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Matrix-core forward for float32 spectra and C <= 48 channels (the MFCC / log-mel front end of
// SURVEY 8(f) row 1): s H is a 16 x K by K x 48 product per tile of 16 frames on
// v_mfma_f32_16x16x4_f32 (float32 products, float32 accumulation: no precision is given up).
//   * the 16 x K tile is one contiguous stretch of the spectrum: copied to LDS with 16-byte loads,
//     one tile AHEAD of the products (the loads of tile n+1 are in flight while tile n multiplies);
//   * product step s pairs the 4 k-slots of the instruction with the bins
//     64 (s / 16) + s % 16 + 16 kq: with the odd row stride K = 257 the 64 operand reads of a step
//     fall on 32 distinct banks twice (the minimum);
//   * H is expanded ONCE per workgroup into a per-channel-tile PROGRAM in LDS: the list of steps whose
//     4 x 16 block of H is not all zero (the triangular filters leave more than half of the blocks
//     empty; discovered from the data -- a dense H simply lists every step) and, per listed step, the
//     B operand in lane order (zero beyond K and C).  The inner loop is branch-free: 8 entries per
//     batch, the next batch's bin offsets fetched while this batch multiplies;
//   * with use_power and a free column (C < 48) the log-energy weights w_k / (2 (K - 1)) ride as one
//     more column of H, so E falls out of the same accumulators; otherwise one pass of operand reads
//     sums the energy on the vector unit;
//   * epilogue: floor, glog, 64-byte runs of 16 channels per frame straight from the accumulators.
// The same kernel serves other "transform of the spectrum, then a K x C matrix" front ends: use_power selects the
// input transform (0: sqrt x, 1: x, 2: log x), H may be a column slice of a wider matrix (row stride ldh), and
// post_mode 1 replaces floor + glog by a plain scaling with the first coefficient halved (cepstral analysis,
// fftcep.py:122-135 with n_iter = 0: log x against the first M + 1 columns of the even cosine matrix).
constexpr int kFmRows = 16, kFmWaves = 4, kFmPre = 20, kFmU = 8;  // K <= 320
#ifdef DSA_FBANK_TIMING
__device__ unsigned long long g_fbank_stamps[16];
#define FB_STAMP(n) do { if (blockIdx.x == 0 && threadIdx.x == 0 && tl == stride) g_fbank_stamps[n] = __builtin_readcyclecounter(); } while (0)
#define FB_STAMP0(n) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_fbank_stamps[n] = __builtin_readcyclecounter(); } while (0)
#else
#define FB_STAMP(n)
#define FB_STAMP0(n)
#endif
typedef float fm_f4 __attribute__((ext_vector_type(4)));

template <int NQ>   // 16-byte loads per lane and tile: ceil(4 K / 64) rounded up to one of 5, 9, 17, 20
__global__ __launch_bounds__(kFmWaves * 64) void fbank_mfma_fwd_kernel(const float* __restrict__ x, long F, int K,
                                                                      const float* __restrict__ H, int C, float floor,
                                                                      float gamma, int use_power, float* __restrict__ y,
                                                                      float* __restrict__ E, int nsteps, int cap,
                                                                      int tile_floats, int vec4, int ldh, int post_mode,
                                                                      float post_scale, const float* __restrict__ W2, int Mo,
                                                                      float* __restrict__ z, int ldy)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fb_smem[];
    float* himg = reinterpret_cast<float*>(fb_smem);                         // [3][cap][64]
    int* prog = reinterpret_cast<int*>(himg + (size_t)3 * cap * 64);         // [3][cap]: bin offset of k-slot 0, or -1
    int* cnt = prog + 3 * cap;                                               // [8]: entries per tile | unchecked entries per tile
    unsigned* smask = reinterpret_cast<unsigned*>(cnt + 8);                  // [nsteps rounded]
    float* tiles = reinterpret_cast<float*>(smask + ((nsteps + 3) & ~3));
    float* wimg = tiles + (size_t)kFmWaves * tile_floats;   // [12][64]: second product's B operands (MFCC: DCT x lifter)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, kq = lane >> 4;
    FB_STAMP0(0);
    float* tile = tiles + (size_t)wave * tile_floats;
    const float* arow = tile + i * K;
    const float* arow_kq = arow + (kq << 4);
    const int klim = K - (kq << 4);   // k-slot kq of an entry is inside the row iff off < klim
    const long ntiles = (F + kFmRows - 1) / kFmRows;
    const long stride = (long)gridDim.x * kFmWaves;
    const int n4 = (kFmRows * K) >> 2;  // 16 K floats = 4 K float4
    fm_f4 pre[NQ];
    auto issue = [&](long tl) {  // 16-byte loads of a full tile into registers (no wait here)
        if (tl < ntiles && vec4 && (tl + 1) * kFmRows <= F) {
            const fm_f4* src = reinterpret_cast<const fm_f4*>(x + tl * (long)kFmRows * K);
#pragma unroll
            for (int q = 0; q < NQ; ++q)   // no predicate, no branch (index clamped): either makes the compiler wait per load
                pre[q] = src[q * 64 + lane < n4 ? q * 64 + lane : n4 - 1];   // (nontemporal loads: 10x slower to issue)
        }
    };
    long tl = (long)blockIdx.x * kFmWaves + wave;
    issue(tl);   // the first tile is in flight while the program is built
    FB_STAMP0(7);
    const int ecol = (use_power == 1 && C < 48 && E) ? C : -1;
    const bool yvec4 = (((size_t)y) & 15) == 0;   // every tile starts 64 C bytes further
    const float ew = 1.f / (float)(2 * (K - 1));
    {   // H -> LDS (the tile buffers are free until the first tile is staged): the program is built from LDS
        float* Hs = tiles;
        const int nh = K * C;
        if (ldh != C) {
#pragma unroll 4
            for (int q = threadIdx.x; q < nh; q += kFmWaves * 64) {
                const int bin = q / C;
                Hs[q] = H[(long)bin * ldh + (q - bin * C)];
            }
        } else if ((((size_t)H) & 15) == 0) {
            const int nh4 = nh >> 2;
#pragma unroll 4
            for (int q = threadIdx.x; q < nh4; q += kFmWaves * 64)
                reinterpret_cast<fm_f4*>(Hs)[q] = reinterpret_cast<const fm_f4*>(H)[q];
            for (int q = (nh4 << 2) + threadIdx.x; q < nh; q += kFmWaves * 64) Hs[q] = H[q];
        } else {
#pragma unroll 4
            for (int q = threadIdx.x; q < nh; q += kFmWaves * 64) Hs[q] = H[q];
        }
    }
    __syncthreads();
    FB_STAMP0(8);
    const float* Hl = tiles;
    auto hval = [&](int bin, int ch) -> float {   // branch-free (clamped read + selects): lets the reads of a batch overlap
        const float h = Hl[(bin < K ? bin : K - 1) * C + (ch < C ? ch : C - 1)];
        const float e = (ch == ecol) ? ((bin == 0 || bin == K - 1) ? ew : 2.f * ew) : 0.f;   // fbank.py:319-320
        return bin < K ? (ch < C ? h : e) : 0.f;
    };
    for (int s0 = wave * 4; s0 < nsteps; s0 += kFmWaves * 4) {   // 4 steps x 3 channel tiles per wave and round
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + u;
            const int bin = s < nsteps ? ((s >> 4) << 6) + (s & 15) + (kq << 4) : K;
#pragma unroll
            for (int t = 0; t < 3; ++t) v[u][t] = hval(bin, 16 * t + i);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned m = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (__ballot(v[u][t] != 0.f) != 0ull) m |= 1u << t;
            if (lane == 0 && s0 + u < nsteps) smask[s0 + u] = m;
        }
    }
    __syncthreads();
    FB_STAMP0(9);
    if (wave < 3) {   // wave t compacts the step list of channel tile t (ballot + prefix count)
        const int t = wave;
        int n = 0, nfull = 0;
        for (int s0 = 0; s0 < nsteps; s0 += 64) {
            const int s = s0 + lane;
            const bool bit = s < nsteps && ((smask[s < nsteps ? s : 0] >> t) & 1u);
            const unsigned long long bits = __ballot(bit);
            const int off = ((s >> 4) << 6) + (s & 15);
            if (bit) prog[t * cap + n + __popcll(bits & ((1ull << lane) - 1ull))] = off;
            n += __popcll(bits);
            nfull += __popcll(__ballot(bit && off + 48 < K));   // a prefix of the list (offsets ascend)
        }
        for (int e = n + lane; e < cap; e += 64) prog[t * cap + e] = -1;   // batch padding and the read-ahead region
        if (lane == 0) {
            cnt[t] = (n + kFmU - 1) / kFmU * kFmU;
            cnt[4 + t] = nfull / kFmU * kFmU;   // whole batches whose 4 k-slots all lie inside the row: no range checks
        }
    }
    __syncthreads();
    FB_STAMP0(10);
    for (int t = 0; t < 3; ++t) {
        const int n = cnt[t];   // a multiple of kFmU = 8
        for (int e0 = wave * 8; e0 < n; e0 += kFmWaves * 8) {
            int off[8];
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) off[u] = prog[t * cap + e0 + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = hval(off[u] >= 0 ? off[u] + (kq << 4) : K, 16 * t + i);
#pragma unroll
            for (int u = 0; u < 8; ++u) himg[(size_t)(t * cap + e0 + u) * 64 + lane] = v[u];
        }
    }
    __syncthreads();
    FB_STAMP0(11);
    if (W2) {   // z = y W2 (C x Mo, Mo <= 16): k-slot kq of step s is channel 4 s + kq, column = lane & 15
        for (int q = threadIdx.x; q < 12 * 64; q += kFmWaves * 64) {
            const int ch = 4 * (q >> 6) + ((q & 63) >> 4), col = q & 15;
            wimg[q] = (ch < C && col < Mo) ? W2[ch * Mo + col] : 0.f;
        }
        __syncthreads();
    }
    int n_t[3], nf_t[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        n_t[t] = __builtin_amdgcn_readfirstlane(cnt[t]);
        nf_t[t] = __builtin_amdgcn_readfirstlane(cnt[4 + t]);
    }

    FB_STAMP0(1);
    for (; tl < ntiles; tl += stride) {
        const long f0 = tl * kFmRows;
        FB_STAMP(2);
        __builtin_amdgcn_wave_barrier();
        if (vec4 && f0 + kFmRows <= F) {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q * 64 + lane < n4) {
                    fm_f4 v = pre[q];
                    if (use_power == 0) v = fm_f4{__builtin_amdgcn_sqrtf(v.x), __builtin_amdgcn_sqrtf(v.y), __builtin_amdgcn_sqrtf(v.z), __builtin_amdgcn_sqrtf(v.w)};  // fbank.py:315 (v_sqrt_f32: 1 ulp)
                    if (use_power == 2) v = fm_f4{dsa_log(v.x), dsa_log(v.y), dsa_log(v.z), dsa_log(v.w)};   // fftcep.py:122
                    reinterpret_cast<fm_f4*>(tile)[q * 64 + lane] = v;
                }
        } else {  // ragged last tile or unaligned spectrum: element loads, missing rows read as 1
            const long have = (F - f0 < kFmRows ? F - f0 : kFmRows) * (long)K;
            for (int e = lane; e < kFmRows * K; e += 64) {
                const float v = e < have ? x[f0 * K + e] : 1.f;
                tile[e] = use_power == 1 ? v : (use_power == 2 ? dsa_log(v) : __builtin_amdgcn_sqrtf(v));
            }
        }
        __builtin_amdgcn_wave_barrier();
        FB_STAMP(3);
        issue(tl + stride);
        FB_STAMP(4);
        fm_f4 acc[3][2];
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t][0] = acc[t][1] = fm_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int* pg = prog + t * cap;
            const float* hb = himg + (size_t)t * cap * 64 + lane;
            // float32 MFMA and the vector ALU share one datapath on this chip: every vector instruction in this
            // loop costs matrix time, so the common case (all 4 k-slots inside the row) is one address add per
            // product and the range-checked form only runs for the last, partial block of bins.
            // Software pipeline: batch n+1's operand reads are issued before batch n's products, batch n+2's bin
            // offsets before that (the lists are padded with -1 for two batches beyond their end).
            auto run = [&](int eb, int ee, auto chk) {
                constexpr bool kCheck = decltype(chk)::value;
                if (eb >= ee) return;
                int off[kFmU];
                float a[kFmU], b[kFmU];
                auto fetch = [&](int e0) {
#pragma unroll
                    for (int u = 0; u < kFmU; ++u) {
                        if (kCheck) {
                            const bool ok = (unsigned)off[u] < (unsigned)klim;   // off = -1 wraps to a huge value
                            const float v = arow_kq[ok ? off[u] : 0];
                            a[u] = ok ? v : 0.f;
                        } else {
                            a[u] = arow_kq[off[u]];
                        }
                        b[u] = hb[(size_t)(e0 + u) * 64];
                    }
                };
#pragma unroll
                for (int u = 0; u < kFmU; ++u) off[u] = pg[eb + u];
                fetch(eb);
#pragma unroll
                for (int u = 0; u < kFmU; ++u) off[u] = pg[eb + kFmU + u];
                asm volatile("" ::: "memory");
                for (int e0 = eb; e0 < ee; e0 += kFmU) {
                    float ac[kFmU], bc[kFmU];
#pragma unroll
                    for (int u = 0; u < kFmU; ++u) {
                        ac[u] = a[u];
                        bc[u] = b[u];
                    }
                    fetch(e0 + kFmU);
#pragma unroll
                    for (int u = 0; u < kFmU; ++u) off[u] = pg[e0 + 2 * kFmU + u];
                    asm volatile("" ::: "memory");   // the reads above stay above the products (the scheduler sinks them otherwise)
#pragma unroll
                    for (int u = 0; u < kFmU; ++u)
                        acc[t][u & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[u], bc[u], acc[t][u & 1], 0, 0, 0);
                }
            };
            run(0, nf_t[t], std::false_type{});
            run(nf_t[t], n_t[t], std::true_type{});
        }
        FB_STAMP(5);
        float es = 0.f;
        if (E && ecol < 0) {   // energy on the vector unit: every bin of the row once
#pragma unroll 8
            for (int s = 0; s < nsteps; ++s) {
                const int bin = ((s >> 4) << 6) + (s & 15) + (kq << 4);
                const float a = bin < K ? arow[bin < K ? bin : 0] : 0.f;
                const float w = (bin == 0 || bin == K - 1) ? 1.f : 2.f;
                es += use_power ? w * a : w * a * a;
            }
            es += __shfl_xor(es, 16);
            es += __shfl_xor(es, 32);
        }
        FB_STAMP(6);
        // the 16 x C results are one contiguous stretch of y: through the (now free) tile buffer, 16-byte stores
        __builtin_amdgcn_wave_barrier();
        float* ost = tile;                      // [16][C]
        float* est = tile + kFmRows * C;        // [16]
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const fm_f4 d = acc[t][0] + acc[t][1];
            const int ch = 16 * t + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int fr = 4 * kq + r;
                if (ch < C) {
                    if (post_mode) {
                        const bool halve = !(post_mode & 4) && (ch == 0 || ((post_mode & 2) && ch == C - 1));   // 4: a plain product
                        ost[fr * C + ch] = d[r] * (halve ? 0.5f * post_scale : post_scale);
                    } else {
                        const float v = d[r] < floor ? floor : d[r];                // fbank.py:317 (NaN stays NaN)
                        ost[fr * C + ch] = glog_fwd(v, gamma);
                    }
                } else if (ch == ecol) {
                    est[fr] = dsa_log(d[r]);                                        // fbank.py:320
                }
            }
        }
        if (E && ecol < 0 && lane < kFmRows) est[lane] = dsa_log(es * ew);
        __builtin_amdgcn_wave_barrier();
        if (W2) {   // MFCC: the DCT-II x lifter product straight from the staged log filter-bank outputs (mfcc.py:249-252)
            fm_f4 zc = fm_f4{0.f, 0.f, 0.f, 0.f};
            const int csteps = (C + 3) >> 2;
            for (int sidx = 0; sidx < csteps; ++sidx) {
                const int ch = 4 * sidx + kq;
                const float a = ch < C ? ost[i * C + (ch < C ? ch : 0)] : 0.f;
                zc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wimg[sidx * 64 + lane], zc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long f = f0 + 4 * kq + r;
                if (f < F && i < Mo) z[f * Mo + i] = zc[r];
            }
        }
        if (y) {
            const int rows = (int)(F - f0 < kFmRows ? F - f0 : kFmRows);
            const int n = rows * C;
            float* dst = y + f0 * C;
            if (ldy != C) {   // a column slice of a wider output (row stride ldy): row by row
                for (int q = lane; q < n; q += 64) {
                    const int r = q / C;
                    y[(f0 + r) * (long)ldy + (q - r * C)] = ost[q];
                }
            } else if (yvec4) {
                for (int q = lane; q < (n >> 2); q += 64) reinterpret_cast<fm_f4*>(dst)[q] = reinterpret_cast<const fm_f4*>(ost)[q];
                for (int q = (n & ~3) + lane; q < n; q += 64) dst[q] = ost[q];
            } else {
                for (int q = lane; q < n; q += 64) dst[q] = ost[q];
            }
        }
        if (E && lane < kFmRows && f0 + lane < F) E[f0 + lane] = est[lane];
    }
}

static size_t fbank_mfma_lds_bytes(int K, bool second_product)
{
    const int nsteps = 16 * (K / 64) + ((K % 64) < 16 ? (K % 64) : 16);
    const int tile_floats = (kFmRows * K + 3) & ~3;
    const int cap = ((nsteps + kFmU - 1) / kFmU + 3) * kFmU;
    return ((size_t)3 * cap * 64 + 3 * cap + 8 + ((nsteps + 3) & ~3) + (size_t)kFmWaves * tile_floats +
            (second_product ? 12 * 64 : 0)) * 4;
}

int fbank_mfma_launch_ex(const void* x, int64_t F, int K, const void* H, int C, int ldh, double floor, double gamma,
                         int use_power, int post_mode, double post_scale, void* y, void* E, hipStream_t st, const char* name,
                         const void* W2, int Mo, void* z, int ldy)
{
    const int nsteps = 16 * (K / 64) + ((K % 64) < 16 ? (K % 64) : 16);
    const int tile_floats = (kFmRows * K + 3) & ~3;
    const int cap = ((nsteps + kFmU - 1) / kFmU + 3) * kFmU;   // every list padded to a batch, plus the read-ahead batches
    const size_t lds = fbank_mfma_lds_bytes(K, W2 != nullptr);
    if (lds > 160 * 1024) return fail(DSA_ERR_UNSUPPORTED, "fbank: spectrum too long for the matrix-core kernel's LDS%s");
    const long ntiles = (long)((F + kFmRows - 1) / kFmRows);
    long blocks = (ntiles + kFmWaves - 1) / kFmWaves;
    if (blocks > 256) blocks = 256;
    const int vec4 = (((size_t)x) & 15) == 0;
    const int nq = (4 * K + 63) / 64;
#define DSA_FM_LAUNCH(NQ)                                                                                              \
    do {                                                                                                               \
        static std::atomic<uint64_t> attr_devices{0};                                                                  \
        if (!ensure_dynamic_lds((const void*)fbank_mfma_fwd_kernel<NQ>, 160 * 1024, attr_devices))                     \
            return fail(DSA_ERR_LAUNCH, "fbank: cannot reserve LDS for the operand images%s");                        \
        hipLaunchKernelGGL(fbank_mfma_fwd_kernel<NQ>, dim3((unsigned)blocks), dim3(kFmWaves * 64), lds, st,            \
                           (const float*)x, (long)F, K, (const float*)H, C, (float)floor, (float)gamma, use_power,     \
                           (float*)y, (float*)E, nsteps, cap, tile_floats, vec4, ldh, post_mode, (float)post_scale,    \
                           (const float*)W2, Mo, (float*)z, ldy > 0 ? ldy : C);                                       \
    } while (0)
    if (nq <= 5) DSA_FM_LAUNCH(5);
    else if (nq <= 9) DSA_FM_LAUNCH(9);
    else if (nq <= 17) DSA_FM_LAUNCH(17);
    else DSA_FM_LAUNCH(20);
#undef DSA_FM_LAUNCH
    return check_launch(name);
}

static int fbank_mfma_launch(const void* x, int64_t F, int K, const void* H, int C, double floor, double gamma,
                             int use_power, void* y, void* E, hipStream_t st)
{
    return fbank_mfma_launch_ex(x, F, K, H, C, C, floor, gamma, use_power ? 1 : 0, 0, 1.0, y, E, st, "fbank_mfma_fwd", nullptr, 0,
                                nullptr);
}

template <typename T>
static int fbank_launch(bool bwd, const void* gy, const void* gE, const void* x, int64_t F, int K, const void* H, int C,
                        double floor, double gamma, int use_power, void* y, void* E, void* gx, hipStream_t st)
{
    if (F == 0) return DSA_OK;
    const size_t tile1 = sizeof(T) * kFbFrames * (size_t)(bwd ? K + C : K);   // per wave
    const size_t ranges = sizeof(int) * 2 * (size_t)(bwd ? K + C : C);
    const size_t hmat = sizeof(T) * (size_t)K * C;
    const size_t budget = 150 * 1024;
    if (4 * tile1 + ranges > budget) return fail(DSA_ERR_UNSUPPORTED, "fbank: spectrum too long for LDS%s");
    const bool hlds = 4 * tile1 + ranges + hmat <= budget;   // else H stays in global memory (full ranges)
    long waves = (long)((budget - ranges - (hlds ? hmat : 0)) / tile1);
    if (waves > kFbMaxWaves) waves = kFbMaxWaves;
    const long groups = (F + kFbFrames - 1) / kFbFrames;
    if (waves > groups) waves = groups < 1 ? 1 : groups;
    const size_t lds = (size_t)waves * tile1 + ranges + (hlds ? hmat : 0);
    long blocks = (groups + waves - 1) / waves;
    if (blocks > 256) blocks = 256;   // one persistent workgroup per CU; H is copied once each
    const int kFbWaves = (int)waves;
#define DSA_FB_ATTR(kern)                                                                                        \
    do {                                                                                                         \
        static std::atomic<uint64_t> attr_devices{0};                                                            \
        if (lds > 48 * 1024 && !ensure_dynamic_lds((const void*)kern, 150 * 1024, attr_devices))                 \
            return fail(DSA_ERR_LAUNCH, "fbank: cannot reserve LDS for the filter matrix%s");                   \
    } while (0)
    if (!bwd) {
        if (hlds) {
            DSA_FB_ATTR((fbank_fwd_kernel<T, true>));
            hipLaunchKernelGGL((fbank_fwd_kernel<T, true>), dim3((unsigned)blocks), dim3(kFbWaves * 64), lds, st, (const T*)x,
                               (long)F, K, (const T*)H, C, (T)floor, (T)gamma, use_power, (T*)y, (T*)E);
        } else {
            hipLaunchKernelGGL((fbank_fwd_kernel<T, false>), dim3((unsigned)blocks), dim3(kFbWaves * 64), lds, st, (const T*)x,
                               (long)F, K, (const T*)H, C, (T)floor, (T)gamma, use_power, (T*)y, (T*)E);
        }
        return check_launch("fbank_fwd");
    }
    if (hlds) {
        DSA_FB_ATTR((fbank_bwd_kernel<T, true>));
        hipLaunchKernelGGL((fbank_bwd_kernel<T, true>), dim3((unsigned)blocks), dim3(kFbWaves * 64), lds, st, (const T*)gy,
                           (const T*)gE, (const T*)x, (long)F, K, (const T*)H, C, (T)floor, (T)gamma, use_power, (T*)gx);
    } else {
        hipLaunchKernelGGL((fbank_bwd_kernel<T, false>), dim3((unsigned)blocks), dim3(kFbWaves * 64), lds, st, (const T*)gy,
                           (const T*)gE, (const T*)x, (long)F, K, (const T*)H, C, (T)floor, (T)gamma, use_power, (T*)gx);
    }
#undef DSA_FB_ATTR
    return check_launch("fbank_bwd");
}

}  // namespace dsa

using namespace dsa;

DSA_EXPORT int dsa_fbank_fwd(const void* x, int64_t F, int32_t K, const void* H, int32_t C, double floor, double gamma,
                             int32_t use_power, int32_t dtype, void* y, void* E, void* stream)
{
    DSA_REQUIRE(F >= 0 && K >= 2 && C >= 1, "fbank: sizes must be positive");
    DSA_REQUIRE(floor > 0 && gamma >= -1 && gamma <= 1, "fbank: floor must be positive and gamma in [-1, 1]");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSA_F32 && C <= 48 && K <= 16 * kFmPre && K > C && F > 0 && !env_flag("DSA_FBANK_GENERIC"))
        return fbank_mfma_launch(x, F, K, H, C, floor, gamma, use_power, y, E, st);
    if (dtype == DSA_F32) return fbank_launch<float>(false, nullptr, nullptr, x, F, K, H, C, floor, gamma, use_power, y, E, nullptr, st);
    if (dtype == DSA_F64) return fbank_launch<double>(false, nullptr, nullptr, x, F, K, H, C, floor, gamma, use_power, y, E, nullptr, st);
    return fail(DSA_ERR_UNSUPPORTED, "fbank: unsupported dtype%s");
}

DSA_EXPORT int dsa_fbank_bwd(const void* gy, const void* gE, const void* x, int64_t F, int32_t K, const void* H, int32_t C,
                             double floor, double gamma, int32_t use_power, int32_t dtype, void* gx, void* stream)
{
    DSA_REQUIRE(F >= 0 && K >= 2 && C >= 1, "fbank_bwd: sizes must be positive");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSA_F32) return fbank_launch<float>(true, gy, gE, x, F, K, H, C, floor, gamma, use_power, nullptr, nullptr, gx, st);
    if (dtype == DSA_F64) return fbank_launch<double>(true, gy, gE, x, F, K, H, C, floor, gamma, use_power, nullptr, nullptr, gx, st);
    return fail(DSA_ERR_UNSUPPORTED, "fbank_bwd: unsupported dtype%s");
}

// MFCC front end (mfcc.py:244-256): z = glog(max(s H, floor)) W, W:(C, Mo) = DCT-II x truncation x lifter; the filter-bank
// outputs themselves are not written.  Fused into the matrix-core kernel when it applies (float32, C <= 48, Mo <= 16),
// else the two generic launches with a stream-ordered temporary.
DSA_EXPORT int dsa_fbank_dct_fwd(const void* x, int64_t F, int32_t K, const void* H, int32_t C, const void* W, int32_t Mo,
                                 double floor, double gamma, int32_t use_power, int32_t dtype, void* z, void* E, void* stream)
{
    DSA_REQUIRE(F >= 0 && K >= 2 && C >= 1 && Mo >= 1, "fbank_dct: sizes must be positive");
    DSA_REQUIRE(floor > 0 && gamma >= -1 && gamma <= 1, "fbank_dct: floor must be positive and gamma in [-1, 1]");
    if (F == 0) return DSA_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DSA_F32 && C <= 48 && Mo <= 16 && K <= 16 * kFmPre && K > C && fbank_mfma_lds_bytes(K, true) <= 160 * 1024 &&
        !env_flag("DSA_FBANK_GENERIC"))
        return fbank_mfma_launch_ex(x, F, K, H, C, C, floor, gamma, use_power ? 1 : 0, 0, 1.0, nullptr, E, st, "fbank_dct_mfma_fwd",
                                    W, Mo, z);
    const size_t esz = dtype == DSA_F64 ? 8 : 4;
    void* y = nullptr;
    if (hipMallocAsync(&y, esz * (size_t)F * C, st) != hipSuccess) return fail(DSA_ERR_LAUNCH, "fbank_dct: workspace allocation failed%s");
    int rc = dsa_fbank_fwd(x, F, K, H, C, floor, gamma, use_power, dtype, y, E, stream);
    if (rc == DSA_OK) rc = dsa_freqt_fwd(y, F, C, W, Mo, dtype, z, stream);
    (void)hipFreeAsync(y, st);
    return rc;
}

// ---- cotangent of the filter-bank INPUT from the cotangent of its OUTPUT, for matrices with at most two adjacent
// channels per bin (the mel / auditory filters): the backward of the fused STFT -> filter-bank launch, which keeps no
// spectrum.  With y = glog(max(s H, floor)) saved,  d y_c / d (s H)_c = exp(-y_c)  (gamma = 0)  resp.
// (1 + gamma y_c)^((gamma - 1) / gamma), and 0 where the floor clamped (torch.clip in fbank.py:312), so
//     g[f][k] = w0[k] Q(f, c_k) + w1[k] Q(f, c_k + 1),   Q(f, c) = gy[f][c] * that factor
// -- two multiply-adds per bin instead of a (C x K) product (as a dense row product this step cost more than the STFT
// backward it feeds).  Memory-bound: reads 8 C bytes, writes 4 K bytes per frame.
namespace dsa {
__global__ __launch_bounds__(256) void fbank_bins_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, long F, int K,
                                                             int C, const float4* __restrict__ table, float thr, float gamma,
                                                             float* __restrict__ g)
{
    const long total = F * K;
    const float ex = gamma == 0.f ? 0.f : (gamma - 1.f) / gamma;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long f = i / K;
        const int k = (int)(i - f * K);
        const float4 t = table[k];
        const int c0 = __float_as_int(t.x);
        const int c1 = c0 + 1 < C ? c0 + 1 : c0;
        const float y0 = y[f * C + c0], y1 = y[f * C + c1];
        const float d0 = gamma == 0.f ? __expf(-y0) : dsa_pow(1.f + gamma * y0, ex);
        const float d1 = gamma == 0.f ? __expf(-y1) : dsa_pow(1.f + gamma * y1, ex);
        const float q0 = y0 > thr ? gy[f * C + c0] * d0 : 0.f;
        const float q1 = y1 > thr ? gy[f * C + c1] * d1 : 0.f;
        g[i] = t.y * q0 + t.z * q1;
    }
}
}  // namespace dsa

// table[k] = {bits(c_k), w0, w1, 0}: H[k][c_k] = w0, H[k][c_k + 1] = w1 are the only non-zero entries of row k
// (c_k + 1 == C: w1 = 0).  DSA_ERR_UNSUPPORTED when a row has more, or non-adjacent, non-zero entries.
DSA_EXPORT int dsa_fbank_bins_plan(const double* H, int32_t K, int32_t C, float* table)
{
    DSA_REQUIRE(H && table && K >= 1 && C >= 1, "fbank_bins_plan: bad arguments");
    for (int k = 0; k < K; ++k) {
        int first = -1, count = 0, last = -1;
        for (int c = 0; c < C; ++c) {
            const double v = H[(size_t)k * C + c];
            if (!(v == v) || v > 1.7e308 || v < -1.7e308) return fail(DSA_ERR_UNSUPPORTED, "fbank_bins_plan: non-finite matrix%s");
            if (v != 0.0) {
                if (first < 0) first = c;
                last = c;
                ++count;
            }
        }
        if (count > 2 || (count == 2 && last != first + 1))
            return fail(DSA_ERR_UNSUPPORTED, "fbank_bins_plan: a bin feeds more than two adjacent channels%s");
        int c0 = first < 0 ? 0 : first;
        float w0 = first < 0 ? 0.f : (float)H[(size_t)k * C + c0];
        float w1 = count == 2 ? (float)H[(size_t)k * C + c0 + 1] : 0.f;
        int32_t bits = c0;
        float fb;
        memcpy(&fb, &bits, 4);
        table[4 * k + 0] = fb;
        table[4 * k + 1] = w0;
        table[4 * k + 2] = w1;
        table[4 * k + 3] = 0.f;
    }
    return DSA_OK;
}

DSA_EXPORT int dsa_fbank_bins_bwd(const void* gy, const void* y, int64_t F, int32_t K, int32_t C, const void* table, double floor,
                                  double gamma, int32_t dtype, void* g, void* stream)
{
    DSA_REQUIRE(F >= 0 && K >= 1 && C >= 1, "fbank_bins_bwd: sizes must be positive");
    DSA_REQUIRE(floor > 0 && gamma >= -1 && gamma <= 1, "fbank_bins_bwd: floor must be positive and gamma in [-1, 1]");
    if (dtype != DSA_F32) return fail(DSA_ERR_UNSUPPORTED, "fbank_bins_bwd: float32 only (the fused forward it differentiates is)%s");
    if (F == 0) return DSA_OK;
    // glog(floor) as the forward kernels produce it, plus four units in the last place: their logarithm is the hardware's
    double thr_d = gamma == 0 ? log(floor) : (pow(floor, gamma) - 1.0) / gamma;
    const float thr = (float)(thr_d + 4.8e-7 * fabs(thr_d) + 1e-30);
    long blocks = ((long)F * K + 255) / 256;
    if (blocks > 256L * 32) blocks = 256L * 32;
    hipLaunchKernelGGL(dsa::fbank_bins_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)gy,
                       (const float*)y, (long)F, K, C, (const float4*)table, thr, (float)gamma, (float*)g);
    return dsa::check_launch("fbank_bins_bwd");
}
