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

namespace dsa {

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
                    const T v = acc[fi] > floor ? acc[fi] : floor;                                   // fbank.py:317
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
        static bool done = false;                                                                                \
        if (!done && lds > 48 * 1024) {                                                                          \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != \
                hipSuccess)                                                                                      \
                return fail(DSA_ERR_LAUNCH, "fbank: cannot reserve LDS for the filter matrix%s");               \
            done = true;                                                                                         \
        }                                                                                                        \
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
