// Round 6: the ADJOINT of the spectral half of a Newton step of MelCepstralAnalysis at the geometries the tile kernels do not cover
// (the 48 kHz set-ups: fft_length 1024 / 2048, orders 32 .. 54), on the binary16 matrix pipe (included by mcep_mfma.hip).
//
//   forward (mcep_resid_f16.h):   tau = logx - 2 mc D,  e = exp(tau),  rt = e E                                  (mcep.py:210-215)
//   this launch, given grt:(F, N): ebar = grt E^T,  z = ebar * e,  glogx += z,  gmc = -2 z D^T                    (their autograd)
//
// Until now a gradient at these geometries ran the step as differentiable pieces -- two float32 row products with e:(F, K) kept for
// the backward, their transposed products, two element-wise launches and the stock additions of the gradient graph: 2.0 ms per step
// and 102 400 frames, 4.2 GB of saved e per analysis.  Here the step's backward is dsa_mcep_newton_update_bwd (the solve on the
// cotangent + the diagonal sums: grt) and ONE launch of this kernel, which recomputes e from the iterate exactly as the forward formed
// it and never stores it.  Anatomy of mcep_resid_h_kernel: one wave = 16 frames, the bins in stages of 32 whose operand images
// (prepared once per configuration) are staged through LDS for the workgroup's four waves; per stage
//   first chain   t = log2(e) logx + (-2 log2(e) D)^T mc          2 tiles x KS1 k-steps x 3 terms  (the forward's image and scales)
//   ebar chain    E[bins, :] grt^T                                 2 tiles x KSE k-steps x 3 terms, grt scaled per frame by a power of two
//   z = ebar exp2(t), added into glogx (read-modify-write of the stage's 32 bins: the sum over the Newton steps lives in memory --
//                 16 frames x 1025 bins do not fit a wave's registers, and the steps are sequentially dependent)
//   third chain   (-2 D)[coef, bins] z                             NT3 tiles x 3 terms, z scaled per stage by a power of two
// The first chain's C/D tiles are the k-slots of the third chain's B operand (as e is of the forward's second chain).
#pragma once

namespace dsa {

namespace mrb {
using namespace mrh;
constexpr int LOG2_SDB = 9;     // scale of the -2 D image of the third chain (|2 D| <= ~40 for |alpha| <= 0.9)
constexpr int VMAX_LOG2 = 12;   // scaled cotangent vectors stay below 2^12
// (rounded up to 256 sixteen-byte pieces: every thread of the workgroup stages the same number of pieces, no conditional store)
constexpr int stage_halves_b(int ks1, int kse, int nt3) { return ((4 * ks1 + 4 * kse + 2 * nt3) * 512 + 2047) / 2048 * 2048; }
constexpr int kse_of(int M1) { return (2 * M1 - 1 + 31) / 32; }
constexpr int nt3_of(int M1) { return (M1 + 15) / 16; }
}  // namespace mrb

// images: per stage j (bins 32 j ..):
//   [2 t][KS1 ks][2 (hi, lo)][64 lane][8 i]  first chain, as mcep_resid_h_prep_kernel: row = bin 32 j + 16 t + (lane & 15), k-slot (g, i) <->
//                                            coefficient 32 ks + 8 g + i, value -2 log2(e) D[c][bin] 2^LOG2_SD
//   [2 t][KSE ks][2][64][8]                  ebar chain: row = the same bin, k-slot (g, i) <-> column m = 32 ks + 8 g + i of rt, value E[bin][m] 2^LOG2_SE
//   [NT3 tc][2][64][8]                       third chain: row = coefficient 16 tc + (lane & 15), k-slot (g, i = 4 t + r) <-> bin 32 j + 16 t + 4 g + r,
//                                            value -2 D[c][bin] 2^LOG2_SDB
// zero outside the matrices
__global__ __launch_bounds__(256) void mcep_resid_bwd_prep_kernel(const float* __restrict__ D, int ldd, const float* __restrict__ E, int lde, int K,
                                                                 int M1, int N, int ks1, int kse, int nt3, _Float16* __restrict__ img)
{
    using namespace mrb;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int sh = stage_halves_b(ks1, kse, nt3);
    const int nstage = (K + 31) / 32;
    if (idx >= (long)nstage * sh / 2) return;   // one thread per (hi, lo) pair of one value
    const int j = (int)(idx / (sh / 2));
    int e = (int)(idx - (long)j * (sh / 2));
    const int c1_pairs = 2 * ks1 * 512, c2_pairs = 2 * kse * 512;
    if (e >= c1_pairs + c2_pairs + nt3 * 512) {   // the stage's padding (never read as an operand)
        img[(long)j * sh + 2 * (c1_pairs + c2_pairs + nt3 * 512) + 2 * (e - (c1_pairs + c2_pairs + nt3 * 512))] = (_Float16)0.f;
        img[(long)j * sh + 2 * (c1_pairs + c2_pairs + nt3 * 512) + 2 * (e - (c1_pairs + c2_pairs + nt3 * 512)) + 1] = (_Float16)0.f;
        return;
    }
    float v = 0.f;
    long base;
    if (e < c1_pairs) {
        const int i = e & 7, l = (e >> 3) & 63, ks = (e >> 9) % ks1, t = (e >> 9) / ks1;
        const int bin = 32 * j + 16 * t + (l & 15), c = 32 * ks + 8 * (l >> 4) + i;
        if (bin < K && c < M1) v = -2.885390081777926815f * D[(long)c * ldd + bin];
        v = __builtin_ldexpf(v, LOG2_SD);
        base = (long)j * sh + (((long)(t * ks1 + ks) * 2) * 64 + l) * 8 + i;
    } else if (e < c1_pairs + c2_pairs) {
        e -= c1_pairs;
        const int i = e & 7, l = (e >> 3) & 63, ks = (e >> 9) % kse, t = (e >> 9) / kse;
        const int bin = 32 * j + 16 * t + (l & 15), m = 32 * ks + 8 * (l >> 4) + i;
        if (bin < K && m < N) v = E[(long)bin * lde + m];
        v = __builtin_ldexpf(v, LOG2_SE);
        base = (long)j * sh + 4 * ks1 * 512 + (((long)(t * kse + ks) * 2) * 64 + l) * 8 + i;
    } else {
        e -= c1_pairs + c2_pairs;
        const int i = e & 7, l = (e >> 3) & 63, tc = e >> 9;
        const int bin = 32 * j + 16 * (i >> 2) + 4 * (l >> 4) + (i & 3), c = 16 * tc + (l & 15);
        if (bin < K && c < M1) v = -2.f * D[(long)c * ldd + bin];
        v = __builtin_ldexpf(v, LOG2_SDB);
        base = (long)j * sh + 4 * ks1 * 512 + 4 * kse * 512 + (((long)tc * 2) * 64 + l) * 8 + i;
    }
    split1(v, img[base], img[base + 512]);
}

// GX = false (0.2.2): glogx is NOT touched -- z = ebar e only feeds the third chain; the sum over the Newton steps is then formed in one
// pass over the bins by mcep_glogx_h_kernel (mcep_glogx_f16.h) from the saved iterates and cotangents, and a step's launch moves 420 MB
// instead of 1.26 GB per 102 400 frames at 1025 bins.
template <int KS1, int KSE, int NT3, bool GX = true>
__global__ __launch_bounds__(256, 2) DSA_PK_TARGET void mcep_resid_bwd_h_kernel(const float* __restrict__ logx, long F, int K, const float* __restrict__ mc,
                                                                                 int M1, const float* __restrict__ grt, int N,
                                                                                 const _Float16* __restrict__ img, float* glogx,
                                                                                 float* __restrict__ gmc)
{
    using namespace mrb;
    constexpr int SH = stage_halves_b(KS1, KSE, NT3);
    constexpr int PIECES = SH / 8;                       // 16-byte pieces per stage
    constexpr int PER = PIECES / 256;
    static_assert(PIECES % 256 == 0, "whole rounds of the workgroup's 256 threads");
    constexpr int OFF2 = (4 * KS1 * 512) / 8, OFF3 = OFF2 + (4 * KSE * 512) / 8;   // (f16x8 units) where the ebar / third-chain images start
    extern __shared__ __attribute__((aligned(16))) _Float16 sbuf_b[];   // [2][SH]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int nstage = (K + 31) / 32;
    const long t16 = ((long)blockIdx.x * WAVES + wave) * 16;      // uniform; the grid covers ceil(F / 64) workgroups
    const bool tile_ok = t16 < F;
    const long tb = tile_ok ? t16 : 0;
    const int rows_here = (int)((F - tb < 16) ? F - tb : 16);
    const int rn = n < rows_here ? n : rows_here - 1;
    const bool row_ok = tile_ok && n < rows_here;
    const float* xt = logx + tb * (long)K + (long)rn * K;
    float* gxt = GX ? glogx + tb * (long)K + (long)rn * K : nullptr;
    const f32x4* img4 = reinterpret_cast<const f32x4*>(img);
    // (Measured and not kept, profiles/r06_48khz_gradient_one_node.txt: every load and store of the stage loop unconditional -- clamped
    // addresses, raw-buffer stores pushed out of range where nothing is to be written, the partial last stage's bin in registers of its
    // own -- so that the compiler's wait counts stay exact instead of `vmcnt(0)`: 2-5 % faster at 12 800 frames, 9 % SLOWER per launch at
    // 102 400 (539 against 495 us: the launch is bound by its 1.26 GB, and the always-issued extra store costs more than the drained
    // waits), and only for K = 32 m (+ 1) bins.)
    f32x4 st0[PER];   // one stage ahead in registers, the stage in use in the other LDS buffer
    auto fetch = [&](int j, f32x4 (&sv)[PER]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int p = tid + 256 * q;
            const long src = (long)j * PIECES + p, last = (long)nstage * PIECES - 1;
            sv[q] = img4[src < last ? src : last];
        }
    };
    auto stage = [&](int buf, const f32x4 (&sv)[PER]) __attribute__((always_inline)) {
        f32x4* d = reinterpret_cast<f32x4*>(sbuf_b + buf * SH);
#pragma unroll
        for (int q = 0; q < PER; ++q) d[tid + 256 * q] = sv[q];
    };
    fetch(0, st0);
    // B operands of the first chain (mc) and of the ebar chain (grt): this lane's frame, scaled per frame by a power of two
    f16x8 bh[KS1], bl[KS1], rh[KSE], rl[KSE];
    int k1, ke;
    {
        float bv[KS1][8];
        float bmax = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = 32 * ks + 8 * g + i;
                bv[ks][i] = c < M1 ? mc[(tb + rn) * (long)M1 + c] : 0.f;
                bmax = __builtin_fmaxf(bmax, __builtin_fabsf(bv[ks][i]));
            }
        bmax = rows_max4(bmax);
        const int s_b = 12 - __builtin_amdgcn_frexp_expf(bmax);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            float ms[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ms[i] = __builtin_ldexpf(bv[ks][i], s_b);
            split8(ms, bh[ks], bl[ks]);
        }
        k1 = -s_b - LOG2_SD;     // the first chain's accumulators x 2^k1 = (-2 log2(e) D)^T mc
        float rv[KSE][8];
        float rmax = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSE; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = 32 * ks + 8 * g + i;
                rv[ks][i] = m < N ? grt[(tb + rn) * (long)N + m] : 0.f;
                rmax = __builtin_fmaxf(rmax, __builtin_fabsf(rv[ks][i]));
            }
        rmax = rows_max4(rmax);
        const int s_g = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(rmax);
#pragma unroll
        for (int ks = 0; ks < KSE; ++ks) {
            float ms[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ms[i] = __builtin_ldexpf(rv[ks][i], s_g);
            split8(ms, rh[ks], rl[ks]);
        }
        ke = -s_g - LOG2_SE;     // the ebar chain's accumulators x 2^ke = E grt^T
    }
    f32x4 acc[NT3];
#pragma unroll
    for (int t = 0; t < NT3; ++t) acc[t] = zero4;
    // the lane's log-spectrum values and gradient sums of a stage: bins 32 j + 16 t + 4 g + r; bins past the end read the row's last
    // value, are masked to e = 0 below and never stored
    f32x4 x0[2], x1[2], a0[2], a1[2];
    auto xfetch = [&](int j, f32x4 (&xr)[2], f32x4 (&ar)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int b0 = 32 * j + 16 * t + 4 * g;
            if (b0 + 3 < K) {
                xr[t] = *reinterpret_cast<const f32x4_u4*>(xt + b0);
                if constexpr (GX) ar[t] = *reinterpret_cast<const f32x4_u4*>(gxt + b0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    xr[t][r] = xt[b0 + r < K ? b0 + r : K - 1];
                    if constexpr (GX) ar[t][r] = gxt[b0 + r < K ? b0 + r : K - 1];
                }
            }
            if constexpr (!GX) ar[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    xfetch(0, x0, a0);
    xfetch(nstage > 1 ? 1 : 0, x1, a1);
    stage(0, st0);
    if (nstage > 1) fetch(1, st0);
    __syncthreads();
    // stage j: `sv` holds stage j + 1 (requested during stage j - 1) and takes stage j + 2 once staged; (xr, ar) hold the rows of stage j
    auto body = [&](int j, f32x4 (&sv)[PER], f32x4 (&xr)[2], f32x4 (&ar)[2]) __attribute__((always_inline)) {
        const int buf = j & 1;
        const f32x4 xv[2] = {xr[0], xr[1]};
        const f32x4 av_[2] = {ar[0], ar[1]};
        if (j + 1 < nstage) stage(buf ^ 1, sv);   // the other buffer: its readers finished before the barrier that ended stage j - 1
        if (j + 2 < nstage) {
            xfetch(j + 2, xr, ar);
            fetch(j + 2, sv);
        }
        if (tile_ok) {
            const f16x8* c1 = reinterpret_cast<const f16x8*>(sbuf_b + buf * SH) + lane;
            const f16x8* c2 = c1 + OFF2;
            const f16x8* c3 = c1 + OFF3;
            // first chain and ebar chain: independent, interleaved tile by tile
            f32x4 s[2] = {zero4, zero4}, eb[2] = {zero4, zero4};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    const f16x8 dh = c1[((t * KS1 + ks) * 2 + 0) * 64], dl = c1[((t * KS1 + ks) * 2 + 1) * 64];
                    s[t] = mfma_h(dl, bh[ks], s[t]);
                    s[t] = mfma_h(dh, bl[ks], s[t]);
                    s[t] = mfma_h(dh, bh[ks], s[t]);
                }
#pragma unroll
                for (int ks = 0; ks < KSE; ++ks) {
                    const f16x8 eh_ = c2[((t * KSE + ks) * 2 + 0) * 64], el_ = c2[((t * KSE + ks) * 2 + 1) * 64];
                    eb[t] = mfma_h(el_, rh[ks], eb[t]);
                    eb[t] = mfma_h(eh_, rl[ks], eb[t]);
                    eb[t] = mfma_h(eh_, rh[ks], eb[t]);
                }
            }
            // e = exp(tau) (dead bins: 0), z = ebar e, glogx += z
            float zv[8];
            float zm = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 o = av_[t];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool live = 32 * j + 16 * t + 4 * g + r < K;
                    const float tv = __builtin_fmaf(xv[t][r], 1.4426950408889634f, __builtin_ldexpf(s[t][r], k1));
                    const float e_ = live ? __builtin_amdgcn_exp2f(tv) : 0.f;
                    const float z = __builtin_ldexpf(eb[t][r] * e_, ke);
                    zv[4 * t + r] = z;
                    zm = __builtin_fmaxf(zm, __builtin_fabsf(z));
                    o[r] += z;
                }
                if (GX && row_ok) {
                    const int b0 = 32 * j + 16 * t + 4 * g;
                    if (b0 + 3 < K) {
                        *reinterpret_cast<f32x4_u4*>(gxt + b0) = o;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (b0 + r < K) gxt[b0 + r] = o[r];
                    }
                }
            }
            // third chain: gmc += (-2 D) z with the stage's own scale
            zm = rows_max4(zm);
            const int s_z = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(zm);
            float zs[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) zs[i] = __builtin_ldexpf(zv[i], s_z);
            f16x8 zh, zl;
            split8(zs, zh, zl);
            const int k3 = -s_z - LOG2_SDB;
#pragma unroll
            for (int tc = 0; tc < NT3; ++tc) {
                const f16x8 wh = c3[(tc * 2 + 0) * 64], wlo = c3[(tc * 2 + 1) * 64];
                f32x4 a_ = mfma_h(wlo, zh, zero4);
                a_ = mfma_h(wh, zl, a_);
                a_ = mfma_h(wh, zh, a_);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[tc][r] += __builtin_ldexpf(a_[r], k3);
            }
        }
        __syncthreads();
    };
#pragma unroll 1
    for (int j = 0; j < nstage; j += 2) {
        body(j, st0, x0, a0);
        if (j + 1 < nstage) body(j + 1, st0, x1, a1);
    }
    if (!row_ok) return;
    // C/D layout: lane (n, g) register r of tile tc <-> coefficient 16 tc + 4 g + r of frame n
    float* orow = gmc + (tb + n) * (long)M1;
#pragma unroll
    for (int tc = 0; tc < NT3; ++tc)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * tc + 4 * g + r;
            if (c < M1) orow[c] = acc[tc][r];
        }
}

int64_t mcep_resid_bwd_images_bytes(int K, int M1)
{
    const int ks1 = (M1 + 31) / 32;
    return (int64_t)((K + 31) / 32) * mrb::stage_halves_b(ks1, mrb::kse_of(M1), mrb::nt3_of(M1)) * 2;
}

int mcep_resid_bwd_prepare(const void* D, int ldd, const void* E, int lde, int K, int M1, void* images, hipStream_t st)
{
    const int ks1 = (M1 + 31) / 32, kse = mrb::kse_of(M1), nt3 = mrb::nt3_of(M1);
    const long pairs = (long)((K + 31) / 32) * mrb::stage_halves_b(ks1, kse, nt3) / 2;
    hipLaunchKernelGGL(mcep_resid_bwd_prep_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, (const float*)D, ldd, (const float*)E,
                       lde, K, M1, 2 * M1 - 1, ks1, kse, nt3, (_Float16*)images);
    return check_launch("mcep_resid_bwd_prepare");
}

// orders 32 .. 54 (M1 33 .. 55: two k-steps of coefficients); DSA_ERR_UNSUPPORTED otherwise -- the caller keeps the composed gradient
int mcep_resid_bwd_h(const void* logx, int64_t F, int K, const void* mc, int M1, const void* grt, const void* images, void* glogx, void* gmc,
                     hipStream_t st)
{
    const int ks1 = (M1 + 31) / 32, kse = mrb::kse_of(M1), nt3 = mrb::nt3_of(M1), N = 2 * M1 - 1;
    if (!(ks1 == 2 && M1 >= 33 && M1 <= 55 && K >= 4)) return DSA_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((F + 63) / 64));
#define DSA_RESID_BWD(KSEV, NT3V)                                                                                                         \
    do {                                                                                                                                  \
        constexpr int lds_b = 2 * mrb::stage_halves_b(2, KSEV, NT3V) * 2;                                                                 \
        static std::atomic<uint64_t> attr{0}, attr0{0};                                                                                   \
        if (glogx) {                                                                                                                      \
            if (!ensure_dynamic_lds((const void*)mcep_resid_bwd_h_kernel<2, KSEV, NT3V, true>, lds_b, attr))                              \
                return fail(DSA_ERR_LAUNCH, "mcep_resid_bwd_h: cannot reserve LDS%s");                                                    \
            hipLaunchKernelGGL((mcep_resid_bwd_h_kernel<2, KSEV, NT3V, true>), grid, dim3(256), lds_b, st, (const float*)logx, (long)F, K, \
                               (const float*)mc, M1, (const float*)grt, N, (const _Float16*)images, (float*)glogx, (float*)gmc);          \
        } else {                                                                                                                          \
            if (!ensure_dynamic_lds((const void*)mcep_resid_bwd_h_kernel<2, KSEV, NT3V, false>, lds_b, attr0))                            \
                return fail(DSA_ERR_LAUNCH, "mcep_resid_bwd_h: cannot reserve LDS%s");                                                    \
            hipLaunchKernelGGL((mcep_resid_bwd_h_kernel<2, KSEV, NT3V, false>), grid, dim3(256), lds_b, st, (const float*)logx, (long)F, K, \
                               (const float*)mc, M1, (const float*)grt, N, (const _Float16*)images, (float*)nullptr, (float*)gmc);        \
        }                                                                                                                                 \
    } while (0)
    (void)nt3;
    if (kse == 3) DSA_RESID_BWD(3, 3);   // M1 33 .. 48: N = 2 M1 - 1 <= 95 columns of rt, three tiles of coefficients
    else DSA_RESID_BWD(4, 4);            // M1 49 .. 55
#undef DSA_RESID_BWD
    return check_launch("mcep_resid_bwd_h");
}

}  // namespace dsa
