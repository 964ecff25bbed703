// Round 5: the spectral half of a Newton step of MelCepstralAnalysis for the geometries the tile kernel does not cover (the 48 kHz
// set-ups: fft_length 1024 / 2048, orders 34 / 49), matrix chains as 3-term binary16 splits (included by mcep_mfma.hip):
//   rt:(F, N) = exp(logx - 2 mc D) E,   logx:(F, K) natural logarithms, mc:(F, M1), D:(M1 x K), E:(K x N), N = 2 M1 - 1   (mcep.py:210-215)
// mcep_resid_mfma_kernel (rows_gemm.hip, round 4) runs both products on v_mfma_f32_16x16x4_f32, i.e. on the float32 datapath: 82
// matrix instructions of 32 cycles per 32 bins and 16 frames at order 49 (76 us per step and 12 800 frames at 2048 / 49, two thirds
// of the step).  Here, as in the tile kernels: one wave = 16 frames, the bins in STAGES of 32, a stage's operand images (prepared
// once per configuration by mcep_resid_h_prep_kernel: -2 log2(e) D^T and E^T split into binary16 hi / lo in matrix-instruction
// lane order) staged through LDS for the four waves of a workgroup (double-buffered, one barrier per stage);
//   first chain   t = log2(e) logx + (-2 log2(e) D)^T mc: 2 tiles x KS1 k-steps x 3 terms, mc scaled per frame by a power of two;
//   e             exp2(t - ceil(max t)) of the stage's 32 bins, scaled to 2^13 and split: the first chain's C/D tiles ARE the
//                 k-slots of the second chain's B operand;
//   second chain  NT column tiles x 3 terms, fresh accumulators per stage, added into float32 sums with the stage's scale.
// 12 KS1 / 2 + 3 NT binary16 products per stage instead of 82 float32 ones.
#pragma once

namespace dsa {

namespace mrh {
constexpr int WAVES = 4;
constexpr int LOG2_SD = 9;     // scale of the -2 log2(e) D image (|D| <= ~20 for |alpha| <= 0.9)
constexpr int LOG2_SE = 16;    // scale of the E image (|E| <= ~0.03)
constexpr int EMAX_LOG2 = 13;  // scaled e is at most 2^13
constexpr int stage_halves(int ks1, int nt) { return (4 * ks1 + 2 * nt) * 512; }
}  // namespace mrh

// images: per stage j (bins 32 j ..): [2 t][KS1 ks][2 (hi, lo)][64 lane][8 i] first chain (row = bin 32 j + 16 t + (lane & 15), k-slot
// (g, i) <-> coefficient 32 ks + 8 g + i) | [NT tc][2 (hi, lo)][64 lane][8 i] second chain (row = column 16 tc + (lane & 15),
// k-slot (g, i = 4 t + r) <-> bin 32 j + 16 t + 4 g + r); zero outside the matrices
__global__ __launch_bounds__(256) void mcep_resid_h_prep_kernel(const float* __restrict__ D, int ldd, const float* __restrict__ E, int lde,
                                                               int K, int M1, int N, int ks1, int nt, _Float16* __restrict__ img)
{
    using namespace mrh;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int sh = stage_halves(ks1, nt);
    const int nstage = (K + 31) / 32;
    if (idx >= (long)nstage * sh / 2) return;   // one thread per (hi, lo) pair of one value
    const int j = (int)(idx / (sh / 2));
    int e = (int)(idx - (long)j * (sh / 2));
    const int c1_pairs = 2 * ks1 * 512;
    float v = 0.f;
    long o_hi, o_lo;
    if (e < c1_pairs) {
        const int i = e & 7, l = (e >> 3) & 63, ks = (e >> 9) % ks1, t = (e >> 9) / ks1;
        const int bin = 32 * j + 16 * t + (l & 15), c = 32 * ks + 8 * (l >> 4) + i;
        if (bin < K && c < M1) v = -2.885390081777926815f * D[(long)c * ldd + bin];
        v = __builtin_ldexpf(v, LOG2_SD);
        const long base = (long)j * sh + (((long)(t * ks1 + ks) * 2) * 64 + l) * 8 + i;
        o_hi = base;
        o_lo = base + 512;
    } else {
        e -= c1_pairs;
        const int i = e & 7, l = (e >> 3) & 63, tc = e >> 9;
        const int bin = 32 * j + 16 * (i >> 2) + 4 * (l >> 4) + (i & 3), col = 16 * tc + (l & 15);
        if (bin < K && col < N) v = E[(long)bin * lde + col];
        v = __builtin_ldexpf(v, LOG2_SE);
        const long base = (long)j * sh + 4 * ks1 * 512 + (((long)tc * 2) * 64 + l) * 8 + i;
        o_hi = base;
        o_lo = base + 512;
    }
    split1(v, img[o_hi], img[o_lo]);
}

template <int KS1, int NT>
__global__ __launch_bounds__(256, 2) DSA_PK_TARGET void mcep_resid_h_kernel(const float* __restrict__ logx, long F, int K, const float* __restrict__ mc, int M1,
                                                             const _Float16* __restrict__ img, int N, float* __restrict__ out, int ldo)
{
    using namespace mrh;
    constexpr int SH = stage_halves(KS1, NT);
    constexpr int PIECES = SH / 8;                       // 16-byte pieces per stage
    constexpr int PER = (PIECES + 255) / 256;
    __shared__ __attribute__((aligned(16))) _Float16 sbuf[2][SH];
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
    const float* xt = logx + tb * (long)K + (long)rn * K;
    const f32x4* img4 = reinterpret_cast<const f32x4*>(img);
    // Staging runs TWO stages ahead of the arithmetic (one stage in registers, one in the other LDS buffer) and the log-spectrum
    // rows two stages ahead as well: a small batch leaves one wave per SIMD, nothing hides a round trip to memory but distance
    // (one stage ahead: 76 -> 59.5 us per step at 12 800 frames, a stage no shorter than the trip).
    f32x4 st0[PER], st1[PER];
    auto fetch = [&](int j, f32x4 (&sv)[PER]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int p = tid + 256 * q;
            if (p < PIECES) sv[q] = img4[(long)j * PIECES + p];
        }
    };
    auto stage = [&](int buf, const f32x4 (&sv)[PER]) __attribute__((always_inline)) {
        f32x4* d = reinterpret_cast<f32x4*>(sbuf[buf]);
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int p = tid + 256 * q;
            if (p < PIECES) d[p] = sv[q];
        }
    };
    fetch(0, st0);
    // B operands of the first chain: mc[32 ks + 8 g + i] of this lane's frame, scaled per frame
    f16x8 bh[KS1], bl[KS1];
    int k1;
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
    }
    // TWO sets of sums, the even and the odd stages': added once at the end.  The one-launch kernel of round 6 (mcep_big_f16.h) gives
    // the stages of a tile to two waves alternately and adds their partial sums -- with the same order here, rt is bit-identical
    // whichever of the two runs (the host chooses by batch size: a frame's bits must not depend on that choice).
    // (only where that kernel has instantiations -- orders 35 .. 54: KS1 = 2, NT = 5 .. 7; the second set costs 4 NT registers and
    //  ~7 % of this kernel's time at large batches)
    constexpr bool SPLIT = KS1 == 2 && NT >= 5;
    f32x4 acc[NT], acc_odd[SPLIT ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = zero4;
#pragma unroll
    for (int t = 0; t < (SPLIT ? NT : 1); ++t) acc_odd[t] = zero4;
    // the lane's log-spectrum values of a stage: bins 32 j + 16 t + 4 g + r; bins past the end read the row's last value and are
    // masked below
    f32x4 x0[2], x1[2];
    auto xfetch = [&](int j, f32x4 (&xr)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int b0 = 32 * j + 16 * t + 4 * g;
            if (b0 + 3 < K) {
                xr[t] = *reinterpret_cast<const f32x4_u4*>(xt + b0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) xr[t][r] = xt[b0 + r < K ? b0 + r : K - 1];
            }
        }
    };
    xfetch(0, x0);
    xfetch(nstage > 1 ? 1 : 0, x1);
    stage(0, st0);
    if (nstage > 1) fetch(1, st0);
    __syncthreads();
    // stage j: `sa` holds stage j + 1 (requested during stage j - 1), `sb` takes stage j + 2, `xr` holds the rows of stage j
    auto body = [&](int j, f32x4 (&sa)[PER], f32x4 (&sb)[PER], f32x4 (&xr)[2], f32x4* accj) __attribute__((always_inline)) {
        const int buf = j & 1;
        const f32x4 xv[2] = {xr[0], xr[1]};
        if (j + 2 < nstage) {
            xfetch(j + 2, xr);
            fetch(j + 2, sb);
        }
        if (tile_ok) {
            const f16x8* c1 = reinterpret_cast<const f16x8*>(sbuf[buf]) + lane;
            const f16x8* w2 = c1 + (4 * KS1 * 512) / 8;
            // first chain
            f32x4 s[2] = {zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f16x8 dh = c1[((t * KS1 + ks) * 2 + 0) * 64], dl = c1[((t * KS1 + ks) * 2 + 1) * 64];
                    s[t] = mfma_h(dl, bh[ks], s[t]);
                    s[t] = mfma_h(dh, bl[ks], s[t]);
                    s[t] = mfma_h(dh, bh[ks], s[t]);
                }
            // t = log2(e) logx - 2 log2(e) (mc D); the stage's shift; e
            float tv[8];
            float tmax = -3.0e38f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool live = 32 * j + 16 * t + 4 * g + r < K;
                    const float v = __builtin_fmaf(xv[t][r], 1.4426950408889634f, __builtin_ldexpf(s[t][r], k1));
                    tv[4 * t + r] = live ? v : -3.0e38f;
                    tmax = __builtin_fmaxf(tmax, tv[4 * t + r]);
                }
            tmax = rows_max4(tmax);
            const float mi = __builtin_ceilf(tmax);
            const float shf = (float)EMAX_LOG2 - mi;
            float ev[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ev[i] = __builtin_amdgcn_exp2f(tv[i] + shf);   // (dead bins: exp2(-huge) = 0)
            f16x8 eh, el;
            split8(ev, eh, el);
            // second chain
            const int k2 = (int)mi - EMAX_LOG2 - LOG2_SE;
#pragma unroll
            for (int tc = 0; tc < NT; ++tc) {
                const f16x8 wh = w2[(tc * 2 + 0) * 64], wlo = w2[(tc * 2 + 1) * 64];
                f32x4 a_ = mfma_h(wlo, eh, zero4);
                a_ = mfma_h(wh, el, a_);
                a_ = mfma_h(wh, eh, a_);
#pragma unroll
                for (int r = 0; r < 4; ++r) accj[tc][r] += __builtin_ldexpf(a_[r], k2);
            }
        }
        if (j + 1 < nstage) stage(buf ^ 1, sa);   // the other buffer: its readers finished before the barrier that ended stage j - 1
        __syncthreads();
    };
#pragma unroll 1
    for (int j = 0; j < nstage; j += 2) {
        body(j, st0, st1, x0, acc);
        if (j + 1 < nstage) body(j + 1, st1, st0, x1, SPLIT ? acc_odd : acc);
    }
    if (SPLIT) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] += acc_odd[t];
    }
    if (!tile_ok || n >= rows_here) return;
    // C/D layout: lane (n, g) register r of tile tc <-> column 16 tc + 4 g + r of frame n
    float* orow = out + (tb + n) * (long)ldo;
#pragma unroll
    for (int tc = 0; tc < NT; ++tc)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 16 * tc + 4 * g + r;
            if (col < N) orow[col] = acc[tc][r];
        }
}

int64_t mcep_resid_h_images_bytes(int K, int M1)
{
    const int ks1 = (M1 + 31) / 32, nt = (2 * M1 - 1 + 15) / 16;
    return (int64_t)((K + 31) / 32) * mrh::stage_halves(ks1, nt) * 2;
}

int mcep_resid_h_prepare(const void* D, int ldd, const void* E, int lde, int K, int M1, void* images, hipStream_t st)
{
    const int ks1 = (M1 + 31) / 32, nt = (2 * M1 - 1 + 15) / 16;
    const long pairs = (long)((K + 31) / 32) * mrh::stage_halves(ks1, nt) / 2;
    hipLaunchKernelGGL(mcep_resid_h_prep_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, (const float*)D, ldd, (const float*)E, lde,
                       K, M1, 2 * M1 - 1, ks1, nt, (_Float16*)images);
    return check_launch("mcep_resid_prepare");
}

int mcep_resid_h_fwd(const void* logx, int64_t F, int K, const void* mc, int M1, const void* images, void* out, int ldo, hipStream_t st)
{
    const int ks1 = (M1 + 31) / 32, nt = (2 * M1 - 1 + 15) / 16, N = 2 * M1 - 1;
    const dim3 grid((unsigned)((F + 63) / 64));
#define DSA_RESID_H(KS, NTV)                                                                                                           \
    hipLaunchKernelGGL((mcep_resid_h_kernel<KS, NTV>), grid, dim3(256), 0, st, (const float*)logx, (long)F, K, (const float*)mc, M1, \
                       (const _Float16*)images, N, (float*)out, ldo)
    if (ks1 == 1) {
        switch (nt) {
            case 1: DSA_RESID_H(1, 1); break;
            case 2: DSA_RESID_H(1, 2); break;
            case 3: DSA_RESID_H(1, 3); break;
            default: DSA_RESID_H(1, 4); break;
        }
    } else {
        switch (nt) {
            case 5: DSA_RESID_H(2, 5); break;
            case 6: DSA_RESID_H(2, 6); break;
            default: DSA_RESID_H(2, 7); break;
        }
    }
#undef DSA_RESID_H
    return check_launch("mcep_resid_h");
}

}  // namespace dsa
