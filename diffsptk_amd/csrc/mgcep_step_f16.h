// Round 5: one Newton step of the mel-generalized cepstral analysis (gamma != 0) in ONE launch, matrix chains as 3-term binary16
// splits (included by mcep_mfma.hip: the block elimination, the split helpers and the quad-layout Toeplitz-plus-Hankel solve live
// there).  mgcep.py:199-230 for fft_length 512, cep_order 24, float32:
//   (re, im) = b1 (Cr, Ci);  X = 1 + g re, Y = g im, D = X^2 + Y^2, pp = x D^(-1/g - 1), qq = pp / D                 (:199-209)
//   pt = pp Pr,  qt = (1 + g)(qq (X^2 - Y^2) Qr + qq 2XY Qi),  r = pp X Rr + pp Y Ri                                   (:212-220)
//   b1 <- b1 + solve(toeplitz(pt) + hankel(qt), r[1:])                                                                 (:226-230)
// Rounds 2-4 ran this as dsa_mgcep_step (60 float32 matrix instructions per 16 bins on the float32 datapath: 95 us per 51 200
// frames) + dsa_thsolve (32 us) with (pt, qt, r) through memory.  Here one wave = 16 frames, eight waves per workgroup (two per
// SIMD), the bins in nine STAGES of 32: the stage's operand images (32 KB, tables.mgcep_step_h_images) are staged through LDS for
// the eight waves (double-buffered, one barrier per stage);
//   first chain   2 tiles x (re, im) x 3 terms on v_mfma_f32_16x16x32_f16, b1 scaled per frame by a power of two;
//   spectra       40 values per lane (5 spectra x 2 tiles x 4), scaled by the STAGE's own power of two (from their maximum over
//                 the frame's 32 bins) and split into binary16 hi + lo: the C/D tiles of the first chain ARE the k-slots of the
//                 second chain's B operands (as in the mel-cepstral kernels);
//   second chain  12 (matrix tile, spectrum) x 3 terms, fresh accumulators per stage, added into float32 sums with the stage's scale;
// then the 24 x 24 system of the wave's 16 frames in the quad layout (the 4 x 4 x 1 block elimination of thsolve_quad24_kernel,
// with its pivoted re-solve for systems that are not positive definite) and the update of b1.  r is written out as well (the gain
// of the last step, mgcep.py:221, reads it).
#pragma once

namespace dsa {

namespace mgh {
using namespace mm;
constexpr int WAVES = 8;
constexpr int STAGES = 9;                       // stages of the images (the backward walks all nine)
constexpr int FSTAGES = 8;                      // the forward's loop: bins 0 .. 255; bin 256 is a handful of float32 multiply-adds (below)
constexpr int NYQ_FLOATS = 240;                 // behind the images: Cr | Ci at bin 256 [24 + 24], rows 256 of Pr [32] Qr [48] Qi [48] Rr [32] Ri [32]
constexpr int C1_HALVES = 2 * 2 * 2 * 512;      // [t][Cr | Ci][hi | lo][64 lane][8]
constexpr int W2_HALVES = 12 * 2 * 512;         // [c][hi | lo][64 lane][8]
constexpr int STAGE_HALVES = C1_HALVES + W2_HALVES;   // 16 384 halves = 32 KB
constexpr int LOG2_SC = 12, LOG2_SW = 20;       // tables.MGCEP_STEP_H_LOG2_SC / _SW
constexpr int VMAX_LOG2 = 13;                   // scaled B operands are below 2^13
// LDS (floats): two stage buffers | per-wave solve records (16 x kTq) | constants [0, 28) zeros, [32, 57) e_24
constexpr int L_STAGE = 0;
constexpr int L_WAVE = 2 * STAGE_HALVES / 2;
constexpr int L_B = L_WAVE + WAVES * 16 * kTq;        // per wave: the 16 frames' coefficients b1 (16 x 24), carried from step to step
constexpr int L_CST = L_B + WAVES * 16 * 24;
constexpr int L_NYQ = L_CST + 64;
constexpr int LDS_FLOATS = L_NYQ + NYQ_FLOATS;
static_assert(LDS_FLOATS * 4 <= 160 * 1024 && L_WAVE % 4 == 0, "the mgcep step's LDS carve-up");
}  // namespace mgh

#ifndef MGH_ABL
#define MGH_ABL 0   // measurement builds only: 1 no staging after stage 0, 2 no barrier per stage (with 1), 4 no solve, 8 no second chain
#endif
__global__ __launch_bounds__(512, 2) void mgcep_step_h_kernel(const float* __restrict__ x, const float* b1, long F, float gamma,
                                                             const _Float16* __restrict__ img, float* b1_out,   // (b1_out may be b1)
                                                             float* __restrict__ r_out, float* __restrict__ pt_out, float* __restrict__ qt_out,   // pt / qt: NULL, or where a graph keeps the system
                                                             int n_steps, float* __restrict__ b1_prev_out)   // n_steps Newton steps per launch; b1_prev: NULL, or the LAST step's input
{
    using namespace mgh;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int nq = lane >> 2, gs = lane & 3;
    if (tid < 64) lds[L_CST + tid] = tid == 32 + 24 ? 1.f : 0.f;
    if (tid < NYQ_FLOATS) lds[L_NYQ + tid] = reinterpret_cast<const float*>(img + STAGES * STAGE_HALVES)[tid];
    const float* nyq = lds + L_NYQ;
    float* wl = lds + L_WAVE + wave * 16 * kTq;
    float* bsl = lds + L_B + wave * 16 * 24;
    const float* cst = lds + L_CST;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float ex = -1.f / gamma - 1.f;
    const float og = 1.f + gamma;
    const long ntiles = (F + 15) / 16;
    const long nrounds = (ntiles + (long)gridDim.x * WAVES - 1) / ((long)gridDim.x * WAVES);
    // staging: a stage = 2048 16-byte pieces, 4 per thread
    const f32x4* img4 = reinterpret_cast<const f32x4*>(img);
    f32x4 st[4];
    auto fetch = [&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) st[q] = img4[(long)j * (STAGE_HALVES / 8) + tid + 512 * q];
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        f32x4* d = reinterpret_cast<f32x4*>(lds + L_STAGE) + buf * (STAGE_HALVES / 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) d[tid + 512 * q] = st[q];
    };
    for (long round = 0; round < nrounds; ++round) {
        // wave-major within a round: a partly filled last round gives every workgroup its share on its FIRST waves (one per SIMD),
        // instead of filling some workgroups with two waves per SIMD and leaving the others idle
        const long tile_raw = (round * WAVES + wave) * gridDim.x + blockIdx.x;   // uniform
        const bool tile_ok = tile_raw < ntiles;
        const long tile = tile_ok ? tile_raw : ntiles - 1;
        const long t16 = tile * 16;
        const int rows_here = (int)((F - t16 < 16) ? F - t16 : 16);
        const bool f_ok = tile_ok && n < rows_here;
        const int rn = n < rows_here ? n : rows_here - 1;
        const float* xt = x + t16 * 257;
        const float* bt = b1 + t16 * 24;
        // the tile's coefficients into the wave's LDS slots (rows past the end: the last row again; their systems are identities)
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int idx = lane + 64 * it, fr = idx / 24, k = idx - fr * 24;
            bsl[idx] = bt[(fr < rows_here ? fr : rows_here - 1) * 24 + k];
        }
        __builtin_amdgcn_wave_barrier();
        for (int step = 0; step < n_steps; ++step) {
        const bool last = step + 1 == n_steps;
        fetch(0);
        // B operand of the first chain: b1[8 g + i] of this lane's frame, scaled per frame
        float bv[8];
        float bmax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bv[i] = g < 3 ? bsl[n * 24 + 8 * g + i] : 0.f;
            bmax = __builtin_fmaxf(bmax, __builtin_fabsf(bv[i]));
        }
        if (last && b1_prev_out && f_ok && g < 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) b1_prev_out[(t16 + n) * 24 + 8 * g + i] = bv[i];
        }
        bmax = rows_max4(bmax);
        const int s_b = 12 - __builtin_amdgcn_frexp_expf(bmax);
        f16x8 bh, bl;
        {
            float ms[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ms[i] = __builtin_ldexpf(bv[i], s_b);
            split8(ms, bh, bl);
        }
        const int k1 = -s_b - LOG2_SC;      // (re, im) = 2^k1 x the first chain's accumulators
        f32x4 acc[7];                       // pt 0-1 | qt 2-4 | r 5-6: C/D layout, lane (n, g) register r <-> column 16 t + 4 g + r
#pragma unroll
        for (int t = 0; t < 7; ++t) acc[t] = zero4;
        // the lane's spectrum values of a stage: bins 32 j + 16 t + 4 g + r (only bin 256 exists in the last stage); requested a whole
        // stage ahead (a request that goes to memory takes thousands of cycles, and loads return in order)
        f32x4 xn[2] = {zero4, zero4};
        auto xfetch = [&](int j) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 2; ++t) xn[t] = *reinterpret_cast<const f32x4_u4*>(xt + rn * 257 + 32 * j + 16 * t + 4 * g);
        };
        xfetch(0);
        const float x256 = xt[rn * 257 + 256];
        stage(0);
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < FSTAGES; ++j) {
            const int buf = (MGH_ABL & 1) ? 0 : (j & 1);
            const f32x4 xv[2] = {xn[0], xn[1]};
            if (j + 1 < FSTAGES) {
                xfetch(j + 1);
                if (!(MGH_ABL & 1)) fetch(j + 1);
            }
            const f16x8* c1 = reinterpret_cast<const f16x8*>(lds + L_STAGE) + buf * (STAGE_HALVES / 8) + lane;
            const f16x8* w2 = c1 + C1_HALVES / 8;
            if (tile_ok) {
            // first chain
            f32x4 re[2], im[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f16x8 crh = c1[((t * 2 + 0) * 2 + 0) * 64], crl = c1[((t * 2 + 0) * 2 + 1) * 64];
                const f16x8 cih = c1[((t * 2 + 1) * 2 + 0) * 64], cil = c1[((t * 2 + 1) * 2 + 1) * 64];
                re[t] = mfma_h(crl, bh, zero4);
                im[t] = mfma_h(cil, bh, zero4);
                re[t] = mfma_h(crh, bl, re[t]);
                im[t] = mfma_h(cih, bl, im[t]);
                re[t] = mfma_h(crh, bh, re[t]);
                im[t] = mfma_h(cih, bh, im[t]);
            }
            // the five spectra
            float s[5][8];
            float smax = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float X = __builtin_fmaf(gamma, __builtin_ldexpf(re[t][r], k1), 1.f), Y = gamma * __builtin_ldexpf(im[t][r], k1);
                    const float XX = X * X, YY = Y * Y, D = XX + YY;
                    const float dp = __builtin_amdgcn_exp2f(ex * __builtin_amdgcn_logf(D));   // D > 0; 1 ulp each (as dsa_mgcep_step)
                    const float pp = xv[t][r] * dp;
                    const float qq = pp * __builtin_amdgcn_rcpf(D);
                    const int i = 4 * t + r;
                    s[0][i] = pp;
                    s[1][i] = qq * (XX - YY);
                    s[2][i] = qq * (2.f * X * Y);
                    s[3][i] = pp * X;
                    s[4][i] = pp * Y;
                    smax = __builtin_fmaxf(smax, __builtin_fmaxf(__builtin_fabsf(s[0][i]), __builtin_fabsf(s[1][i])));
                    smax = __builtin_fmaxf(smax, __builtin_fmaxf(__builtin_fabsf(s[2][i]), __builtin_fabsf(s[3][i])));
                    smax = __builtin_fmaxf(smax, __builtin_fabsf(s[4][i]));
                }
            smax = rows_max4(smax);
            const int s_g = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(smax);
            // second chain: k-slot (g, i = 4 t + r) <-> bin 32 j + 16 t + 4 g + r: the scaled values above ARE the B operands
            f32x4 ag[7];
#pragma unroll
            for (int t = 0; t < 7; ++t) ag[t] = zero4;
#pragma unroll
            for (int in = 0; in < ((MGH_ABL & 8) ? 1 : 5); ++in) {
                float ms[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) ms[i] = __builtin_ldexpf(s[in][i], s_g);
                f16x8 sh, sl;
                split8(ms, sh, sl);
                const int c0 = in == 0 ? 0 : (in == 1 ? 2 : (in == 2 ? 5 : (in == 3 ? 8 : 10)));
                const int nt = (in == 1 || in == 2) ? 3 : 2;
                const int t0 = in == 0 ? 0 : (in < 3 ? 2 : 5);
#pragma unroll
                for (int tc = 0; tc < nt; ++tc) {
                    const f16x8 wh = w2[((c0 + tc) * 2 + 0) * 64], wlo = w2[((c0 + tc) * 2 + 1) * 64];
                    ag[t0 + tc] = mfma_h(wlo, sh, ag[t0 + tc]);
                    ag[t0 + tc] = mfma_h(wh, sl, ag[t0 + tc]);
                    ag[t0 + tc] = mfma_h(wh, sh, ag[t0 + tc]);
                }
            }
            const int k2 = -s_g - LOG2_SW;
#pragma unroll
            for (int t = 0; t < 7; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] += __builtin_ldexpf(ag[t][r], k2);
            }
            if (j + 1 < FSTAGES && !(MGH_ABL & 1)) stage(buf ^ 1);   // the other buffer: its readers finished before the barrier that ended stage j - 1
            if (!(MGH_ABL & 2)) __syncthreads();
        }
        if (tile_ok) {
        // ---------------- bin 256 (a ninth stage of 32 bins for one bin cost a ninth of the loop): float32 multiply-adds ----------------
        {
            float re = 0.f, im = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = g < 3 ? 8 * g + i : 0;
                const float bc = g < 3 ? bsl[n * 24 + c] : 0.f;   // (re-read from the wave's slots: eight registers less across the stages)
                re = __builtin_fmaf(bc, nyq[c], re);
                im = __builtin_fmaf(bc, nyq[24 + c], im);
            }
            re = rows_sum4(re);
            im = rows_sum4(im);
            const float X = __builtin_fmaf(gamma, re, 1.f), Y = gamma * im;
            const float XX = X * X, YY = Y * Y, D = XX + YY;
            const float dp = __builtin_amdgcn_exp2f(ex * __builtin_amdgcn_logf(D));
            const float pp = x256 * dp;
            const float qq = pp * __builtin_amdgcn_rcpf(D);
            const float s1 = qq * (XX - YY), s2 = qq * (2.f * X * Y), s3 = pp * X, s4 = pp * Y;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 16 * t + 4 * g + r;   // < 48
                    if (t < 2) {
                        acc[t][r] = __builtin_fmaf(pp, nyq[48 + col], acc[t][r]);
                        acc[5 + t][r] = __builtin_fmaf(s3, nyq[176 + col], __builtin_fmaf(s4, nyq[208 + col], acc[5 + t][r]));
                    }
                    acc[2 + t][r] = __builtin_fmaf(s1, nyq[80 + col], __builtin_fmaf(s2, nyq[128 + col], acc[2 + t][r]));
                }
        }
        // ---------------- the wave's 16 systems: windows (q | mirrored p | r[1:]) in the quad-layout solve's record ----------------
        for (int e = lane; e < 16 * kTq; e += 64) wl[e] = 0.f;
        __builtin_amdgcn_wave_barrier();
        {
            float* rec = wl + n * kTq;
            const bool row_ok = n < rows_here && tile_ok;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 16 * t + 4 * g + r;
                    // a missing system: the identity (p = e_0), right-hand side 0
                    const float pv = row_ok ? acc[t][r] : (col == 0 ? 1.f : 0.f);
                    if (col < 24) { rec[52 + 27 + col] = pv; rec[52 + 27 - col] = pv; }
                    if (last && pt_out && f_ok && col < 24) pt_out[(t16 + n) * 24 + col] = pv;
                    const float rv = row_ok ? acc[5 + t][r] : 0.f;
                    if (col >= 1 && col < 25) rec[104 + col - 1] = rv;
                    if (last && f_ok && col < 25) r_out[(t16 + n) * 25 + col] = rv;
                }
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 16 * t + 4 * g + r;
                    if (col < 47) rec[col] = row_ok ? og * acc[2 + t][r] : 0.f;
                    if (last && qt_out && f_ok && col < 47) qt_out[(t16 + n) * 47 + col] = og * acc[2 + t][r];
                }
        }
        __builtin_amdgcn_wave_barrier();
        if (!(MGH_ABL & 4)) {
            const GroupMask gq = make_group_mask(gs);
            const float* rt_q = wl + nq * kTq;
            const float* rr_q = rt_q + 52;
            float xq[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, keep_if(gq.m[1], -1.f)};
            bool bad = false;
            {
                f32x4 a[blk::NBLK];
                float ninvs[M1];
                int gsv = gs;
                asm volatile("" : "+v"(gsv));
                const float* zr = cst;
                const float* pa6 = gsv == 0 ? cst + 32 : (gsv == 1 ? rt_q + 104 : zr);   // column 24: e_24 | right-hand side | 0
                blk_build_rows<0>(a, rt_q, rr_q, pa6, zr, gs);
                blk_elim_all(a, gq, ninvs, std::make_integer_sequence<int, M1>{});
                blk_backsub_all(a, xq, gq, ninvs, std::make_integer_sequence<int, blk::NG>{});
                // no pivoting: sound for the positive definite systems of the analysis; a pivot that is not positive marks the system
                // for the pivoted re-solve below (as thsolve_quad24_kernel)
#pragma unroll
                for (int k = 0; k < M1 - 1; ++k) bad |= !(ninvs[k] < 0.f && ninvs[k] > -3.0e38f);
            }
            const long fq = t16 + nq;
            const bool q_ok = tile_ok && nq < rows_here;
            if (!bad) {   // (rows past the end: identities, xq = 0)
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const float bn = bsl[nq * 24 + gs + 4 * c] + xq[c];
                    bsl[nq * 24 + gs + 4 * c] = bn;
                    if (last && q_ok) b1_out[fq * 24 + gs + 4 * c] = bn;
                }
            }
            unsigned long long marked = __ballot(bad && gs == 0 && q_ok);
            while (marked) {   // uniform; normally empty
                const int bl_ = __builtin_ctzll(marked);
                marked &= marked - 1;
                const int sy = bl_ >> 2;
                const float* qs2 = wl + sy * kTq;               // q window
                const float* ps2 = qs2 + 52 + 27;               // p[d] at the centre of the mirrored window
                const float rhs = lane < 24 ? qs2[104 + lane] : 0.f;
                int col;
                float sol;
                th_solve_reg<float, 24>(ps2, qs2, rhs, 24, lane, col, sol);
                const long fs = t16 + sy;
                if (lane < 24) {
                    const float bn = bsl[sy * 24 + col] + sol;
                    bsl[sy * 24 + col] = bn;
                    if (last) b1_out[fs * 24 + col] = bn;
                }
            }
        }
        }
        __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Backward of (pt, qt, r) = mgcep_step(x, b1) (mgcep.py:199-220) in ONE launch on the binary16 matrix pipe -- the counterpart of
// mgcep_step_bwd_kernel (csrc/mgc.hip: 144 float32 matrix instructions per 32 bins and 16 frames; here 66 binary16 ones).  Same
// tiling as the forward above: a wave = 16 frames, nine stages of 32 bins, the stage's images (44 KB, tables.mgcep_step_bwd_h_images)
// staged through LDS for the workgroup's eight waves.  Per stage: (re, im) recomputed (first chain), the cotangents of the five
// spectra at the stage's bins as products of the bin-row matrices with the cotangent vectors (gpt | (1 + gamma) gqt | gr: the B
// operands, scaled per frame and split once per launch), the element-wise chain rule (as mgcep_step_bwd_kernel), gx written, and
// the cotangent of (re, im) -- scaled by the STAGE's own power of two and split -- accumulated into gb1 = gamma (gX Cr^T + gY Ci^T).
// ---------------------------------------------------------------------------------------------------------------------------
namespace mgh {
constexpr int BW_WT = C1_HALVES;                       // [2 t][7 ks][hi | lo][64][8]
constexpr int BW_CT = BW_WT + 2 * 7 * 2 * 512;         // [Cr | Ci][2 tc][hi | lo][64][8]
constexpr int BW_STAGE_HALVES = BW_CT + 2 * 2 * 2 * 512;   // 22 528 halves = 44 KB
static_assert(2 * BW_STAGE_HALVES * 2 <= 160 * 1024, "the mgcep step backward's two stage buffers");
}  // namespace mgh

__global__ __launch_bounds__(512, 2) DSA_PK_TARGET void mgcep_step_bwd_h_kernel(const float* __restrict__ x, const float* __restrict__ b1,
                                                                 const float* __restrict__ gpt, const float* __restrict__ gqt,
                                                                 const float* __restrict__ grr, long F, float gamma,
                                                                 const _Float16* __restrict__ img, const float* __restrict__ gx_in,
                                                                 float* __restrict__ gx, float* __restrict__ gb1)
{
    using namespace mgh;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PIECES = BW_STAGE_HALVES / 8;            // 2816 16-byte pieces per stage
    constexpr int PER = (PIECES + 511) / 512;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float ex = -1.f / gamma - 1.f;
    const float og = 1.f + gamma;
    const long ntiles = (F + 15) / 16;
    const long nrounds = (ntiles + (long)gridDim.x * WAVES - 1) / ((long)gridDim.x * WAVES);
    const f32x4* img4 = reinterpret_cast<const f32x4*>(img);
    f32x4 st[PER];
    auto fetch = [&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int p = tid + 512 * q;
            if (p < PIECES) st[q] = img4[(long)j * PIECES + p];
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        f32x4* d = reinterpret_cast<f32x4*>(lds) + buf * PIECES;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int p = tid + 512 * q;
            if (p < PIECES) d[p] = st[q];
        }
    };
    // eight values of one row as a scaled binary16 pair: returns the scale's exponent (value = 2^-s x operand)
    auto operand = [&](const float (&v)[8], f16x8& hi, f16x8& lo) __attribute__((always_inline)) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) m = __builtin_fmaxf(m, __builtin_fabsf(v[i]));
        m = rows_max4(m);
        const int sx = 12 - __builtin_amdgcn_frexp_expf(m);
        float ms[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ms[i] = __builtin_ldexpf(v[i], sx);
        split8(ms, hi, lo);
        return sx;
    };
    for (long round = 0; round < nrounds; ++round) {
        const long tile_raw = (round * WAVES + wave) * gridDim.x + blockIdx.x;   // uniform; wave-major within a round (see the forward)
        const bool tile_ok = tile_raw < ntiles;
        const long tile = tile_ok ? tile_raw : ntiles - 1;
        const long t16 = tile * 16;
        const int rows_here = (int)((F - t16 < 16) ? F - t16 : 16);
        const bool f_ok = tile_ok && n < rows_here;
        const int rn = n < rows_here ? n : rows_here - 1;
        const float* xt = x + t16 * 257 + rn * 257;
        fetch(0);
        // B operands held for the whole tile: lane (n, g) holds entries 8 g + i of its frame's vectors
        f16x8 bh, bl, ph, pl, qh[2], ql[2], rh, rl;
        int k1, kp, kq, kr;
        {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = g < 3 ? b1[(t16 + rn) * 24 + 8 * g + i] : 0.f;
            k1 = -operand(v, bh, bl) - LOG2_SC;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = g < 3 ? gpt[(t16 + rn) * 24 + 8 * g + i] : 0.f;
            kp = -operand(v, ph, pl) - LOG2_SW;
            float w0[8], w1[8];
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                w0[i] = og * gqt[(t16 + rn) * 47 + 8 * g + i];
                w1[i] = 32 + 8 * g + i < 47 ? og * gqt[(t16 + rn) * 47 + 32 + 8 * g + i] : 0.f;
                m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(w0[i]), __builtin_fabsf(w1[i])));
            }
            m = rows_max4(m);
            const int sq = 12 - __builtin_amdgcn_frexp_expf(m);
#pragma unroll
            for (int i = 0; i < 8; ++i) { w0[i] = __builtin_ldexpf(w0[i], sq); w1[i] = __builtin_ldexpf(w1[i], sq); }
            split8(w0, qh[0], ql[0]);
            split8(w1, qh[1], ql[1]);
            kq = -sq - LOG2_SW;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 8 * g + i < 25 ? grr[(t16 + rn) * 25 + 8 * g + i] : 0.f;
            kr = -operand(v, rh, rl) - LOG2_SW;
        }
        f32x4 accb[2] = {zero4, zero4};   // gb1: C/D layout, lane (n, g) register r <-> coefficient 16 tc + 4 g + r
        f32x4 xn[2] = {zero4, zero4};
        auto xfetch = [&](int j) __attribute__((always_inline)) {
            if (j < 8) {
#pragma unroll
                for (int t = 0; t < 2; ++t) xn[t] = *reinterpret_cast<const f32x4_u4*>(xt + 32 * j + 16 * t + 4 * g);
            } else {
                xn[0] = zero4;
                xn[1] = zero4;
                if (g == 0) xn[0][0] = xt[256];
            }
        };
        xfetch(0);
        stage(0);
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < STAGES; ++j) {
            const int buf = j & 1;
            const f32x4 xv[2] = {xn[0], xn[1]};
            if (j + 1 < STAGES) {
                xfetch(j + 1);
                fetch(j + 1);
            }
            if (tile_ok) {
                const f16x8* c1 = reinterpret_cast<const f16x8*>(lds) + buf * PIECES + lane;
                const f16x8* wt = c1 + BW_WT / 8;
                const f16x8* ct = c1 + BW_CT / 8;
                float gre[8], gim[8];
                float gmax = 0.f;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    // (re, im) of the tile
                    const f16x8 crh = c1[((t * 2 + 0) * 2 + 0) * 64], crl = c1[((t * 2 + 0) * 2 + 1) * 64];
                    const f16x8 cih = c1[((t * 2 + 1) * 2 + 0) * 64], cil = c1[((t * 2 + 1) * 2 + 1) * 64];
                    f32x4 re = mfma_h(crl, bh, zero4), im = mfma_h(cil, bh, zero4);
                    re = mfma_h(crh, bl, re);
                    im = mfma_h(cih, bl, im);
                    re = mfma_h(crh, bh, re);
                    im = mfma_h(cih, bh, im);
                    // cotangents of the five spectra at the tile's bins
                    auto prod = [&](int ks, const f16x8& vh, const f16x8& vl, f32x4 a_) __attribute__((always_inline)) {
                        const f16x8 wh = wt[((t * 7 + ks) * 2 + 0) * 64], wlo = wt[((t * 7 + ks) * 2 + 1) * 64];
                        a_ = mfma_h(wlo, vh, a_);
                        a_ = mfma_h(wh, vl, a_);
                        return mfma_h(wh, vh, a_);
                    };
                    const f32x4 g0 = prod(0, ph, pl, zero4);
                    const f32x4 g1 = prod(2, qh[1], ql[1], prod(1, qh[0], ql[0], zero4));
                    const f32x4 g2 = prod(4, qh[1], ql[1], prod(3, qh[0], ql[0], zero4));
                    const f32x4 g3 = prod(5, rh, rl, zero4);
                    const f32x4 g4 = prod(6, rh, rl, zero4);
                    f32x4 gxo;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float gs0 = __builtin_ldexpf(g0[r], kp), gs1 = __builtin_ldexpf(g1[r], kq), gs2 = __builtin_ldexpf(g2[r], kq);
                        const float gs3 = __builtin_ldexpf(g3[r], kr), gs4 = __builtin_ldexpf(g4[r], kr);
                        const float X = __builtin_fmaf(gamma, __builtin_ldexpf(re[r], k1), 1.f), Y = gamma * __builtin_ldexpf(im[r], k1);
                        const float XX = X * X, YY = Y * Y, D = XX + YY;
                        const float dp = __builtin_amdgcn_exp2f(ex * __builtin_amdgcn_logf(D));
                        const float pp = xv[t][r] * dp;
                        const float rD = __builtin_amdgcn_rcpf(D);
                        const float qq = pp * rD;
                        const float g_qq = gs1 * (XX - YY) + gs2 * (2.f * X * Y);
                        const float g_pp = gs0 + gs3 * X + gs4 * Y + g_qq * rD;
                        const float g_D = (g_pp * ex * pp - g_qq * qq) * rD;
                        const float gX = 2.f * qq * (gs1 * X + gs2 * Y) + gs3 * pp + 2.f * X * g_D;
                        const float gY = 2.f * qq * (gs2 * X - gs1 * Y) + gs4 * pp + 2.f * Y * g_D;
                        const bool live = j < 8 || (t == 0 && r == 0 && g == 0);   // (the last stage holds bin 256 only)
                        gre[4 * t + r] = live ? gamma * gX : 0.f;
                        gim[4 * t + r] = live ? gamma * gY : 0.f;
                        gxo[r] = g_pp * dp;
                        gmax = __builtin_fmaxf(gmax, __builtin_fmaxf(__builtin_fabsf(gre[4 * t + r]), __builtin_fabsf(gim[4 * t + r])));
                    }
                    if (f_ok) {
                        if (j < 8) {
                            float* dst = gx + (t16 + n) * 257 + 32 * j + 16 * t + 4 * g;
                            if (gx_in) gxo += *reinterpret_cast<const f32x4_u4*>(gx_in + (t16 + n) * 257 + 32 * j + 16 * t + 4 * g);
                            *reinterpret_cast<f32x4_u4*>(dst) = gxo;
                        } else if (t == 0 && g == 0) {
                            gx[(t16 + n) * 257 + 256] = gxo[0] + (gx_in ? gx_in[(t16 + n) * 257 + 256] : 0.f);
                        }
                    }
                }
                // gb1 += (Cr | Ci) with coefficients as rows x the stage's (gre | gim), scaled by the stage's power of two
                gmax = rows_max4(gmax);
                const int s_g = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(gmax);
                float ms[8];
                f16x8 reh, rel, imh, iml;
#pragma unroll
                for (int i = 0; i < 8; ++i) ms[i] = __builtin_ldexpf(gre[i], s_g);
                split8(ms, reh, rel);
#pragma unroll
                for (int i = 0; i < 8; ++i) ms[i] = __builtin_ldexpf(gim[i], s_g);
                split8(ms, imh, iml);
                const int k3 = -s_g - LOG2_SC;
#pragma unroll
                for (int tc = 0; tc < 2; ++tc) {
                    const f16x8 ah = ct[((0 * 2 + tc) * 2 + 0) * 64], al = ct[((0 * 2 + tc) * 2 + 1) * 64];
                    const f16x8 dh = ct[((1 * 2 + tc) * 2 + 0) * 64], dl = ct[((1 * 2 + tc) * 2 + 1) * 64];
                    f32x4 a_ = mfma_h(al, reh, zero4);
                    a_ = mfma_h(ah, rel, a_);
                    a_ = mfma_h(ah, reh, a_);
                    a_ = mfma_h(dl, imh, a_);
                    a_ = mfma_h(dh, iml, a_);
                    a_ = mfma_h(dh, imh, a_);
#pragma unroll
                    for (int r = 0; r < 4; ++r) accb[tc][r] += __builtin_ldexpf(a_[r], k3);
                }
            }
            if (j + 1 < STAGES) stage(buf ^ 1);
            __syncthreads();
        }
        if (f_ok) {
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = 16 * tc + 4 * g + r;
                    if (col < 24) gb1[(t16 + n) * 24 + col] = accb[tc][r];
                }
        }
    }
}

int mgcep_step_bwd_h(const void* x, const void* b1, const void* gpt, const void* gqt, const void* gr, int64_t F, double gamma,
                     const void* images, const void* gx_in, void* gx, void* gb1, hipStream_t st)
{
    const int lds_bytes = 2 * mgh::BW_STAGE_HALVES * 2;
    static std::atomic<uint64_t> attr{0};
    if (!ensure_dynamic_lds((const void*)mgcep_step_bwd_h_kernel, lds_bytes, attr))
        return fail(DSA_ERR_LAUNCH, "mgcep_step_bwd_h: cannot reserve the LDS stage buffers%s");
    const long ntiles = (long)((F + 15) / 16);
    long blocks = (ntiles + mgh::WAVES - 1) / mgh::WAVES;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(mgcep_step_bwd_h_kernel, dim3((unsigned)blocks), dim3(mgh::WAVES * 64), lds_bytes, st, (const float*)x, (const float*)b1,
                       (const float*)gpt, (const float*)gqt, (const float*)gr, (long)F, (float)gamma, (const _Float16*)images,
                       (const float*)gx_in, (float*)gx, (float*)gb1);
    return check_launch("mgcep_step_bwd_h");
}

int mgcep_step_solve_fwd(const void* x, const void* b1, int64_t F, double gamma, const void* images, void* b1_out, void* r_out, hipStream_t st,
                         void* pt_out, void* qt_out, int n_steps, void* b1_prev_out)
{
    const int lds_bytes = mgh::LDS_FLOATS * 4;
    static std::atomic<uint64_t> attr{0};
    if (!ensure_dynamic_lds((const void*)mgcep_step_h_kernel, lds_bytes, attr))
        return fail(DSA_ERR_LAUNCH, "mgcep_step_solve: cannot reserve the LDS stage buffers%s");
    const long ntiles = (long)((F + 15) / 16);
    long blocks = (ntiles + mgh::WAVES - 1) / mgh::WAVES;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(mgcep_step_h_kernel, dim3((unsigned)blocks), dim3(mgh::WAVES * 64), lds_bytes, st, (const float*)x, (const float*)b1,
                       (long)F, (float)gamma, (const _Float16*)images, (float*)b1_out, (float*)r_out, (float*)pt_out, (float*)qt_out, n_steps, (float*)b1_prev_out);
    return check_launch("mgcep_step_solve");
}

}  // namespace dsa
