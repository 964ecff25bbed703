// Round 6 (0.2.2): glogx of the 48 kHz mel-cepstral analysis in ONE pass over the bins (included by mcep_mfma.hip after
// mcep_resid_bwd_f16.h, whose images, scales and arithmetic it shares; autograd of mcep.py:210-215 summed over the Newton steps).
//
// The reverse sweep's step (mcep_resid_bwd_h_kernel) forms z_s = (grt_s E^T) * exp(logx - 2 mc_s D) over all bins and needs it twice:
// contracted with D for the next cotangent (gmc_s = -2 z_s D^T: must happen inside the sweep, the steps depend on each other through
// it) and ADDED into glogx.  The second use made every step read and write the whole (F, K) gradient -- 840 of the step's 1 260 MB at
// 102 400 frames x 1 025 bins, a launch bound by exactly those bytes (0.50 ms per step).  But the sum over the steps depends on the
// sweep only through two short vectors per frame and step, the iterate mc_s (the forward keeps it) and the cotangent grt_s (now kept
// too: 4 (2 M + 1) bytes per frame and step):
//     glogx[f, k] = sum_s (sum_j grt_s[f, j] E[k, j]) * exp(logx[f, k] - 2 sum_c mc_s[f, c] D[c, k])
// so it is formed here AFTER the sweep, bin stage by bin stage with all steps in the inner loop: logx read once, glogx written once
// (840 MB instead of 12.6 GB per analysis at that size), and the sweep's launches run with GX = false.
// Layout: a workgroup = ONE tile of 16 frames, EIGHT waves (two per SIMD).  Prologue: the B operands of every step -- mc_s and grt_s of
// the 16 frames, scaled per frame by a power of two and split into binary16 pieces exactly as mcep_resid_bwd_h_kernel does -- go to LDS
// once (12 KB per step: the reason for one tile per workgroup and one workgroup per CU).  Then the 16-bin half stages (stage j, tile t)
// are dealt round-robin to the waves: the unit's operand images straight from L2 into registers (this wave is their only reader on the
// CU), and per step the two chains, the exponential and z_s -- the same instructions on the same values as the sweep's kernel --
// summed over the steps IN THE SWEEP'S ORDER (last step first), so the result equals the in-place accumulation bit for bit.
// (First cut: four waves, a whole stage per wave -- one wave per SIMD with nothing to hide its dependent chains behind, and 33 stages
// over 4 waves is 9 against 8.25: 1.45 ms per 102 400 frames at 1 025 bins; see profiles/r06_48khz_gradient_one_node.txt.)
#pragma once

namespace dsa {

namespace mgx {
using namespace mrb;
constexpr int nop(int ks1, int kse) { return 2 * ks1 + 2 * kse; }                     // f16x8 operands per lane and step
constexpr int step_bytes(int ks1, int kse) { return nop(ks1, kse) * 1024 + 128; }     // + [16] k1, [16] ke
constexpr int kMaxLds = 156 * 1024;
}  // namespace mgx

template <int KS1, int KSE, int NT3>
__global__ __launch_bounds__(512, 1) DSA_PK_TARGET void mcep_glogx_h_kernel(const float* __restrict__ logx, long F, int K, const float* __restrict__ mcs,
                                                                             int M1, const float* __restrict__ grts, int N, int n_iter,
                                                                             const _Float16* __restrict__ img, float* __restrict__ glogx)
{
    using namespace mgx;
    constexpr int GW = 8;                                // waves per workgroup
    constexpr int SH = stage_halves_b(KS1, KSE, NT3);
    constexpr int PIECES = SH / 8;                       // 16-byte pieces per stage of the images
    constexpr int NOP = nop(KS1, KSE);
    constexpr int STEP_F = step_bytes(KS1, KSE) / 4;     // floats per step in LDS
    constexpr int NI = 2 * KS1 + 2 * KSE;                // image pieces per lane and unit: first chain (hi, lo) x KS1, ebar chain (hi, lo) x KSE
    extern __shared__ __attribute__((aligned(16))) float smem_gx[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int nunit = (K + 15) / 16;                      // 16-bin units: (stage j = u / 2, tile t = u % 2)
    const long tb = (long)blockIdx.x * 16;                // the workgroup's tile (the grid covers ceil(F / 16) tiles)
    const int rows_here = (int)((F - tb < 16) ? F - tb : 16);
    const int rn = n < rows_here ? n : rows_here - 1;
    const bool row_ok = n < rows_here;
    // ---------------- prologue: the steps' B operands, split as the sweep's kernel splits them, into LDS ----------------
    for (int s = wave; s < n_iter; s += GW) {
        const float* mc = mcs + (long)s * F * M1;
        const float* grt = grts + (long)s * F * N;
        f16x8* ops = reinterpret_cast<f16x8*>(smem_gx + s * STEP_F) + lane;
        int* ks_ = reinterpret_cast<int*>(smem_gx + s * STEP_F + NOP * 256);
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
            f16x8 h_, l_;
            split8(ms, h_, l_);
            ops[(2 * ks + 0) * 64] = h_;
            ops[(2 * ks + 1) * 64] = l_;
        }
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
            f16x8 h_, l_;
            split8(ms, h_, l_);
            ops[(2 * KS1 + 2 * ks + 0) * 64] = h_;
            ops[(2 * KS1 + 2 * ks + 1) * 64] = l_;
        }
        if (g == 0) {
            ks_[n] = -s_b - LOG2_SD;        // k1: the first chain's accumulators x 2^k1 = (-2 log2(e) D)^T mc
            ks_[16 + n] = -s_g - LOG2_SE;   // ke: the ebar chain's accumulators x 2^ke = E grt^T
        }
    }
    __syncthreads();
    // ---------------- this wave's units: images from L2, all steps in the inner loop ----------------
    const float* xt = logx + tb * (long)K + (long)rn * K;
    float* gxt = glogx + tb * (long)K + (long)rn * K;
    const f16x8* img8 = reinterpret_cast<const f16x8*>(img) + lane;
    constexpr int OFF2 = (4 * KS1 * 512) / 8;             // (f16x8 units) where the ebar chain's images of a stage start
    f16x8 ia[NI], ib[NI];                                 // two register sets: the unit in use, the wave's next unit
    auto ifetch = [&](int u, f16x8 (&iv)[NI]) __attribute__((always_inline)) {
        const int uu = u < nunit ? u : nunit - 1;         // (clamped: unconditional loads, exact wait counts)
        const int j = uu >> 1, t = uu & 1;
        const f16x8* p = img8 + (long)j * PIECES;
#pragma unroll
        for (int q = 0; q < 2 * KS1; ++q) iv[q] = p[(t * KS1 * 2 + q) * 64];
#pragma unroll
        for (int q = 0; q < 2 * KSE; ++q) iv[2 * KS1 + q] = p[OFF2 + (t * KSE * 2 + q) * 64];
    };
    f32x4 xa, xb;
    auto xfetch = [&](int u, f32x4& xr) __attribute__((always_inline)) {
        const int uu = u < nunit ? u : nunit - 1;
        const int b0 = 16 * uu + 4 * g;
        if (b0 + 3 < K) {
            xr = *reinterpret_cast<const f32x4_u4*>(xt + b0);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) xr[r] = xt[b0 + r < K ? b0 + r : K - 1];
        }
    };
    auto body = [&](int u, const f16x8 (&iv)[NI], const f32x4& xv) __attribute__((always_inline)) {
        f32x4 zsum = zero4;
#pragma unroll 1
        for (int s = n_iter - 1; s >= 0; --s) {           // the sweep's order: last step first
            const f16x8* ops = reinterpret_cast<const f16x8*>(smem_gx + s * STEP_F) + lane;
            const int* ks_ = reinterpret_cast<const int*>(smem_gx + s * STEP_F + NOP * 256);
            f16x8 bh[KS1], bl[KS1], rh[KSE], rl[KSE];
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) { bh[ks] = ops[(2 * ks + 0) * 64]; bl[ks] = ops[(2 * ks + 1) * 64]; }
#pragma unroll
            for (int ks = 0; ks < KSE; ++ks) { rh[ks] = ops[(2 * KS1 + 2 * ks + 0) * 64]; rl[ks] = ops[(2 * KS1 + 2 * ks + 1) * 64]; }
            const int k1 = ks_[n], ke = ks_[16 + n];
            f32x4 sc = zero4, eb = zero4;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const f16x8 dh = iv[ks * 2 + 0], dl = iv[ks * 2 + 1];
                sc = mfma_h(dl, bh[ks], sc);
                sc = mfma_h(dh, bl[ks], sc);
                sc = mfma_h(dh, bh[ks], sc);
            }
#pragma unroll
            for (int ks = 0; ks < KSE; ++ks) {
                const f16x8 eh_ = iv[2 * KS1 + ks * 2 + 0], el_ = iv[2 * KS1 + ks * 2 + 1];
                eb = mfma_h(el_, rh[ks], eb);
                eb = mfma_h(eh_, rl[ks], eb);
                eb = mfma_h(eh_, rh[ks], eb);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool live = 16 * u + 4 * g + r < K;
                const float tv = __builtin_fmaf(xv[r], 1.4426950408889634f, __builtin_ldexpf(sc[r], k1));
                const float e_ = live ? __builtin_amdgcn_exp2f(tv) : 0.f;
                zsum[r] += __builtin_ldexpf(eb[r] * e_, ke);
            }
        }
        if (row_ok) {
            const int b0 = 16 * u + 4 * g;
            if (b0 + 3 < K) {
                *reinterpret_cast<f32x4_u4*>(gxt + b0) = zsum;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (b0 + r < K) gxt[b0 + r] = zsum[r];
            }
        }
    };
    ifetch(wave, ia);
    xfetch(wave, xa);
#pragma unroll 1
    for (int u = wave; u < nunit; u += 2 * GW) {
        ifetch(u + GW, ib);
        xfetch(u + GW, xb);
        body(u, ia, xa);
        if (u + GW < nunit) {
            ifetch(u + 2 * GW, ia);
            xfetch(u + 2 * GW, xa);
            body(u + GW, ib, xb);
        }
    }
}

// glogx = sum over the steps of (grt_s E^T) * exp(logx - 2 mc_s D): mcs (n_iter, F, M1) the iterates the steps started from, grts
// (n_iter, F, 2 M1 - 1) the cotangents dsa_mcep_newton_update_bwd produced for them; `images` of mcep_resid_bwd_prepare.
// DSA_ERR_UNSUPPORTED (no error text): orders outside 32 .. 54 or more steps than one workgroup's LDS holds -- the caller keeps the
// in-place accumulation of dsa_mcep_newton_resid_h_bwd.
int mcep_glogx_h(const void* logx, int64_t F, int K, const void* mcs, int M1, const void* grts, int n_iter, const void* images, void* glogx,
                 hipStream_t st)
{
    const int ks1 = (M1 + 31) / 32, kse = mrb::kse_of(M1), N = 2 * M1 - 1;
    if (!(ks1 == 2 && M1 >= 33 && M1 <= 55 && K >= 4 && n_iter >= 1)) return DSA_ERR_UNSUPPORTED;
    if ((long)n_iter * mgx::step_bytes(2, kse) > mgx::kMaxLds) return DSA_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((F + 15) / 16));
    const int lds_b = n_iter * mgx::step_bytes(2, kse);
#define DSA_GLOGX(KSEV, NT3V)                                                                                                             \
    do {                                                                                                                                  \
        static std::atomic<uint64_t> attr{0};                                                                                             \
        if (!ensure_dynamic_lds((const void*)mcep_glogx_h_kernel<2, KSEV, NT3V>, mgx::kMaxLds, attr))                                     \
            return fail(DSA_ERR_LAUNCH, "mcep_glogx_h: cannot reserve LDS%s");                                                            \
        hipLaunchKernelGGL((mcep_glogx_h_kernel<2, KSEV, NT3V>), grid, dim3(512), lds_b, st, (const float*)logx, (long)F, K,              \
                           (const float*)mcs, M1, (const float*)grts, N, n_iter, (const _Float16*)images, (float*)glogx);                 \
    } while (0)
    if (kse == 3) DSA_GLOGX(3, 3);
    else DSA_GLOGX(4, 4);
#undef DSA_GLOGX
    return check_launch("mcep_glogx_h");
}

}  // namespace dsa
