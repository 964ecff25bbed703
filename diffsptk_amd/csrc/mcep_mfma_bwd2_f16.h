// Round 5: the mel-cepstral backward at TWO waves per SIMD (included by mcep_mfma.hip after mcep_mfma_bwd_f16.h, whose operand
// images, scales and mathematics it shares: the reverse sweep over the unrolled Newton iteration, mcep.py:189-224 under autograd).
//
// mcep_mfma_bwd_kernel_h keeps four 257-bin arrays of its 16 frames in registers (log2 X, lbar, e, zbar: 256 registers before the
// solve's 109), so one wave fills a SIMD and every latency of its in-order stream is exposed.  With the forward's rt rows saved
// (DSA_ALGO_HIST_HAS_RT) e = exp(log X + D^T mc) is needed in ONE place only -- zbar = ebar * e -- and the step can run bin group by
// bin group (4 MFMA tiles = 64 bins at a time) with nothing but lbar resident:
//   windows <- the saved rt row;  A [g | u] = [rt[:25] - alpha | mbar] by the forward's 4 x 4 x 1 block elimination (the second
//   right-hand side rides in column group 6; g = the difference of two saved iterates on all steps but the first of the sweep);
//   rtbar from the outer product u g^T (as mcep_mfma_bwd_kernel_h);
//   per group:  t = log2 X + D^T mc (X re-read: L2),  e = exp2(t - ceil(max t)),  ebar = E rtbar (streamed image),
//               zbar = ebar * e,  lbar += zbar,  mbar' += (-2 D) zbar with the group's own power-of-two scale.
// 64 resident registers instead of 256: 256 registers per wave, eight waves per CU, and the elimination is the forward's (at 256
// registers the block quadruples stay in the vector file).  Per frame the arithmetic depends on the frame's own data only (batch
// invariant).  The split tail of mcep_mfma_bwd_kernel_h (pieces of Newton steps handed over through memory) is NOT taken over: it assumes
// slots that run at one speed, and measured here it never helps (batch 160 .. 1024: equal or up to 5 % slower).
#pragma once

namespace dsa {

namespace mh2 {
using namespace mhb;
constexpr int WAVES_2 = 8;
// LDS carve-up (float units): D^T images of the first chain | -2 D images (bins contracted) | small tables | per-wave records
constexpr int C_DB = DL_OFF + 16 * 64 * 4;            // DB hi | lo: 2 x 4096 floats
constexpr int C_E256 = C_DB + 2 * IMG_DB / 2;         // [64] unscaled E[256][m] (0 past 48)
constexpr int C_D256 = C_E256 + 64;                   // [32] -2 log2(e) D[c][256], then [32] -2 D[c][256]
constexpr int C_NAV = C_D256 + 64;                    // [28] -alpha_vec, zero-padded
constexpr int C_ZERO = C_NAV + 28;                    // [28] zeros
constexpr int C_SLOT = C_ZERO + 28;                   // [4] the workgroup's first wave-slot number (split tail)
constexpr int C_WAVE = C_SLOT + 4;
constexpr int C_LDS_FLOATS = C_WAVE + WAVES_2 * B_WAVE_FLOATS;
static_assert(C_WAVE % 4 == 0 && C_LDS_FLOATS * 4 <= 160 * 1024, "the two-wave backward's LDS carve-up");
}  // namespace mh2

#ifndef BWD2_ABL
#define BWD2_ABL 0   // measurement builds only: 1 no elimination, 2 no rtbar products, 4 no group chains, 8 X from one line,
                     // 16 history / rt rows from one small region, 32 no back substitution, 64 no rtbar rotations
#endif

__global__ __launch_bounds__(512, 2) DSA_PK_TARGET void mcep_mfma_bwd2_kernel_h(
    const float* __restrict__ gmc, const float* __restrict__ X, const float* __restrict__ hist, long F, int n_iter,
    const float* __restrict__ av, float* gX, long ntiles16, unsigned int* __restrict__ queue,
    const _Float16* __restrict__ img, const float* __restrict__ hist_rt)
{
    using namespace mh2;
    constexpr float kInvSDM = 1.f / (SD * SM);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform to the compiler: tile addresses in scalar registers)
    const __amdgpu_buffer_rsrc_t img_rsrc = image_rsrc(img, IMG_B_BYTES);   // the streamed E / G images (kernel argument: uniform)
    const int n = lane & 15, g = lane >> 4;

    // ---------------- operand images and small tables ----------------
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(img);
        f32x4* dst = reinterpret_cast<f32x4*>(lds + DH_OFF);
        for (int idx = tid; idx < (2 * IMG_D) / 8; idx += WAVES_2 * 64) dst[idx] = src[idx];
        const f32x4* srcb = reinterpret_cast<const f32x4*>(img + IMG_DBH);
        f32x4* dstb = reinterpret_cast<f32x4*>(lds + C_DB);
        for (int idx = tid; idx < 2 * IMG_DB / 8; idx += WAVES_2 * 64) dstb[idx] = srcb[idx];
    }
    const float* tail_f = reinterpret_cast<const float*>(img + IMG_HALVES);     // G[256][c] (forward workspace tail)
    const float* tail_b = reinterpret_cast<const float*>(img + IMG_B_HALVES);   // -2 D[c][256] | E[256][m] | E[bin][48]
    if (tid < 64) lds[C_E256 + tid] = tail_b[32 + tid];          // E[256][m], m < 64 (0 past 48): ebar of the Nyquist bin
    if (tid < 32) {
        lds[C_D256 + tid] = 1.4426950408889634f * tail_b[tid];    // -2 log2(e) D[c][256]
        lds[C_D256 + 32 + tid] = tail_b[tid];                     // -2 D[c][256]
    }
    if (tid < 28) {
        lds[C_NAV + tid] = tid < M1 ? -av[tid] : 0.f;
        lds[C_ZERO + tid] = 0.f;
    }
    __syncthreads();

    float* wave_lds = lds + C_WAVE + wave * B_WAVE_FLOATS;
    float* rt_n = wave_lds + n * FS;           // this lane's frame, MFMA-layout view
    float* rr_n = rt_n + 52;
    float* aux_n = rt_n + 116;
    const int nq = lane >> 2, gs = lane & 3;   // solve layout: a quad per frame
    float* rt_q = wave_lds + nq * FS;
    float* rr_q = rt_q + 52;
    float* aux_q = rt_q + 116;
    int lane_a = lane, lane_c = lane + C_DB / 4;
    asm volatile("" : "+v"(lane_a), "+v"(lane_c));
    const f16x8* DH = reinterpret_cast<const f16x8*>(lds + DH_OFF) + lane_a;
    const f16x8* DL = reinterpret_cast<const f16x8*>(lds + DL_OFF) + lane_a;
    const f16x8* DBH = reinterpret_cast<const f16x8*>(lds) + lane_c;
    const f16x8* DBL = DBH + IMG_DB / 8;
    const unsigned lane16 = (unsigned)lane * 16u;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const long wave_stride = (long)gridDim.x * WAVES_2;
    long tile = (long)blockIdx.x * WAVES_2 + wave;
#define DSA_SB() __builtin_amdgcn_sched_barrier(0x0004)
    while (tile < ntiles16) {
        const int it_hi = n_iter, it_lo = 0;
        // The tile is uniform: every array is addressed as a scalar base of the tile's first row + a 32-bit lane offset (frames past F
        // read the last row; 64-bit per-lane pointers cost registers and carry-chained vector additions)
        const long t16 = tile * 16;
        const int rows_here = (int)((F - t16 < 16) ? F - t16 : 16);
        const bool f_ok = n < rows_here;
        const int rn = f_ok ? n : rows_here - 1;                       // this lane's row of the tile, MFMA layout
        const int rq = (lane >> 2) < rows_here ? (lane >> 2) : rows_here - 1;   // quad layout
        const float* Xt = X + t16 * K;
        float* gXt = gX + t16 * K;
        const int xo = (BWD2_ABL & 8) ? 4 * g : rn * K + 4 * g;        // tile mt of the lane's row: Xt[xo + 16 mt ..]
        auto xload = [&](int mt) __attribute__((always_inline)) { return *reinterpret_cast<const f32x4_u4*>(Xt + xo + 16 * mt); };
        f32x4 lbar[16];
#pragma unroll
        for (int mt = 0; mt < 16; ++mt) lbar[mt] = zero4;
        const float logx256 = __log2f(Xt[rn * K + H]);
        float lbar256 = 0.f;
        // mbar in the C/D layout of a 32-row product: tile it2, register r <-> coefficient 16 it2 + 4 g + r
        f32x4 mbarC[2];
        {
#pragma unroll
            for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = it2 * 16 + 4 * g + r;
                    mbarC[it2][r] = c < M1 ? (gmc + t16 * M1)[rn * M1 + c] : 0.f;
                }
        }
        // the next tile: drawn from the device counter (the waves of a SIMD do not run at one speed)
        long tile_next;
        {
            unsigned int nxt = 0;
            if (lane == 0) nxt = atomicAdd(queue, 1u);
            tile_next = wave_stride + (long)__builtin_amdgcn_readfirstlane((int)nxt);
        }
        // between the steps mbar lives in the frame's exchange window (aux [0, 32)), the step's rt row in the rt / rr windows
#pragma unroll
        for (int it2 = 0; it2 < 2; ++it2) *reinterpret_cast<f32x4*>(aux_n + it2 * 16 + 4 * g) = mbarC[it2];
        int g_it = g;
        asm volatile("" : "+v"(g_it));
        // one saved rt row (lane (n, g): rt[16 it + 4 g + r] of its frame) requested / written into the windows
        f32x4 rw0, rw1, rw2;
        float rw48;
        auto rt_row_request = [&](int it_) __attribute__((always_inline)) {
            const float* hr = hist_rt + ((BWD2_ABL & 16) ? 0L : (long)it_ * F + t16) * M2 + rn * M2;
            rw0 = *reinterpret_cast<const f32x4_u4*>(hr + 4 * g_it);
            rw1 = *reinterpret_cast<const f32x4_u4*>(hr + 16 + 4 * g_it);
            rw2 = *reinterpret_cast<const f32x4_u4*>(hr + 32 + 4 * g_it);
            rw48 = hr[48];
        };
        auto rt_row_to_windows = [&]() __attribute__((always_inline)) {
            float* rtw = rt_n + 4 * g_it;
            float* rra = rr_n + 27 + 4 * g_it;
            float* rrb = rr_n + 24 - 4 * g_it;              // rr[27 - idx], idx = 4 g + r: the lane's four entries reversed
            float* rra1 = g_it < 3 ? rra + 16 : rr_n + 55;
            float* rrb1 = g_it < 3 ? rrb - 16 : rr_n + 59;
            *reinterpret_cast<f32x4*>(rtw) = rw0;
            *reinterpret_cast<f32x4_u4*>(rra) = rw0;
            *reinterpret_cast<f32x4*>(rrb) = __builtin_shufflevector(rw0, rw0, 3, 2, 1, 0);
            *reinterpret_cast<f32x4*>(rtw + 16) = rw1;
            *reinterpret_cast<f32x4_u4*>(rra1) = rw1;
            *reinterpret_cast<f32x4_u4*>(rrb1) = __builtin_shufflevector(rw1, rw1, 3, 2, 1, 0);
            *reinterpret_cast<f32x4*>(rtw + 32) = rw2;
            rt_n[48] = rw48;
        };
        rt_row_request(it_hi - 1);
        rt_row_to_windows();
        // Touch loads: one dword per 128-byte line of the spectrum rows the groups read later in the step, issued behind the loads
        // whose data is needed next (loads return in order) and a build + elimination ahead of the next request
        auto touch_x = [&](int q) __attribute__((always_inline)) {   // bins 64 q .. 64 q + 63 of the tile's rows: lane (row, line)
            const int r = (lane >> 2) < rows_here ? (lane >> 2) : rows_here - 1, j = lane & 3;
            return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(Xt + r * K + 64 * q) + (j < 2 ? 128 * j : 252));
        };
#define touch_done(v) asm volatile("" ::"v"(v))
#ifndef BWD2_TOUCH_X
#define BWD2_TOUCH_X 0   // (measured: 1.276 ms with the touches, 1.252 without)
#endif
#ifdef DSA_MCEP_TIMING   // phase stamps in scalar registers (pinned: nothing moves across), flushed at the end of the step
#define B2STAMP(i) do { __builtin_amdgcn_sched_barrier(0); st2_[i] = (unsigned)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define B2STAMP(i)
#endif

        for (int iter = it_hi - 1; iter >= it_lo; --iter) {
#ifdef DSA_MCEP_TIMING
            unsigned st2_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
            B2STAMP(0);
            const bool g_saved = iter + 1 < n_iter;
            // g = A^-1 (rt[:25] - alpha) as the difference of two SAVED iterates (mcep.py:224: mc <- mc + g): requested here, a build
            // and an elimination ahead of its use
            float h0r[KS], h1r[KS];
            {
                const float* h0 = hist + ((BWD2_ABL & 16) ? 0L : (long)iter * F + t16) * M1 + rq * M1;
                const float* h1 = g_saved ? h0 + F * M1 : h0;
#pragma unroll
                for (int c = 0; c < KS - 1; ++c) { h0r[c] = h0[gs + 4 * c]; h1r[c] = h1[gs + 4 * c]; }
                h0r[KS - 1] = h0[M1 - 1]; h1r[KS - 1] = h1[M1 - 1];
            }
            float t_x[4] = {0.f, 0.f, 0.f, 0.f};
            if (BWD2_TOUCH_X) {
#pragma unroll
                for (int q = 0; q < 4; ++q) t_x[q] = touch_x(q);
            }
            __builtin_amdgcn_wave_barrier();
            B2STAMP(1);

            // ---------------- solve A [gv | uv] = [rt[:25] - alpha | mbar] in the quad layout ----------------
            float xq1[KS], xq2[KS];
            float mcv[8];
            f32x4 xg[4];
            {
                // the lane-group masks live for the solve only (8 registers)
                int gsv = gs;
                asm volatile("" : "+v"(gsv));
                const GroupMask gq = make_group_mask(gsv);
#pragma unroll
                for (int c = 0; c < KS; ++c) { xq1[c] = 0.f; xq2[c] = 0.f; }
                xq1[KS - 1] = keep_if(gq.m[1], -1.f);
                xq2[KS - 1] = keep_if(gq.m[2], -1.f);
                f32x4 a[blk::NBLK];
                float ninvs[M1];   // (dead: the back substitutions take the pivots' reciprocals again, 25 registers fewer across the elimination)
                {
                    const float* zr = lds + C_ZERO;
                    const float* pa6 = gsv == 0 ? rt_q + 24 : (gsv == 1 ? rt_q : (gsv == 2 ? aux_q : zr));
                    const float* pb6 = gsv == 0 ? rr_q + 3 : (gsv == 1 ? lds + C_NAV : zr);
                    blk_build_rows<0>(a, rt_q, rr_q, pa6, pb6, gs);
                }
                __builtin_amdgcn_wave_barrier();
                B2STAMP(2);
                if (!(BWD2_ABL & 1)) blk_elim_all(a, gq, ninvs, std::make_integer_sequence<int, M1>{});
                B2STAMP(3);
                if (BWD2_TOUCH_X) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) touch_done(t_x[q]);
                }
                if (g_saved) {
#pragma unroll
                    for (int c = 0; c < KS; ++c) xq1[c] = h1r[c] - h0r[c];
                    xq1[KS - 1] = keep_if(gq.m[0], xq1[KS - 1]);   // k = 24 on lane 0 only
                } else {
                    blk_backsub_all_r(a, xq1, gq, std::make_integer_sequence<int, blk::NG>{});
                    xq1[KS - 1] = keep_if(gq.m[0], xq1[KS - 1]);
                }
                if (!(BWD2_ABL & 32)) blk_backsub_all_r(a, xq2, gq, std::make_integer_sequence<int, blk::NG>{});
                else {
#pragma unroll
                    for (int c = 0; c < KS; ++c) xq2[c] = a[blk::at(c, c)][0] + a[blk::at(0, c)][1];
                }
                xq2[KS - 1] = keep_if(gq.m[0], xq2[KS - 1]);
            }
            B2STAMP(4);
            // requested here, rtbar and the exchange ahead of their use (the elimination's 112 registers are free again): this step's
            // iterate, the first group's spectrum rows, the NEXT step's rt row
#pragma unroll
            for (int i = 0; i < 8; ++i) mcv[i] = (8 * g + i < M1) ? (hist + ((BWD2_ABL & 16) ? 0L : (long)iter * F + t16) * M1)[rn * M1 + 8 * g + i] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) xg[i] = xload(i);
            if (iter > it_lo) rt_row_request(iter - 1);

            // ---------------- rtbar (49 entries), scaled per frame to below 2^13, into the exchange window ----------------
            // (as mcep_mfma_bwd_kernel_h: 2 x 49 4 x 4 x 1 outer-product blocks, quad rotations for the sums over equal m; the
            // exchange goes through the rr window, which is dead once the rows are built: aux keeps mbar)
            {
                f32x4 DHk[13], DTk[13];
#pragma unroll
                for (int k = 0; k < 13; ++k) { DHk[k] = zero4; DTk[k] = zero4; }
#pragma unroll
                for (int ri = 0; ri < ((BWD2_ABL & 2) ? 1 : KS); ++ri)
#pragma unroll
                    for (int cj = 0; cj < KS; ++cj) {
                        DHk[ri + cj] = mfma441(xq2[ri], xq1[cj], DHk[ri + cj]);
                        DTk[cj - ri + 6] = mfma441(xq2[ri], xq1[cj], DTk[cj - ri + 6]);
                    }
                auto rotR = [](float v, int k) __attribute__((always_inline)) {   // w <- (w - k) mod 4
                    return k == 1 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x93, 0xf, 0xf, true))
                           : k == 2 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true))
                                    : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x39, 0xf, 0xf, true));
                };
                auto rotL = [](float v, int k) __attribute__((always_inline)) {   // w <- (w + k) mod 4
                    return k == 1 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x39, 0xf, 0xf, true))
                           : k == 2 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true))
                                    : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x93, 0xf, 0xf, true));
                };
                auto refl = [](float v, int k) __attribute__((always_inline)) {   // w <- (k - w) mod 4
                    return k == 0 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x6C, 0xf, 0xf, true))
                           : k == 1 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true))
                           : k == 2 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xC6, 0xf, 0xf, true))
                                    : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x1B, 0xf, 0xf, true));
                };
                float rb[13];   // quad layout: lane gs of slot s holds rtbar[4 s + gs]
#pragma unroll
                for (int sl = 0; sl < 13; ++sl) {
                    float hs = DHk[sl][0];
                    if (BWD2_ABL & 64) { rb[sl] = hs + DTk[sl][1]; continue; }
#pragma unroll
                    for (int ip = 1; ip < 4; ++ip) {
                        const float prev = sl > 0 ? DHk[sl - 1][ip] : 0.f;
                        hs += rotR(gs < 4 - ip ? DHk[sl][ip] : prev, ip);
                    }
                    float r = -hs;
                    if (sl < 7) {
                        float tp = DTk[sl + 6][0];
#pragma unroll
                        for (int ip = 1; ip < 4; ++ip) {
                            const float nxt = sl + 7 < 13 ? DTk[sl + 7][ip] : 0.f;
                            tp += rotL(gs >= ip ? DTk[sl + 6][ip] : nxt, ip);
                        }
                        float tn = 0.f;
#pragma unroll
                        for (int ip = 0; ip < 4; ++ip) {
                            const float far = 6 - sl - 1 >= 0 ? DTk[6 - sl - 1][ip] : 0.f;
                            tn += refl(gs <= ip ? DTk[6 - sl][ip] : far, ip);
                        }
                        if (sl == 0) tn = gs == 0 ? 0.f : tn;   // offset 0 is counted once (it is in tp)
                        r = r - tp - tn + xq2[sl];              // + u_m: through the right-hand side rt[:25] - alpha
                    }
                    rb[sl] = r;
                }
                float amax = 0.f;
#pragma unroll
                for (int sl = 0; sl < 13; ++sl) amax = __builtin_fmaxf(amax, __builtin_fabsf(rb[sl]));
                amax = __builtin_fmaxf(amax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(amax), 0xB1, 0xf, 0xf, true)));
                amax = __builtin_fmaxf(amax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(amax), 0x4E, 0xf, 0xf, true)));
                const int s_r = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(amax);
#pragma unroll
                for (int sl = 0; sl < 13; ++sl) rr_q[4 * sl + gs] = __builtin_ldexpf(rb[sl], s_r);
                rr_q[52 + gs] = 0.f;
                rr_q[56 + gs] = 0.f;
                rr_q[60 + gs] = gs == 3 ? __int_as_float(s_r) : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            B2STAMP(5);
            f16x8 rbh[2], rbl[2];
            float eb256 = 0.f;
            const int s_rn = __float_as_int(rr_n[63]);   // the scale of THIS lane's frame in the MFMA layout
            {
                float rv[16];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    rv[i] = rr_n[8 * g + i];
                    rv[8 + i] = g < 3 ? rr_n[32 + 8 * g + i] : 0.f;     // slot 63 of group 3 holds the scale, not data
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    eb256 = __builtin_fmaf(rv[i], lds[C_E256 + 8 * g + i], eb256);
                    eb256 = __builtin_fmaf(rv[8 + i], lds[C_E256 + 32 + 8 * g + i], eb256);
                }
                float lo8[8], hi8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { lo8[i] = rv[i]; hi8[i] = rv[8 + i]; }
                split8(lo8, rbh[0], rbl[0]);
                split8(hi8, rbh[1], rbl[1]);
            }
            __builtin_amdgcn_wave_barrier();
            eb256 = rows_sum4(eb256);
            // the next step's rt row into the windows (their last reader, the exchange above, is done)
            if (iter > it_lo) rt_row_to_windows();

            // ---------------- the bins, 64 at a time: e, ebar, zbar, lbar, this step's contribution to mbar ----------------
            f16x8 bh, bl;
            float d256 = 0.f;
            {
                float ms[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    ms[i] = mcv[i] * SM;
                    d256 = __builtin_fmaf(mcv[i], lds[C_D256 + 8 * g + i], d256);
                }
                split8(ms, bh, bl);
            }
            d256 = rows_sum4(d256);
            f32x4 macc[2] = {zero4, zero4};
            B2STAMP(6);
            if (!(BWD2_ABL & 4)) {
                // Streamed E image (L2), two tiles in flight.  Loads return in order, so the one slow request of a group -- the
                // NEXT group's spectrum rows (memory) -- is made behind the next group's first two image tiles, when this group's
                // products are through: nothing younger than it is needed before its own data is.
                f16x8 ah[2][2], al[2][2];
                auto req = [&](int slot, int mt) __attribute__((always_inline)) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        ah[slot][ks] = gload8(img_rsrc, lane16, 2 * (IMG_EBH + (mt * 2 + ks) * 512));
                        al[slot][ks] = gload8(img_rsrc, lane16, 2 * (IMG_EBL + (mt * 2 + ks) * 512));
                    }
                };
                req(0, 0);
                req(1, 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // first chain: t = log2 X + D^T mc on the group's four tiles
                    f32x4 c[4];
                    {
                        f16x8 dl_[4], dh_[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) { dl_[i] = DL[(4 * q + i) * 64]; dh_[i] = DH[(4 * q + i) * 64]; }
#pragma unroll
                        for (int i = 0; i < 4; ++i) c[i] = mfma_h(dl_[i], bh, zero4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) c[i] = mfma_h(dh_[i], bl, c[i]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) c[i] = mfma_h(dh_[i], bh, c[i]);
                    }
                    // ebar = E rtbar on one tile (independent of e)
                    auto ebar = [&](int slot) __attribute__((always_inline)) {
                        f32x4 a_ = mfma_h(al[slot][0], rbh[0], zero4);
                        a_ = mfma_h(ah[slot][0], rbl[0], a_);
                        a_ = mfma_h(ah[slot][0], rbh[0], a_);
                        a_ = mfma_h(al[slot][1], rbh[1], a_);
                        a_ = mfma_h(ah[slot][1], rbl[1], a_);
                        a_ = mfma_h(ah[slot][1], rbh[1], a_);
                        return a_;
                    };
                    // t, the group's shift
                    float gm = -3.0e38f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4 lx = {__log2f(xg[i][0]), __log2f(xg[i][1]), __log2f(xg[i][2]), __log2f(xg[i][3])};
                        const f32x2v ta = fma2(lo2(c[i]), kInvSDM, lo2(lx)), tb = fma2(hi2(c[i]), kInvSDM, hi2(lx));
                        c[i] = f32x4{ta[0], ta[1], tb[0], tb[1]};
                        gm = __builtin_fmaxf(__builtin_fmaxf(gm, ta[0]), ta[1]);
                        gm = __builtin_fmaxf(__builtin_fmaxf(gm, tb[0]), tb[1]);
                    }
                    gm = rows_max4(gm);
                    const float mi = __builtin_ceilf(gm);
                    const int kz = (int)mi - s_rn - SEB_LOG2;   // zbar = acc e' 2^kz, e = 2^mi e'
                    float zm = 0.f;
                    // e, zbar = ebar * e, lbar += zbar of one tile
                    auto zbar = [&](int i, f32x4 acc) __attribute__((always_inline)) {
                        const f32x2v ta = lo2(c[i]) - f32x2v{mi, mi}, tb = hi2(c[i]) - f32x2v{mi, mi};
                        const f32x2v ea = {__builtin_amdgcn_exp2f(ta[0]), __builtin_amdgcn_exp2f(ta[1])};
                        const f32x2v eb = {__builtin_amdgcn_exp2f(tb[0]), __builtin_amdgcn_exp2f(tb[1])};
                        const f32x2v ma = lo2(acc) * ea, mb = hi2(acc) * eb;
                        const f32x4 z = {__builtin_ldexpf(ma[0], kz), __builtin_ldexpf(ma[1], kz), __builtin_ldexpf(mb[0], kz),
                                         __builtin_ldexpf(mb[1], kz)};
                        c[i] = z;
                        lbar[4 * q + i] += z;
                        zm = __builtin_fmaxf(__builtin_fmaxf(zm, __builtin_fabsf(z[0])), __builtin_fabsf(z[1]));
                        zm = __builtin_fmaxf(__builtin_fmaxf(zm, __builtin_fabsf(z[2])), __builtin_fabsf(z[3]));
                    };
                    if (q == 1) B2STAMP(8);
                    {
                        const f32x4 a0 = ebar(0);
                        req(0, 4 * q + 2);
                        const f32x4 a1 = ebar(1);
                        req(1, 4 * q + 3);
                        zbar(0, a0);
                        zbar(1, a1);
                        const f32x4 a2 = ebar(0);
                        const f32x4 a3 = ebar(1);
                        if (q < 3) {
                            req(0, 4 * q + 4);
                            req(1, 4 * q + 5);
#pragma unroll
                            for (int i = 0; i < 4; ++i) xg[i] = xload(4 * q + 4 + i);
                        }
                        zbar(2, a2);
                        zbar(3, a3);
                    }
                    if (q == 1) B2STAMP(9);
                    // mbar contribution of the group with the group's own scale
                    zm = rows_max4(zm);
                    const int s_z = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(zm);
                    f32x4 acc2[2] = {zero4, zero4};
#pragma unroll
                    for (int ks2 = 0; ks2 < 2; ++ks2) {
                        f16x8 zh, zl;
#pragma unroll
                        for (int t_ = 0; t_ < 2; ++t_)
#pragma unroll
                            for (int r = 0; r < 4; r += 2) {
                                f16x2 hh, ll;
                                split2(__builtin_ldexpf(c[2 * ks2 + t_][r], s_z), __builtin_ldexpf(c[2 * ks2 + t_][r + 1], s_z), hh, ll);
                                zh[4 * t_ + r] = hh[0]; zh[4 * t_ + r + 1] = hh[1];
                                zl[4 * t_ + r] = ll[0]; zl[4 * t_ + r + 1] = ll[1];
                            }
                        const int j = 2 * q + ks2;
#pragma unroll
                        for (int it2 = 0; it2 < 2; ++it2) {
                            const f16x8 dh_ = DBH[(it2 * 8 + j) * 64], dl_ = DBL[(it2 * 8 + j) * 64];
                            acc2[it2] = mfma_h(dl_, zh, acc2[it2]);
                            acc2[it2] = mfma_h(dh_, zl, acc2[it2]);
                            acc2[it2] = mfma_h(dh_, zh, acc2[it2]);
                        }
                    }
#pragma unroll
                    for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) macc[it2][r] += __builtin_ldexpf(acc2[it2][r], -s_z - SDB_LOG2);
                    if (q == 0) B2STAMP(7);
                    if (q == 1) B2STAMP(10);
                }
            }
            // the Nyquist bin; mbar <- mbar + this step's contribution (in the exchange window)
            {
                const float t256 = logx256 + d256;
                const float m256 = __builtin_ceilf(t256);
                const float zb256 = __builtin_ldexpf(eb256 * __builtin_amdgcn_exp2f(t256 - m256), (int)m256 - s_rn);
                lbar256 += zb256;
#pragma unroll
                for (int it2 = 0; it2 < 2; ++it2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c_ = it2 * 16 + 4 * g + r;   // < 32 (the table is 0 past c = 24)
                        macc[it2][r] = __builtin_fmaf(zb256, lds[C_D256 + 32 + c_], macc[it2][r]);
                    }
                    f32x4* mp = reinterpret_cast<f32x4*>(aux_n + it2 * 16 + 4 * g);
                    *mp = *mp + macc[it2];
                }
            }
            B2STAMP(11);
#ifdef DSA_MCEP_TIMING
            if (blockIdx.x == 0 && threadIdx.x == 0 && tile == (long)blockIdx.x * WAVES_2 + wave && iter == n_iter - 2)
                for (int i_ = 0; i_ < 16; ++i_) g_mcep_stamps[40 + i_] = st2_[i_];
#endif
        }

        // ---------------- lbar += G mbar_0 (mcep.py:204-207 adjoint); gX = lbar / X ----------------
        __builtin_amdgcn_wave_barrier();
        float m0[8];
        float mmax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            m0[i] = (8 * g + i < M1) ? aux_n[8 * g + i] : 0.f;
            mmax = __builtin_fmaxf(mmax, __builtin_fabsf(m0[i]));
        }
        __builtin_amdgcn_wave_barrier();
        mmax = rows_max4(mmax);
        const int s_m = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(mmax);
        float part256 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) part256 = __builtin_fmaf(m0[i], tail_f[8 * g + i], part256);   // G[256][c] (0 past c = 24)
        part256 = rows_sum4(part256);
        lbar256 += part256;
        {
            float ms[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ms[i] = __builtin_ldexpf(m0[i], s_m);
            f16x8 mh8, ml8;
            split8(ms, mh8, ml8);
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
                const f16x8 ah = gload8(img_rsrc, lane16, 2 * (IMG_GBH + mt * 512)), al = gload8(img_rsrc, lane16, 2 * (IMG_GBL + mt * 512));
                const f32x4 xv = xload(mt);
                f32x4 acc = {0, 0, 0, 0};
                acc = mfma_h(al, mh8, acc);
                acc = mfma_h(ah, ml8, acc);
                acc = mfma_h(ah, mh8, acc);
                if (f_ok) {
                    float* dst = gXt + rn * K + mt * 16 + 4 * g;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dst[r] = (lbar[mt][r] + __builtin_ldexpf(acc[r], -s_m - SGB_LOG2)) * __builtin_amdgcn_rcpf(xv[r]);   // d log X = dX / X
                }
            }
        }
        if (f_ok && g == 0) gXt[rn * K + H] = lbar256 * __builtin_amdgcn_rcpf(Xt[rn * K + H]);
        tile = tile_next;
    }
#undef DSA_SB
}

}  // namespace dsa
