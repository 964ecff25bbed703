// Round 5: the mel-cepstral backward rebuilt around TWO waves per SIMD (included by mcep_mfma.hip after mcep_mfma_bwd_f16.h, whose
// images, scales and mathematics it shares: the reverse sweep over the unrolled Newton iteration from the saved iterates,
// mcep.py:189-224 under autograd).
//
// mcep_mfma_bwd_kernel_h holds log2 X, lbar, e and zbar of its 16 frames in registers -- four 257-bin arrays, 256 registers before
// the solve's 109 -- so one wave fills a SIMD and every latency of its in-order stream is exposed (1.53 ms per 204 800 frames for
// three rounds; the forward kernel, two waves per SIMD, does two chains and a solve in a seventh of the time).  Here a PAIR of
// waves shares the 16 frames of a tile and splits the BINS: wave h of the pair (h = wave & 1; the two sit on different SIMDs)
// owns bins 128 h .. 128 h + 127 (h = 1 also the Nyquist bin), i.e. 8 of the 16 MFMA tiles of every bin array -- 32 registers
// per array instead of 64, 256 registers per wave, eight waves per CU, each SIMD hosting waves of two different pairs.  Per step:
//   both   t = log2 X + D^T mc (own tiles), e = exp2(t + own shift), partial rt = E^T e over the own bins;
//   h = 1  hands its partial rt to h = 0 through LDS; h = 0 adds the two, writes the pair's rt / rr windows, signals;
//   both   the SAME 25 x 25 system, solved redundantly with both right-hand sides (4 x 4 x 1 block elimination of the forward
//          kernel: at 256 registers the quadruples stay in the vector file), rtbar redundantly;
//   both   ebar = E rtbar, zbar = ebar * e, lbar += zbar on the own tiles; partial mbar = -2 D zbar over the own bins;
//   h = 1  hands its partial to h = 0, which adds and publishes mbar for both.
// Two hand-overs each way per step (LDS counters, the waiting wave sleeps); everything else a wave does alone.  The duplicated
// solve and rtbar are the price of symmetric, barrier-free halves.  Operand images: D^T and E^T (forward layout) stay in LDS
// (80 KB), the three backward-layout images (E, -2 D, G with bins as rows / contracted) stream from L2, each wave its own half.
#pragma once

namespace dsa {

namespace mhp {
using namespace mhb;
constexpr int WAVES_P = 8, PAIRS = 4;
constexpr int P_E48 = EL_OFF + 24 * 64 * 4;          // [16 mt][4 g][4 r]
constexpr int P_E256 = P_E48 + 256;                  // as B_E256: [48] scaled, [48] unscaled, then [64] unscaled E[256][m]
constexpr int P_D256 = P_E256 + 52 + 64;             // [32] -2 log2(e) D[c][256], [32] -2 D[c][256]
constexpr int P_AV = P_D256 + 64;
constexpr int P_NAV = P_AV + 28;
constexpr int P_ZERO = P_NAV + 28;
constexpr int P_FLAG = P_ZERO + 28;                  // [PAIRS][4] hand-over counters
constexpr int P_PAIR = P_FLAG + 4 * PAIRS;
constexpr int WS = 116;                              // per-frame record of the pair's windows: rt [0,52) | rr [52,116)
constexpr int AS = 68;                               // per-frame record of a wave's exchange window (68 % 32 = 4)
constexpr int XS = 52;                               // per-frame record of the hand-over area (partial rt: 49; partial / full mbar: 32)
constexpr int PAIR_FLOATS = 16 * WS + 2 * 16 * AS + 16 * XS;
constexpr int P_LDS_FLOATS = P_PAIR + PAIRS * PAIR_FLOATS;
static_assert(P_LDS_FLOATS * 4 <= 160 * 1024, "the paired backward's LDS carve-up");
}  // namespace mhp

__device__ __forceinline__ void pair_signal(unsigned* flag, unsigned v, int lane)
{
    // LDS operations of a wave execute in order and LDS has no cache in front of it: once this wave's earlier writes have been
    // issued and counted down, a later write of the counter cannot overtake them
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) *reinterpret_cast<volatile unsigned*>(flag) = v;
    asm volatile("" ::: "memory");
}
#ifndef PAIR_ABL
#define PAIR_ABL 0   // measurement builds only: 1 no hand-over waits, 2 streamed images from a fixed line, 4 no rtbar, 8 no solve, 16 no chains
#endif
__device__ __forceinline__ void pair_wait(unsigned* flag, unsigned v)
{
    if (PAIR_ABL & 1) return;
    for (;;) {
        const unsigned x = (unsigned)__builtin_amdgcn_readfirstlane((int)*reinterpret_cast<volatile unsigned*>(flag));
        if ((int)(x - v) >= 0) break;
        __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(512, 2) void mcep_mfma_bwd_pair_kernel(
    const float* __restrict__ gmc, const float* __restrict__ X, const float* __restrict__ hist, long F, int n_iter,
    const float* __restrict__ av, float* gX, long ntiles16, const _Float16* __restrict__ img)
{
    using namespace mhp;
    constexpr float kInvSDM = 1.f / (SD * SM);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wave >> 1, h = wave & 1;   // pair, half
    const __amdgpu_buffer_rsrc_t img_rsrc = image_rsrc(img, IMG_B_BYTES);
    const int n = lane & 15, g = lane >> 4;

    // ---------------- operand images (forward layout) and small tables ----------------
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(img);
        f32x4* dst = reinterpret_cast<f32x4*>(lds + DH_OFF);
        for (int idx = tid; idx < (2 * IMG_D + 2 * IMG_E) / 8; idx += WAVES_P * 64) dst[idx] = src[idx];
    }
    const float* tail_f = reinterpret_cast<const float*>(img + IMG_HALVES);     // G[256][c]
    const float* tail_b = reinterpret_cast<const float*>(img + IMG_B_HALVES);   // -2 D[c][256] | E[256][m] | E[bin][48]
    if (tid < 256) {
        const int r = tid & 3, gg = (tid >> 2) & 3, mt = tid >> 4;
        lds[P_E48 + tid] = tail_b[96 + mt * 16 + gg * 4 + r];
    }
    if (tid < 48) lds[P_E256 + tid] = SE * tail_b[32 + tid];
    if (tid == 48) lds[P_E256 + 48] = tail_b[32 + 48];
    if (tid < 64) lds[P_E256 + 52 + tid] = tail_b[32 + tid];
    if (tid < 32) {
        lds[P_D256 + tid] = 1.4426950408889634f * tail_b[tid];
        lds[P_D256 + 32 + tid] = tail_b[tid];
    }
    if (tid < 28) {
        const float a_ = tid < M1 ? av[tid] : 0.f;
        lds[P_AV + tid] = a_;
        lds[P_NAV + tid] = -a_;
        lds[P_ZERO + tid] = 0.f;
    }
    if (tid < 4 * PAIRS) reinterpret_cast<unsigned*>(lds + P_FLAG)[tid] = 0u;
    __syncthreads();   // the only workgroup barrier

    float* pair_lds = lds + P_PAIR + pr * PAIR_FLOATS;
    float* win = pair_lds;                                   // the pair's rt / rr windows
    float* aux = pair_lds + 16 * WS + h * 16 * AS;           // this wave's exchange window
    float* xch = pair_lds + 16 * WS + 2 * 16 * AS;           // hand-over area (h = 1 -> h = 0: partials; h = 0 -> h = 1: mbar)
    unsigned* f10 = reinterpret_cast<unsigned*>(lds + P_FLAG) + 4 * pr;       // written by h = 1
    unsigned* f01 = f10 + 1;                                                   // written by h = 0
    unsigned n10 = 0, n01 = 0;                                                 // counts so far (both waves keep both)
    float* rt_n = win + n * WS;
    float* rr_n = rt_n + 52;
    float* aux_n = aux + n * AS;
    float* x_n = xch + n * XS;
    const int nq = lane >> 2, gs = lane & 3;
    float* rt_q = win + nq * WS;
    float* rr_q = rt_q + 52;
    float* aux_q = aux + nq * AS;
    const GroupMask gq = make_group_mask(gs);
    int lane_a = lane + h * 8 * 64, lane_b = lane + EL_OFF / 4;
    asm volatile("" : "+v"(lane_a), "+v"(lane_b));
    const f16x8* DH = reinterpret_cast<const f16x8*>(lds + DH_OFF) + lane_a;    // this half's tiles of D^T: [8 mt][64]
    const f16x8* DL = reinterpret_cast<const f16x8*>(lds + DL_OFF) + lane_a;
    const f16x8* EH = reinterpret_cast<const f16x8*>(lds + EH_OFF) + lane + h * 4 * 64;   // this half's bodies: [3 it][8 j][64], j = 4 h ..
    const f16x8* EL = reinterpret_cast<const f16x8*>(lds) + lane_b + h * 4 * 64;
    const f32x4* E484 = reinterpret_cast<const f32x4*>(lds + P_E48) + h * 32;             // [(2 jj + t) 4 + g], jj = 4 h + j
    const unsigned lane16 = (unsigned)lane * 16u;
    const int eb_half = h * (8 * 2 * 512) * 2;     // byte offsets of this half inside the streamed images
    const int db_half = h * (4 * 512) * 2;
    const int gb_half = h * (8 * 512) * 2;
    const long npairs = (long)gridDim.x * PAIRS;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#define DSA_SB() __builtin_amdgcn_sched_barrier(0x0004)
#define PLOAD(off) gload8(img_rsrc, lane16, (PAIR_ABL & 2) ? 0 : (off))

    for (long tile = (long)blockIdx.x * PAIRS + pr; tile < ntiles16; tile += npairs) {
        const long f_raw = tile * 16 + n;
        const bool f_ok = f_raw < F;
        const long f = f_ok ? f_raw : F - 1;
        const float* xf = X + f * K + h * 128;
        f32x4 logx[8], lbar[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const float* p = xf + mt * 16 + 4 * g;
            logx[mt] = f32x4{__log2f(p[0]), __log2f(p[1]), __log2f(p[2]), __log2f(p[3])};
            lbar[mt] = zero4;
        }
        const float logx256 = __log2f(X[f * K + H]);   // (used by h = 1 only)
        float lbar256 = 0.f;
        f32x4 mbarC[2];   // C/D layout of a 32-row product: tile it2, register r <-> coefficient 16 it2 + 4 g + r (both waves: all of it)
#pragma unroll
        for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = it2 * 16 + 4 * g + r;
                mbarC[it2][r] = c < M1 ? gmc[f * M1 + c] : 0.f;
            }
        float mcv_n[8], h0_n[KS], h1_n[KS];
        auto load_step = [&](int it_) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) mcv_n[i] = (8 * g + i < M1) ? hist[((long)it_ * F + f) * M1 + 8 * g + i] : 0.f;
            const long fq_raw = tile * 16 + nq;
            const float* h0 = hist + ((long)it_ * F + (fq_raw < F ? fq_raw : F - 1)) * M1;
            const float* h1 = it_ + 1 < n_iter ? h0 + F * M1 : h0;
#pragma unroll
            for (int c = 0; c < KS - 1; ++c) { h0_n[c] = h0[gs + 4 * c]; h1_n[c] = h1[gs + 4 * c]; }
            h0_n[KS - 1] = h0[M1 - 1]; h1_n[KS - 1] = h1[M1 - 1];
        };
        if (n_iter > 0) load_step(n_iter - 1);
        for (int iter = n_iter - 1; iter >= 0; --iter) {
            float mcv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) mcv[i] = mcv_n[i];
            const bool g_saved = iter + 1 < n_iter;
            float gh[KS];
#pragma unroll
            for (int c = 0; c < KS; ++c) gh[c] = h1_n[c] - h0_n[c];
            gh[KS - 1] = keep_if(gq.m[0], gh[KS - 1]);
            if (iter > 0) load_step(iter - 1);
            // ---------------- forward quantities on the own tiles: e (kept, scaled by 2^sh), partial rt ----------------
            f16x8 bh, bl;
            {
                float ms[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) ms[i] = mcv[i] * SM;
                split8(ms, bh, bl);
            }
            f32x4 ep[8];
            float t256 = -3.0e38f;
            if (h) {
                float d256 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) d256 = __builtin_fmaf(mcv[i], lds[P_D256 + 8 * g + i], d256);
                d256 = rows_sum4(d256);
                t256 = logx256 + d256;
            }
            float tmax = t256;
            {
                f16x8 al[4], ah[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { al[i] = DL[i * 64]; ah[i] = DH[i * 64]; }
                f32x4 c[4] = {zero4, zero4, zero4, zero4}, pc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const bool pm = q < 2, vw = q > 0;
                    f16x8 ah_n[4] = {ah[0], ah[1], ah[2], ah[3]};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (q < 1) ah_n[i] = DH[(4 * q + 4 + i) * 64];
                        if (pm) c[i] = mfma_h(al[i], bh, zero4);
                        DSA_SB();
                        if (vw) {
                            const f32x2v ta = fma2(lo2(pc[i]), kInvSDM, lo2(logx[4 * q - 4 + i]));
                            const f32x2v tb = fma2(hi2(pc[i]), kInvSDM, hi2(logx[4 * q - 4 + i]));
                            ep[4 * q - 4 + i] = f32x4{ta[0], ta[1], tb[0], tb[1]};
                        }
                        DSA_SB();
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (pm) c[i] = mfma_h(ah[i], bl, c[i]);
                        if (q < 1) al[i] = DL[(4 * q + 4 + i) * 64];
                        DSA_SB();
                        if (vw) {
                            const f32x4 v = ep[4 * q - 4 + i];
                            tmax = __builtin_fmaxf(__builtin_fmaxf(tmax, v[0]), v[1]);
                            tmax = __builtin_fmaxf(__builtin_fmaxf(tmax, v[2]), v[3]);
                        }
                        DSA_SB();
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (pm) c[i] = mfma_h(ah[i], bh, c[i]);
                        DSA_SB();
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) { pc[i] = c[i]; ah[i] = ah_n[i]; }
                }
            }
            f16x8 eah[3], eal[3];
#pragma unroll
            for (int it = 0; it < 3; ++it) { eah[it] = EH[(it * 8) * 64]; eal[it] = EL[(it * 8) * 64]; }
            tmax = rows_max4(tmax);
            const float mi = __builtin_ceilf(tmax);
            const float sh = (float)EMAX_LOG2 - mi;
            const int back = (int)mi - EMAX_LOG2;   // e = 2^back ep (this half's own scale)
            const float e256 = h ? __builtin_amdgcn_exp2f(t256 + sh) : 0.f;
            f32x4 accB[3];
#pragma unroll
            for (int it = 0; it < 3; ++it) accB[it] = *reinterpret_cast<const f32x4*>(lds + P_E256 + it * 16 + 4 * g) * e256;   // Nyquist: h = 1
            f32x2v rt48v = {0.f, 0.f};
            float rt48 = 0.f;
            f16x8 eh_p = {}, el_p = {};
            DSA_SB();
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                f16x8 eh = eh_p, el = el_p;
                f16x8 eah_n[3] = {eah[0], eah[1], eah[2]};
                f32x4 c48[2] = {zero4, zero4};
                if (j < 4) {
                    c48[0] = E484[(2 * j) * 4 + g];
                    c48[1] = E484[(2 * j + 1) * 4 + g];
                }
                if (j > 0 && j < 4) {
#pragma unroll
                    for (int it = 0; it < 3; ++it) eah_n[it] = EH[(it * 8 + j) * 64];
                }
                auto prodE = [&](int i) __attribute__((always_inline)) {
                    if (j > 0) {
                        const int it = i % 3, term = i / 3;
                        accB[it] = mfma_h(term == 0 ? eal[it] : eah[it], term == 1 ? el_p : eh_p, accB[it]);
                    }
                };
                auto vecA = [&](int t_) __attribute__((always_inline)) {
                    const int mt = (2 * j + t_) & 7;
                    const f32x2v ta = lo2(ep[mt]) + f32x2v{sh, sh}, tb = hi2(ep[mt]) + f32x2v{sh, sh};
                    ep[mt] = f32x4{__builtin_amdgcn_exp2f(ta[0]), __builtin_amdgcn_exp2f(ta[1]), __builtin_amdgcn_exp2f(tb[0]),
                                   __builtin_amdgcn_exp2f(tb[1])};
                };
                auto vecD = [&](int t_) __attribute__((always_inline)) {
                    const int mt = (2 * j + t_) & 7;
                    rt48v = lo2(ep[mt]) * lo2(c48[t_]) + rt48v;
                    rt48v = hi2(ep[mt]) * hi2(c48[t_]) + rt48v;
                };
                auto vecE = [&](int t_, int r) __attribute__((always_inline)) {
                    const int mt = (2 * j + t_) & 7;
                    f16x2 hh, ll;
                    split2(ep[mt][r], ep[mt][r + 1], hh, ll);
                    eh[4 * t_ + r] = hh[0]; eh[4 * t_ + r + 1] = hh[1];
                    el[4 * t_ + r] = ll[0]; el[4 * t_ + r + 1] = ll[1];
                };
                const bool vw = j < 4;
                prodE(0); DSA_SB(); if (vw) vecA(0); DSA_SB();
                prodE(1); DSA_SB(); if (vw) vecA(1); DSA_SB();
                if (j == 4) rt48 = rows_sum4(rt48v[0] + rt48v[1]);
                prodE(2);
                if (j > 0 && j < 4) {
#pragma unroll
                    for (int it = 0; it < 3; ++it) eal[it] = EL[(it * 8 + j) * 64];
                }
                DSA_SB(); if (vw) vecD(0); DSA_SB();
                prodE(3); DSA_SB(); if (vw) vecE(0, 0); DSA_SB();
                prodE(4); DSA_SB(); if (vw) vecD(1); DSA_SB();
                prodE(5); DSA_SB(); if (vw) vecE(0, 2); DSA_SB();
                prodE(6); DSA_SB(); if (vw) vecE(1, 0); DSA_SB();
                prodE(7); DSA_SB(); if (vw) vecE(1, 2); DSA_SB();
                prodE(8); DSA_SB();
                eh_p = eh; el_p = el;
#pragma unroll
                for (int it = 0; it < 3; ++it) eah[it] = eah_n[it];
            }
            rt48 = __builtin_fmaf(e256, lds[P_E256 + 48], rt48);   // (e256 = 0 on h = 0)
            rt48 = __builtin_ldexpf(rt48, back);
            // ---------------- the two partial rt meet: h = 1 -> hand-over area -> h = 0 -> the pair's windows ----------------
            {
                const int bk = back - SE_LOG2;
                float v[3][4];
#pragma unroll
                for (int it = 0; it < 3; ++it)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[it][r] = __builtin_ldexpf(accB[it][r], bk);
                if (h) {
#pragma unroll
                    for (int it = 0; it < 3; ++it) *reinterpret_cast<f32x4*>(x_n + 16 * it + 4 * g) = f32x4{v[it][0], v[it][1], v[it][2], v[it][3]};
                    __builtin_amdgcn_wave_barrier();
                    if (g == 0) x_n[48] = rt48;
                    pair_signal(f10, ++n10, lane);
                    ++n01;
                    pair_wait(f01, n01);       // the windows hold rt
                } else {
                    ++n10;
                    pair_wait(f10, n10);
#pragma unroll
                    for (int it = 0; it < 3; ++it) {
                        const f32x4 p1 = *reinterpret_cast<const f32x4*>(x_n + 16 * it + 4 * g);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[it][r] += p1[r];
                    }
                    rt48 += x_n[48];
                    int g_it = g;
                    asm volatile("" : "+v"(g_it));
                    float* rtw = rt_n + 4 * g_it;
                    float* rra = rr_n + 27 + 4 * g_it;
                    float* rrb = rr_n + 27 - 4 * g_it;
                    float* rra1 = g_it < 3 ? rra + 16 : rr_n + 55;
                    float* rrb1 = g_it < 3 ? rrb - 16 : rr_n + 62;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        rtw[r] = v[0][r];
                        rra[r] = v[0][r];
                        rrb[-r] = v[0][r];
                        rtw[16 + r] = v[1][r];
                        rra1[r] = v[1][r];
                        rrb1[-r] = v[1][r];
                        rtw[32 + r] = v[2][r];
                    }
                    rt_n[48] = rt48;
                    pair_signal(f01, ++n01, lane);
                }
            }
            // mbar to this wave's exchange window (C/D layout writer -> quad-layout reader)
#pragma unroll
            for (int it2 = 0; it2 < 2; ++it2) *reinterpret_cast<f32x4*>(aux_n + it2 * 16 + 4 * g) = mbarC[it2];
            __builtin_amdgcn_wave_barrier();

            // ---------------- solve A [gv | uv] = [rt[:25] - alpha | mbar] in the quad layout (both waves, the same system) ----------------
            float xq1[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, keep_if(gq.m[1], -1.f)};
            float xq2[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, keep_if(gq.m[2], -1.f)};
            {
                f32x4 a[blk::NBLK];
                float ninvs[M1];
                {
                    int gsv = gs;
                    asm volatile("" : "+v"(gsv));
                    const float* zr = lds + P_ZERO;
                    const float* pa6 = gsv == 0 ? rt_q + 24 : (gsv == 1 ? rt_q : (gsv == 2 ? aux_q : zr));
                    const float* pb6 = gsv == 0 ? rr_q + 3 : (gsv == 1 ? lds + P_NAV : zr);
                    blk_build_rows<0>(a, rt_q, rr_q, pa6, pb6, gs);
                }
                __builtin_amdgcn_wave_barrier();
                if (!(PAIR_ABL & 8)) blk_elim_all(a, gq, ninvs, std::make_integer_sequence<int, M1>{});
                else { for (int k = 0; k < M1; ++k) ninvs[k] = a[k % blk::NBLK][0]; }
                if (g_saved) {
#pragma unroll
                    for (int c = 0; c < KS; ++c) xq1[c] = gh[c];
                } else {
                    blk_backsub_all(a, xq1, gq, ninvs, std::make_integer_sequence<int, blk::NG>{});
                    xq1[KS - 1] = keep_if(gq.m[0], xq1[KS - 1]);
                }
                blk_backsub_all(a, xq2, gq, ninvs, std::make_integer_sequence<int, blk::NG>{});
                xq2[KS - 1] = keep_if(gq.m[0], xq2[KS - 1]);
            }
            // ---------------- rtbar (49 entries), scaled per frame to below 2^13, into the exchange window (as mcep_mfma_bwd_kernel_h) ----------------
            int s_r;
            {
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                f32x4 DHk[13], DTk[13];
#pragma unroll
                for (int k = 0; k < 13; ++k) { DHk[k] = z4; DTk[k] = z4; }
#pragma unroll
                for (int ri = 0; ri < ((PAIR_ABL & 4) ? 1 : KS); ++ri)
#pragma unroll
                    for (int cj = 0; cj < KS; ++cj) {
                        DHk[ri + cj] = mfma441(xq2[ri], xq1[cj], DHk[ri + cj]);
                        DTk[cj - ri + 6] = mfma441(xq2[ri], xq1[cj], DTk[cj - ri + 6]);
                    }
                auto rotR = [](float v, int k) __attribute__((always_inline)) {
                    return k == 1 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x93, 0xf, 0xf, true))
                           : k == 2 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true))
                                    : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x39, 0xf, 0xf, true));
                };
                auto rotL = [](float v, int k) __attribute__((always_inline)) {
                    return k == 1 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x39, 0xf, 0xf, true))
                           : k == 2 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true))
                                    : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x93, 0xf, 0xf, true));
                };
                auto refl = [](float v, int k) __attribute__((always_inline)) {
                    return k == 0 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x6C, 0xf, 0xf, true))
                           : k == 1 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true))
                           : k == 2 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xC6, 0xf, 0xf, true))
                                    : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x1B, 0xf, 0xf, true));
                };
                float rb[13];
#pragma unroll
                for (int sl = 0; sl < 13; ++sl) {
                    float hs = DHk[sl][0];
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
                        if (sl == 0) tn = gs == 0 ? 0.f : tn;
                        r = r - tp - tn + xq2[sl];
                    }
                    rb[sl] = r;
                }
                float amax = 0.f;
#pragma unroll
                for (int sl = 0; sl < 13; ++sl) amax = __builtin_fmaxf(amax, __builtin_fabsf(rb[sl]));
                amax = __builtin_fmaxf(amax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(amax), 0xB1, 0xf, 0xf, true)));
                amax = __builtin_fmaxf(amax, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(amax), 0x4E, 0xf, 0xf, true)));
                s_r = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(amax);
#pragma unroll
                for (int sl = 0; sl < 13; ++sl) aux_q[4 * sl + gs] = __builtin_ldexpf(rb[sl], s_r);
                aux_q[52 + gs] = 0.f;
                aux_q[56 + gs] = 0.f;
                aux_q[60 + gs] = gs == 3 ? __int_as_float(s_r) : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            f16x8 rbh[2], rbl[2];
            float eb256 = 0.f;
            const int s_rn = __float_as_int(aux_n[63]);
            {
                float rv[16];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    rv[i] = aux_n[8 * g + i];
                    rv[8 + i] = g < 3 ? aux_n[32 + 8 * g + i] : 0.f;
                }
                if (h) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        eb256 = __builtin_fmaf(rv[i], lds[P_E256 + 52 + 8 * g + i], eb256);
                        eb256 = __builtin_fmaf(rv[8 + i], lds[P_E256 + 52 + 32 + 8 * g + i], eb256);
                    }
                }
                float lo8[8], hi8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { lo8[i] = rv[i]; hi8[i] = rv[8 + i]; }
                split8(lo8, rbh[0], rbl[0]);
                split8(hi8, rbh[1], rbl[1]);
            }
            __builtin_amdgcn_wave_barrier();
            if (h) eb256 = rows_sum4(eb256);

            // ---------------- ebar^T = E rtbar^T on the own tiles ; zbar = ebar * e ; lbar += zbar ----------------
            const int kz = back - s_rn - SEB_LOG2;
            f32x4 zb[8];
            float zmax = 0.f;
            {
                f16x8 ah[4][2], al[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        ah[i][ks] = PLOAD(eb_half + 2 * (IMG_EBH + (i * 2 + ks) * 512));
                        al[i][ks] = PLOAD(eb_half + 2 * (IMG_EBL + (i * 2 + ks) * 512));
                    }
                f32x4 acc[4] = {zero4, zero4, zero4, zero4}, pacc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const bool pm = q < 2, vw = q > 0;
                    f16x8 ah_n[4][2], al_n[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) { ah_n[i][ks] = ah[i][ks]; al_n[i][ks] = al[i][ks]; }
                    auto vec = [&](int i, int half) __attribute__((always_inline)) {
                        const int mt = 4 * q - 4 + i;
                        const f32x2v a2 = half ? hi2(pacc[i]) : lo2(pacc[i]), e2 = half ? hi2(ep[mt]) : lo2(ep[mt]);
                        const f32x2v m2 = a2 * e2;
                        const float z0 = __builtin_ldexpf(m2[0], kz), z1 = __builtin_ldexpf(m2[1], kz);
                        zb[mt][2 * half] = z0; zb[mt][2 * half + 1] = z1;
                        const f32x2v l2 = (half ? hi2(lbar[mt]) : lo2(lbar[mt])) + f32x2v{z0, z1};
                        lbar[mt][2 * half] = l2[0]; lbar[mt][2 * half + 1] = l2[1];
                        zmax = __builtin_fmaxf(__builtin_fmaxf(zmax, __builtin_fabsf(z0)), __builtin_fabsf(z1));
                    };
#pragma unroll
                    for (int term = 0; term < 6; ++term) {
                        const int ks = term / 3, tr = term % 3;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (pm) acc[i] = mfma_h(tr == 0 ? al[i][ks] : ah[i][ks], tr == 1 ? rbl[ks] : rbh[ks], term == 0 ? zero4 : acc[i]);
                            if (q < 1 && tr == 0) al_n[i][ks] = PLOAD(eb_half + 2 * (IMG_EBL + ((4 * q + 4 + i) * 2 + ks) * 512));
                            if (q < 1 && tr == 2) ah_n[i][ks] = PLOAD(eb_half + 2 * (IMG_EBH + ((4 * q + 4 + i) * 2 + ks) * 512));
                            DSA_SB();
                            if (vw && term < 2) vec(i, term);
                            DSA_SB();
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        pacc[i] = acc[i];
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) { ah[i][ks] = ah_n[i][ks]; al[i][ks] = al_n[i][ks]; }
                    }
                }
            }
            const float zb256 = h ? __builtin_ldexpf(eb256 * e256, back - s_rn) : 0.f;
            lbar256 += zb256;
            // ---------------- partial mbar^T = (-2 D) zbar^T over the own bins (streamed image), then the two partials meet ----------------
            zmax = rows_max4(zmax);
            const int s_z = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(zmax);
            {
                f32x4 acc2[2][2] = {{zero4, zero4}, {zero4, zero4}};
                f16x8 zh_p = {}, zl_p = {};
                f16x8 dh[2], dl[2];
#pragma unroll
                for (int it2 = 0; it2 < 2; ++it2) {
                    dh[it2] = PLOAD(db_half + 2 * (IMG_DBH + (it2 * 8) * 512));
                    dl[it2] = PLOAD(db_half + 2 * (IMG_DBL + (it2 * 8) * 512));
                }
                DSA_SB();
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    f16x8 zh = zh_p, zl = zl_p;
                    f16x8 dh_n[2] = {dh[0], dh[1]}, dl_n[2] = {dl[0], dl[1]};
                    if (j > 0 && j < 4) {
#pragma unroll
                        for (int it2 = 0; it2 < 2; ++it2) {
                            dh_n[it2] = PLOAD(db_half + 2 * (IMG_DBH + (it2 * 8 + j) * 512));
                            dl_n[it2] = PLOAD(db_half + 2 * (IMG_DBL + (it2 * 8 + j) * 512));
                        }
                    }
                    auto prodM = [&](int i) __attribute__((always_inline)) {
                        if (j > 0) {
                            const int it2 = i & 1, term = i >> 1;
                            f32x4& a2 = acc2[it2][(j - 1) & 1];
                            a2 = mfma_h(term == 0 ? dl[it2] : dh[it2], term == 1 ? zl_p : zh_p, a2);
                        }
                    };
                    auto vecZ = [&](int t_, int r) __attribute__((always_inline)) {
                        const int mt = (2 * j + t_) & 7;
                        f16x2 hh, ll;
                        split2(__builtin_ldexpf(zb[mt][r], s_z), __builtin_ldexpf(zb[mt][r + 1], s_z), hh, ll);
                        zh[4 * t_ + r] = hh[0]; zh[4 * t_ + r + 1] = hh[1];
                        zl[4 * t_ + r] = ll[0]; zl[4 * t_ + r + 1] = ll[1];
                    };
                    const bool vw = j < 4;
                    prodM(0); DSA_SB(); if (vw) vecZ(0, 0); DSA_SB();
                    prodM(1); DSA_SB(); if (vw) vecZ(0, 2); DSA_SB();
                    prodM(2); DSA_SB(); if (vw) vecZ(1, 0); DSA_SB();
                    prodM(3); DSA_SB(); if (vw) vecZ(1, 2); DSA_SB();
                    prodM(4); DSA_SB();
                    prodM(5); DSA_SB();
                    zh_p = zh; zl_p = zl;
#pragma unroll
                    for (int it2 = 0; it2 < 2; ++it2) { dh[it2] = dh_n[it2]; dl[it2] = dl_n[it2]; }
                }
                f32x4 part[2];
#pragma unroll
                for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = it2 * 16 + 4 * g + r;
                        part[it2][r] = __builtin_ldexpf(acc2[it2][0][r] + acc2[it2][1][r], -s_z - SDB_LOG2);
                        part[it2][r] = __builtin_fmaf(zb256, lds[P_D256 + 32 + c], part[it2][r]);   // Nyquist bin (zb256 = 0 on h = 0)
                    }
                if (h) {
#pragma unroll
                    for (int it2 = 0; it2 < 2; ++it2) *reinterpret_cast<f32x4*>(x_n + it2 * 16 + 4 * g) = part[it2];
                    pair_signal(f10, ++n10, lane);
                    ++n01;
                    pair_wait(f01, n01);       // the area holds the new mbar
#pragma unroll
                    for (int it2 = 0; it2 < 2; ++it2) mbarC[it2] = *reinterpret_cast<const f32x4*>(x_n + it2 * 16 + 4 * g);
                    __builtin_amdgcn_wave_barrier();
                } else {
                    ++n10;
                    pair_wait(f10, n10);
#pragma unroll
                    for (int it2 = 0; it2 < 2; ++it2) {
                        const f32x4 p1 = *reinterpret_cast<const f32x4*>(x_n + it2 * 16 + 4 * g);
                        mbarC[it2] += part[it2] + p1;
                        *reinterpret_cast<f32x4*>(x_n + it2 * 16 + 4 * g) = mbarC[it2];
                    }
                    pair_signal(f01, ++n01, lane);
                }
            }
        }
        // ---------------- lbar += G mbar_0 on the own tiles (mcep.py:204-207 adjoint); gX = lbar / X ----------------
#pragma unroll
        for (int it2 = 0; it2 < 2; ++it2) *reinterpret_cast<f32x4*>(aux_n + it2 * 16 + 4 * g) = mbarC[it2];
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
        if (h) {
            float part256 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) part256 = __builtin_fmaf(m0[i], tail_f[8 * g + i], part256);
            part256 = rows_sum4(part256);
            lbar256 += part256;
        }
        {
            float ms[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ms[i] = __builtin_ldexpf(m0[i], s_m);
            f16x8 mh8, ml8;
            split8(ms, mh8, ml8);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const f16x8 ah = PLOAD(gb_half + 2 * (IMG_GBH + mt * 512)), al = PLOAD(gb_half + 2 * (IMG_GBL + mt * 512));
                f32x4 acc = {0, 0, 0, 0};
                acc = mfma_h(al, mh8, acc);
                acc = mfma_h(ah, ml8, acc);
                acc = mfma_h(ah, mh8, acc);
                if (f_ok) {
                    float* dst = gX + f * K + h * 128 + mt * 16 + 4 * g;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dst[r] = (lbar[mt][r] + __builtin_ldexpf(acc[r], -s_m - SGB_LOG2)) * __builtin_amdgcn_exp2f(-logx[mt][r]);
                }
            }
        }
        if (h && f_ok && g == 0) gX[f * K + H] = lbar256 * __builtin_amdgcn_exp2f(-logx256);
    }
#undef DSA_SB
}

}  // namespace dsa
