// EXPERIMENT (round 2), NOT in the product build: backward of the tuned mel-cepstral analysis at TWO waves per SIMD.
// To try it, include it from csrc/mcep_mfma.hip after mcep_mfma_bwd_f16.h (whose mathematics, operand images and
// scalings it shares) and launch mcep_mfma_bwd_kernel_s<0> with 512 threads and mhs::S_LDS_FLOATS * 4 bytes of LDS.
// Measured on MI355X, 204 800 frames, n_iter 10 (tools/bench_fwdbwd.py 1024): 2.91 ms against 2.24 ms of the
// one-wave-per-SIMD kernel it was meant to replace.  Ablations (ABL mask): without the read-modify-write of lbar in
// gX 1.90 ms; with the two streamed images read from LDS instead of L2 2.36 ms; both 1.57 ms; without the solve
// 1.79 ms (it spills: 340 B of scratch at 256 registers); all three 1.14 ms.  I.e. the vector work alone
// (chains 1.14 ms + a spill-free two-right-hand-side solve ~0.6 ms) is ~1.75 ms at perfect two-wave overlap, and
// what the second wave buys is paid back by the L2 round trips (lbar accumulated in memory, 96 KB of operand
// images per tile and step with eight waves per CU pulling them).  It is also not bit-reproducible across
// concurrent streams (tests/test_gpu_configs.py::test_mcep_many_launches_two_tables_four_streams), which the
// resident-register kernel is.  Kept for the ideas that did work: the RUNNING per-frame scale of the zbar chain
// (no stored zbar, no second pass) and the e re-derivation in the adjoint phase.
//
// mcep_mfma_bwd_kernel_h keeps four per-bin arrays of its 16 frames in registers for the whole sweep -- log2 X,
// lbar (the accumulated cotangent of log X), e and zbar: 256 of its 434 registers -- and therefore runs ONE wave per
// SIMD: nothing overlaps its LDS / L2 / matrix-core latencies (PMC: VALU busy 41 %, the wave waits 39 % of its
// cycles), and a quarter of its vector instructions only shuttle values between the two register-file halves.
// Here a wave keeps ONE of them:
//   log2 X   stays in registers (64);
//   e        is not kept across the solve: the adjoint phase re-derives it from log2 X with one more pass of the
//            (cheap, binary16 matrix-core) first chain -- its scale 2^sh is known from the forward phase;
//   zbar     is not kept at all: the chain mbar -= 2 D zbar^T consumes it 32 bins at a time with a RUNNING
//            per-frame power-of-two scale (when a later bin group raises the frame's maximum, the partial sums are
//            rescaled -- exact, powers of two -- like an online softmax);
//   lbar     accumulates in the OUTPUT buffer gX itself: every Newton step reads, adds to and rewrites the frame's
//            row (1028 B each way per frame and step, L2 / Infinity-Cache resident: the tile is private to the wave).
// That is <= 256 registers: two waves per SIMD (8 per workgroup), each hiding the other's latencies.  LDS holds
// the two forward images (80 KB) and 9.75 KB of per-frame windows per wave (the rtbar exchange reuses the rt / rr
// region, dead once the system is assembled); the transposed E image (64 KB) and the -2 D image (32 KB) are
// streamed from L2 in operand order.
#pragma once

namespace dsa {

namespace mhs {
using namespace mhb;
constexpr int WAVES_S = 8;
// LDS carve-up (float units): forward images as in mh, then the small tables, then the per-wave windows
constexpr int S_E48 = EL_OFF + 24 * 64 * 4;          // [16 mt][4 g][4 r]  E[bin][48]
constexpr int S_E256 = S_E48 + 256;                  // [48] SE E[256][m] | [48] = E[256][48] | +52: [64] E[256][m] (0 past 48)
constexpr int S_D256 = S_E256 + 52 + 64;             // [32] -2 log2(e) D[c][256] | [32] -2 D[c][256]
constexpr int S_AV = S_D256 + 64;                    // [28]
constexpr int S_WAVE = S_AV + 28;
constexpr int FS2 = 156;                             // per-frame record: rt [0,52) | rr [52,116) | aux [116,156)
constexpr int S_WAVE_FLOATS = 16 * FS2;
constexpr int S_LDS_FLOATS = S_WAVE + WAVES_S * S_WAVE_FLOATS;
static_assert(S_LDS_FLOATS * 4 <= 160 * 1024, "backward kernel: LDS carve-up exceeds 160 KB");
}  // namespace mhs

typedef float f32x4_u4 __attribute__((ext_vector_type(4), aligned(4)));   // four floats at a 4-byte aligned address

// ABL (ablation bit mask for tools/bench_fwdbwd.py via DSA_MCEP_BWD_ABL; 0 in the product): 1 no lbar read-modify-write |
// 2 backward images read from the LDS-resident forward images instead of L2 (wrong numbers, right instruction mix) |
// 4 no solve
template <int ABL>
__global__ __launch_bounds__(512, 2) void mcep_mfma_bwd_kernel_s(
    const float* __restrict__ gmc, const float* __restrict__ X, const float* __restrict__ hist, long F, int n_iter,
    const float* __restrict__ av, float* __restrict__ gX, long ntiles16, unsigned int* __restrict__ queue,
    const _Float16* __restrict__ img)
{
    using namespace mhs;
    constexpr float kInvSDM = 1.f / (SD * SM);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;

    // ---------------- forward operand images and small tables ----------------
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(img);
        f32x4* dst = reinterpret_cast<f32x4*>(lds + DH_OFF);
        for (int idx = tid; idx < (2 * IMG_D + 2 * IMG_E) / 8; idx += WAVES_S * 64) dst[idx] = src[idx];
    }
    const float* tail_f = reinterpret_cast<const float*>(img + IMG_HALVES);     // G[256][c] (forward workspace tail)
    const float* tail_b = reinterpret_cast<const float*>(img + IMG_B_HALVES);   // -2 D[c][256] | E[256][m] | E[bin][48]
    if (tid < 256) {
        const int r = tid & 3, gg = (tid >> 2) & 3, mt = tid >> 4;
        lds[S_E48 + tid] = tail_b[96 + mt * 16 + gg * 4 + r];
    }
    if (tid < 48) lds[S_E256 + tid] = SE * tail_b[32 + tid];     // Nyquist k-step of the rt chain (scaled image)
    if (tid == 48) lds[S_E256 + 48] = tail_b[32 + 48];            // rt[48] is a float32 dot product
    if (tid < 64) lds[S_E256 + 52 + tid] = tail_b[32 + tid];      // E[256][m], m < 64 (0 past 48): ebar of the Nyquist bin
    if (tid < 32) {
        lds[S_D256 + tid] = 1.4426950408889634f * tail_b[tid];    // -2 log2(e) D[c][256]
        lds[S_D256 + 32 + tid] = tail_b[tid];                     // -2 D[c][256]
    }
    if (tid < 28) lds[S_AV + tid] = tid < M1 ? av[tid] : 0.f;
    __syncthreads();   // the only workgroup barrier

    float* wave_lds = lds + S_WAVE + wave * S_WAVE_FLOATS;
    float* rt_n = wave_lds + n * FS2;          // this lane's frame, MFMA-layout view
    float* rr_n = rt_n + 52;
    float* aux_n = rt_n + 116;
    const int nq = lane >> 2, gs = lane & 3;   // solve layout: a quad per frame
    float* rt_q = wave_lds + nq * FS2;
    float* rr_q = rt_q + 52;
    float* aux_q = rt_q + 116;
    const GroupMask gq = make_group_mask(gs);
    const unsigned g_eq0 = g == 0 ? 0xffffffffu : 0u;
    int lane_a = lane, lane_b = lane + EL_OFF / 4;
    asm volatile("" : "+v"(lane_a), "+v"(lane_b));
    const f16x8* DH = reinterpret_cast<const f16x8*>(lds + DH_OFF) + lane_a;
    const f16x8* DL = reinterpret_cast<const f16x8*>(lds + DL_OFF) + lane_a;
    const f16x8* EH = reinterpret_cast<const f16x8*>(lds + EH_OFF) + lane_a;
    const f16x8* EL = reinterpret_cast<const f16x8*>(lds) + lane_b;
    const f32x4* E484 = reinterpret_cast<const f32x4*>(lds + S_E48);
    const long wave_id = (long)blockIdx.x * WAVES_S + wave;
    const long wave_stride = (long)gridDim.x * WAVES_S;

    for (long tile = wave_id; tile < ntiles16;) {
        const long f_raw = tile * 16 + n;
        const bool f_ok = f_raw < F;
        const long f = f_ok ? f_raw : F - 1;
        const float* xf = X + f * K;
        float* gxf = gX + f * K;
        f32x4 logx[16];
#pragma unroll
        for (int mt = 0; mt < 16; ++mt) {
            const float* p = xf + mt * 16 + 4 * g;
            logx[mt] = f32x4{__log2f(p[0]), __log2f(p[1]), __log2f(p[2]), __log2f(p[3])};
        }
        const float logx256 = __log2f(xf[H]);
        float lbar256 = 0.f;
        // mbar in the C/D layout of a 32-row product: tile it2, register r <-> coefficient 16 it2 + 4 g + r
        f32x4 mbarC[2];
#pragma unroll
        for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = it2 * 16 + 4 * g + r;
                mbarC[it2][r] = c < M1 ? gmc[f * M1 + c] : 0.f;
            }
        unsigned int nxt = 0;
        if (lane == 0) nxt = atomicAdd(queue, 1u);
        const long tile_next = wave_stride + (long)__builtin_amdgcn_readfirstlane((int)nxt);

        for (int iter = n_iter - 1; iter >= 0; --iter) {
            float mcv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) mcv[i] = (8 * g + i < M1) ? hist[((long)iter * F + f) * M1 + 8 * g + i] : 0.f;
            // ================= forward quantities of this step (the forward kernel's two passes) =================
            f16x8 bh, bl;
            {
                float ms[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) ms[i] = mcv[i] * SM;
                split8(ms, bh, bl);
            }
            auto dtile = [&](int mt, float cinit) __attribute__((always_inline)) {
                const f16x8 ah = DH[mt * 64], al = DL[mt * 64];
                f32x4 c = {cinit, cinit, cinit, cinit};
                c = mfma_h(al, bh, c);
                c = mfma_h(ah, bl, c);
                c = mfma_h(ah, bh, c);
                return c;
            };
            float d256 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) d256 = __builtin_fmaf(mcv[i], lds[S_D256 + 8 * g + i], d256);
            d256 += __shfl_xor(d256, 16, 64);
            d256 += __shfl_xor(d256, 32, 64);
            const float t256 = logx256 + d256;
            float tmax = t256;
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {   // pass 1: the frame's maximum of t (t itself is not kept)
                const f32x4 c = dtile(mt, 0.f);
                const f32x2v ta = fma2(lo2(c), kInvSDM, lo2(logx[mt])), tb = fma2(hi2(c), kInvSDM, hi2(logx[mt]));
                tmax = __builtin_fmaxf(tmax, __builtin_fmaxf(__builtin_fmaxf(ta[0], ta[1]), __builtin_fmaxf(tb[0], tb[1])));
            }
            tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float mi = __builtin_ceilf(tmax);
            const float sh = (float)EMAX_LOG2 - mi;
            const int back = (int)mi - EMAX_LOG2;   // e = 2^back (scaled e)
            const float cinit = sh * (SD * SM);     // the shift rides in the accumulator preload
            const float e256 = __builtin_amdgcn_exp2f(t256 + sh);   // scaled like e
            {
                f32x4 accB[3] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
                f32x2v rt48v = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 8; ++j) {   // pass 2: e = exp2(t + sh), rt^T += E^T e^T
                    f16x8 eh, el;
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int mt = 2 * j + tt;
                        const f32x4 c = dtile(mt, cinit);
                        const f32x4 c48 = E484[mt * 4 + g];
                        const f32x2v ta = fma2(lo2(c), kInvSDM, lo2(logx[mt])), tb = fma2(hi2(c), kInvSDM, hi2(logx[mt]));
                        const float e[4] = {__builtin_amdgcn_exp2f(ta[0]), __builtin_amdgcn_exp2f(ta[1]),
                                            __builtin_amdgcn_exp2f(tb[0]), __builtin_amdgcn_exp2f(tb[1])};
                        rt48v = f32x2v{e[0], e[1]} * lo2(c48) + rt48v;
                        rt48v = f32x2v{e[2], e[3]} * hi2(c48) + rt48v;
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            f16x2 h, l;
                            split2(e[r], e[r + 1], h, l);
                            eh[4 * tt + r] = h[0]; eh[4 * tt + r + 1] = h[1];
                            el[4 * tt + r] = l[0]; el[4 * tt + r + 1] = l[1];
                        }
                    }
#pragma unroll
                    for (int it = 0; it < 3; ++it) {
                        const f16x8 ah = EH[(it * 8 + j) * 64], al = EL[(it * 8 + j) * 64];
                        accB[it] = mfma_h(al, eh, accB[it]);
                        accB[it] = mfma_h(ah, el, accB[it]);
                        accB[it] = mfma_h(ah, eh, accB[it]);
                    }
                }
#pragma unroll
                for (int it = 0; it < 3; ++it)
                    accB[it] = mfma4(keep_if(g_eq0, lds[S_E256 + it * 16 + n]), keep_if(g_eq0, e256), accB[it]);
                float rt48 = rt48v[0] + rt48v[1];
                rt48 += __shfl_xor(rt48, 16, 64);
                rt48 += __shfl_xor(rt48, 32, 64);
                rt48 = __builtin_fmaf(e256, lds[S_E256 + 48], rt48);
                rt48 = __builtin_ldexpf(rt48, back);
                int g_it = g;
                asm volatile("" : "+v"(g_it));
                float* rtw = rt_n + 4 * g_it;
                float* rra = rr_n + 27 + 4 * g_it;
                float* rrb = rr_n + 27 - 4 * g_it;
                float* rra1 = g_it < 3 ? rra + 16 : rr_n + 55;
                float* rrb1 = g_it < 3 ? rrb - 16 : rr_n + 62;
                const int bk = back - SE_LOG2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v0 = __builtin_ldexpf(accB[0][r], bk);
                    const float v1 = __builtin_ldexpf(accB[1][r], bk);
                    rtw[r] = v0;
                    rra[r] = v0;
                    rrb[-r] = v0;
                    rtw[16 + r] = v1;
                    rra1[r] = v1;
                    rrb1[-r] = v1;
                    rtw[32 + r] = __builtin_ldexpf(accB[2][r], bk);
                }
                rt_n[48] = rt48;
                // mbar to the exchange window (C/D layout writer -> quad-layout reader)
#pragma unroll
                for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) aux_n[it2 * 16 + 4 * g_it + r] = mbarC[it2][r];
            }
            __builtin_amdgcn_wave_barrier();

            // ================= solve A [gv | uv] = [rt[:25] - alpha | mbar] in the quad layout =================
            float xq1[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, keep_if(gq.m[1], -1.f)};
            float xq2[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, keep_if(gq.m[2], -1.f)};
            {
                float a[colm::TOTAL];
                col_build_rows<0>(a, rt_q, rr_q, lds + S_AV, aux_q, gs, gq);
                __builtin_amdgcn_wave_barrier();
                if (!(ABL & 4)) {
                    col_elim_all(a, std::make_integer_sequence<int, M1>{});
                    col_backsub_all(a, xq1, gq, std::make_integer_sequence<int, M1>{});
                    col_backsub_all(a, xq2, gq, std::make_integer_sequence<int, M1>{});
                } else {
#pragma unroll
                    for (int c = 0; c < KS; ++c) xq1[c] = a[c], xq2[c] = a[c + 7];
                }
                xq1[KS - 1] = keep_if(gq.m[0], xq1[KS - 1]);
                xq2[KS - 1] = keep_if(gq.m[0], xq2[KS - 1]);
            }
            // ================= rtbar (49 entries), scaled per frame to below 2^13, into the rt / rr region =================
            {
                const float (&uq)[KS] = xq2;
                // exchange window: [0,3) zeros | g[0..24] at 3..27 | zeros to 33
#pragma unroll
                for (int k = 0; k < 3; ++k) aux_q[k] = 0.f;
#pragma unroll
                for (int c = 0; c < KS; ++c) aux_q[3 + gs + 4 * c] = xq1[c];
#pragma unroll
                for (int k = 31; k < 34; ++k) aux_q[k] = 0.f;
                __builtin_amdgcn_wave_barrier();
                float gsh1[28], gsh2[31];                      // gsh2 index k + 3, k = -3 .. 27
                const float* w1 = aux_q + 3 - gs;
                const float* w2 = aux_q + gs;
#pragma unroll
                for (int k = 0; k < 28; ++k) gsh1[k] = w1[k];
#pragma unroll
                for (int k = 0; k < 31; ++k) gsh2[k] = w2[k];
                __builtin_amdgcn_wave_barrier();
                float rb[M2];
#pragma unroll
                for (int m = 0; m < M2; ++m) {
                    float acc = 0.f;
#pragma unroll
                    for (int c = 0; c < KS; ++c) {
                        if (m - 4 * c >= 0 && m - 4 * c <= 27) acc = __builtin_fmaf(-uq[c], gsh1[m - 4 * c], acc);          // i + j = m
                        if (m < M1 && 4 * c + m <= 27) acc = __builtin_fmaf(-uq[c], gsh2[4 * c + m + 3], acc);               // j - i = m
                        if (m < M1 && m > 0 && 4 * c - m >= -3) acc = __builtin_fmaf(-uq[c], gsh2[4 * c - m + 3], acc);      // i - j = m
                    }
                    if (m < M1) acc += keep_if(gq.m[m & 3], uq[m >> 2]);   // through the right-hand side rt[:25] - alpha
                    acc += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
                    acc += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
                    rb[m] = acc;
                }
                float amax = 0.f;
#pragma unroll
                for (int m = 0; m < M2; ++m) amax = __builtin_fmaxf(amax, __builtin_fabsf(rb[m]));
                const int s_r = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(amax);
                // the four lanes of a quad hold identical sums: all of them store (16-byte pieces, same values) into
                // the rt / rr region of the frame (dead since the rows were assembled): 64 floats, slot 63 = the scale
                f32x4* dst4 = reinterpret_cast<f32x4*>(rt_q);
#pragma unroll
                for (int q4 = 0; q4 < 16; ++q4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int m = 4 * q4 + e;
                        v[e] = m < M2 ? __builtin_ldexpf(rb[m < M2 ? m : 0], s_r) : (m == 63 ? __int_as_float(s_r) : 0.f);
                    }
                    dst4[q4] = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
            f16x8 rbh[2], rbl[2];
            float eb256 = 0.f;
            const int s_rn = __float_as_int(rt_n[63]);   // the scale of THIS lane's frame in the MFMA layout
            {
                float rv[16];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    rv[i] = rt_n[8 * g + i];
                    rv[8 + i] = g < 3 ? rt_n[32 + 8 * g + i] : 0.f;     // slot 63 of group 3 holds the scale, not data
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    eb256 = __builtin_fmaf(rv[i], lds[S_E256 + 52 + 8 * g + i], eb256);
                    eb256 = __builtin_fmaf(rv[8 + i], lds[S_E256 + 52 + 32 + 8 * g + i], eb256);
                }
                float lo8[8], hi8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { lo8[i] = rv[i]; hi8[i] = rv[8 + i]; }
                split8(lo8, rbh[0], rbl[0]);
                split8(hi8, rbh[1], rbl[1]);
            }
            __builtin_amdgcn_wave_barrier();
            eb256 += __shfl_xor(eb256, 16, 64);
            eb256 += __shfl_xor(eb256, 32, 64);

            // ================= adjoint phase: e again, ebar^T = E rtbar^T, zbar = ebar * e, lbar += zbar (in gX),
            // mbar^T += (-2 D) zbar^T with a running per-frame scale =================
            const int kz = back - s_rn - SEB_LOG2;   // zbar = acc * e_scaled * 2^kz
            const bool first = iter == n_iter - 1;   // the first step of the sweep initialises lbar
            f32x4 acc2[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
            float zmax = 0.f;
            int s_z = VMAX_LOG2;
            {
                const _Float16* imgt = img;
                asm volatile("" : "+s"(imgt));
                const gf16x8_ptr EBH = (gf16x8_ptr)(imgt + IMG_EBH) + lane;
                const gf16x8_ptr EBL = (gf16x8_ptr)(imgt + IMG_EBL) + lane;
                const gf16x8_ptr DBH = (gf16x8_ptr)(imgt + IMG_DBH) + lane;
                const gf16x8_ptr DBL = (gf16x8_ptr)(imgt + IMG_DBL) + lane;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float zv[8];
                    float m8 = 0.f;
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int mt = 2 * j + tt;
                        const f32x4 c = dtile(mt, cinit);
                        const f32x2v ta = fma2(lo2(c), kInvSDM, lo2(logx[mt])), tb = fma2(hi2(c), kInvSDM, hi2(logx[mt]));
                        const float e[4] = {__builtin_amdgcn_exp2f(ta[0]), __builtin_amdgcn_exp2f(ta[1]),
                                            __builtin_amdgcn_exp2f(tb[0]), __builtin_amdgcn_exp2f(tb[1])};
                        f32x4 acc = {0, 0, 0, 0};
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const f16x8 ah = (ABL & 2) ? EH[(mt * 2 + ks) % 24 * 64] : EBH[(mt * 2 + ks) * 64];
                            const f16x8 al = (ABL & 2) ? EL[(mt * 2 + ks) % 24 * 64] : EBL[(mt * 2 + ks) * 64];
                            acc = mfma_h(al, rbh[ks], acc);
                            acc = mfma_h(ah, rbl[ks], acc);
                            acc = mfma_h(ah, rbh[ks], acc);
                        }
                        f32x4 z;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            z[r] = __builtin_ldexpf(acc[r] * e[r], kz);
                            zv[4 * tt + r] = z[r];
                            m8 = __builtin_fmaxf(m8, __builtin_fabsf(z[r]));
                        }
                        // lbar += zbar, accumulated in the output row (this wave's private tile)
                        f32x4_u4* lp = reinterpret_cast<f32x4_u4*>(gxf + mt * 16 + 4 * g);
                        if (f_ok && !(ABL & 1)) {
                            if (first) {
                                *lp = z;
                            } else {
                                // (volatile: the row was written by this wave one step ago; a plain load may hit a stale
                                // line in the CU's write-through vector cache)
                                const f32x4 old = *reinterpret_cast<volatile f32x4_u4*>(lp);
                                *lp = old + z;
                            }
                        }
                    }
                    // running per-frame scale of zbar: when this bin group raises the maximum, the sums so far shrink
                    m8 = __builtin_fmaxf(m8, __shfl_xor(m8, 16, 64));
                    m8 = __builtin_fmaxf(m8, __shfl_xor(m8, 32, 64));
                    zmax = __builtin_fmaxf(zmax, m8);
                    const int s_new = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(zmax);
                    const int ds = s_new - s_z;   // <= 0
                    s_z = s_new;
#pragma unroll
                    for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc2[it2][r] = __builtin_ldexpf(acc2[it2][r], ds);
                    float zs[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) zs[i] = __builtin_ldexpf(zv[i], s_z);
                    f16x8 zh, zl;
                    split8(zs, zh, zl);
#pragma unroll
                    for (int it2 = 0; it2 < 2; ++it2) {
                        const f16x8 ah = (ABL & 2) ? DH[(it2 * 8 + j) * 64] : DBH[(it2 * 8 + j) * 64];
                        const f16x8 al = (ABL & 2) ? DL[(it2 * 8 + j) * 64] : DBL[(it2 * 8 + j) * 64];
                        acc2[it2] = mfma_h(al, zh, acc2[it2]);
                        acc2[it2] = mfma_h(ah, zl, acc2[it2]);
                        acc2[it2] = mfma_h(ah, zh, acc2[it2]);
                    }
                }
            }
            const float zb256 = __builtin_ldexpf(eb256 * e256, back - s_rn);
            lbar256 += zb256;
#pragma unroll
            for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = it2 * 16 + 4 * g + r;   // < 32
                    mbarC[it2][r] += __builtin_ldexpf(acc2[it2][r], -s_z - SDB_LOG2);
                    mbarC[it2][r] = __builtin_fmaf(zb256, lds[S_D256 + 32 + c], mbarC[it2][r]);   // Nyquist bin (table is 0 past c = 24)
                }
        }

        // ---------------- lbar += G mbar_0 (mcep.py:204-207 adjoint); gX = lbar / X ----------------
#pragma unroll
        for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
            for (int r = 0; r < 4; ++r) aux_n[it2 * 16 + 4 * g + r] = mbarC[it2][r];
        __builtin_amdgcn_wave_barrier();
        float m0[8];
        float mmax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            m0[i] = (8 * g + i < M1) ? aux_n[8 * g + i] : 0.f;
            mmax = __builtin_fmaxf(mmax, __builtin_fabsf(m0[i]));
        }
        __builtin_amdgcn_wave_barrier();
        mmax = __builtin_fmaxf(mmax, __shfl_xor(mmax, 16, 64));
        mmax = __builtin_fmaxf(mmax, __shfl_xor(mmax, 32, 64));
        const int s_m = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(mmax);
        float part256 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) part256 = __builtin_fmaf(m0[i], tail_f[8 * g + i], part256);   // G[256][c] (0 past c = 24)
        part256 += __shfl_xor(part256, 16, 64);
        part256 += __shfl_xor(part256, 32, 64);
        lbar256 += part256;
        {
            float ms[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ms[i] = __builtin_ldexpf(m0[i], s_m);
            f16x8 mh8, ml8;
            split8(ms, mh8, ml8);
            const _Float16* imgt = img;
            asm volatile("" : "+s"(imgt));
            const gf16x8_ptr GBH = (gf16x8_ptr)(imgt + IMG_GBH) + lane;
            const gf16x8_ptr GBL = (gf16x8_ptr)(imgt + IMG_GBL) + lane;
            const bool have_lbar = n_iter > 0;
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
                const f16x8 ah = GBH[mt * 64], al = GBL[mt * 64];
                f32x4 acc = {0, 0, 0, 0};
                acc = mfma_h(al, mh8, acc);
                acc = mfma_h(ah, ml8, acc);
                acc = mfma_h(ah, mh8, acc);
                if (f_ok) {
                    f32x4_u4* lp = reinterpret_cast<f32x4_u4*>(gxf + mt * 16 + 4 * g);
                    f32x4 lb = {0, 0, 0, 0};
                    if (have_lbar) lb = *reinterpret_cast<volatile f32x4_u4*>(lp);
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o[r] = (lb[r] + __builtin_ldexpf(acc[r], -s_m - SGB_LOG2)) * __builtin_amdgcn_exp2f(-logx[mt][r]);
                    *lp = o;
                }
            }
        }
        if (f_ok && g == 0) gxf[H] = lbar256 * __builtin_amdgcn_exp2f(-logx256);
        tile = tile_next;
    }
}

}  // namespace dsa
