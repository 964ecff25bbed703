// Round 6, third cut of the one-launch Newton kernel for the 48 kHz set-ups (mcep.py:208-222; included by mcep_mfma.hip after
// mcep_big_f16.h, whose solver wrappers, record geometry and images it shares): TWIN workgroups.
//
// The eight-wave kernels of mcep_big_f16.h run a step in lock step: every wave of the CU is in the products (binary16 matrix pipe +
// LDS operand reads, 17-33 barriers) and then every wave is in the solve (float32 matrix instructions + vector unit).  The two
// phases use different units, and with all eight waves in the same phase one set of units idles: stamps of a wide step at 2048 / 49
// say stage loop 122-139 k cycles against 35 k of matrix-pipe time and 47 k of LDS operand reads, solve 2 x 50 k against 58 k of
// float32 datapath time (profiles/r06_mcep_big_wide.txt).
// Here a workgroup is FOUR waves, each with its own 16 frames (the stage body of mcep_resid_h_kernel: one stage per barrier, images
// one stage ahead in registers, even / odd stages summed separately -- the same sums in the same order as every other path: the same
// bits), and TWO workgroups share a CU (74-76 KB of LDS each, one wave of each per SIMD, 256 registers).  The second workgroup of a
// CU starts half a step late (`stagger`: the workgroups of the second half of the grid sleep once, before their first step), so while
// one workgroup's waves are in the products the other's are in the solve; barriers tie four waves, not eight.
#pragma once

#include "mcep_big_f16.h"

namespace dsa {

namespace mbg4 {
constexpr int WAVES = 4;
template <int KS1, int NT, int NG, bool QUAD>
constexpr int region_floats()
{
    constexpr int sh = mrh::stage_halves(KS1, NT);   // two single-stage buffers: 2 SH halves = SH floats
    constexpr int solve = WAVES * 16 * mbg::rts(NT) + (QUAD ? WAVES * 16 : WAVES * 8) * mbg::Geo<NG, QUAD>::REC;
    return sh > solve ? sh : solve;
}
template <int KS1, int NT, int NG, bool QUAD>
constexpr int lds_floats()
{
    return region_floats<KS1, NT, NG, QUAD>() + WAVES * 16 * mbg::Geo<NG, QUAD>::MS + 64;
}
}  // namespace mbg4

template <int KS1, int NT, int NG, int NMIN, bool QUAD>
__global__ __launch_bounds__(256, 2) void mcep_big_newton4_kernel(const float* __restrict__ logx, long F, int K, const float* __restrict__ mc_in,
                                                                  int M1, const _Float16* __restrict__ img, const float* __restrict__ av,
                                                                  int n_iter, float* __restrict__ mc_out, int stagger)
{
    using namespace mrh;
    using G = mbg::Geo<NG, QUAD>;
    constexpr int NTH = mbg4::WAVES * 64;
    constexpr int SH = stage_halves(KS1, NT);
    constexpr int PIECES = SH / 8;                          // 16-byte pieces per stage
    constexpr int PER = (PIECES + NTH - 1) / NTH;
    constexpr int PO = G::PO, RO = G::RO, REC = G::REC, MS = G::MS, SYS = G::SYS, BACK = G::BACK;
    constexpr int RTS = mbg::rts(NT);
    constexpr int ROUNDS = QUAD ? 1 : 2;                    // solves per wave and step (octet layout: 8 systems a round)
    extern __shared__ __attribute__((aligned(16))) float smem_big4[];
    _Float16* sbuf0 = reinterpret_cast<_Float16*>(smem_big4);   // [2][SH halves]
    float* recs_all = smem_big4 + mbg4::WAVES * 16 * RTS;     // the parked rt rows and the records live INSIDE the staging buffers
    float* mcs_all = smem_big4 + mbg4::region_floats<KS1, NT, NG, QUAD>();
    float* avs = mcs_all + mbg4::WAVES * 16 * MS;            // [64]: alpha_vec, zero-padded
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int nstage = (K + 31) / 32;
    const int N = 2 * M1 - 1;
    float* mcs = mcs_all + wave * 16 * MS;
    float* wl = recs_all + wave * (QUAD ? 16 : 8) * REC;      // this wave's records
    float* park = smem_big4 + wave * 16 * RTS;                // this wave's rt rows
    const f32x4* img4 = reinterpret_cast<const f32x4*>(img);
    if (threadIdx.x < 64) avs[threadIdx.x] = (int)threadIdx.x < M1 ? av[threadIdx.x] : 0.f;
    const long ntiles = (F + 16 * mbg4::WAVES - 1) / (16 * mbg4::WAVES);
    bool first = true;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long t16 = (tile * mbg4::WAVES + wave) * 16;    // uniform
        const bool tile_ok = t16 < F;
        const long tb = tile_ok ? t16 : 0;
        const int rows_here = (int)((F - tb < 16) ? F - tb : 16);
        // ---- the wave's 16 rows of mc into LDS (rows past the batch repeat the last one: finite, never stored) ----
        for (int e = (threadIdx.x & 63); e < 16 * MS; e += 64) {
            const int row = e / MS, col = e % MS;
            const int rr = row < rows_here ? row : rows_here - 1;
            mcs[row * MS + col] = col < M1 ? mc_in[(tb + rr) * (long)M1 + col] : 0.f;
        }
        if (first) {
            // the CU's second workgroup (dispatch order: the second half of the grid lands on CUs that hold one of the first half
            // already) starts `stagger` x ~8 k cycles late, once: its products then meet the first workgroup's solve and vice versa
            first = false;
            if (stagger > 0 && 2 * blockIdx.x >= gridDim.x && gridDim.x > 256)
                for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
        }
        for (int step = 0; step < n_iter; ++step) {
            __syncthreads();   // every wave has left the previous step's solve (and the first step's avs / mc rows are written): the
                               // staging buffers are free again
            // (everything derived from the lane index is derived again per step from an opaque copy: see mcep_big_newton_kernel)
            int tid_s = threadIdx.x;
            asm volatile("" : "+v"(tid_s));
            const int tid = tid_s, lane = tid_s & 63, n = lane & 15, g = lane >> 4;
            const int rn = n < rows_here ? n : rows_here - 1;
            const float* xt = logx + tb * (long)K + (long)rn * K;
            // ================= rt = exp(logx - 2 mc D) E: the stage body of mcep_resid_h_kernel =================
            // ONE register set: a stage's images are requested at the head of the stage before it.  (Measured, quad-layout orders: a
            // second set requested two stages ahead -- the stage body of mcep_resid_h_kernel -- shortens the bare staging skeleton,
            // 286 -> 260 us per 32 768 frames, and lengthens the whole kernel, 578 -> 604 us; at the octet-layout orders it spills.)
            f32x4 st0[PER];
            auto fetch = [&](int j, f32x4 (&sv)[PER]) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    // (unconditional, from a clamped index: a conditional element assignment keeps the whole array in private memory)
                    const long src = (long)j * PIECES + tid + NTH * q, last = (long)nstage * PIECES - 1;
                    sv[q] = img4[src < last ? src : last];
                }
            };
            auto stage = [&](int buf, const f32x4 (&sv)[PER]) __attribute__((always_inline)) {
                f32x4* d = reinterpret_cast<f32x4*>(sbuf0 + buf * SH);
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int p = tid + NTH * q;
                    if (p < PIECES) d[p] = sv[q];
                }
            };
            fetch(0, st0);
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
                        bv[ks][i] = c < MS ? mcs[n * MS + (c < MS ? c : 0)] : 0.f;   // (columns M1 .. MS - 1 hold zeros)
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
                k1 = -s_b - LOG2_SD;
            }
            f32x4 acc[2][NT];   // [0] the even stages' sums, [1] the odd stages'
#pragma unroll
            for (int hs = 0; hs < 2; ++hs)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[hs][t] = zero4;
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
            __syncthreads();
            // stage j: the images of stage j + 1 are requested at its head and staged at its end; `xr` holds the rows of stage j and takes
            // those of stage j + 2 (every request unconditional, from clamped indices: the compiler's wait counts stay exact)
            auto body = [&](int j, f32x4 (&xr)[2], f32x4* accj) __attribute__((always_inline)) {
                const int buf = j & 1;
                const f32x4 xv[2] = {xr[0], xr[1]};
                fetch(j + 1, st0);
                xfetch(j + 2 < nstage ? j + 2 : nstage - 1, xr);
                if (tile_ok && !(DSA_BIG_ABL & 2)) {
                    const f16x8* c1 = reinterpret_cast<const f16x8*>(sbuf0 + buf * SH) + lane;
                    const f16x8* w2 = c1 + (4 * KS1 * 512) / 8;
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
                stage(buf ^ 1, st0);   // the other buffer: its readers finished before the barrier that ended stage j - 1 (after the last
                                       // stage: the clamped piece again, never read)
                __syncthreads();
            };
#pragma unroll 1
            for (int j = 0; j < nstage; j += 2) {
                body(j, x0, acc[0]);
                if (j + 1 < nstage) body(j + 1, x1, acc[1]);
            }
            // (the barrier that ended the last stage: every wave is done with the staging buffers -- the rt rows may take them)
            // C/D layout: lane (n, g), register r of tile tc <-> rt[16 tc + 4 g + r] of frame n; even + odd stages' sums
#pragma unroll
            for (int tc = 0; tc < NT; ++tc) *reinterpret_cast<f32x4*>(park + n * RTS + 16 * tc + 4 * g) = acc[0][tc] + acc[1][tc];
            __builtin_amdgcn_wave_barrier();
            // ================= mc += solve(T(rt[:n]) + H(rt), rt[:n] - alpha_vec) =================
            if (tile_ok && !(DSA_BIG_ABL & 1)) {
#pragma unroll 1
                for (int rnd = 0; rnd < ROUNDS; ++rnd) {
                    int ln = lane;
                    asm volatile("" : "+v"(ln));
                    const int row0 = QUAD ? 0 : 8 * rnd;
                    {
                        f32x4* z4 = reinterpret_cast<f32x4*>(wl);
                        for (int e = ln; e < SYS * REC / 4; e += 64) z4[e] = zero4;
                    }
                    __builtin_amdgcn_wave_barrier();
                    {
                        const int s0 = (ln >> 4) * (SYS / 4);                                    // lane -> (SYS / 4 records, 16 columns apart)
                        for (int s_ = s0; s_ < s0 + SYS / 4; ++s_) {
                            float* rec = wl + s_ * REC;
                            const float* prow = park + (row0 + s_) * RTS;
                            for (int col = ln & 15; col < N; col += 16) {
                                const float v = prow[col];
                                rec[col] = v;                                                    // q window: q[k] at k
                                if (col < M1) {
                                    rec[PO + col] = v;                                           // p window: p[|d|] at PO + d
                                    if (col >= 1 && col <= BACK) rec[PO - col] = v;
                                    rec[RO + col] = v - avs[col];                                // right-hand side
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    if constexpr (QUAD) big_solve16q<NG, NMIN>((mbg::lds_f*)wl, (mbg::lds_f*)mcs, M1);
                    else big_solve8<NG, NMIN>((mbg::lds_f*)wl, (mbg::lds_f*)(mcs + row0 * MS), M1);
                    __builtin_amdgcn_wave_barrier();   // (the records are rebuilt for the second round)
                }
            }
        }
        // ---- the result: the wave's rows of mc (its own rows: no barrier; the next tile's first step starts with one) ----
        __builtin_amdgcn_wave_barrier();
        if (tile_ok) {
            for (int e = (threadIdx.x & 63); e < 16 * M1; e += 64) {
                const int row = e / M1, col = e % M1;
                if (row < rows_here) mc_out[(tb + row) * (long)M1 + col] = mcs[row * MS + col];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace dsa
