// Round 6: ALL Newton steps of MelCepstralAnalysis (mcep.py:208-222) at the 48 kHz set-ups (fft_length 2048 / order 49 and every
// order 36 .. 55 the octet-layout solver covers) in ONE persistent launch (included by mcep_mfma.hip).
//
// Rounds 4-5 ran a step as two launches -- dsa_mcep_newton_resid_h (rt = exp(logx - 2 mc D) E on binary16 splits, mcep_resid_f16.h)
// and dsa_mcep_newton_update (the batched Toeplitz-plus-Hankel solve, thsolve_tq.h) -- with rt:(F, 2 M + 1) and mc:(F, M + 1)
// going through memory between them: 22 launches per analysis; at 12 800 frames (one workgroup per CU) the kernels take 37 + 35 us
// of a 92 us step, the rest is the boundary between two dependent launches, twenty times.  A frame's iteration depends on the
// frame alone, so here a wave keeps its 16 frames for all steps:
//   * the step's products exactly as mcep_resid_h_kernel computes them (same images, same stages of 32 bins staged through LDS for
//     the workgroup's four waves, same splits and scales: rt is bit-identical);
//   * EIGHT waves per workgroup, 64 frames: waves w and w + 4 share 16 frames.  In the products they take alternate stages (the
//     double-buffered staging holds two consecutive stages anyway) -- a lone wave's stage is a chain of dependent matrix products and
//     an exp, so two waves halve the phase; the two partial sums meet in LDS (even stages + odd stages: a fixed order, but not the
//     stage-by-stage order of mcep_resid_h_kernel -- rt differs from the two-launch step in the last bit);
//   * each of the two then solves 8 of the 16 systems: rt rows -> the solver's LDS records (q window, p window, right-hand side
//     rt[k] - alpha_vec[k]), the octet layout of thsolve_octn_kernel with its own elimination and back substitution templates, incl.
//     the pivoted re-solve of a system whose elimination meets a bad pivot;
//   * the update mc += x lands in the pair's LDS copy of mc, from which the next step's first-chain operands are split.
// One workgroup per CU (two waves per SIMD, 256 registers: the solver's spills of the stand-alone kernel come with it), 105 KB of LDS
// (the records and the parked rows alias the four staging buffers).
// First cut of the round (four waves, each solving its 16 systems in two rounds): correct and bit-identical to the two-launch step,
// but 1.68 ms per 12 800 frames against 0.93 -- a lone wave per SIMD ran the two eliminations one after the other.
#pragma once

#include "thsolve_tq.h"

#ifndef DSA_BIG_ABL
#define DSA_BIG_ABL 0   // measurement builds only: 1 no solve, 2 no products (tools/build_variant.sh ... -DDSA_BIG_ABL=3)
#endif

namespace dsa {

namespace mbg {
constexpr int PAIRS = 4;    // 16-frame groups per workgroup
constexpr int WAVES = 8;    // two waves per group
// the solver's LDS record of a system: octet layout (orders 35 .. 54: thsolve_octn_kernel, 8 systems per wave) or -- QUAD -- the quad
// layout (orders up to 34: thsolve_quadn_kernel, 16 systems per wave)
template <int NG, bool QUAD = false>
struct Geo {
    using O = tq::Oct<NG>;
    static constexpr int NCP = O::NCP;
    static constexpr int CN = 4 * NG - 1;
    static constexpr int QW = QUAD ? 8 * NG - 1 : 4 * NG + 8 * NCP - 1;
    static constexpr int BACK = QUAD ? 3 : 7;                             // how far a block's views reach below the diagonal
    static constexpr int PO = QW + BACK;
    static constexpr int RO = QUAD ? QW + 4 * NG + 3 : PO + 8 * NCP;
    static constexpr int REC = QUAD ? ((16 * NG + 3 - 4 + 31) / 32) * 32 + 4     // as thsolve_quadn_kernel: stride 4 (mod 32)
                                    : ((RO + 4 * NG - 8 + 31) / 32) * 32 + 8;    // as thsolve_octn_kernel: stride 8 (mod 32)
    static constexpr int SYS = QUAD ? 16 : 8;                             // systems a solving wave takes
    static constexpr int MS = 4 * NG + 4;                                 // row stride of the LDS copy of mc (floats, 16-byte rows)
};
constexpr int rts(int nt) { return 16 * nt + 4; }   // row stride of the rt rows parked in LDS between the products and the solve (floats)
// the region the four staging buffers (two stage pairs, SH halves per stage = 2 SH floats together) share with the parked rt rows of
// the four groups and the eight waves' records -- all of them idle while the other is in use
// WIDE (see the kernel): EIGHT 16-frame groups per workgroup, one per wave.  The parked rows are then 128; the records stay 64 in the
// octet layout (a wave solves its 16 systems in two rounds of 8) and become 128 in the quad layout (16 per wave at once).
constexpr int groups(bool wide) { return wide ? WAVES : PAIRS; }
template <int KS1, int NT, int NG, bool QUAD = false, bool WIDE = false>
constexpr int stage_floats()
{
    constexpr int sh = 2 * mrh::stage_halves(KS1, NT);
    constexpr int solve = groups(WIDE) * 16 * rts(NT) + (QUAD ? groups(WIDE) * 16 : WAVES * 8) * Geo<NG, QUAD>::REC;
    return sh > solve ? sh : solve;
}
template <int KS1, int NT, int NG, bool QUAD = false, bool WIDE = false>
constexpr int lds_floats()
{
    return stage_floats<KS1, NT, NG, QUAD, WIDE>() + groups(WIDE) * 16 * Geo<NG, QUAD>::MS + 64;
}
}  // namespace mbg

namespace mbg {
typedef __attribute__((address_space(3))) float lds_f;
}
// Eight systems of the records `wl` solved in the octet layout (thsolve_octn_kernel's construction, elimination, back substitution
// and pivoted re-solve), the solutions ADDED to the eight rows `mrow8` of the LDS copy of mc (mcep.py:222).
template <int NG, int NMIN>
__device__ __attribute__((noinline)) void big_solve8(mbg::lds_f* wl, mbg::lds_f* mrow8, int M1)
{
    using G = mbg::Geo<NG>;
    using O = tq::Oct<NG>;
    constexpr int NCP = G::NCP, CN = G::CN, PO = G::PO, RO = G::RO, REC = G::REC, MS = G::MS;
    constexpr int CPN = (NG - 1) >> 1, HN = (NG - 1) & 1;   // where column CN (the right-hand side) lives
    const int ln = threadIdx.x & 63;
    const int sy = ln >> 3, h = (ln >> 2) & 1, gs = ln & 3;
    const int view = sy * REC + 4 * h + gs;
    const mbg::lds_f* rs = wl + sy * REC + RO;
    f32x4 a[O::N];
#pragma unroll
    for (int rg = 0; rg < NG; ++rg) {
        const mbg::lds_f* qs = wl + view;
        const mbg::lds_f* pw = qs + PO;
#pragma unroll
        for (int cp = rg >> 1; cp < NCP; ++cp) {
            const bool below = 2 * cp < rg;
            const bool cin = 8 * cp + 7 < NMIN || 8 * cp + 4 * h + gs < M1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * rg + i;
                float v = pw[4 * (2 * cp - rg) - i] + qs[4 * (rg + 2 * cp) + i];
                if (8 * cp + 7 >= NMIN) v = cin ? v : 0.f;
                if (below) v = h == 1 ? v : 0.f;
                if (2 * cp == rg || 2 * cp + 1 == rg) {
                    if (row >= NMIN && row < CN) v = (h == (rg & 1) && gs == i && row >= M1) ? 1.f : v;
                }
                if (cp == CPN) v = (h == HN && gs == 3) ? rs[row] : v;
                a[O::at(rg, cp)][i] = v;
            }
        }
    }
    bool bad = false;
    tq::oct_elim_all<NG>(a, gs, bad, std::make_integer_sequence<int, CN>{});
    float xq[NCP];
#pragma unroll
    for (int c = 0; c < NCP; ++c) xq[c] = (c == CPN && h == HN && gs == 3) ? -1.f : 0.f;
    tq::oct_backsub_all<NG>(a, xq, gs, h, std::make_integer_sequence<int, NG>{});
    mbg::lds_f* mrow = mrow8 + sy * MS;
#pragma unroll
    for (int c = 0; c < NCP; ++c) {
        const int col = 8 * c + 4 * h + gs;
        if (col < M1 && !bad) mrow[col] += xq[c];                                // mcep.py:222
    }
    unsigned long long marked = __ballot(bad && (ln & 7) == 0);
    while (marked) {   // uniform; normally empty: the whole wave re-solves the system with row pivoting (th_solve_reg.h)
        const int bl_ = __builtin_ctzll(marked);
        marked &= marked - 1;
        const int sb = bl_ >> 3;
        const float* qs2 = (const float*)(wl + sb * REC);   // (flat view of the LDS record for the cold path)
        const float* ps2 = qs2 + PO;
        const float rhs = ln < M1 ? qs2[RO + ln] : 0.f;
        int col;
        float sol;
        th_solve_reg<float, CN <= 48 ? 48 : 64>(ps2, qs2, rhs, M1, ln, col, sol);
        if (ln < M1) mrow8[sb * MS + col] += sol;
    }
}

// Sixteen systems of the records `wl` solved in the QUAD layout (thsolve_quadn_kernel's construction, elimination, back substitution and
// pivoted re-solve; orders up to 34), the solutions ADDED to the sixteen rows `mrow16` of the LDS copy of mc.
template <int NG, int NMIN>
__device__ __attribute__((noinline)) void big_solve16q(mbg::lds_f* wl, mbg::lds_f* mrow16, int M1)
{
    using G = mbg::Geo<NG, true>;
    using B = tq::Blk<NG>;
    constexpr int CN = G::CN, PO = G::PO, RO = G::RO, REC = G::REC, MS = G::MS;
    const int ln = threadIdx.x & 63;
    const int nq = ln >> 2, gs = ln & 3;
    const mbg::lds_f* qs = wl + nq * REC + gs;        // this lane's views: column offset gs folded in
    const mbg::lds_f* pw = qs + PO;
    const mbg::lds_f* rs = wl + nq * REC + RO;
    f32x4 a[B::N];
#pragma unroll
    for (int rg = 0; rg < NG; ++rg) {
#pragma unroll
        for (int cg = rg; cg < NG; ++cg) {
            const bool cin = 4 * cg + 3 < NMIN || 4 * cg + gs < M1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * rg + i;
                float v = pw[4 * (cg - rg) - i] + qs[4 * (rg + cg) + i];
                if (4 * cg + 3 >= NMIN) v = cin ? v : 0.f;
                if (cg == rg && row >= NMIN && row < CN) v = (gs == i && row >= M1) ? 1.f : v;
                if (cg == NG - 1) v = gs == 3 ? rs[row] : v;
                a[B::at(rg, cg)][i] = v;
            }
        }
    }
    bool bad = false;
    tq::elim_all<NG>(a, gs, bad, std::make_integer_sequence<int, CN>{});
    float xq[NG];
#pragma unroll
    for (int c = 0; c < NG; ++c) xq[c] = (4 * c + gs == CN) ? -1.f : 0.f;
    tq::backsub_all<NG>(a, xq, gs, std::make_integer_sequence<int, NG>{});
    mbg::lds_f* mrow = mrow16 + nq * MS;
#pragma unroll
    for (int c = 0; c < NG; ++c) {
        const int col = 4 * c + gs;
        if (col < M1 && !bad) mrow[col] += xq[c];                                // mcep.py:222
    }
    unsigned long long marked = __ballot(bad && gs == 0);
    while (marked) {   // uniform; normally empty: the whole wave re-solves the system with row pivoting (th_solve_reg.h)
        const int bl_ = __builtin_ctzll(marked);
        marked &= marked - 1;
        const int sb = bl_ >> 2;
        const float* qs2 = (const float*)(wl + sb * REC);   // (flat view of the LDS record for the cold path)
        const float* ps2 = qs2 + PO;
        const float rhs = ln < M1 ? qs2[RO + ln] : 0.f;
        int col;
        float sol;
        th_solve_reg<float, CN <= 32 ? 32 : (CN <= 48 ? 48 : 64)>(ps2, qs2, rhs, M1, ln, col, sol);
        if (ln < M1) mrow16[sb * MS + col] += sol;
    }
}

// WIDE (second cut of the round): EIGHT 16-frame groups per workgroup -- every wave keeps its OWN 16 frames and runs BOTH stages of a
// staged pair (two accumulator sets, even and odd stages, added at the end: the same sums in the same order as the two waves of a
// narrow group, the same bits), then solves its 16 systems itself (octet layout: two rounds of 8).  The narrow kernel is a LATENCY
// design (a tile's step as short as possible: small batches are one round of tiles); at more than one round of tiles what counts is
// frames per staged image byte and per barrier -- the same 17 stage pairs, barriers and LDS staging now serve 128 frames, nothing is
// exchanged between waves, and in the quad layout (orders up to 34) no wave idles through the solve.  Chosen by the batch size
// (mcep_big_newton); both give the same bits (tests/test_gpu_configs.py).
template <int KS1, int NT, int NG, int NMIN, bool QUAD = false, bool WIDE = false>   // QUAD: orders up to 34 -- narrow: the group's first wave solves all 16 systems
__global__ __launch_bounds__(512, 1) void mcep_big_newton_kernel(const float* __restrict__ logx, long F, int K, const float* __restrict__ mc_in,
                                                                 int M1, const _Float16* __restrict__ img, const float* __restrict__ av,
                                                                 int n_iter, float* __restrict__ mc_out)
{
    using namespace mrh;
    using G = mbg::Geo<NG, QUAD>;
    constexpr int NTH = mbg::WAVES * 64;
    constexpr int GROUPS = mbg::groups(WIDE);              // 16-frame groups per workgroup
    constexpr int NSTG = WIDE ? 2 : 1;                     // stages of a pair a wave runs
    constexpr int RPW = WIDE ? 16 : 8;                     // rows of mc a wave moves in / out
    constexpr int SH = stage_halves(KS1, NT);
    constexpr int PIECES = SH / 8;
    constexpr int PER = (2 * PIECES + NTH - 1) / NTH;      // a stage PAIR per staging step
    constexpr int PO = G::PO, RO = G::RO, REC = G::REC, MS = G::MS, SYS = G::SYS, BACK = G::BACK;
    constexpr int RTS = mbg::rts(NT);
    constexpr int ROUNDS = (WIDE && !QUAD) ? 2 : 1;        // solves per wave and step
    extern __shared__ __attribute__((aligned(16))) float smem_big[];
    _Float16* sbuf0 = reinterpret_cast<_Float16*>(smem_big);   // [2 sets][2 stages of a pair][SH halves] = 2 SH floats
    float* recs_all = smem_big + GROUPS * 16 * RTS;          // the records and the parked rt rows live INSIDE the staging buffers
    float* mcs_all = smem_big + mbg::stage_floats<KS1, NT, NG, QUAD, WIDE>();
    float* avs = mcs_all + GROUPS * 16 * MS;                // [64]: alpha_vec, zero-padded
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = WIDE ? wave : (wave & 3), hsel = WIDE ? 0 : (wave >> 2);   // the 16-frame group; narrow: which half of its stages / systems this wave takes
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int nstage = (K + 31) / 32;
    const int N = 2 * M1 - 1;
    float* mcs = mcs_all + pair * 16 * MS;
    float* wl = recs_all + ((QUAD ? pair * 16 : wave * 8)) * REC;   // this wave's records (narrow QUAD: the group's 16, solved by its first wave)
    float* park = smem_big + pair * 16 * RTS;                // the group's rt rows (inside the staging buffers, idle during the solve)
    const f32x4* img4 = reinterpret_cast<const f32x4*>(img);
    if (tid < 64) avs[tid] = tid < M1 ? av[tid] : 0.f;
    const long ntiles = (F + 16 * GROUPS - 1) / (16 * GROUPS);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long t16 = (tile * GROUPS + pair) * 16;    // uniform
        const bool tile_ok = t16 < F;
        const long tb = tile_ok ? t16 : 0;
        const int rows_here = (int)((F - tb < 16) ? F - tb : 16);
        __syncthreads();   // the previous tile's result rows have left the LDS copy of mc
        // ---- the group's 16 rows of mc into LDS, RPW per wave (rows past the batch repeat the last one: finite, never stored) ----
        for (int e = (tid & 63); e < RPW * MS; e += 64) {
            const int row = 8 * hsel + e / MS, col = e % MS;
            const int rr = row < rows_here ? row : rows_here - 1;
            mcs[row * MS + col] = col < M1 ? mc_in[(tb + rr) * (long)M1 + col] : 0.f;
        }
#ifdef DSA_BIG_STAMPS   // (measurement builds: cycle stamps of wave 0 / wave 4 of workgroup 0 in step 2, returned in mc_out's first rows)
        long long tsv[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define BIG_STAMP(i) do { if (step == 2) tsv[i] = __builtin_readcyclecounter(); } while (0)
#define BIG_STAMP_P(i) do { if (step == 2 && ip == 4) tsv[i] = __builtin_readcyclecounter(); } while (0)   // inside pair 4
#else
#define BIG_STAMP(i)
#define BIG_STAMP_P(i)
#endif
        for (int step = 0; step < n_iter; ++step) {
            BIG_STAMP(0);
            __syncthreads();   // every wave has left the previous step's solve: mc is updated, the staging buffers are free again
            // everything derived from the lane index is derived again per step from an opaque copy: hoisted out of the step loop such
            // values live across the elimination, go to scratch there and come back through scratch loads inside the stage loop
            int tid_s = threadIdx.x;
            asm volatile("" : "+v"(tid_s));
            const int tid = tid_s, lane_s = tid_s & 63;
            const int lane = lane_s, n = lane_s & 15, g = lane_s >> 4;
            const int rn = n < rows_here ? n : rows_here - 1;
            const float* xt = logx + tb * (long)K + (long)rn * K;
            // ================= rt = exp(logx - 2 mc D) E: the stage body of mcep_resid_h_kernel, two stages per barrier =================
            // A stage PAIR (2 i, 2 i + 1) is staged together (four buffers: the pair in use, the pair being written); between two
            // barriers the group's first wave runs the even stage, its second wave the odd one -- concurrently (WIDE: the group's one
            // wave runs both).  (First 8-wave cut: one stage per barrier, the waves alternating -- a stage's arithmetic never
            // overlapped the next one's: 1.05 ms.)
            const int npair = (nstage + 1) / 2;
            f32x4 st0[PER];   // ONE register set: a pair's images are requested one pair ahead.  (Two sets, two pairs ahead, do not fit
                              // 256 registers next to the hoisted image reads of a stage: 62 scratch accesses inside the loop.  Stamps of a
                              // step, cycles: head 4-19 k, this loop 67 k (17 pairs), rows + records 6 k, the solve 50-62 k.)
            auto fetch = [&](int ip, f32x4 (&sv)[PER]) __attribute__((always_inline)) {   // stages 2 ip, 2 ip + 1: 2 PIECES pieces
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int p = tid + NTH * q;
                    // (unconditional, from a clamped index: a conditional element assignment kept the whole array in private memory)
                    const long src = (long)(2 * ip) * PIECES + p, last = (long)nstage * PIECES - 1;
                    sv[q] = img4[src < last ? src : last];
                }
            };
            auto stage = [&](int set, const f32x4 (&sv)[PER]) __attribute__((always_inline)) {
                f32x4* d = reinterpret_cast<f32x4*>(sbuf0 + set * 2 * SH);
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const int p = tid + NTH * q;
                    if (p < 2 * PIECES) d[p] = sv[q];
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
            f32x4 acc[NSTG][NT];   // (WIDE: [0] the even stages' sums, [1] the odd stages')
#pragma unroll
            for (int hs = 0; hs < NSTG; ++hs)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[hs][t] = zero4;
            // the lane's log-spectrum values of ITS stage(s) of a pair (narrow: 2 ip + hsel), requested two pairs ahead
            f32x4 x0[NSTG][2], x1[NSTG][2];
            auto xfetch = [&](int ip, f32x4 (&xr)[NSTG][2]) __attribute__((always_inline)) {
#pragma unroll
                for (int hs = 0; hs < NSTG; ++hs) {
                    const int j = 2 * ip + (WIDE ? hs : hsel);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int b0 = 32 * j + 16 * t + 4 * g;
                        if (b0 + 3 < K) {
                            xr[hs][t] = *reinterpret_cast<const f32x4_u4*>(xt + b0);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) xr[hs][t][r] = xt[b0 + r < K ? b0 + r : K - 1];
                        }
                    }
                }
            };
            xfetch(0, x0);
            xfetch(npair > 1 ? 1 : 0, x1);
            stage(0, st0);
            __syncthreads();
            BIG_STAMP(1);
            // pair ip: the images of pair ip + 1 are requested at its head and staged at its end; `xr` holds this wave's rows.
            // (Measured and NOT adopted, profiles/r06_mcep_big_wide.txt: every load of the loop unconditional from clamped indices and the
            // body straight-line -- the compiler's wait-counter analysis then counts exactly, where this form waits `vmcnt(0)` behind the
            // image fetch it has just issued, 0.9-2.1 k cycles of a pair's 4.1 k by the stamps -- together with the two stages of a
            // wide wave run phase by phase interleaved: the stamped wave's phases got shorter and its wait at the barrier longer by as
            // much, 672 -> 718 us per 12 800 frames: the pair's time is set by the YOUNGER wave of each SIMD.)
            auto body = [&](int ip, f32x4 (&xr)[NSTG][2]) __attribute__((always_inline)) {
                const int set = ip & 1;
                f32x4 xv[NSTG][2];
#pragma unroll
                for (int hs = 0; hs < NSTG; ++hs) { xv[hs][0] = xr[hs][0]; xv[hs][1] = xr[hs][1]; }
                BIG_STAMP_P(8);
                if (ip + 1 < npair) fetch(ip + 1, st0);
                if (ip + 2 < npair) xfetch(ip + 2, xr);
                BIG_STAMP_P(9);
#ifndef DSA_BIG_WIDE_INTERLEAVE
#define DSA_BIG_WIDE_INTERLEAVE 1   // (0: the wide wave's two stages one after the other, for A/B builds)
#endif
                bool done_ = false;
                if constexpr (WIDE && !QUAD && DSA_BIG_WIDE_INTERLEAVE) {   // (measured, profiles/r06_mcep_big_wide.txt: 2048 / 49 -1 .. -4 %; the quad-layout orders +3 .. +6 %: off there)
                    if (tile_ok && 2 * ip + 1 < nstage && !(DSA_BIG_ABL & 2)) {
                        // Both stages of the pair, software-pipelined INSIDE the wave: the matrix pipe and the vector unit are separate
                        // pipes, and a stage alone uses them one after the other (stamps: first chain, t / max / exp / split, second chain
                        // ~0.45 / 0.6 / 0.7 k cycles, each at its own pipe's rate when both waves of a SIMD are in the same phase).  Here the
                        // second stage's first chain issues between the first stage's exponentials, the first stage's second chain between
                        // the second stage's: [c1 a] [c1 b | exp a] [c2 a | exp b] [c2 b].  Same instructions, same operands, same order
                        // of every sum: the same bits.
                        done_ = true;
                        const f16x8* c1a = reinterpret_cast<const f16x8*>(sbuf0 + (set * 2 + 0) * SH) + lane;
                        const f16x8* c1b = c1a + SH / 8;
                        auto chain1 = [&](const f16x8* c1, f32x4 (&s2)[2]) __attribute__((always_inline)) {
#pragma unroll
                            for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
                                for (int t = 0; t < 2; ++t) {
                                    const f16x8 dh = c1[((t * KS1 + ks) * 2 + 0) * 64], dl = c1[((t * KS1 + ks) * 2 + 1) * 64];
                                    s2[t] = mfma_h(dl, bh[ks], s2[t]);
                                    s2[t] = mfma_h(dh, bl[ks], s2[t]);
                                    s2[t] = mfma_h(dh, bh[ks], s2[t]);
                                }
                        };
                        auto expsplit = [&](int j, const f32x4 (&xq)[2], const f32x4 (&s2)[2], f16x8& eh, f16x8& el, int& k2) __attribute__((always_inline)) {
                            float tv[8];
                            float tmax = -3.0e38f;
#pragma unroll
                            for (int t = 0; t < 2; ++t)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const bool live = 32 * j + 16 * t + 4 * g + r < K;
                                    const float v = __builtin_fmaf(xq[t][r], 1.4426950408889634f, __builtin_ldexpf(s2[t][r], k1));
                                    tv[4 * t + r] = live ? v : -3.0e38f;
                                    tmax = __builtin_fmaxf(tmax, tv[4 * t + r]);
                                }
                            tmax = rows_max4(tmax);
                            const float mi = __builtin_ceilf(tmax);
                            const float shf = (float)EMAX_LOG2 - mi;
                            float ev[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) ev[i] = __builtin_amdgcn_exp2f(tv[i] + shf);
                            split8(ev, eh, el);
                            k2 = (int)mi - EMAX_LOG2 - LOG2_SE;
                        };
                        auto chain2 = [&](const f16x8* w2, const f16x8& eh, const f16x8& el, int k2, f32x4 (&ac)[NT]) __attribute__((always_inline)) {
#pragma unroll
                            for (int tc = 0; tc < NT; ++tc) {
                                const f16x8 wh = w2[(tc * 2 + 0) * 64], wlo = w2[(tc * 2 + 1) * 64];
                                f32x4 a_ = mfma_h(wlo, eh, zero4);
                                a_ = mfma_h(wh, el, a_);
                                a_ = mfma_h(wh, eh, a_);
#pragma unroll
                                for (int r = 0; r < 4; ++r) ac[tc][r] += __builtin_ldexpf(a_[r], k2);
                            }
                        };
                        f32x4 sa[2] = {zero4, zero4}, sb[2] = {zero4, zero4};
                        f16x8 eha, ela, ehb, elb;
                        int k2a, k2b;
                        __builtin_amdgcn_sched_barrier(0);
                        chain1(c1a, sa);
                        __builtin_amdgcn_sched_barrier(0);
                        BIG_STAMP_P(10);
                        chain1(c1b, sb);
                        expsplit(2 * ip, xv[0], sa, eha, ela, k2a);
#pragma unroll
                        for (int i_ = 0; i_ < 6 * KS1; ++i_) {   // one product, then a handful of the exponentials' vector instructions
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        BIG_STAMP_P(11);
                        chain2(c1a + (4 * KS1 * 512) / 8, eha, ela, k2a, acc[0]);
                        expsplit(2 * ip + 1, xv[NSTG - 1], sb, ehb, elb, k2b);
#pragma unroll
                        for (int i_ = 0; i_ < 3 * NT; ++i_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        BIG_STAMP_P(12);
                        chain2(c1b + (4 * KS1 * 512) / 8, ehb, elb, k2b, acc[NSTG - 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (!done_) {
#pragma unroll
                for (int hs = 0; hs < NSTG; ++hs) {
                    const int hsx = WIDE ? hs : hsel;
                    const int j = 2 * ip + hsx;
                    if (tile_ok && j < nstage && !(DSA_BIG_ABL & 2)) {
                        const f16x8* c1 = reinterpret_cast<const f16x8*>(sbuf0 + (set * 2 + hsx) * SH) + lane;
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
                        if (hs == 0) BIG_STAMP_P(10);
                        float tv[8];
                        float tmax = -3.0e38f;
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const bool live = 32 * j + 16 * t + 4 * g + r < K;
                                const float v = __builtin_fmaf(xv[hs][t][r], 1.4426950408889634f, __builtin_ldexpf(s[t][r], k1));
                                tv[4 * t + r] = live ? v : -3.0e38f;
                                tmax = __builtin_fmaxf(tmax, tv[4 * t + r]);
                            }
                        tmax = rows_max4(tmax);
                        const float mi = __builtin_ceilf(tmax);
                        const float shf = (float)EMAX_LOG2 - mi;
                        float ev[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) ev[i] = __builtin_amdgcn_exp2f(tv[i] + shf);
                        f16x8 eh, el;
                        split8(ev, eh, el);
                        if (hs == 0) BIG_STAMP_P(11);
                        const int k2 = (int)mi - EMAX_LOG2 - LOG2_SE;
#pragma unroll
                        for (int tc = 0; tc < NT; ++tc) {
                            const f16x8 wh = w2[(tc * 2 + 0) * 64], wlo = w2[(tc * 2 + 1) * 64];
                            f32x4 a_ = mfma_h(wlo, eh, zero4);
                            a_ = mfma_h(wh, el, a_);
                            a_ = mfma_h(wh, eh, a_);
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[hs][tc][r] += __builtin_ldexpf(a_[r], k2);
                        }
                        if (hs == 0) BIG_STAMP_P(12);
                    }
                }
                }
                BIG_STAMP_P(13);
                if (ip + 1 < npair) stage(set ^ 1, st0);   // the other set: its readers finished before the barrier that ended pair ip - 1
                BIG_STAMP_P(14);
                __syncthreads();
                BIG_STAMP_P(15);
            };
#pragma unroll 1
            for (int ip = 0; ip < npair; ip += 2) {
                body(ip, x0);
                if (ip + 1 < npair) body(ip + 1, x1);
            }
            BIG_STAMP(2);
            // (the barrier that ended the last stage: every wave is done with the staging buffers -- the rt rows may take them)
            // C/D layout: lane (n, g), register r of tile tc <-> rt[16 tc + 4 g + r] of frame n
            if constexpr (WIDE) {
                // ---- the even and the odd stages' sums, added in the order the two waves of a narrow group add them ----
#pragma unroll
                for (int tc = 0; tc < NT; ++tc) *reinterpret_cast<f32x4*>(park + n * RTS + 16 * tc + 4 * g) = acc[0][tc] + acc[NSTG - 1][tc];
                __builtin_amdgcn_wave_barrier();
            } else {
                // ---- the two partial sums meet: odd stages' wave parks its rows, even stages' wave adds its own and parks the sums ----
                if (hsel == 1) {
#pragma unroll
                    for (int tc = 0; tc < NT; ++tc) *reinterpret_cast<f32x4*>(park + n * RTS + 16 * tc + 4 * g) = acc[0][tc];
                }
                __syncthreads();
                if (hsel == 0) {
#pragma unroll
                    for (int tc = 0; tc < NT; ++tc) {
                        f32x4* p4 = reinterpret_cast<f32x4*>(park + n * RTS + 16 * tc + 4 * g);
                        *p4 = acc[0][tc] + *p4;
                    }
                }
                __syncthreads();
            }
            BIG_STAMP(3);
            // ================= mc += solve(T(rt[:n]) + H(rt), rt[:n] - alpha_vec): eight systems per wave and round =================
            if (tile_ok && (WIDE || !QUAD || hsel == 0) && !(DSA_BIG_ABL & 1)) {
#pragma unroll 1
                for (int rnd = 0; rnd < ROUNDS; ++rnd) {
                    int ln = lane;
                    asm volatile("" : "+v"(ln));   // (lane-derived values re-derived here: see thsolve_octn_kernel)
                    const int row0 = WIDE ? 8 * rnd : (QUAD ? 0 : 8 * hsel);   // the group's rows this solve takes (QUAD: all 16)
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
                    BIG_STAMP(4 + 2 * rnd);
                    // the solve is a FUNCTION CALL: inlined, its 220-register matrix set the whole kernel's allocation and the stage loop's
                    // staged image pieces went through scratch (154 k cycles per step); behind a call boundary nothing of the stage loop
                    // is live here and nothing of the elimination is live there
                    if constexpr (QUAD) big_solve16q<NG, NMIN>((mbg::lds_f*)wl, (mbg::lds_f*)mcs, M1);
                    else big_solve8<NG, NMIN>((mbg::lds_f*)wl, (mbg::lds_f*)(mcs + row0 * MS), M1);
                    BIG_STAMP(5 + 2 * rnd);
                    __builtin_amdgcn_wave_barrier();   // (WIDE: the records are rebuilt for the second round)
                }
            }
        }
        // ---- the result: the group's rows of mc, RPW per wave ----
        __syncthreads();
        if (tile_ok) {
            for (int e = (tid & 63); e < RPW * M1; e += 64) {
                const int row = 8 * hsel + e / M1, col = e % M1;
                if (row < rows_here) mc_out[(tb + row) * (long)M1 + col] = mcs[row * MS + col];
            }
        }
#ifdef DSA_BIG_STAMPS
        if (blockIdx.x == 0 && pair == 0 && (tid & 63) == 0 && tile == 0)
            for (int i = 1; i < 16; ++i) mc_out[(8 * hsel) * (long)M1 + i] = (float)(tsv[i] - tsv[i - 1]);
#endif
    }
}

}  // namespace dsa
