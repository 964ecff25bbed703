// Split-precision backward of the tuned mel-cepstral analysis (included by mcep_mfma.hip after
// mcep_mfma_f16.h).  Same mathematics as mcep_mfma_bwd_kernel (reverse sweep over the unrolled Newton
// iteration from the saved iterates, mcep.py:189-224 under autograd), with every matrix chain on
// v_mfma_f32_16x16x32_f16 as three binary16 products per float32 operand pair:
//   forward re-computation   d^T = D^T mc_k^T,  rt^T = E^T e^T          (images DH/DL, EH/EL in LDS)
//   backward                 ebar^T = E rtbar^T                          (image EB, 64 KB, streamed from L2)
//                            mbar^T += (-2 D) zbar^T,  zbar = ebar * e   (image DB in LDS)
//   epilogue                 lbar^T += G mbar_0^T                        (image GB, streamed once per tile)
// The cotangent operands (rtbar, zbar, mbar_0) have no a-priori range, so each is scaled per frame by
// a power of two taken from its own maximum before the binary16 split (frames are MFMA columns: a
// per-frame scale factors out of the product exactly); e is scaled as in the forward kernel.
// One wave = 16 frames, 4 waves per workgroup (512 registers: log2 X, lbar, e and zbar stay resident),
// tiles drawn from a device counter.
#pragma once

namespace dsa {

namespace mhb {
using namespace mh;
constexpr int WAVES_B = 4;
constexpr float SEB = 65536.f;   // scale of the E image with bins as rows (|E| <= ~0.0044)
constexpr int SEB_LOG2 = 16;
constexpr float SDB = 256.f;     // scale of the -2 D image (|2 D| <= ~2.2)
constexpr int SDB_LOG2 = 8;
constexpr float SGB = 4096.f;    // scale of the G image with bins as rows (|G| <= 0.13)
constexpr int SGB_LOG2 = 12;
constexpr int VMAX_LOG2 = 13;    // per-frame scaled cotangents are below 2^13
// backward images (binary16 elements) behind the forward ones in the per-launch workspace
constexpr int IMG_EB = 16 * 2 * 64 * 8, IMG_DB = 2 * 8 * 64 * 8, IMG_GB = 16 * 64 * 8;
constexpr int B_BASE = IMG_BYTES / 2;                 // halves: forward images + the Nyquist row of G
constexpr int IMG_EBH = B_BASE, IMG_EBL = IMG_EBH + IMG_EB;
constexpr int IMG_DBH = IMG_EBL + IMG_EB, IMG_DBL = IMG_DBH + IMG_DB;
constexpr int IMG_GBH = IMG_DBL + IMG_DB, IMG_GBL = IMG_GBH + IMG_GB;
constexpr int IMG_B_HALVES = IMG_GBL + IMG_GB;
constexpr int TAIL_B = 32 + 64 + 256;                 // float32: -2 D[c][256] | E[256][m] | E[bin][48]
constexpr int IMG_B_BYTES = IMG_B_HALVES * 2 + TAIL_B * 4;
// LDS carve-up (float units)
constexpr int B_DB = EL_OFF + 24 * 64 * 4;            // DB hi | lo: 2 x 4096 floats
constexpr int B_E48 = B_DB + 2 * IMG_DB / 2;          // [16 mt][4 g][4 r]
constexpr int B_E256 = B_E48 + 256;                   // [48] scaled by SE, [48] unscaled, then [64] unscaled E[256][m] (0 past 48)
constexpr int B_D256 = B_E256 + 52 + 64;              // [32] -2 log2(e) D[c][256], then [32] -2 D[c][256]
constexpr int B_AV = B_D256 + 64;                     // [28]
constexpr int B_NAV = B_AV + 28;                      // [28] -alpha_vec, zero-padded
constexpr int B_ZERO = B_NAV + 28;                    // [28] zeros
constexpr int B_SLOT = B_ZERO + 28;                   // [4] the workgroup's first wave-slot number (split tail)
constexpr int B_WAVE = B_SLOT + 4;
constexpr int FS = 180;                               // per-frame record: rt [0,52) | rr [52,116) | aux [116,180)
constexpr int B_WAVE_FLOATS = 16 * FS;
constexpr int B_LDS_FLOATS = B_WAVE + WAVES_B * B_WAVE_FLOATS;
}  // namespace mhb

__global__ __launch_bounds__(256) void mcep_hb_prep_kernel(const float* __restrict__ G, const float* __restrict__ D,
                                                           const float* __restrict__ E, _Float16* __restrict__ img)
{
    using namespace mhb;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int i = idx & 7, l = (idx >> 3) & 63;
    if (idx < IMG_EB) {
        // ebar^T = E rtbar^T: rows = bins 16 mt + (l & 15), k-slot (g, i) of k-step ks <-> m = 32 ks + 8 g + i
        const int ks = (idx >> 9) & 1, mt = idx >> 10;
        const int m = 32 * ks + 8 * (l >> 4) + i;
        const float v = m < M2 ? SEB * E[(mt * 16 + (l & 15)) * M2 + m] : 0.f;
        split1(v, img[IMG_EBH + idx], img[IMG_EBL + idx]);
    } else if (idx < IMG_EB + IMG_DB) {
        // mbar^T += (-2 D) zbar^T: rows = coefficients 16 it2 + (l & 15), k-slot (g, i = 4 t + r) of step j
        // <-> bin 32 j + 16 t + 4 g + r (C/D register r of tile 2 j + t)
        const int e = idx - IMG_EB;
        const int j = (e >> 9) & 7, it2 = e >> 12;
        const int bin = 32 * j + 16 * (i >> 2) + 4 * (l >> 4) + (i & 3);
        const int c = it2 * 16 + (l & 15);
        const float v = c < M1 ? (-2.f * SDB) * D[c * K + bin] : 0.f;
        split1(v, img[IMG_DBH + e], img[IMG_DBL + e]);
    } else if (idx < IMG_EB + IMG_DB + IMG_GB) {
        // lbar^T += G mbar_0^T: rows = bins, one k-step: k-slot (g, i) <-> coefficient 8 g + i
        const int e = idx - IMG_EB - IMG_DB;
        const int mt = e >> 9;
        const int c = 8 * (l >> 4) + i;
        const float v = c < M1 ? SGB * G[(mt * 16 + (l & 15)) * M1 + c] : 0.f;
        split1(v, img[IMG_GBH + e], img[IMG_GBL + e]);
    }
    float* tail = reinterpret_cast<float*>(img + IMG_B_HALVES);
    if (idx < 32) tail[idx] = idx < M1 ? -2.f * D[idx * K + H] : 0.f;
    if (idx >= 32 && idx < 96) tail[idx] = idx - 32 < M2 ? E[H * M2 + idx - 32] : 0.f;
    if (idx >= 96 && idx < 96 + 256) tail[idx] = E[(idx - 96) * M2 + 48];
}

// v = hi + lo (binary16), eight values of one MFMA operand
__device__ __forceinline__ void split8(const float (&v)[8], f16x8& hi, f16x8& lo)
{
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        f16x2 h, l;
        split2(v[i], v[i + 1], h, l);
        hi[i] = h[0]; hi[i + 1] = h[1];
        lo[i] = l[0]; lo[i + 1] = l[1];
    }
}

// SAVED_RT (round 5): the forward kept every step's rt row (DSA_ALGO_HIST_HAS_RT, `hist_rt`: (n_iter, F, 49)); the step then loads it
// into the windows instead of re-running the second forward chain (72 binary16 products, the split of e, 27 image reads per step).
template <bool SAVED_RT>
__global__ __launch_bounds__(256, 1) DSA_PK_TARGET void mcep_mfma_bwd_kernel_h(
    const float* __restrict__ gmc, const float* __restrict__ X, const float* __restrict__ hist, long F, int n_iter,
    const float* __restrict__ av, float* gX, long ntiles16, unsigned int* __restrict__ queue,
    const _Float16* __restrict__ img, int split_tiles, int split_pieces, float* ws, const float* __restrict__ hist_rt)
{
    using namespace mhb;
    constexpr float kNeg2Log2e = -2.885390081777926815f;
    constexpr float kInvSDM = 1.f / (SD * SM);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t img_rsrc = image_rsrc(img, IMG_B_BYTES);   // the streamed E / G images (kernel argument: uniform)
    const int n = lane & 15, g = lane >> 4;

    // ---------------- operand images and small tables ----------------
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(img);
        f32x4* dst = reinterpret_cast<f32x4*>(lds + DH_OFF);
        for (int idx = tid; idx < (2 * IMG_D + 2 * IMG_E) / 8; idx += WAVES_B * 64) dst[idx] = src[idx];
        const f32x4* srcb = reinterpret_cast<const f32x4*>(img + IMG_DBH);
        f32x4* dstb = reinterpret_cast<f32x4*>(lds + B_DB);
        for (int idx = tid; idx < 2 * IMG_DB / 8; idx += WAVES_B * 64) dstb[idx] = srcb[idx];
    }
    const float* tail_f = reinterpret_cast<const float*>(img + IMG_HALVES);     // G[256][c] (forward workspace tail)
    const float* tail_b = reinterpret_cast<const float*>(img + IMG_B_HALVES);   // -2 D[c][256] | E[256][m] | E[bin][48]
    {
        const int r = tid & 3, gg = (tid >> 2) & 3, mt = tid >> 4;
        lds[B_E48 + tid] = tail_b[96 + mt * 16 + gg * 4 + r];
    }
    if (tid < 48) lds[B_E256 + tid] = SE * tail_b[32 + tid];     // Nyquist k-step of the rt chain (scaled image)
    if (tid == 48) lds[B_E256 + 48] = tail_b[32 + 48];            // rt[48] is a float32 dot product
    if (tid < 64) lds[B_E256 + 52 + tid] = tail_b[32 + tid];      // E[256][m], m < 64 (0 past 48): ebar of the Nyquist bin
    if (tid < 32) {
        lds[B_D256 + tid] = 1.4426950408889634f * tail_b[tid];    // -2 log2(e) D[c][256]
        lds[B_D256 + 32 + tid] = tail_b[tid];                     // -2 D[c][256]
    }
    if (tid < 28) {
        const float a_ = tid < M1 ? av[tid] : 0.f;
        lds[B_AV + tid] = a_;
        lds[B_NAV + tid] = -a_;
        lds[B_ZERO + tid] = 0.f;
    }
    // Split tail (split_tiles = R > 0; host: the launch would end in a round of R <= slots / 2 tiles, most wave slots idle for
    // a whole tile).  The backward sweep of a tile carries only (lbar, mbar) from step to step, so the LAST R tiles are cut into
    // P pieces of ~n_iter / P steps that hand (lbar -> the tile's rows of gX, mbar -> ws) over through memory, and R P slots each
    // take ONE piece next to their whole tiles: slot s = p R + j runs piece p of split tile j after its first p whole tiles, a
    // whole tile apart from piece p - 1.  Slot numbers come in ARRIVAL order (one atomic per workgroup): a wave only ever waits
    // for slots that started before it, whatever the dispatch order.  One wave per SIMD here: the slots run at one speed.
    if (split_tiles > 0 && tid == 0) reinterpret_cast<unsigned*>(lds + B_SLOT)[0] = atomicAdd(queue + 12, (unsigned)WAVES_B);
    __syncthreads();

    float* wave_lds = lds + B_WAVE + wave * B_WAVE_FLOATS;
    float* rt_n = wave_lds + n * FS;           // this lane's frame, MFMA-layout view
    float* rr_n = rt_n + 52;
    float* aux_n = rt_n + 116;
    const int nq = lane >> 2, gs = lane & 3;   // solve layout: a quad per frame
    float* rt_q = wave_lds + nq * FS;
    float* rr_q = rt_q + 52;
    float* aux_q = rt_q + 116;
    const GroupMask gq = make_group_mask(gs);
    const unsigned g_eq0 = g == 0 ? 0xffffffffu : 0u;
    int lane_a = lane, lane_b = lane + EL_OFF / 4, lane_c = lane + B_DB / 4;
    asm volatile("" : "+v"(lane_a), "+v"(lane_b), "+v"(lane_c));
    const f16x8* DH = reinterpret_cast<const f16x8*>(lds + DH_OFF) + lane_a;
    const f16x8* DL = reinterpret_cast<const f16x8*>(lds + DL_OFF) + lane_a;
    const f16x8* EH = reinterpret_cast<const f16x8*>(lds + EH_OFF) + lane_a;
    const f16x8* EL = reinterpret_cast<const f16x8*>(lds) + lane_b;
    const f16x8* DBH = reinterpret_cast<const f16x8*>(lds) + lane_c;
    const f16x8* DBL = DBH + IMG_DB / 8;
    const f32x4* E484 = reinterpret_cast<const f32x4*>(lds + B_E48);
    long wave_id = (long)blockIdx.x * WAVES_B + wave;
    const long wave_stride = (long)gridDim.x * WAVES_B;
    const long nwhole = ntiles16 - split_tiles;   // tiles [0, nwhole) run whole, [nwhole, ntiles16) in pieces
    int piece_mine = -1, split_j = 0;
    if (split_tiles > 0) {
        wave_id = (long)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const unsigned*>(lds + B_SLOT)[0]) + wave;
        const int pm = (int)(wave_id / split_tiles);
        if (pm < split_pieces) { piece_mine = pm; split_j = (int)(wave_id - (long)pm * split_tiles); }
    }
    long tile_whole = wave_id;   // the next whole tile of this wave (>= nwhole: none left)
    int whole_started = 0;
    for (;;) {
        // ---- this round's work item: a whole tile, or this wave's piece of a split tile (steps it_hi - 1 .. it_lo) ----
        long tile;
        int it_hi = n_iter, it_lo = 0;
        bool is_piece = false;
        if (piece_mine >= 0 && (whole_started >= piece_mine || tile_whole >= nwhole)) {
            is_piece = true;
            tile = nwhole + split_j;
            it_hi = n_iter - (int)((long)piece_mine * n_iter / split_pieces);
            it_lo = n_iter - (int)((long)(piece_mine + 1) * n_iter / split_pieces);
        } else if (tile_whole < nwhole) {
            tile = tile_whole;
            ++whole_started;
        } else {
            break;
        }
        const long f_raw = tile * 16 + n;
        const bool f_ok = f_raw < F;
        const long f = f_ok ? f_raw : F - 1;
        const float* xf = X + f * K;
        f32x4 logx[16], lbar[16];
#pragma unroll
        for (int mt = 0; mt < 16; ++mt) {
            const float* p = xf + mt * 16 + 4 * g;
            logx[mt] = f32x4{__log2f(p[0]), __log2f(p[1]), __log2f(p[2]), __log2f(p[3])};
            lbar[mt] = f32x4{0, 0, 0, 0};
        }
        const float logx256 = __log2f(xf[H]);
        float lbar256 = 0.f;
        // mbar in the C/D layout of a 32-row product: tile it2, register r <-> coefficient 16 it2 + 4 g + r
        f32x4 mbarC[2];
        if (it_hi < n_iter) {
            // a later piece: (lbar, mbar) as the previous piece left them, once ALL pieces of that level have been published
            // (device-coherent loads of just these values: a device-scope fence would flush the XCD's L2 under everyone)
            if (lane == 0)
                while (__hip_atomic_load(queue + 2 + piece_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)split_tiles)
                    __builtin_amdgcn_s_sleep(64);
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            const float* gxf = gX + f * K;
#pragma unroll
            for (int mt = 0; mt < 16; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lbar[mt][r] = __hip_atomic_load(gxf + mt * 16 + 4 * g + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lbar256 = __hip_atomic_load(gxf + H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float* wsf = ws + ((long)split_j * 16 + n) * 32;
#pragma unroll
            for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    mbarC[it2][r] = __hip_atomic_load(wsf + it2 * 16 + 4 * g + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
#pragma unroll
            for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = it2 * 16 + 4 * g + r;
                    mbarC[it2][r] = c < M1 ? gmc[f * M1 + c] : 0.f;
                }
        }
        if (!is_piece) {   // (a piece leaves the ticket already held untouched)
            unsigned int nxt = 0;
            if (lane == 0) nxt = atomicAdd(queue, 1u);
            tile_whole = wave_stride + (long)__builtin_amdgcn_readfirstlane((int)nxt);
        }

#ifdef DSA_MCEP_TIMING   // branch-free phase stamps in scalar registers, flushed at the end of the step (see DSA_STAMP)
#define BSTAMP(i) bst_[i] = (unsigned)__builtin_readcyclecounter()
#else
#define BSTAMP(i)
#endif
        float mcv_n[8], h0_n[KS], h1_n[KS];
        auto load_step = [&](int it_) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 8; ++i) mcv_n[i] = (8 * g + i < M1) ? hist[((long)it_ * F + f) * M1 + 8 * g + i] : 0.f;
            const long fq_raw = tile * 16 + nq;
            const float* h0 = hist + ((long)it_ * F + (fq_raw < F ? fq_raw : F - 1)) * M1;
            const float* h1 = it_ + 1 < n_iter ? h0 + F * M1 : h0;   // the last forward step's result is not in the history
#pragma unroll
            for (int c = 0; c < KS - 1; ++c) { h0_n[c] = h0[gs + 4 * c]; h1_n[c] = h1[gs + 4 * c]; }
            h0_n[KS - 1] = h0[M1 - 1]; h1_n[KS - 1] = h1[M1 - 1];
        };
        load_step(it_hi - 1);
        for (int iter = it_hi - 1; iter >= it_lo; --iter) {
#ifdef DSA_MCEP_TIMING
            unsigned bst_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
            BSTAMP(0);
            // this step's iterate and (for every step but the last of the forward sweep) the step's own solution
            // g = A^-1 (rt[:25] - alpha) as the difference of two SAVED iterates (mcep.py:224: mc <- mc + g): both were requested
            // one step ahead (a load from the history costs a round trip to memory at the head of every step otherwise)
            float mcv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) mcv[i] = mcv_n[i];
            const bool g_saved = iter + 1 < n_iter;
            float gh[KS];
#pragma unroll
            for (int c = 0; c < KS; ++c) gh[c] = h1_n[c] - h0_n[c];
            gh[KS - 1] = keep_if(gq.m[0], gh[KS - 1]);   // k = 24 on lane 0 only
            if (iter > 0) load_step(iter - 1);
            // SAVED_RT: this step's rt row, rt[16 it + 4 g + r] of this lane's frame, requested here -- behind the next step's
            // history requests, whose address reloads wait for everything in flight -- and consumed behind the first chain
            // (held a whole step ahead like the iterates, its 13 registers spilled: 1.61 ms against 1.57 without the saved row)
            f32x4 rt_c4[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            float rt48_c = 0.f;
            if (SAVED_RT) {
                const float* hr = hist_rt + ((long)iter * F + f) * M2 + 4 * g;
#pragma unroll
                for (int it = 0; it < 3; ++it) rt_c4[it] = *reinterpret_cast<const f32x4_u4*>(hr + 16 * it);
                rt48_c = hist_rt[((long)iter * F + f) * M2 + 48];
            }
            // ---------------- forward quantities of this step: e (kept, scaled by 2^sh), rt -> LDS windows ----------------
            f16x8 bh, bl;
            {
                float ms[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) ms[i] = mcv[i] * SM;
                split8(ms, bh, bl);
            }
            f32x4 ep[16];
            float d256 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) d256 = __builtin_fmaf(mcv[i], lds[B_D256 + 8 * g + i], d256);
            d256 = rows_sum4(d256);
            const float t256 = logx256 + d256;
            // The two forward chains as an explicit software pipeline (see the forward kernel): one binary16 product per slot, the
            // vector work of the previous group of tiles between the products, operand images read one body ahead; products
            // into one accumulator four slots apart (a dependent product waits out the full latency of its predecessor).
#define DSA_SB() __builtin_amdgcn_sched_barrier(0x0004)
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            float tmax = t256;
            {
                f16x8 al[4], ah[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { al[i] = DL[i * 64]; ah[i] = DH[i * 64]; }
                f32x4 c[4] = {zero4, zero4, zero4, zero4}, pc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const bool pm = q < 4, vw = q > 0;   // products of group q, vector work of group q - 1
                    f16x8 ah_n[4] = {ah[0], ah[1], ah[2], ah[3]};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (q < 3) ah_n[i] = DH[(4 * q + 4 + i) * 64];
#ifdef DSA_BWD_ABL_NOCHAIN1   // (timing only: the bound of hiding the first chain's products in another phase)
                        if (pm) c[i] = f32x4{mcv[0], mcv[1], mcv[2], mcv[3]} * (float)(4 * q + i);
#else
                        if (pm) c[i] = mfma_h(al[i], bh, zero4);
#endif
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
#ifndef DSA_BWD_ABL_NOCHAIN1
                        if (pm) c[i] = mfma_h(ah[i], bl, c[i]);
#endif
                        if (q < 3) al[i] = DL[(4 * q + 4 + i) * 64];
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
#ifndef DSA_BWD_ABL_NOCHAIN1
                        if (pm) c[i] = mfma_h(ah[i], bh, c[i]);
#endif
                        DSA_SB();
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) { pc[i] = c[i]; ah[i] = ah_n[i]; }
                }
            }
            BSTAMP(6);
            f16x8 eah[3], eal[3];
#pragma unroll
            for (int it = 0; it < 3; ++it) { eah[it] = EH[(it * 8) * 64]; eal[it] = EL[(it * 8) * 64]; }
            tmax = rows_max4(tmax);
            const float mi = __builtin_ceilf(tmax);
            const float sh = (float)EMAX_LOG2 - mi;
            const int back = (int)mi - EMAX_LOG2;   // e = 2^back ep
            const float e256 = __builtin_amdgcn_exp2f(t256 + sh);   // scaled like ep
            // the Nyquist bin preloads the accumulators of the second chain
            f32x4 accB[3];
#pragma unroll
            for (int it = 0; it < 3; ++it) accB[it] = *reinterpret_cast<const f32x4*>(lds + B_E256 + it * 16 + 4 * g) * e256;
            f32x2v rt48v = {0.f, 0.f};
            float rt48 = 0.f;
            f16x8 eh_p = {}, el_p = {};
            DSA_SB();
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                f16x8 eh = eh_p, el = el_p;
                f16x8 eah_n[3] = {eah[0], eah[1], eah[2]};
                f32x4 c48[2] = {zero4, zero4};
                if (j < 8) {
                    c48[0] = E484[(2 * j) * 4 + g];
                    c48[1] = E484[(2 * j + 1) * 4 + g];
                }
                if (j > 0 && j < 8) {
#pragma unroll
                    for (int it = 0; it < 3; ++it) eah_n[it] = EH[(it * 8 + j) * 64];
                }
                auto prodE = [&](int i) __attribute__((always_inline)) {   // i = 3 term + it
#ifndef DSA_BWD_ABL_NOCHAIN2
                    if (!SAVED_RT && j > 0) {
                        const int it = i % 3, term = i / 3;
                        accB[it] = mfma_h(term == 0 ? eal[it] : eah[it], term == 1 ? el_p : eh_p, accB[it]);
                    }
#endif
                };
                auto vecA = [&](int t_) __attribute__((always_inline)) {   // e = exp2(t + sh), kept for zbar = ebar * e
                    const int mt = (2 * j + t_) & 15;
                    const f32x2v ta = lo2(ep[mt]) + f32x2v{sh, sh}, tb = hi2(ep[mt]) + f32x2v{sh, sh};
                    ep[mt] = f32x4{__builtin_amdgcn_exp2f(ta[0]), __builtin_amdgcn_exp2f(ta[1]), __builtin_amdgcn_exp2f(tb[0]),
                                   __builtin_amdgcn_exp2f(tb[1])};
                };
                auto vecD = [&](int t_) __attribute__((always_inline)) {
                    const int mt = (2 * j + t_) & 15;
                    rt48v = lo2(ep[mt]) * lo2(c48[t_]) + rt48v;
                    rt48v = hi2(ep[mt]) * hi2(c48[t_]) + rt48v;
                };
                auto vecE = [&](int t_, int r) __attribute__((always_inline)) {
                    const int mt = (2 * j + t_) & 15;
                    f16x2 h, l;
                    split2(ep[mt][r], ep[mt][r + 1], h, l);
                    eh[4 * t_ + r] = h[0]; eh[4 * t_ + r + 1] = h[1];
                    el[4 * t_ + r] = l[0]; el[4 * t_ + r + 1] = l[1];
                };
                const bool vw = j < 8;   // body 8 only drains the second chain
                prodE(0); DSA_SB(); if (vw) vecA(0); DSA_SB();
                prodE(1); DSA_SB(); if (vw) vecA(1); DSA_SB();
                if (j == 8) rt48 = rows_sum4(rt48v[0] + rt48v[1]);
                prodE(2);
                if (j > 0 && j < 8) {
#pragma unroll
                    for (int it = 0; it < 3; ++it) eal[it] = EL[(it * 8 + j) * 64];
                }
                DSA_SB(); if (vw) vecD(0); DSA_SB();
                prodE(3); DSA_SB(); if (vw && !SAVED_RT) vecE(0, 0); DSA_SB();
                prodE(4); DSA_SB(); if (vw) vecD(1); DSA_SB();
                prodE(5); DSA_SB(); if (vw && !SAVED_RT) vecE(0, 2); DSA_SB();
                prodE(6); DSA_SB(); if (vw && !SAVED_RT) vecE(1, 0); DSA_SB();
                prodE(7); DSA_SB(); if (vw && !SAVED_RT) vecE(1, 2); DSA_SB();
                prodE(8); DSA_SB();
                eh_p = eh; el_p = el;
#pragma unroll
                for (int it = 0; it < 3; ++it) eah[it] = eah_n[it];
            }
            BSTAMP(7);
            rt48 = __builtin_fmaf(e256, lds[B_E256 + 48], rt48);
            rt48 = __builtin_ldexpf(rt48, back);
            if (SAVED_RT) rt48 = rt48_c;
            {
                int g_it = g;
                asm volatile("" : "+v"(g_it));
                float* rtw = rt_n + 4 * g_it;
                float* rra = rr_n + 27 + 4 * g_it;
                float* rrb = rr_n + 24 - 4 * g_it;              // rr[27 - idx], idx = 4 g + r: the lane's four entries reversed
                float* rra1 = g_it < 3 ? rra + 16 : rr_n + 55;
                float* rrb1 = g_it < 3 ? rrb - 16 : rr_n + 59;
                const int bk = back - SE_LOG2;
                // the lane's four consecutive entries as ONE 16-byte store per window (as the forward)
                f32x4 w0, w1, w2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    w0[r] = SAVED_RT ? rt_c4[0][r] : __builtin_ldexpf(accB[0][r], bk);
                    w1[r] = SAVED_RT ? rt_c4[1][r] : __builtin_ldexpf(accB[1][r], bk);
                    w2[r] = SAVED_RT ? rt_c4[2][r] : __builtin_ldexpf(accB[2][r], bk);
                }
                *reinterpret_cast<f32x4*>(rtw) = w0;
                *reinterpret_cast<f32x4_u4*>(rra) = w0;
                *reinterpret_cast<f32x4*>(rrb) = __builtin_shufflevector(w0, w0, 3, 2, 1, 0);
                *reinterpret_cast<f32x4*>(rtw + 16) = w1;
                *reinterpret_cast<f32x4_u4*>(rra1) = w1;
                *reinterpret_cast<f32x4_u4*>(rrb1) = __builtin_shufflevector(w1, w1, 3, 2, 1, 0);
                *reinterpret_cast<f32x4*>(rtw + 32) = w2;
                rt_n[48] = rt48;
                // mbar to the exchange window (C/D layout writer -> quad-layout reader)
#pragma unroll
                for (int it2 = 0; it2 < 2; ++it2)
                    *reinterpret_cast<f32x4*>(aux_n + it2 * 16 + 4 * g_it) = f32x4{mbarC[it2][0], mbarC[it2][1], mbarC[it2][2], mbarC[it2][3]};
            }
            __builtin_amdgcn_wave_barrier();

            BSTAMP(1);
            // ---------------- solve A [gv | uv] = [rt[:25] - alpha | mbar] in the quad layout ----------------
            // quad-layout solutions: xq1[c] = g[gs + 4 c], xq2[c] = u[gs + 4 c]
            float xq1[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, keep_if(gq.m[1], -1.f)};
            float xq2[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, keep_if(gq.m[2], -1.f)};
            {
                // The column-cyclic v_fmac_f32_dpp elimination of rounds 1-2 stays HERE: this kernel holds one wave per SIMD with
                // 512 registers, the compiler keeps the register quadruples of the 4 x 4 x 1 form (blk_*, the forward's solve) in
                // the accumulation file, and every vector instruction that touches an element then needs a v_accvgpr_read first
                // (two back substitutions read all 112): measured 2.19 ms against 2.03 ms with this form.
                float a[colm::TOTAL];
                {
                    // slot 6 of every row through two per-lane pointers (col_build_rows_p): lane 0 column 24, lane 1 the
                    // right-hand side rt[:25] - alpha, lane 2 the second right-hand side mbar (+ 0), lane 3 zeros
                    int gsv = gs;
                    asm volatile("" : "+v"(gsv));
                    const float* zr = lds + B_ZERO;
                    const float* pa6 = gsv == 0 ? rt_q + 24 : (gsv == 1 ? rt_q : (gsv == 2 ? aux_q : zr));
                    const float* pb6 = gsv == 0 ? rr_q + 3 : (gsv == 1 ? lds + B_NAV : zr);
                    col_build_rows_p<0>(a, rt_q, rr_q, pa6, pb6, gs);
                }
                __builtin_amdgcn_wave_barrier();
                col_elim_all(a, std::make_integer_sequence<int, M1>{});
                if (g_saved) {
#pragma unroll
                    for (int c = 0; c < KS; ++c) xq1[c] = gh[c];
                } else {
                    col_backsub_all(a, xq1, gq, std::make_integer_sequence<int, M1>{});
                    xq1[KS - 1] = keep_if(gq.m[0], xq1[KS - 1]);
                }
                col_backsub_all(a, xq2, gq, std::make_integer_sequence<int, M1>{});
                // slot 6 holds x[24] on lane 0 only (the other lanes' slot 6 are the right-hand-side markers)
                xq2[KS - 1] = keep_if(gq.m[0], xq2[KS - 1]);
            }
            BSTAMP(2);
            // ---------------- rtbar (49 entries), scaled per frame to below 2^13, into the exchange window ----------------
            int s_r;   // rtbar = 2^-s_r (window contents)
            {
                // rtbar[m] = -sum_{i+j=m} u_i g_j - [m<25] (sum_{|i-j|=m} u_i g_j - u_m), m = 0 .. 48, from the outer product
                // P = u g^T formed on the float32 matrix instruction: one v_mfma_f32_4x4x1 is the 4 x 4 block
                // u[4 ri ..] (x) g[4 cj ..] of all 16 frames of the wave (A = slot ri of u, B = slot cj of g: the solve's quad
                // layout as it stands).  Blocks on one block anti-diagonal (ri + cj = s) accumulate into DH[s], blocks on one
                // block diagonal (cj - ri = d) into DT[d + 6]: 2 x 49 products.  Entry [i'][j'] (register i', lane j') of DH[s]
                // belongs to m = 4 s + i' + j', of DT[d + 6] to the signed offset j - i = 4 d + j' - i'; the sums over the
                // entries of equal m are three quad rotations per accumulator (DPP operands of the additions) with a lane select
                // for the entries that spill into the neighbouring slot.  Result in the quad layout: lane w of slot s holds m = 4 s + w.
                // (Rounds 1-2: ~420 multiply-adds and 98 quad reductions per frame quad on the vector ALU, every lane forming all 49.)
                const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                f32x4 DH[13], DT[13];
#pragma unroll
                for (int k = 0; k < 13; ++k) { DH[k] = z4; DT[k] = z4; }
#pragma unroll
                for (int ri = 0; ri < KS; ++ri)
#pragma unroll
                    for (int cj = 0; cj < KS; ++cj) {
                        DH[ri + cj] = mfma441(xq2[ri], xq1[cj], DH[ri + cj]);
                        DT[cj - ri + 6] = mfma441(xq2[ri], xq1[cj], DT[cj - ri + 6]);
                    }
                // quad rotations: dst lane w <- src lane perm[w]
                auto rotR = [](float v, int k) __attribute__((always_inline)) {   // w <- (w - k) mod 4
                    const int ctrl = k == 1 ? 0x93 : (k == 2 ? 0x4E : 0x39);    // [3,0,1,2] [2,3,0,1] [1,2,3,0]
                    return k == 1 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x93, 0xf, 0xf, true))
                           : k == 2 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true))
                                    : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x39, 0xf, 0xf, true));
                    (void)ctrl;
                };
                auto rotL = [](float v, int k) __attribute__((always_inline)) {   // w <- (w + k) mod 4
                    return k == 1 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x39, 0xf, 0xf, true))   // [1,2,3,0]
                           : k == 2 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true)) // [2,3,0,1]
                                    : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x93, 0xf, 0xf, true)); // [3,0,1,2]
                };
                auto refl = [](float v, int k) __attribute__((always_inline)) {   // w <- (k - w) mod 4
                    return k == 0 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x6C, 0xf, 0xf, true))   // [0,3,2,1]
                           : k == 1 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true)) // [1,0,3,2]
                           : k == 2 ? __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xC6, 0xf, 0xf, true)) // [2,1,0,3]
                                    : __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x1B, 0xf, 0xf, true)); // [3,2,1,0]
                };
                float rb[13];   // quad layout: lane gs of slot s holds rtbar[4 s + gs]
#pragma unroll
                for (int sl = 0; sl < 13; ++sl) {
                    // Hankel part: entry (i', j') of DH[sl] with i' + j' < 4 stays in slot sl, i' + j' >= 4 belongs to slot sl + 1
                    float h = DH[sl][0];
#pragma unroll
                    for (int ip = 1; ip < 4; ++ip) {
                        const float prev = sl > 0 ? DH[sl - 1][ip] : 0.f;
                        h += rotR(gs < 4 - ip ? DH[sl][ip] : prev, ip);
                    }
                    float r = -h;
                    if (sl < 7) {
                        // Toeplitz part.  j - i = 4 d + (j' - i') >= 0 -> m = j - i: slot d (j' >= i') or d - 1 (j' < i')
                        float tp = DT[sl + 6][0];
#pragma unroll
                        for (int ip = 1; ip < 4; ++ip) {
                            const float nxt = sl + 7 < 13 ? DT[sl + 7][ip] : 0.f;
                            tp += rotL(gs >= ip ? DT[sl + 6][ip] : nxt, ip);
                        }
                        // i - j = -4 d + (i' - j') > 0 -> m = i - j: slot -d (j' <= i') or -d - 1 (j' > i')
                        float tn = 0.f;
#pragma unroll
                        for (int ip = 0; ip < 4; ++ip) {
                            const float far = 6 - sl - 1 >= 0 ? DT[6 - sl - 1][ip] : 0.f;
                            tn += refl(gs <= ip ? DT[6 - sl][ip] : far, ip);
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
                s_r = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(amax);
                // every lane stores its own entries: m = 4 s + gs for s < 13 (entries 49 .. 51 are sums of nothing: 0), zeros to 62,
                // the scale in slot 63
#pragma unroll
                for (int sl = 0; sl < 13; ++sl) aux_q[4 * sl + gs] = __builtin_ldexpf(rb[sl], s_r);
                aux_q[52 + gs] = 0.f;
                aux_q[56 + gs] = 0.f;
                aux_q[60 + gs] = gs == 3 ? __int_as_float(s_r) : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
            f16x8 rbh[2], rbl[2];
            float eb256 = 0.f;
            const int s_rn = __float_as_int(aux_n[63]);   // the scale of THIS lane's frame in the MFMA layout
            {
                float rv[16];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    rv[i] = aux_n[8 * g + i];
                    rv[8 + i] = g < 3 ? aux_n[32 + 8 * g + i] : 0.f;     // slot 63 of group 3 holds the scale, not data
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    eb256 = __builtin_fmaf(rv[i], lds[B_E256 + 52 + 8 * g + i], eb256);
                    eb256 = __builtin_fmaf(rv[8 + i], lds[B_E256 + 52 + 32 + 8 * g + i], eb256);
                }
                float lo8[8], hi8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { lo8[i] = rv[i]; hi8[i] = rv[8 + i]; }
                split8(lo8, rbh[0], rbl[0]);
                split8(hi8, rbh[1], rbl[1]);
            }
            __builtin_amdgcn_wave_barrier();
            eb256 = rows_sum4(eb256);

            BSTAMP(3);
            // ---------------- ebar^T = E rtbar^T ; zbar = ebar * e ; lbar += zbar ----------------
            // zbar = acc * ep * 2^kz,  kz = -(s_r + SEB_LOG2) + back
            const int kz = back - s_rn - SEB_LOG2;
            f32x4 zb[16];
            float zmax = 0.f;
            {
                // groups of four tiles: 24 products (2 k-steps x 3 terms x 4 tiles; the four tiles' accumulators take turns, so a
                // product is four slots behind its predecessor in the same accumulator), the streamed E image of the NEXT group
                // requested while this group multiplies, the vector work of the PREVIOUS group between the products
                const unsigned lane16 = (unsigned)lane * 16u;
                f16x8 ah[4][2], al[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        ah[i][ks] = gload8(img_rsrc, lane16, 2 * (IMG_EBH + (i * 2 + ks) * 512));
                        al[i][ks] = gload8(img_rsrc, lane16, 2 * (IMG_EBL + (i * 2 + ks) * 512));
                    }
                f32x4 acc[4] = {zero4, zero4, zero4, zero4}, pacc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const bool pm = q < 4, vw = q > 0;
                    f16x8 ah_n[4][2], al_n[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) { ah_n[i][ks] = ah[i][ks]; al_n[i][ks] = al[i][ks]; }
                    auto vec = [&](int i, int half) __attribute__((always_inline)) {
                        // zbar = ebar * e (mcep.py:212 adjoint): two bins per instruction, exact power-of-two rescale
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
                    for (int term = 0; term < 6; ++term) {   // (k-step, term) = (0, lo hi) (0, hi lo) (0, hi hi) (1, ..) ..
                        const int ks = term / 3, tr = term % 3;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (pm) acc[i] = mfma_h(tr == 0 ? al[i][ks] : ah[i][ks], tr == 1 ? rbl[ks] : rbh[ks], term == 0 ? zero4 : acc[i]);
                            // the next group's image: each operand register is requested right after its last use
                            if (q < 3 && tr == 0) al_n[i][ks] = gload8(img_rsrc, lane16, 2 * (IMG_EBL + ((4 * q + 4 + i) * 2 + ks) * 512));
                            if (q < 3 && tr == 2) ah_n[i][ks] = gload8(img_rsrc, lane16, 2 * (IMG_EBH + ((4 * q + 4 + i) * 2 + ks) * 512));
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
            const float zb256 = __builtin_ldexpf(eb256 * e256, back - s_rn);
            lbar256 += zb256;
            BSTAMP(4);
            // ---------------- mbar^T += (-2 D) zbar^T : zbar scaled per frame to below 2^13 ----------------
            zmax = rows_max4(zmax);
            const int s_z = VMAX_LOG2 - __builtin_amdgcn_frexp_expf(zmax);
            {
                // body j: the six products of bins 32 (j - 1) .. into FOUR accumulators (coefficient tile x parity of j: a product
                // is four slots behind its predecessor in the same accumulator), around the rescale + binary16 split of bins 32 j ..
                f32x4 acc2[2][2] = {{zero4, zero4}, {zero4, zero4}};
                f16x8 zh_p = {}, zl_p = {};
                f16x8 dh[2], dl[2];
#pragma unroll
                for (int it2 = 0; it2 < 2; ++it2) { dh[it2] = DBH[(it2 * 8) * 64]; dl[it2] = DBL[(it2 * 8) * 64]; }
                DSA_SB();
                BSTAMP(8);
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    f16x8 zh = zh_p, zl = zl_p;
                    f16x8 dh_n[2] = {dh[0], dh[1]}, dl_n[2] = {dl[0], dl[1]};
                    if (j > 0 && j < 8) {
#pragma unroll
                        for (int it2 = 0; it2 < 2; ++it2) { dh_n[it2] = DBH[(it2 * 8 + j) * 64]; dl_n[it2] = DBL[(it2 * 8 + j) * 64]; }
                    }
                    auto prodM = [&](int i) __attribute__((always_inline)) {   // i = 2 term + it2
                        if (j > 0) {
                            const int it2 = i & 1, term = i >> 1;
                            f32x4& a2 = acc2[it2][(j - 1) & 1];
                            a2 = mfma_h(term == 0 ? dl[it2] : dh[it2], term == 1 ? zl_p : zh_p, a2);
                        }
                    };
                    auto vecZ = [&](int t_, int r) __attribute__((always_inline)) {
                        const int mt = (2 * j + t_) & 15;
                        f16x2 h, l;
                        split2(__builtin_ldexpf(zb[mt][r], s_z), __builtin_ldexpf(zb[mt][r + 1], s_z), h, l);
                        zh[4 * t_ + r] = h[0]; zh[4 * t_ + r + 1] = h[1];
                        zl[4 * t_ + r] = l[0]; zl[4 * t_ + r + 1] = l[1];
                    };
                    const bool vw = j < 8;
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
                BSTAMP(9);
#pragma unroll
                for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = it2 * 16 + 4 * g + r;   // < 32
                        mbarC[it2][r] += __builtin_ldexpf(acc2[it2][0][r] + acc2[it2][1][r], -s_z - SDB_LOG2);
                        mbarC[it2][r] = __builtin_fmaf(zb256, lds[B_D256 + 32 + c], mbarC[it2][r]);   // Nyquist bin (table is 0 past c = 24)
                    }
            }
#undef DSA_SB
            BSTAMP(5);
#ifdef DSA_MCEP_TIMING
            if (blockIdx.x == 0 && threadIdx.x == 0 && tile == wave_id && iter == n_iter - 2)
                for (int i_ = 0; i_ < 16; ++i_) g_mcep_stamps[40 + i_] = bst_[i_];
#endif
        }

        if (it_lo > 0) {
            // hand over: lbar into the tile's rows of gX (the last piece overwrites them with gX), mbar into ws; written through
            // to device scope and acknowledged, then the level's counter
            float* gxf = gX + f * K;
            float* wsf = ws + ((long)split_j * 16 + n) * 32;
            if (f_ok) {
#pragma unroll
                for (int mt = 0; mt < 16; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __hip_atomic_store(gxf + mt * 16 + 4 * g + r, lbar[mt][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (g == 0) __hip_atomic_store(gxf + H, lbar256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int it2 = 0; it2 < 2; ++it2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __hip_atomic_store(wsf + it2 * 16 + 4 * g + r, mbarC[it2][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) atomicAdd(queue + 3 + piece_mine, 1u);
            piece_mine = -1;
            continue;
        }
        if (is_piece) piece_mine = -1;
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
            const unsigned lane16 = (unsigned)lane * 16u;
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
                const f16x8 ah = gload8(img_rsrc, lane16, 2 * (IMG_GBH + mt * 512)), al = gload8(img_rsrc, lane16, 2 * (IMG_GBL + mt * 512));
                f32x4 acc = {0, 0, 0, 0};
                acc = mfma_h(al, mh8, acc);
                acc = mfma_h(ah, ml8, acc);
                acc = mfma_h(ah, mh8, acc);
                if (f_ok) {
                    float* dst = gX + f * K + mt * 16 + 4 * g;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dst[r] = (lbar[mt][r] + __builtin_ldexpf(acc[r], -s_m - SGB_LOG2)) * __builtin_amdgcn_exp2f(-logx[mt][r]);
                }
            }
        }
        if (f_ok && g == 0) gX[f * K + H] = lbar256 * __builtin_amdgcn_exp2f(-logx256);
    }
}

}  // namespace dsa
