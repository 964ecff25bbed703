// MOVED OUT OF THE PRODUCT TREE in round 3 (was diffsptk_amd/csrc/stft_mfma.h, compiled out since round 2): the matrix-core STFT
// below leaves one stale bin in about one launch of three and is slower than the packed register FFT of stft_pk.h (119-129 us
// against 69-73 us per 204 800 frames).  Kept for the record; nothing builds it.
// Matrix-core STFT for fft_length 512, float32 (included by stft.hip).
// ShortTimeFourierTransform._forward, stft.py:237-241 = Frame (frame.py:120-141) -> Window
// (window.py:185-193) -> rfft (fftr.py:136-151) -> |.|^2 + eps / formatter (spec.py:152-178),
// for the real-valued output formats without relative floor and without per-frame mean removal
// (those options keep the register-FFT kernel stft512_fwd_kernel).
//
// The register FFT is bound by vector issue (786 VALU instructions per 4 frames) and by the LDS
// transposes between its radix-16 stages.  Here the 512-point real DFT is decimated in time by 8:
//     n = 8 m + p :   S_p[kk] = sum_m xw[8 m + p] W64^(m kk)        (eight 64-point real-input DFTs)
//     X[kk + 64 j]  = sum_p W8^(p j) ( W512^(p kk) S_p[kk] )         (twiddle + 8-point DFT over p)
// The sub-DFTs of 16 frames are ONE real matrix product per phase, 64 x 64 (rows: Re/Im of
// kk = 0..32, real input => the other half is the conjugate) times 64 x 16 (columns: frames), on
// v_mfma_f32_16x16x32_f16 in split precision (every float32 operand = binary16 hi + lo, three
// products accumulated in float32: see mcep_mfma_f16.h; the data operand is scaled per frame by a
// power of two, so dynamic range between neighbouring frames costs nothing).  Frames are the MFMA N
// dimension and every phase has its own accumulators, hence the 8 values S_0..7[kk] of one frame
// land in the SAME lane and register index of 8 accumulators: the twiddle and the 8-point DFT
// are plain in-lane vector code, Re/Im in adjacent registers (rows are ordered for that).
// Because the input is real, X[512 - k] = conj X[k]: the 8 outputs of kk = 1..31 are the bins
// kk + 64 j (j = 0..3) and 64 - kk + 64 (7 - j) (j = 4..7); the two real-only rows (kk = 0 and 32)
// share slot 0 and get the bins 0, 64, .., 256 and 32, 96, 160, 224.
//   * a lane reads its k-slots straight from global memory: for the k-slot order m = 32 ks + 8 g + i
//     the 64 samples 8 m + p (i, p = 0..7) of a lane are one contiguous 256-byte run;
//   * the 16 x 257 output tile is staged in wave-private LDS (row stride 257 = layout in HBM) and
//     leaves as one contiguous run of 16-byte stores.
// Per 16 frames: 192 MFMAs (3.1 k matrix-pipe cycles) + ~1.3 k vector instructions, against
// 3.1 k vector instructions for the register FFT.  tools/proto_stft_mfma.py is the numerical model.
//
// STATUS: EXPERIMENTAL, opt-in (DSA_STFT_VARIANT=1).  Parity-correct (tools/ab_stft.py: max error
// 4.6e-7 of the frame maximum) and 10-15 % faster than the register FFT (119-129 us vs 131-150 us per
// 204 800 frames), but NOT deterministic: in about one launch out of three at that size ONE output bin
// of ONE 16-frame tile (always a slot of lane group 3, first register pair) keeps the value a
// previous tile left in the LDS staging row -- 16 of 5.3e7 outputs (tools/dbg_stft3.py).  Ruled out so
// far: MFMA-result latency (128-cycle s_sleep between the products and their first use), the inline
// assembly below (a plain-vector-code build shows it too), strict aliasing, LDS bank-conflict layout
// (a phase-major staging variant shows it more often), s_waitcnt vmcnt(0) before the staged samples
// are consumed, s_waitcnt lgkmcnt(0) after every group of staging writes, early-clobber outputs on
// the asm helpers (destination != source).  A sentinel-filled staging tile shows the bad bins ARE
// written, with a wrong value: an intermediate of the twiddle / DFT-8 block is wrong in lanes 48..63
// (the last quarter of a wave64 vector operation) of one instruction -- tools/dbg_stft4.py.  Until that is root-caused the dispatcher keeps
// the register-FFT kernel as the default.
#pragma once

#include <mutex>
#include <vector>

namespace dsa {

typedef _Float16 sf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sf16x2 __attribute__((ext_vector_type(2)));
typedef float sf32x2 __attribute__((ext_vector_type(2)));
typedef float sf32x4 __attribute__((ext_vector_type(4)));

namespace sm {
constexpr int WAVES = 8;
constexpr int K = 257;
constexpr float SA = 1024.f;     // scale of the DFT-64 operand image (entries in [-1, 1])
constexpr int SA_LOG2 = 10;
constexpr int XMAX_LOG2 = 14;    // the largest scaled sample of a frame is in [2^13, 2^14)
constexpr int IMG_HALVES = 4 * 2 * 64 * 8;   // [4 mt][2 ks][64 lanes][8]  (hi image, then lo image)
// constant tables in global memory, built once (float32 words): image hi | image lo | tw[32][8] | tw16[8]
constexpr int TAB_IMG_WORDS = 2 * IMG_HALVES / 2;
constexpr int TAB_TW = TAB_IMG_WORDS;            // float2 [32 slots][8 phases]: W512^(p kk)
constexpr int TAB_TW16 = TAB_TW + 32 * 8 * 2;    // float2 [8]: W16^p
constexpr int TAB_WORDS = TAB_TW16 + 16;
// LDS carve-up (float words)
constexpr int L_IMG = 0;                      // hi | lo images: 4096 + 4096 words
constexpr int L_TW = L_IMG + TAB_IMG_WORDS;   // 512 words
constexpr int L_TW16 = L_TW + 512;            // 16 words
constexpr int L_WIN = L_TW16 + 16;            // 512 words (window, zero past frame_length)
constexpr int L_STAGE = L_WIN + 512;          // per wave: 16 x 257 words (+ 4 pad)
constexpr int STAGE_WORDS = 16 * K + 4;
constexpr int LDS_WORDS = L_STAGE + WAVES * STAGE_WORDS;
}  // namespace sm

// ---- constant tables: float64 on the host, uploaded once ----
static const float* stft_mfma_tables()
{
    static float* dev = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        using namespace sm;
        std::vector<float> tab(TAB_WORDS, 0.f);
        _Float16* hi = reinterpret_cast<_Float16*>(tab.data());
        _Float16* lo = hi + IMG_HALVES;
        const double two_pi = 6.283185307179586476925286766559;
        for (int idx = 0; idx < IMG_HALVES; ++idx) {
            const int i = idx & 7, l = (idx >> 3) & 63, ks = (idx >> 9) & 1, mt = idx >> 10;
            const int m = 32 * ks + 8 * (l >> 4) + i;        // k-slot (g, i) of k-step ks
            const int rho = 16 * mt + (l & 15);               // row: slot rho >> 1, part rho & 1
            const int slot = rho >> 1, part = rho & 1;
            double v;
            if (slot == 0) v = part == 0 ? 1.0 : ((m & 1) ? -1.0 : 1.0);  // kk = 0 | kk = 32 (both real)
            else v = part == 0 ? cos(two_pi * m * slot / 64.0) : -sin(two_pi * m * slot / 64.0);
            const float vs = (float)(v * SA);
            const _Float16 h = (_Float16)vs;
            hi[idx] = h;
            lo[idx] = (_Float16)(vs - (float)h);
        }
        for (int slot = 0; slot < 32; ++slot)
            for (int p = 0; p < 8; ++p) {
                const double a = -two_pi * p * slot / 512.0;   // slot 0 -> 1
                tab[TAB_TW + 2 * (slot * 8 + p)] = (float)cos(a);
                tab[TAB_TW + 2 * (slot * 8 + p) + 1] = (float)sin(a);
            }
        for (int p = 0; p < 8; ++p) {
            const double a = -two_pi * p / 16.0;
            tab[TAB_TW16 + 2 * p] = (float)cos(a);
            tab[TAB_TW16 + 2 * p + 1] = (float)sin(a);
        }
        float* d = nullptr;
        if (hipMalloc((void**)&d, TAB_WORDS * sizeof(float)) != hipSuccess) return;
        if (hipMemcpy(d, tab.data(), TAB_WORDS * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            hipFree(d);
            return;
        }
        dev = d;
    });
    return dev;
}

// Packed complex arithmetic on (re, im) register pairs.  The half-swaps and sign flips of complex
// products and of multiplications by -i are operand modifiers of the packed instructions (op_sel /
// neg_lo / neg_hi); hipcc materialises them as extra v_mov / v_xor, hence the inline assembly.
__device__ __forceinline__ sf32x2 cmul2(sf32x2 a, sf32x2 t)
{
    // t1 = (a.x t.x, a.y t.x) in plain C: `a` comes straight out of MFMA accumulators, and only an
    // instruction the compiler knows gets the MFMA-result wait states in front of it;
    // r = (t1.x - a.y t.y, t1.y + a.x t.y)
    const sf32x2 t1 = a * sf32x2{t.x, t.x};
    sf32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(t), "v"(t1));
    return r;
}
__device__ __forceinline__ sf32x2 add_negi(sf32x2 a, sf32x2 b)  // a - i b = (a.x + b.y, a.y - b.x)
{
    sf32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ sf32x2 add_posi(sf32x2 a, sf32x2 b)  // a + i b = (a.x - b.y, a.y + b.x)
{
    sf32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ sf32x2 diff_sum(sf32x2 b)  // (b.x - b.y, b.x + b.y)
{
    sf32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(r) : "v"(b), "v"(b));
    return r;
}

// out[j] = sum_p W8^(p j) u[p]: 26 packed instructions
__device__ __forceinline__ void dft8(const sf32x2 (&u)[8], sf32x2 (&y)[8])
{
    const sf32x2 r2 = {0.70710678118654752f, 0.70710678118654752f};
    const sf32x2 s04 = u[0] + u[4], d04 = u[0] - u[4], s26 = u[2] + u[6], d26 = u[2] - u[6];
    const sf32x2 s15 = u[1] + u[5], d15 = u[1] - u[5], s37 = u[3] + u[7], d37 = u[3] - u[7];
    const sf32x2 a0 = s04 + s26, a2 = s04 - s26, a1 = add_negi(d04, d26), a3 = add_posi(d04, d26);
    const sf32x2 b0 = s15 + s37, b2 = s15 - s37, b1 = add_negi(d15, d37), b3 = add_posi(d15, d37);
    // W8^1 b1 = r2 (b1.x + b1.y, b1.y - b1.x);  W8^2 b2 = -i b2;  W8^3 b3 = -r2 (b3.x - b3.y, b3.x + b3.y)
    const sf32x2 t1 = add_negi(b1, b1), t3 = diff_sum(b3);
    y[0] = a0 + b0; y[4] = a0 - b0;
    y[1] = t1 * r2 + a1; y[5] = a1 - t1 * r2;
    y[2] = add_negi(a2, b2); y[6] = add_posi(a2, b2);
    y[3] = a3 - t3 * r2; y[7] = t3 * r2 + a3;
}

template <bool PLAIN>
__global__ __launch_bounds__(sm::WAVES * 64, 2) void stft512_mfma_kernel(
    const float* __restrict__ x, long Tlen, long N, int L, int P, int left, int mode,
    const float* __restrict__ w, const float* __restrict__ tab, float eps, int fmt,
    float* __restrict__ y, long total_tiles, int tiles_per_utt)
{
    using namespace sm;
    (void)mode;  // constant padding only (the dispatcher keeps the other modes on the register-FFT kernel)
    extern __shared__ __attribute__((aligned(16))) float slds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: tiles, descriptors
    const int c = lane & 15, g = lane >> 4;

    for (int i = tid; i < TAB_IMG_WORDS + 512 + 16; i += WAVES * 64) slds[i] = tab[i];
    for (int i = tid; i < 512; i += WAVES * 64) slds[L_WIN + i] = i < L ? w[i] : 0.f;
    __syncthreads();  // the only workgroup barrier

    const sf16x8* IMGH = reinterpret_cast<const sf16x8*>(slds + L_IMG) + lane;
    const sf16x8* IMGL = IMGH + IMG_HALVES / 8;
    const sf32x2* TW = reinterpret_cast<const sf32x2*>(slds + L_TW);
    const sf32x2* TW16 = reinterpret_cast<const sf32x2*>(slds + L_TW16);
    float* stage = slds + L_STAGE + wave * STAGE_WORDS;
    float* stage_row = stage + c * K;
    const unsigned g_is0 = g == 0 ? 0xffffffffu : 0u;

    const long wave_id = (long)blockIdx.x * WAVES + wave;
    const long wave_stride = (long)gridDim.x * WAVES;
#ifdef DSA_STFT_TIMING
#define MSTAMP(i)                                                                                  \
    do {                                                                                           \
        if (blockIdx.x == 0 && threadIdx.x == 0 && tile == wave_id + 2 * wave_stride)              \
            g_stft_stamps[i] = __builtin_readcyclecounter();                                       \
    } while (0)
#else
#define MSTAMP(i)
#endif
    // Stage-in of a tile: the 15 P + L samples its 16 frames share, in fully coalesced 16-byte loads.
    // (Letting every lane load its own 256-byte run from global memory costs one L1 tag lookup per lane
    // and instruction: the kernel then runs at the texture pipe's pace, 3x slower.)  The loads go
    // through a buffer descriptor of the utterance [xb, xb + T): the hardware range check returns 0 for
    // every dword outside it, which IS constant (zero) padding at both edges -- no branches.  They are
    // issued one tile ahead (8 x 4 registers), so HBM latency hides behind the matrix products.
    constexpr int PRE = 8;  // 16-byte pieces per lane: up to 2048 samples per tile
    const int n4 = (15 * P + L + 3) >> 2;
    sf32x4 pre[PRE];
    auto prefetch = [&](long tl) __attribute__((always_inline)) {
        // branch-free: past the last tile the descriptor covers 0 bytes (every load returns 0), and the
        // pieces past the 15 P + L samples get an offset outside any utterance
        const bool live = tl < total_tiles;
        const long tb = live ? tl / tiles_per_utt : 0;
        const long t_tile = (tl - tb * tiles_per_utt) * 16 * (long)P - left;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)(x + tb * Tlen), 0, live ? (int)(Tlen * 4) : 0, 0x00020000);
#pragma unroll
        for (int it = 0; it < PRE; ++it) {
            const int s4i = lane + 64 * it;
            const unsigned off = s4i < n4 ? (unsigned)((t_tile + 4 * s4i) * 4) : 0xfffffff0u;
            pre[it] = __builtin_bit_cast(sf32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0));
        }
    };
    prefetch(wave_id);
    const float* run = stage + c * P + 64 * g;           // this lane's run: samples 0 .. 63 (+ 256) of its frame
    const bool head_full = L >= 256;                      // every sample of k-step 0 is inside the frame
    for (long tile = wave_id; tile < total_tiles; tile += wave_stride) {
        MSTAMP(0);
        const long b = tile / tiles_per_utt;
        const long f0 = (tile - b * tiles_per_utt) * 16;
        const int nvalid = (int)((N - f0) < 16 ? (N - f0) : 16);
        const bool fvalid = c < nvalid;

        // ---------------- this lane's 2 x 64 samples: window, per-frame scale, binary16 split ----------------
        // B operand of phase p, k-step ks: slot i <-> sample n = 256 ks + 64 g + 8 i + p.
        // Two passes over the same 2 x 16 pieces of 16 bytes: first the per-frame maximum of |x w|, then
        // scale, split and pack -- the raw samples are never all live next to the 128 registers of
        // packed operands.
        {
            sf32x4* st4 = reinterpret_cast<sf32x4*>(stage);
#pragma unroll
            for (int it = 0; it < PRE; ++it) {
                const int s4i = lane + 64 * it;
                st4[s4i] = pre[it];   // (pieces past the 15 P + L samples hold zeros; the tile has room for all 8 KB)
            }
        }
        __builtin_amdgcn_wave_barrier();
        // a piece = 4 consecutive samples of this lane's run, times the window; samples past the frame are
        // never read (zero padding is exact; non-finite neighbours stay out of frames that do not
        // contain them: frame.py:135-138).  frame_length and frame_period are multiples of 4 here.
        auto piece = [&](int ks, int q, unsigned pass_tag) __attribute__((always_inline)) {
            (void)pass_tag;
            const int n = 256 * ks + 64 * g + 4 * q;   // frame-relative index of the first sample
            sf32x4 v = {0.f, 0.f, 0.f, 0.f};
            // (k-step 0 of the usual frame lengths >= 256 needs no per-lane test: one uniform branch)
            if ((ks == 0 && head_full) ? fvalid : (fvalid && n < L))
                v = *reinterpret_cast<const sf32x4*>(run + 256 * ks + 4 * q) * *reinterpret_cast<const sf32x4*>(slds + L_WIN + n);   // window.py:190
            return v;
        };
        float amax = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const sf32x4 v = piece(ks, q, 0u);
                amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v.x)), __builtin_fabsf(v.y));
                amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v.z)), __builtin_fabsf(v.w));
            }
        MSTAMP(1);
        amax = __builtin_fmaxf(amax, __shfl_xor(amax, 16, 64));
        amax = __builtin_fmaxf(amax, __shfl_xor(amax, 32, 64));
        // amax = 2^(ex-1) [1, 2):  scaled samples are below 2^14
        const int ex = __builtin_amdgcn_frexp_expf(amax);
        const float scale = __builtin_ldexpf(1.f, XMAX_LOG2 - ex);
        // |X|^2 = (accumulated value)^2 * 4^-(SA_LOG2 + XMAX_LOG2 - ex)
        const float c2 = __builtin_ldexpf(1.f, 2 * (ex - XMAX_LOG2 - SA_LOG2));
        sf16x8 bh[2][8], bl[2][8];
        const unsigned tag = 0u;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int qq = 0; qq < 16; qq += 4)
#pragma unroll
                for (int qo = 0; qo < 2; ++qo) {
                    // pieces q and q + 2 hold slots i = q >> 1 and i + 1 of the phases 4 (q & 1) .. + 3
                    const int q = qq + qo, i = q >> 1, pb = 4 * (q & 1);
                    const sf32x4 va = piece(ks, q, tag) * scale, vb = piece(ks, q + 2, tag) * scale;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const sf16x2 h = __builtin_convertvector(sf32x2{va[j], vb[j]}, sf16x2);
                        const sf16x2 l = __builtin_convertvector(sf32x2{va[j] - (float)h[0], vb[j] - (float)h[1]}, sf16x2);
                        bh[ks][pb + j][i] = h[0]; bh[ks][pb + j][i + 1] = h[1];
                        bl[ks][pb + j][i] = l[0]; bl[ks][pb + j][i + 1] = l[1];
                    }
                }

        MSTAMP(2);
        __builtin_amdgcn_wave_barrier();  // the staged samples are consumed: the tile now collects the output
        // ---------------- per 16 rows: eight sub-DFT products, then twiddle + DFT-8 over the phases ----------------
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            sf32x4 acc[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[p] = sf32x4{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const sf16x8 ah = IMGH[(mt * 2 + ks) * 64], al = IMGL[(mt * 2 + ks) * 64];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[ks][p], acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[ks][p], acc[p], 0, 0, 0);
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[ks][p], acc[p], 0, 0, 0);
                }
            }
            if (mt == 3) {
                // The next tile's samples are requested as soon as the packed operands are dead (after the
                // last matrix product): HBM latency hides behind the last twiddle / DFT-8 block and the
                // output copy, and the 32 prefetch registers never coexist with the 128 operand registers.
                __builtin_amdgcn_sched_barrier(0);
                prefetch(tile + wave_stride);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int slot0 = 8 * mt + h;           // + 2 g: this lane's slot kk
                const int kk = slot0 + 2 * g;
                const sf32x2* twl = TW + (slot0 + 2 * g) * 8;
                sf32x2 u[8], yv[8];
                u[0] = sf32x2{acc[0][2 * h], acc[0][2 * h + 1]};
#pragma unroll
                for (int p = 1; p < 8; ++p) u[p] = cmul2(sf32x2{acc[p][2 * h], acc[p][2 * h + 1]}, twl[p]);
                dft8(u, yv);
                float pw[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pw[j] = yv[j].x * yv[j].x + yv[j].y * yv[j].y;
                int kb[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    kb[j] = kk + 64 * j;
                    kb[4 + j] = 64 - kk + 64 * (3 - j);
                }
                float p9 = 0.f;
                if (mt == 0 && h == 0) {
                    // slot 0 (lane group 0 only): the rows are Re S_p[0] and Re S_p[32], the twiddle above was 1.
                    //   a = DFT8 of the real sequence Re S_p[0]  =  (G_j + conj G_(8-j)) / 2  ->  bins 0, 64, .., 256
                    //   hh = DFT8 of W16^p Re S_p[32]                                           ->  bins 32, 96, 160, 224
                    sf32x2 v[8], hh[8];
                    v[0] = sf32x2{acc[0][1], 0.f};
#pragma unroll
                    for (int p = 1; p < 8; ++p) v[p] = TW16[p] * sf32x2{acc[p][1], acc[p][1]};
                    dft8(v, hh);
                    float pa[5];
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const sf32x2 gj = yv[j], gc = yv[(8 - j) & 7];
                        const float ar = 0.5f * (gj.x + gc.x), ai = 0.5f * (gj.y - gc.y);
                        pa[j] = ar * ar + ai * ai;
                    }
                    const float ph[4] = {hh[0].x * hh[0].x + hh[0].y * hh[0].y, hh[1].x * hh[1].x + hh[1].y * hh[1].y,
                                         hh[2].x * hh[2].x + hh[2].y * hh[2].y, hh[3].x * hh[3].x + hh[3].y * hh[3].y};
                    // lane group 0: outputs 0..4 <- pa -> bins 64 j; outputs 5..7 and the ninth <- ph -> bins 32 + 64 j
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        pw[j] = __uint_as_float((__float_as_uint(pa[j]) & g_is0) | (__float_as_uint(pw[j]) & ~g_is0));
                        kb[j] = g == 0 ? 64 * j : kb[j];
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        pw[5 + j] = __uint_as_float((__float_as_uint(ph[j]) & g_is0) | (__float_as_uint(pw[5 + j]) & ~g_is0));
                        kb[5 + j] = g == 0 ? 32 + 64 * j : kb[5 + j];
                    }
                    p9 = ph[3];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float sv = __builtin_fmaf(pw[j], c2, eps);   // spec.py:173
                    if (!PLAIN) sv = spec_format(sv, fmt);
                    stage_row[kb[j]] = sv;
                }
                if (mt == 0 && h == 0) {
                    // bin 224 of lane group 0; the other groups rewrite their own bin 64 - kk + 192 (same value)
                    float sv = __builtin_fmaf(p9, c2, eps);
                    if (!PLAIN) sv = spec_format(sv, fmt);
                    const float other = __builtin_fmaf(pw[4], c2, eps);
                    stage_row[g == 0 ? 224 : kb[4]] = g == 0 ? sv : (PLAIN ? other : spec_format(other, fmt));
                }
            }
        }
        MSTAMP(3);
        __builtin_amdgcn_wave_barrier();
        // ---------------- the staged 16 x 257 tile leaves as one contiguous run ----------------
        const long out0 = (b * N + f0) * K;
        if (nvalid == 16 && (out0 & 3) == 0) {
            sf32x4* y4 = reinterpret_cast<sf32x4*>(y + out0);
            const sf32x4* s4 = reinterpret_cast<const sf32x4*>(stage);
#pragma unroll 4
            for (int it = 0; it < 16; ++it) y4[lane + 64 * it] = s4[lane + 64 * it];   // (4 in flight: the prefetch
            if (lane < 16 * K / 4 - 1024) y4[lane + 1024] = s4[lane + 1024];           //  registers stay resident)
        } else {
            const int total = nvalid * K;
            for (int t = lane; t < total; t += 64) y[out0 + t] = stage[t];
        }
        __builtin_amdgcn_wave_barrier();
        MSTAMP(4);
    }
}

static int stft512_mfma_launch(const float* x, long B, long Tlen, long N, int L, int P, int left, int mode,
                               const float* w, float eps, int fmt, float* y, hipStream_t st)
{
    using namespace sm;
    const float* tab = stft_mfma_tables();
    if (!tab) return fail(DSA_ERR_LAUNCH, "stft512_mfma: cannot set up the constant tables%s");
    const int lds_bytes = LDS_WORDS * 4;
    static std::once_flag once;
    static bool attr_ok = true;
    std::call_once(once, [&] {
        attr_ok = hipFuncSetAttribute((const void*)stft512_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      lds_bytes) == hipSuccess &&
                  hipFuncSetAttribute((const void*)stft512_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      lds_bytes) == hipSuccess;
    });
    if (!attr_ok) return fail(DSA_ERR_LAUNCH, "stft512_mfma: cannot reserve %s of LDS", "150 KB");
    const int tiles_per_utt = (int)((N + 15) / 16);
    const long total_tiles = B * tiles_per_utt;
    long blocks = (total_tiles + WAVES - 1) / WAVES;
    long grid = blocks < 256 ? blocks : 256;  // one persistent workgroup per CU
    if (fmt == DSA_SPEC_POWER)
        hipLaunchKernelGGL((stft512_mfma_kernel<true>), dim3((unsigned)grid), dim3(WAVES * 64), lds_bytes, st, x, Tlen, N,
                           L, P, left, mode, w, tab, eps, fmt, y, total_tiles, tiles_per_utt);
    else
        hipLaunchKernelGGL((stft512_mfma_kernel<false>), dim3((unsigned)grid), dim3(WAVES * 64), lds_bytes, st, x, Tlen, N,
                           L, P, left, mode, w, tab, eps, fmt, y, total_tiles, tiles_per_utt);
    return check_launch("stft512_mfma_fwd");
}

}  // namespace dsa
