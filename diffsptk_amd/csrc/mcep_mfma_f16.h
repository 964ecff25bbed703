// Split-precision variant of the tuned mel-cepstral forward (included by mcep_mfma.hip).
//
// fp32 MFMA runs at the fp32 VECTOR rate and shares the vector datapath, so in
// mcep_mfma_fwd_kernel_v2 the two matrix chains of a Newton step (d = mc D, rt = e E) cost as much
// as the whole 25 x 25 solve.  binary16 MFMA is a separate unit, 16x faster per flop.  Here each
// float32 operand is split into two binary16 pieces, x = hi + lo (|x - hi - lo| <= 2^-22 |x|), and a
// product is three MFMAs accumulated in float32:  a b ~= ah bh + ah bl + al bh  (the dropped al bl
// term is 2^-22 |a b|) -- float32-grade accuracy (tools/proto_f16split.py: error of the converged
// mel-cepstrum against the float64 golden is unchanged), at 3/16 of the fp32 MFMA cost.
//   * the constant operands (D^T, E^T images) are split once per workgroup into LDS, scaled by
//     powers of two so that their lo pieces stay normal binary16 numbers;
//   * mc is scaled by 2^10 (|mc| < 64 for any float32 power spectrum), e = exp2(t) by a per-frame
//     power of two 2^(15 - ceil(max t)), so the largest e is in (2^14, 2^15] (binary16 max 65504);
//     the scalings are exact and undone with v_ldexp_f32 / inside an FMA;
//   * the k-slots of v_mfma_f32_16x16x32_f16 pair lane group g, element i of A with the same of B,
//     so TWO 16-bin C/D tiles of the first chain are, converted and packed, ONE B operand of the
//     second chain: as in the fp32 kernel, e never leaves registers.
// Everything outside the two chains (log2 X, mc0 chain, Nyquist bin, rt[48], the quad-layout
// elimination) is the code of mcep_mfma_fwd_kernel_v2.
#pragma once

#include "pk_math.h"

namespace dsa {

// The waveform side of the fused STFT -> mel-cepstrum launch (stft.py:237-241 feeding mcep.py:189-224 without the (B, N, 257)
// spectrogram's round trip through memory: 320 + 100 bytes per frame instead of 1348 + 1128).
struct StftIn {
    const float* x;        // (B, Tlen) waveforms
    long Tlen, N;          // samples per utterance, frames per utterance
    int P, left;           // frame period; samples of left padding (L / 2 with center, 0 without)
    const float* w;        // (400) window
    const float* twiddle;  // (512, 2) = (cos, -sin)(2 pi m / 512)
    float eps;             // spec.py:173
    float* X_out;          // NULL, or (B N, 257): the power spectrogram as a side product (kept for the backward)
    int pad_mode;          // frame.py:130-137 (DSA_PAD_*): what positions outside the utterance read
    int zmean;             // frame.py:139-140: the frame's mean removed before the window
    float floor_lin;       // spec.py:174-176: < 0 none, else every bin at least (the frame's largest value) x this (10^(dB / 10))
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// explicit global address space: a pointer that went through an opaque asm statement is otherwise loaded from
// with FLAT instructions, which count on the LDS counter too and stall every LDS wait behind an L2 round trip
typedef const __attribute__((address_space(1))) f16x8* gf16x8_ptr;
// Streaming the operand images from L2: a buffer descriptor over the image block (wave-uniform) + a 32-bit lane offset +
// a constant -- buffer_load_dwordx4 takes all three without a vector instruction (as 64-bit per-lane pointers the steps
// beyond the 4 KB immediate cost two carry-chained vector additions per window: 94 vector instructions per tile and step)
typedef int i32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const _Float16* base, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f16x8 gload8(__amdgpu_buffer_rsrc_t rsrc, unsigned lane_bytes, int const_bytes)
{
    const i32x4v r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_bytes, const_bytes, 0);
    return __builtin_bit_cast(f16x8, r);
}
typedef const __attribute__((address_space(1))) float* gf32_ptr;

namespace mh {
using namespace mm;
constexpr float SD = 16.f;      // scale of the D^T image (|-2 log2(e) D| <= ~7)
constexpr float SM = 1024.f;    // scale of mc
constexpr float SE = 65536.f;   // scale of the E^T image (|E| <= ~0.01)
constexpr int SE_LOG2 = 16;
constexpr float SG = 4096.f;    // scale of the (ln 2) G^T image (|G| <= 0.13 for |alpha| <= 0.95)
constexpr float SL = 128.f;     // scale of log2 X (|log2 X| < 150 for any float32 input)
// operand images in global memory (binary16 elements), written by mcep_h_prep_kernel once per launch
constexpr int IMG_D = 16 * 64 * 8, IMG_E = 24 * 64 * 8, IMG_G = 16 * 64 * 8;
constexpr int IMG_DH = 0, IMG_DL = IMG_DH + IMG_D, IMG_EH = IMG_DL + IMG_D, IMG_EL = IMG_EH + IMG_E;
constexpr int IMG_GH = IMG_EL + IMG_E, IMG_GL = IMG_GH + IMG_G, IMG_HALVES = IMG_GL + IMG_G;
constexpr int IMG_BYTES = IMG_HALVES * 2 + 32 * 4;  // + the Nyquist row G[256][0..24] as float32
constexpr int EMAX_LOG2 = 15;   // largest scaled e is <= 2^15
// LDS carve-up (float units; same region sizes as namespace mm: the binary16 hi + lo images take
// exactly the room of one float32 image)
constexpr int DH_OFF = 0;                    // [16 mt][64 lane] f16x8
constexpr int DL_OFF = DH_OFF + 16 * 64 * 4;
constexpr int EH_OFF = DL_OFF + 16 * 64 * 4;  // [3 it][8 j][64 lane] f16x8
constexpr int EL_OFF = EH_OFF + 24 * 64 * 4;
constexpr int H_E48 = EL_OFF + 24 * 64 * 4;  // [16 mt][4 g][4 r] float
constexpr int H_E256 = H_E48 + 256;      // [48] scaled by SE, [48] = unscaled E[256][48]
constexpr int H_D256 = H_E256 + 52;      // [32]
constexpr int H_AV = H_D256 + 32;        // [28]
constexpr int H_NAV = H_AV + 28;         // [28] -alpha_vec, zero-padded
constexpr int H_ZERO = H_NAV + 28;       // [28] zeros
constexpr int H_WAVE = H_ZERO + 28;
constexpr int h_lds_floats(int waves) { return H_WAVE + waves * WAVE_FLOATS; }
// fused launch: three small tables behind the per-wave regions, shared by the workgroup
constexpr int FU_LC = 400;                       // frame length the prologue is built for
constexpr int FU_NR = (FU_LC + 31) / 32;         // sample pairs a lane reads (13)
constexpr int FU_S = 272;                        // per-frame stride of the staged log-spectra (floats; 272 % 64 = 16: the 16 reading
                                                 // lanes (4 frames x 4 groups) x 4 words cover the 64 banks exactly once)
constexpr int FU_T256 = 0;                       // [16 k1][16 j] v2f: W256^(j k1) / 2
constexpr int FU_WTAB = FU_T256 + 512;           // [16 j][13] v2f window pairs
constexpr int FU_TWS = FU_WTAB + 2 * 16 * FU_NR; // [64 lane] (W512^(2l+1), W512^(2l+2)) as v4f
constexpr int FU_FLOATS = FU_TWS + 256;
static_assert(4 * 272 * 2 == WAVE_FLOATS, "a wave's rt / rr windows double as the STFT tile of four frames");
static_assert(FU_TWS % 4 == 0 && H_WAVE % 4 == 0 && WAVE_FLOATS % 4 == 0, "16-byte aligned LDS tables");
constexpr int h_lds_floats_fused(int waves) { return h_lds_floats(waves) + FU_FLOATS; }
}  // namespace mh

// The fused prologue's arithmetic, per stage on the packed (v_pk_*_f32) or the scalar forms of pk_math.h: bit 0 window multiply,
// 1 first 16-point FFT, 2 the W256 twiddles, 3 second 16-point FFT, 4 real-FFT split.
// Rounds 4-5 shipped every stage on the scalar forms (mask 0): with the packed helpers of those rounds ~58-170 of 204 800 frames per
// launch came out wrong.  Round 6 found the cause stand-alone (tools/hazard/repro_min.cpp): a packed float32 form with a
// SET op_sel bit fails next to the 16-bit-input matrix products of the SIMD's other wave; forms without one never do.  The helpers
// no longer contain such a form (DSA_PK_CROSSED = 0: the rotations, the complex product's second instruction and the split's sums are
// one-component instructions), so the packed stages are back: mask 31, 0.5891 -> 0.5811 ms per 204 800 frames, 0 wrong frames in 300
// launches of tools/hazard/hazard_check.py (profiles/r06_fused_prologue_packed_again.txt); bit-identical to mask 0 by construction.
#ifndef DSA_FUSED_PK_MASK
#define DSA_FUSED_PK_MASK 31
#endif
#ifndef DSA_FUSED_DBG
#define DSA_FUSED_DBG 0   // reduction builds: 1 no matrix chains, 2 no solve, 4 no back substitution, 8 self-check of the first FFT (log in `hist`)
#endif
template <bool PK> __device__ __forceinline__ v2f fu_mul(v2f a, v2f b) { if constexpr (PK) return pk_mul(a, b); else return sc_mul(a, b); }
template <bool PK> __device__ __forceinline__ v2f fu_cmul(v2f a, v2f t) { if constexpr (PK) return pk_cmul(a, t); else return sc_cmul(a, t); }
template <bool PK, bool ZT> __device__ __forceinline__ void fu_fft16(v2f (&v)[16]) { if constexpr (PK) pk_fft16<ZT>(v); else sc_fft16<ZT>(v); }

__device__ __forceinline__ f32x4 mfma_h(f16x8 a, f16x8 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// x = hi + lo in binary16 (round to nearest): two values at a time
__device__ __forceinline__ void split2(float x0, float x1, f16x2& hi, f16x2& lo)
{
    hi = __builtin_convertvector(f32x2v{x0, x1}, f16x2);
    // lo = binary16(x - float(hi)), the difference exact in float32: ONE v_fma_mixlo_f16 / v_fma_mixhi_f16 per value (binary16
    // operand taken from its half register, float32 multiply-add, result rounded into the low / high half of the destination).
    // Round 2 used v_fma_mix_f32 + a packed conversion (three instructions per pair), the compiler's own form is five.
    const unsigned hb = __builtin_bit_cast(unsigned, hi);
    // (one statement: the compiler does not see a half-register write inside inline assembly, and gfx950 wants a wait state
    // between such a write and the next vector instruction touching the register -- the second half's read-modify-write)
    unsigned lb;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\ts_nop 0\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0"
        : "=&v"(lb) : "v"(hb), "v"(x0), "v"(x1));
    lo = __builtin_bit_cast(f16x2, lb);
}
// four values at a time: the two pairs' half-register writes interleaved, so that the wait state between the low-half write and
// the high half's read-modify-write of one destination is the other pair's instruction (3 of 4 s_nop fewer per four values)
__device__ __forceinline__ void split4(float x0, float x1, float x2, float x3, f16x2& hiA, f16x2& hiB, f16x2& loA, f16x2& loB)
{
    hiA = __builtin_convertvector(f32x2v{x0, x1}, f16x2);
    hiB = __builtin_convertvector(f32x2v{x2, x3}, f16x2);
    const unsigned ha = __builtin_bit_cast(unsigned, hiA), hb = __builtin_bit_cast(unsigned, hiB);
    unsigned la, lb;
    asm("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %3, -1.0, %6 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %3, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0"
        : "=&v"(la), "=&v"(lb) : "v"(ha), "v"(hb), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    loA = __builtin_bit_cast(f16x2, la);
    loB = __builtin_bit_cast(f16x2, lb);
}
// c * s + b on both halves of a register pair: v_pk_fma_f32 (two flops per lane and issue slot)
#ifdef DSA_MCEP_NOPK
__device__ __forceinline__ f32x2v fma2(f32x2v c, float s, f32x2v b)
{
    float r0 = __builtin_fmaf(c[0], s, b[0]), r1 = __builtin_fmaf(c[1], s, b[1]);
    asm volatile("" : "+v"(r0));
    return f32x2v{r0, r1};
}
#else
__device__ __forceinline__ f32x2v fma2(f32x2v c, float s, f32x2v b) { return c * f32x2v{s, s} + b; }
#endif
__device__ __forceinline__ f32x2v lo2(f32x4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f32x2v hi2(f32x4 v) { return __builtin_shufflevector(v, v, 2, 3); }

__device__ __forceinline__ void split1(float x, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// Operand images of the three constant matrices, split into binary16 hi / lo, in the exact
// per-lane order the MFMAs consume them (one 16-byte element per lane and k-step).  One tiny launch
// ahead of the main kernel; the workgroups of the main kernel then copy D / E images straight into
// LDS and stream the G image (used once per tile) from L2.
__global__ __launch_bounds__(256) void mcep_h_prep_kernel(const float* __restrict__ G, const float* __restrict__ D,
                                                          const float* __restrict__ E, _Float16* __restrict__ img)
{
    using namespace mh;
    constexpr float kNeg2Log2e = -2.885390081777926815f, kLn2 = 0.693147180559945309f;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int i = idx & 7, l = (idx >> 3) & 63;
    if (idx < IMG_D) {
        const int mt = idx >> 9;
        const int k = 8 * (l >> 4) + i;  // k-slot (g, i) <-> coefficient 8 g + i
        const float v = k < M1 ? (kNeg2Log2e * SD) * D[k * K + mt * 16 + (l & 15)] : 0.f;
        split1(v, img[IMG_DH + idx], img[IMG_DL + idx]);
    } else if (idx < IMG_D + IMG_E) {
        const int e = idx - IMG_D;
        const int j = (e >> 9) & 7, it = e >> 12;
        // k-slot (g, i = 4 t + r) <-> bin 32 j + 16 t + 4 g + r: C/D register r of tile 2 j + t
        const int bin = 32 * j + 16 * (i >> 2) + 4 * (l >> 4) + (i & 3);
        const float v = SE * E[bin * M2 + it * 16 + (l & 15)];
        split1(v, img[IMG_EH + e], img[IMG_EL + e]);
    } else if (idx < IMG_D + IMG_E + IMG_G) {
        const int e = idx - IMG_D - IMG_E;
        const int j = (e >> 9) & 7, it = e >> 12;
        const int bin = 32 * j + 16 * (i >> 2) + 4 * (l >> 4) + (i & 3);
        const int coef = it * 16 + (l & 15);
        const float v = coef < M1 ? (kLn2 * SG) * G[bin * M1 + coef] : 0.f;
        split1(v, img[IMG_GH + e], img[IMG_GL + e]);
    }
    if (idx < 32) reinterpret_cast<float*>(img + IMG_HALVES)[idx] = idx < M1 ? G[H * M1 + idx] : 0.f;
}

// A lane's eight consecutive coefficients mc[8 g .. 8 g + 7] of its frame (only mc[24] in lane group 3) as two 16-byte stores
// (rows are 100 bytes apart: 4-byte aligned) instead of eight 4-byte ones: a quarter of the store instructions of the history
// the backward needs (11 rows of 25 floats per frame: 225 MB per 204 800 frames).
__device__ __forceinline__ void store_mc_row(float* row, int g, const float (&mcv)[8])
{
    if (g < 3) {
        *reinterpret_cast<f32x4_u4*>(row + 8 * g) = f32x4_u4{mcv[0], mcv[1], mcv[2], mcv[3]};
        *reinterpret_cast<f32x4_u4*>(row + 8 * g + 4) = f32x4_u4{mcv[4], mcv[5], mcv[6], mcv[7]};
    } else {
        row[24] = mcv[0];
    }
}

template <int WAVES, bool FUSED = false, bool HIST_RT = false, bool PADM = false>   // HIST_RT: also keep every step's rt row (its own
                                                                 // instantiation: the plain kernels' code and register allocation are
                                                                 // untouched); PADM: the options instantiation -- reflect / replicate /
                                                                 // circular padding, zmean, relative floor (likewise)
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) DSA_PK_TARGET void mcep_mfma_fwd_kernel_h(
    const float* __restrict__ X, long F, int n_iter, const float* __restrict__ G,
    const float* __restrict__ D, const float* __restrict__ E, const float* __restrict__ av,
    float* __restrict__ mc_out, float* __restrict__ hist, long ntiles16, long tiles_shared,
    unsigned int* __restrict__ queue, const _Float16* __restrict__ img, StftIn sti, float* __restrict__ hist_rt, int tail_wgs)
{
    // hist_rt (round 5; NULL or (n_iter, F, 49)): every step's rt = e E row, kept for the backward, which then skips its own second
    // forward chain (DSA_ALGO_HIST_HAS_RT; 196 more bytes per frame and step)
    using namespace mh;
    constexpr float kNeg2Log2e = -2.885390081777926815f, kLn2 = 0.693147180559945309f;
    constexpr float kInvSDM = 1.f / (SD * SM);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;

    // ---------------- operand images (binary16 hi / lo): copied from the prepared global images ----------------
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(img);
        f32x4* dst = reinterpret_cast<f32x4*>(lds + DH_OFF);
        for (int idx = tid; idx < (2 * IMG_D + 2 * IMG_E) / 8; idx += WAVES * 64) dst[idx] = src[idx];
    }
    {
        const int t2 = tid & 255;
        int r = t2 & 3, gg = (t2 >> 2) & 3, mt = t2 >> 4;
        lds[H_E48 + t2] = E[(mt * 16 + gg * 4 + r) * M2 + 48];
    }
    if (tid < 48) lds[H_E256 + tid] = SE * E[H * M2 + tid];
    if (tid == 48) lds[H_E256 + 48] = E[H * M2 + 48];
    if (tid < 32) lds[H_D256 + tid] = tid < M1 ? kNeg2Log2e * D[tid * K + H] : 0.f;
    if (tid < 28) {
        const float a_ = tid < M1 ? av[tid] : 0.f;
        lds[H_AV + tid] = a_;
        lds[H_NAV + tid] = -a_;
        lds[H_ZERO + tid] = 0.f;
    }
    if (FUSED) {
        // tables of the STFT prologue (csrc/stft_pk.h keeps the same three per wave / in registers)
        float* fu = lds + h_lds_floats(WAVES);
        v2f* t256w = reinterpret_cast<v2f*>(fu + FU_T256);
        v2f* wtabw = reinterpret_cast<v2f*>(fu + FU_WTAB);
        v2f* twsw = reinterpret_cast<v2f*>(fu + FU_TWS);
        if (tid < 256) {
            const int m = 2 * (tid & 15) * (tid >> 4);   // W256^(j k1) = W512^(2 j k1), halved: the 1/2 of the real-FFT split (exact)
            t256w[tid] = v2f{0.5f * sti.twiddle[2 * m], 0.5f * sti.twiddle[2 * m + 1]};
        } else if (tid < 256 + 16 * FU_NR) {
            const int i = tid - 256;
            const int l = 2 * (i / FU_NR) + 32 * (i % FU_NR);
            wtabw[i] = v2f{l < FU_LC ? sti.w[l] : 0.f, l + 1 < FU_LC ? sti.w[l + 1] : 0.f};
        }
        if (tid < 128) {   // split twiddles W512^k of lane l's bins k = 2 l + 1, 2 l + 2
            const int k = 2 * (tid >> 1) + 1 + (tid & 1);
            twsw[tid] = v2f{sti.twiddle[2 * k], sti.twiddle[2 * k + 1]};
        }
    }
    __syncthreads();  // the only workgroup barrier

    float* rt_lds = lds + H_WAVE + wave * WAVE_FLOATS + n * RS;
    float* rr_lds = rt_lds + 16 * RS;
    // Two opaque lane bases for the operand images: ds_read immediates reach 64 KB, the images span
    // 80 KB, and left to itself hipcc materialises (and then spills) one address register per far read
    int lane_a = lane, lane_b = lane + EL_OFF / 4;
    asm volatile("" : "+v"(lane_a), "+v"(lane_b));
    const f16x8* DH = reinterpret_cast<const f16x8*>(lds + DH_OFF) + lane_a;
    const f16x8* DL = reinterpret_cast<const f16x8*>(lds + DL_OFF) + lane_a;
    const f16x8* EH = reinterpret_cast<const f16x8*>(lds + EH_OFF) + lane_a;
    const f16x8* EL = reinterpret_cast<const f16x8*>(lds) + lane_b;
    const f32x4* E484 = reinterpret_cast<const f32x4*>(lds + H_E48);
    const long wave_id = (long)blockIdx.x * WAVES + wave;
    const long wave_stride = (long)gridDim.x * WAVES;
    const int nq = lane >> 2, gs = lane & 3;
    const GroupMask gq = make_group_mask(gs);
    const unsigned g_lt3 = g < 3 ? 0xffffffffu : 0u, g_eq0 = g == 0 ? 0xffffffffu : 0u;
    float* rt_q = lds + H_WAVE + wave * WAVE_FLOATS + nq * RS;
    float* rr_q = rt_q + 16 * RS;
#ifdef DSA_MCEP_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0) g_mcep_stamps[12] = __builtin_readcyclecounter();
#endif

#ifdef DSA_MCEP_TIMING
    int tcount = 0;
#define DSA_STAMP_T(i)                                                                   \
    do {                                                                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0 && tcount == 2) g_mcep_stamps[i] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define DSA_STAMP_T(i)
#endif
#ifdef DSA_MCEP_ABL_ONEWAVE   // timing experiment: one working wave per SIMD (the critical path of a wave on its own)
    const bool abl_idle = wave >= WAVES / 2;
#else
    const bool abl_idle = false;
#endif
#ifdef DSA_MCEP_TIMING
    const unsigned long long slot_t0 = wall_clock64();
    int slot_tiles = 0;
#endif
    for (long tile = abl_idle ? ntiles16 : wave_id; tile < ntiles16;) {
        DSA_STAMP_T(16);
#ifdef DSA_MCEP_TIMING
        if (blockIdx.x == 0 && threadIdx.x == 0 && tcount < 12) { g_mcep_stamps[32 + tcount] = __builtin_readcyclecounter(); g_mcep_stamps[48 + tcount] = (unsigned long long)tile; }
#endif
        const long f_raw = tile * 16 + n;
        const bool f_ok = f_raw < F;
        const long f = f_ok ? f_raw : F - 1;

        f32x4 logx[16];
        float logx256;
        if (!FUSED) {
            const float* xf = X + f * K;
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
                const float* p = xf + mt * 16 + 4 * g;
                logx[mt] = f32x4{__log2f(p[0]), __log2f(p[1]), __log2f(p[2]), __log2f(p[3])};  // mcep.py:203 (base 2)
            }
            logx256 = __log2f(xf[H]);
        } else {
            // ---------------- fused: the tile's 16 power spectra straight from the waveform (stft.py:237-241) ----------------
            // Four passes of four frames, each the pass of stft512_fwd_pk_kernel (csrc/stft_pk.h: 16 lanes per frame, window,
            // radix-16 x 16 complex FFT of the 256 sample pairs through the wave's LDS region, real-FFT split with |.|^2 + eps on
            // float32 with the packed kernel's roundings, so the power values are bit-identical to the stand-alone kernel's; as SCALAR vector
            // instructions: packed ones next to the other wave's 4 x 4 x 1 matrix products deliver stale results, see pk_math.h), then
            // log2 in the split's layout (every lane busy), staged as 4 x 257 floats and picked up by the 16 lanes (n, g) whose
            // frames these are, in the matrix-core layout the chains below consume.  The samples come straight from memory /
            // L2 per frame (neighbouring frames overlap there; a lane reads 13 pairs), requested one pass ahead.
            constexpr int PKM = DSA_FUSED_PK_MASK;
            const int j = lane & 15, fl = lane >> 4;
            float* wreg = lds + H_WAVE + wave * WAVE_FLOATS;
            v2f* zbuf = reinterpret_cast<v2f*>(wreg);
            v2f* zf = zbuf + fl * 272;
            const float* fu = lds + h_lds_floats(WAVES);
            v2f raw[FU_NR];
            auto fetch = [&](int p) __attribute__((always_inline)) {
                // 32-bit arithmetic from an opaque copy of the lane's frame slot, formed here (F, Tlen < 2^31, host-checked): as 64-bit
                // values derived from `fl` these were hoisted out of the pass loop, spilled across the Newton iteration, and each
                // reload was a `s_waitcnt vmcnt(0)` in front of the fetch it feeds
                int flo = fl;
                asm volatile("" : "+v"(flo));
                int fr = (int)(tile * 16) + 4 * p + flo;
                fr = fr < (int)F ? fr : (int)F - 1;
                const unsigned ub = (unsigned)fr / (unsigned)sti.N;
                const int nf = fr - (int)(ub * (unsigned)sti.N);
                const int start = nf * sti.P - sti.left;
                const float* xb = sti.x + (long)ub * sti.Tlen;
                const bool interior = start >= 0 && start + 32 * FU_NR <= (int)sti.Tlen;
                if (__builtin_amdgcn_ballot_w64(!interior) == 0) {
                    const v2f_u4* src = reinterpret_cast<const v2f_u4*>(xb + start + 2 * j);
#pragma unroll
                    for (int m1 = 0; m1 < FU_NR; ++m1) raw[m1] = src[16 * m1];
                } else if (!PADM || sti.pad_mode == (int)DSA_PAD_CONSTANT) {
                           // frames that reach over an end of their utterance: zeros outside (F.pad, constant mode).  Branch-free:
                           // clamped addresses, the out-of-range values selected away (32-bit: Tlen < 2^31, host-checked)
                    const int s00 = start + 2 * j, tl = (int)sti.Tlen;
#pragma unroll
                    for (int m1 = 0; m1 < FU_NR; ++m1) {
                        const int s0 = s00 + 32 * m1, s1 = s0 + 1;
                        const bool ok0 = (unsigned)s0 < (unsigned)tl, ok1 = (unsigned)s1 < (unsigned)tl;
                        const float a0 = xb[ok0 ? s0 : 0], a1 = xb[ok1 ? s1 : 0];
                        raw[m1] = v2f{ok0 ? a0 : 0.f, ok1 ? a1 : 0.f};
                    }
                } else {   // reflect / replicate / circular (round 6): the position outside reads the sample F.pad would have put there
                    // (32-bit pad_src_index: Tlen < 2^31 and |position| < Tlen + 512, host-checked; positions past the frame's 400
                    //  samples are selected away below, their addresses stay inside the row)
                    const int s00 = start + 2 * j, tl = (int)sti.Tlen, md = sti.pad_mode;
                    auto src = [&](int i) __attribute__((always_inline)) -> int {
                        if ((unsigned)i < (unsigned)tl) return i;
                        if (md == (int)DSA_PAD_REPLICATE) return i < 0 ? 0 : tl - 1;
                        if (md == (int)DSA_PAD_CIRCULAR) {
                            int q = i % tl;
                            return q < 0 ? q + tl : q;
                        }
                        if (tl == 1) return 0;                       // reflect (frame.py:134-137 through F.pad)
                        const int period = 2 * (tl - 1);
                        int q = i % period;
                        q = q < 0 ? q + period : q;
                        return q < tl ? q : period - q;
                    };
#pragma unroll
                    for (int m1 = 0; m1 < FU_NR; ++m1) {
                        const int s0 = s00 + 32 * m1;
                        raw[m1] = v2f{xb[src(s0)], xb[src(s0 + 1)]};
                    }
                }
            };
            fetch(0);
#pragma unroll 1
            for (int p = 0; p < 4; ++p) {
                // (opaque per pass: the table reads below are loop invariant, and hoisted out of the pass / tile loops their 62
                // registers would live -- spilled -- across the whole Newton iteration)
                int jo = j * FU_NR, j1 = j, l4 = lane;
                asm volatile("" : "+v"(jo), "+v"(j1), "+v"(l4));
                const v2f* wtab = reinterpret_cast<const v2f*>(fu + FU_WTAB) + jo;
                const v2f* t256 = reinterpret_cast<const v2f*>(fu + FU_T256) + j1;
                v2f v[16];
                if (PADM && sti.zmean) {   // frame.py:139-140, with the packed kernel's summation (pk_math.h: pk_zero_mean)
                    bool in0[FU_NR], in1[FU_NR];
#pragma unroll
                    for (int m1 = 0; m1 < FU_NR; ++m1) {
                        in0[m1] = 32 * m1 + 30 < FU_LC || 32 * m1 + 2 * j < FU_LC;
                        in1[m1] = 32 * m1 + 31 < FU_LC || 32 * m1 + 1 + 2 * j < FU_LC;
                        raw[m1] = v2f{in0[m1] ? raw[m1].x : 0.f, in1[m1] ? raw[m1].y : 0.f};
                    }
                    pk_zero_mean<FU_NR>(raw, in0, in1, FU_LC);
                }
#pragma unroll
                for (int m1 = 0; m1 < FU_NR; ++m1) {
                    // element (m1, e) belongs to the frame iff 32 m1 + e + 2 j < L; selected, never multiplied: zero padding is
                    // exact and non-finite neighbours stay out of frames that do not contain them
                    const bool in0 = 32 * m1 + 30 < FU_LC || 32 * m1 + 2 * j < FU_LC;
                    const bool in1 = 32 * m1 + 31 < FU_LC || 32 * m1 + 1 + 2 * j < FU_LC;
                    v[m1] = fu_mul<(PKM & 1) != 0>(v2f{in0 ? raw[m1].x : 0.f, in1 ? raw[m1].y : 0.f}, wtab[m1]);
                }
#pragma unroll
                for (int m1 = FU_NR; m1 < 16; ++m1) v[m1] = v2f{0.f, 0.f};
                if (p < 3) fetch(p + 1);
#if DSA_FUSED_DBG & 8
                {   // reduction build: the packed transform against its scalar twin on the same inputs; mismatches logged with a re-run
                    v2f vin[16], vs[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) { vin[i] = v[i]; vs[i] = v[i]; }
                    pk_fft16<true>(v);
                    sc_fft16<true>(vs);
                    int nbad = 0, first = -1;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const bool b = __builtin_bit_cast(unsigned, v[i].x) != __builtin_bit_cast(unsigned, vs[i].x) ||
                                       __builtin_bit_cast(unsigned, v[i].y) != __builtin_bit_cast(unsigned, vs[i].y);
                        if (b && first < 0) first = i;
                        nbad += b ? 1 : 0;
                    }
                    if (nbad) {
                        v2f vr[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) vr[i] = vin[i];
                        pk_fft16<true>(vr);
                        unsigned* lg = reinterpret_cast<unsigned*>(hist);
                        const unsigned slot = atomicAdd(lg, 1u);
                        if (slot < 4000) {
                            unsigned hwid, xcc;
                            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                            unsigned* r = lg + 16 + 16 * slot;
                            float pkx = 0, pky = 0, scx = 0, scy = 0, rx = 0, ry = 0;
#pragma unroll
                            for (int i = 0; i < 16; ++i)
                                if (i == first) { pkx = v[i].x; pky = v[i].y; scx = vs[i].x; scy = vs[i].y; rx = vr[i].x; ry = vr[i].y; }
                            r[0] = (unsigned)tile; r[1] = p; r[2] = lane; r[3] = first; r[4] = nbad;
                            r[5] = __builtin_bit_cast(unsigned, pkx); r[6] = __builtin_bit_cast(unsigned, pky);
                            r[7] = __builtin_bit_cast(unsigned, scx); r[8] = __builtin_bit_cast(unsigned, scy);
                            r[9] = __builtin_bit_cast(unsigned, rx); r[10] = __builtin_bit_cast(unsigned, ry);
                            r[11] = hwid; r[12] = xcc; r[13] = wave; r[14] = blockIdx.x; r[15] = 0xabcd0000u;
                        }
#pragma unroll
                        for (int i = 0; i < 16; ++i) v[i] = vs[i];   // carry on with the right values
                    }
                }
#else
                fu_fft16<(PKM & 2) != 0, true>(v);
#endif
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) {
                    zf[k1 * 17 + j] = fu_cmul<(PKM & 4) != 0>(v[FFT16_OUT(k1)], t256[k1 * 16]);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = zf[j * 17 + i];
                __builtin_amdgcn_wave_barrier();
                fu_fft16<(PKM & 8) != 0, false>(v);
#pragma unroll
                for (int k0 = 0; k0 < 16; ++k0) {
                    zf[j + 16 * k0 + (k0 < 8 ? 1 : 2)] = v[FFT16_OUT(k0)];   // Z[k] at k + 1 (k <= 128) / k + 2: 16-byte aligned pair reads
                    if (k0 == 8 && j == 0) zf[129] = v[FFT16_OUT(k0)];
                }
                __builtin_amdgcn_wave_barrier();
                v2f pa[4][2], pb[4][2], z0[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v2f* z = zbuf + q * 272;
                    const v4f a2 = *reinterpret_cast<const v4f*>(z + 2 * lane + 2);     // Z[2l+1], Z[2l+2]
                    const v4f b2 = *reinterpret_cast<const v4f*>(z + 256 - 2 * lane);   // Z[254-2l], Z[255-2l]
                    pa[q][0] = v2f{a2.x, a2.y};
                    pa[q][1] = v2f{a2.z, a2.w};
                    pb[q][1] = v2f{b2.x, b2.y};
                    pb[q][0] = v2f{b2.z, b2.w};
                    z0[q] = z[1];
                }
                const v4f tw2 = reinterpret_cast<const v4f*>(fu + FU_TWS)[l4];
                const v2f twA = v2f{tw2.x, tw2.y}, twB = v2f{tw2.z, tw2.w};
                __builtin_amdgcn_wave_barrier();   // every pair is read: the region now takes the staged log-spectra
                const long fr0 = tile * 16 + 4 * p;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // bins 0 and 256 from Z[0] alone: X[0] = 2 (re + im), X[256] = 2 (re - im) (Z arrives halved)
                    v2f se, sp[2];
                    if constexpr ((PKM & 16) != 0) {   // the packed form of csrc/stft_pk.h, instruction for instruction
                        const v2f eps2 = v2f{sti.eps, sti.eps};
                        v2f Ee;
                        Ee = pk_lo_pm_hi(z0[q], z0[q]);
                        const v2f E4 = pk_mul_s(Ee, v2f{4.f, 4.f});
                        se = pk_fma_sc(E4, Ee, eps2);
#pragma unroll
                        for (int part = 0; part < 2; ++part) {
                            const v2f S = pk_add_conj(pa[q][part], pb[q][part]);
                            const v2f Dd = pk_sub_conj(pa[q][part], pb[q][part]);
                            const v2f Pp = pk_cmul(Dd, part == 0 ? twA : twB);
                            v2f R, I;
                            R = pk_lo_pm_hi(S, Pp);
                            I = pk_hi_mp_lo(S, Pp);
                            sp[part] = pk_fma(I, I, pk_fma_sc(R, R, eps2));
                        }
                    } else {
                    const v2f Ee = v2f{sc_add1(z0[q].x, z0[q].y), sc_sub1(z0[q].x, z0[q].y)};
                    const v2f E4 = v2f{sc_mul1s(Ee.x, 4.f), sc_mul1s(Ee.y, 4.f)};
                    se = v2f{sc_fma1sc(E4.x, Ee.x, sti.eps), sc_fma1sc(E4.y, Ee.y, sti.eps)};
#pragma unroll
                    for (int part = 0; part < 2; ++part) {
                        // S = a + conj(b), Dd = a - conj(b), Pp = W^k Dd; X[k] = (S.re + Pp.im, S.im - Pp.re),
                        // X[256-k] = (S.re - Pp.im, -S.im - Pp.re); |.|^2 + eps (spec.py:173) -- roundings as stft_pk.h
                        const v2f S = sc_add_conj(pa[q][part], pb[q][part]);
                        const v2f Dd = sc_sub_conj(pa[q][part], pb[q][part]);
                        const v2f Pp = sc_cmul(Dd, part == 0 ? twA : twB);
                        const v2f R = v2f{sc_add1(S.x, Pp.y), sc_sub1(S.x, Pp.y)};
                        const v2f I = v2f{sc_sub1(S.y, Pp.x), sc_nsub1(S.y, Pp.x)};
                        const v2f s0_ = v2f{sc_fma1sc(R.x, R.x, sti.eps), sc_fma1sc(R.y, R.y, sti.eps)};
                        sp[part] = v2f{sc_fma1(I.x, I.x, s0_.x), sc_fma1(I.y, I.y, s0_.y)};
                    }
                    }
                    // sp[0] = (bin 2l+1, bin 255-2l), sp[1] = (bin 2l+2, bin 254-2l)
                    if (PADM && sti.floor_lin >= 0.f) {   // spec.py:174-176, as csrc/stft_pk.h applies it
                        float m = sp[0].x > sp[0].y ? sp[0].x : sp[0].y;
                        m = sp[1].x > m ? sp[1].x : m;
                        m = sp[1].y > m ? sp[1].y : m;
                        m = se.x > m ? se.x : m;
                        m = se.y > m ? se.y : m;
                        const float flv = wave64_max(m) * sti.floor_lin;
                        sp[0] = v2f{sp[0].x > flv ? sp[0].x : flv, sp[0].y > flv ? sp[0].y : flv};
                        sp[1] = v2f{sp[1].x > flv ? sp[1].x : flv, sp[1].y > flv ? sp[1].y : flv};
                        se = v2f{se.x > flv ? se.x : flv, se.y > flv ? se.y : flv};
                    }
                    if (sti.X_out && fr0 + q < F) {   // the spectrogram as a side product (a gradient will need it)
                        float* yr = sti.X_out + (fr0 + q) * K;   // (uniform; the lane part below from the per-pass opaque copy l4)
                        *reinterpret_cast<v2f_u4*>(yr + 2 * l4 + 1) = v2f{sp[0].x, sp[1].x};
                        *reinterpret_cast<v2f_u4*>(yr + 254 - 2 * l4) = v2f{sp[1].y, sp[0].y};
                        if (lane == 0) {
                            yr[0] = se.x;
                            yr[256] = se.y;
                        }
                    }
                    float* st = wreg + q * FU_S;
                    *reinterpret_cast<v2f_u4*>(st + 2 * lane + 1) = v2f{__log2f(sp[0].x), __log2f(sp[1].x)};     // mcep.py:203 (base 2)
                    *reinterpret_cast<v2f*>(st + 254 - 2 * lane) = v2f{__log2f(sp[1].y), __log2f(sp[0].y)};
                    if (lane == 0) {
                        st[0] = __log2f(se.x);
                        st[256] = __log2f(se.y);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if ((n >> 2) == p) {
                    const float* st = wreg + (n & 3) * FU_S;
#pragma unroll
                    for (int mt = 0; mt < 16; ++mt) logx[mt] = *reinterpret_cast<const f32x4*>(st + 16 * mt + 4 * g);
                    logx256 = st[256];
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        DSA_STAMP_T(17);

        // ---------------- mc0^T = G^T logx^T  (mcep.py:204-207): split-precision MFMA, the G^T image
        // streamed from global memory (L2-resident, 32 KB, read once per tile) ----------------
        // mcv[i] = mc[8 g + i] of frame n: the B operand slots of the first chain
        float mcv[8];
        {
            // (opaque per tile: hoisted out of the tile loop the 32 operand loads would live in scratch)
            const _Float16* imgt = img;
            asm volatile("" : "+s"(imgt));
            const gf16x8_ptr GH = (gf16x8_ptr)(imgt + IMG_GH) + lane;
            const gf16x8_ptr GL = (gf16x8_ptr)(imgt + IMG_GL) + lane;
            f32x4 accG[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f16x8 lh, ll;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        f16x2 h, l;
                        split2(logx[2 * j + tt][r] * SL, logx[2 * j + tt][r + 1] * SL, h, l);
                        lh[4 * tt + r] = h[0]; lh[4 * tt + r + 1] = h[1];
                        ll[4 * tt + r] = l[0]; ll[4 * tt + r + 1] = l[1];
                    }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const f16x8 ah = GH[(it * 8 + j) * 64], al = GL[(it * 8 + j) * 64];
                    accG[it] = mfma_h(al, lh, accG[it]);
                    accG[it] = mfma_h(ah, ll, accG[it]);
                    accG[it] = mfma_h(ah, lh, accG[it]);
                }
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {  // Nyquist bin: one float32 k-step on k-slot 0
                const int out = it * 16 + n;
                const float gv = out < M1 ? (kLn2 * SG * SL) * ((gf32_ptr)(imgt + IMG_HALVES))[out] : 0.f;
                accG[it] = mfma4(keep_if(g_eq0, gv), keep_if(g_eq0, logx256), accG[it]);
                accG[it] *= 1.f / (SG * SL);
            }
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) rt_lds[it * 16 + 4 * g + r] = accG[it][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) mcv[i] = rt_lds[8 * g + i];  // coefficients >= 25 come out 0
            __builtin_amdgcn_wave_barrier();
        }
        if (!(DSA_FUSED_DBG & 8) && hist && f_ok) store_mc_row(hist + f * M1, g, mcv);

        DSA_STAMP_T(18);
        // Ticket for this wave's next tile, drawn now: the atomic's round trip hides behind the iterations.
        // Tiles [0, tiles_shared) are open to every wave.  When the launch ends in a round that fills at
        // most half of the wave slots, those last tiles sit behind a second counter that only the first
        // wave of each SIMD pair draws from: a SIMD then finishes with ONE wave at full issue rate
        // instead of two waves at half rate each (the tail is one tile long either way).
        long tile_next;
        {
            unsigned int nxt = 0;
            if (lane == 0) nxt = atomicAdd(queue, 1u);
            tile_next = wave_stride + (long)__builtin_amdgcn_readfirstlane((int)nxt);
            if (tile_next >= tiles_shared) {
                tile_next = ntiles16;
                // (tail_wgs > 0, DSA_ALGO_OVERLAPPED_LAUNCHES: the short round on ALL waves of the first tail_wgs workgroups instead --
                //  the other workgroups exit and the next launch, queued on the caller's second stream, takes their CUs.  Measured,
                //  204 800 frames, 200 steps on two streams: 0.5945 -> 0.5685 ms per step.  Two cleaner-looking dealings were built
                //  and measured too, and neither gains anything (profiles/r06_overlapped_launches_ab.txt): ALL tiles dealt statically
                //  by whole workgroups, 0.589; shared-queue tickets per workgroup ROUND of eight tiles -- claimed in LDS by the first
                //  wave to need one, published one step into its tile -- so that a workgroup's waves leave together, 0.594)
                if ((tail_wgs > 0 ? (int)blockIdx.x < tail_wgs : wave < WAVES / 2) && tiles_shared < ntiles16) {
                    if (lane == 0) nxt = atomicAdd(queue + 1, 1u);
                    tile_next = tiles_shared + (long)__builtin_amdgcn_readfirstlane((int)nxt);
                }
            }
        }
        for (int iter = 0; iter < n_iter; ++iter) {
            DSA_STAMPS_DECL;
            DSA_STAMP(0);
#if DSA_FUSED_DBG & 1   // reduction build: no matrix chains; a finite, diagonally dominant system from data at hand
            f32x4 accB[3] = {logx[0], logx[1], logx[2]};
            if (lane == (lane & 15)) accB[0][0] += 65536.f * 4096.f;
            float rt48 = logx256;
            const int back = 0;
#else
            // ------------- first chain: t = log2 X - 2 log2(e) d,  d^T = D^T mc^T  (mcep.py:210-212) -----
            f16x8 bh, bl;
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                f16x2 h, l;
                split2(mcv[i] * SM, mcv[i + 1] * SM, h, l);
                bh[i] = h[0]; bh[i + 1] = h[1];
                bl[i] = l[0]; bl[i + 1] = l[1];
            }
            float d256 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) d256 = __builtin_fmaf(mcv[i], lds[H_D256 + 8 * g + i], d256);
            d256 = rows_sum4(d256);
            const float t256 = logx256 + d256;
            DSA_STAMP(6);
            // Round 3: the two chains as an explicit software pipeline, the first chain run ONCE.
            // A wave issues in order, at most one instruction per ~4.3 cycles whatever its kind (tools/bench_issue.cpp), and a
            // binary16 product occupies the matrix pipe for 16 cycles; a product that accumulates into the result of the
            // previous one waits out its full latency (27 cycles per product measured for two interleaved chains).  Left to the
            // compiler, the products came out in clusters with the wave's own vector work waiting behind them (one wave alone
            // spent 6.8 k cycles in this phase against 2.7 k of matrix time and ~2 k of vector time).  Here every product is
            // followed by a few vector instructions of the PREVIOUS group of tiles (sched_barrier pins the order between the slots,
            // LDS reads included: the operand images are read one body ahead of their products), and products into one
            // accumulator are at least four slots apart.
            // With the elimination in register quadruples that are dead during this phase there is room to KEEP t (64
            // registers): the first chain no longer runs a second time for the shifted exponent (48 products, 16 LDS reads and
            // their issue slots per step); the shift is one packed add per value pair.
#define DSA_SB() __builtin_amdgcn_sched_barrier(0x0004)
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            f32x4 tt[16];
            float tmax = t256;
            {
                // pass A: t = log2 X + (D^T mc^T) / (SD SM) in groups of four tiles, running maximum
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
                        if (pm) c[i] = mfma_h(al[i], bh, zero4);
                        DSA_SB();
                        if (vw) tt[4 * q - 4 + i] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (vw) {
                            const f32x2v ta = fma2(lo2(pc[i]), kInvSDM, lo2(logx[4 * q - 4 + i]));
                            const f32x2v tb = fma2(hi2(pc[i]), kInvSDM, hi2(logx[4 * q - 4 + i]));
                            tt[4 * q - 4 + i] = f32x4{ta[0], ta[1], tb[0], tb[1]};
                        }
                        DSA_SB();
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (pm) c[i] = mfma_h(ah[i], bl, c[i]);
                        if (q < 3) al[i] = DL[(4 * q + 4 + i) * 64];
                        DSA_SB();
                        if (vw) {
                            const f32x4 v = tt[4 * q - 4 + i];
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
            DSA_STAMP(7);
            // operands of the first body of pass B, read behind the reduction of the maximum
            f16x8 eah[3], eal[3];
#pragma unroll
            for (int it = 0; it < 3; ++it) { eah[it] = EH[(it * 8) * 64]; eal[it] = EL[(it * 8) * 64]; }
            tmax = rows_max4(tmax);
            const float mi = __builtin_ceilf(tmax);
            const float sh = (float)EMAX_LOG2 - mi;
            const int back = (int)mi - EMAX_LOG2;  // rt = 2^back (scaled sums)
            DSA_STAMP(8);

            // ------------- pass B: e = exp2(t + sh), second chain rt^T += E^T e^T  (mcep.py:212-215).  Body j: the nine products
            // for bins 32 (j - 1) .. (three terms x three output tiles), one per slot, around the vector work of bins 32 j ..
            // (shift, exp2, the rt[48] column, the binary16 split) -------------
            // the Nyquist bin first: rt[16 it + 4 g + r] = E[256][.] e[256] preloads the accumulators of the second chain (body 0 has
            // no products to wait for), its share of rt[48] is added with the reduction at the end
            const float e256 = __builtin_amdgcn_exp2f(t256 + sh);
            f32x4 accB[3];
#pragma unroll
            for (int it = 0; it < 3; ++it) accB[it] = *reinterpret_cast<const f32x4*>(lds + H_E256 + it * 16 + 4 * g) * e256;
            const float rt48n = e256 * lds[H_E256 + 48];
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
                    if (j > 0) {
                        const int it = i % 3, term = i / 3;
                        accB[it] = mfma_h(term == 0 ? eal[it] : eah[it], term == 1 ? el_p : eh_p, accB[it]);
                    }
                };
                f32x2v ta[2], tb[2];
                float e[2][4];
                auto vecA = [&](int t_) __attribute__((always_inline)) {
                    const f32x4 v = tt[(2 * j + t_) & 15];
                    ta[t_] = lo2(v) + f32x2v{sh, sh};
                    tb[t_] = hi2(v) + f32x2v{sh, sh};
                };
                auto vecB = [&](int t_) __attribute__((always_inline)) {
                    e[t_][0] = __builtin_amdgcn_exp2f(ta[t_][0]);  // mcep.py:212
                    e[t_][1] = __builtin_amdgcn_exp2f(ta[t_][1]);
                };
                auto vecC = [&](int t_) __attribute__((always_inline)) {
                    e[t_][2] = __builtin_amdgcn_exp2f(tb[t_][0]);
                    e[t_][3] = __builtin_amdgcn_exp2f(tb[t_][1]);
                };
                auto vecD = [&](int t_) __attribute__((always_inline)) {
                    rt48v = f32x2v{e[t_][0], e[t_][1]} * lo2(c48[t_]) + rt48v;
                    rt48v = f32x2v{e[t_][2], e[t_][3]} * hi2(c48[t_]) + rt48v;
                };
                auto vecE = [&](int t_, int r) __attribute__((always_inline)) {
                    f16x2 h, l;
                    split2(e[t_][r], e[t_][r + 1], h, l);
                    eh[4 * t_ + r] = h[0]; eh[4 * t_ + r + 1] = h[1];
                    el[4 * t_ + r] = l[0]; el[4 * t_ + r + 1] = l[1];
                };
                const bool vw = j < 8;   // body 8 only drains the second chain
                prodE(0); DSA_SB(); if (vw) { vecA(0); vecB(0); } DSA_SB();
                prodE(1); DSA_SB(); if (vw) { vecC(0); vecA(1); } DSA_SB();
                if (j == 8) rt48 = rows_sum4(rt48v[0] + rt48v[1]) + rt48n;   // the drain body has room for the rt[48] reduction
                prodE(2);
                if (j > 0 && j < 8) {
#pragma unroll
                    for (int it = 0; it < 3; ++it) eal[it] = EL[(it * 8 + j) * 64];
                }
                DSA_SB(); if (vw) { vecD(0); vecB(1); } DSA_SB();
                // the binary16 split of e: in the one-launch kernel a tile's four values go in ONE slot with the half-register writes of
                // the two pairs interleaved (split4: 3 of 4 wait states gone; fused step 0.5927 -> 0.5882 ms); in the spectrogram-in
                // kernel the four separate slots measure faster (0.5223 against 0.5280 ms), so each keeps its own schedule
                auto vecE4 = [&](int t_) __attribute__((always_inline)) {
                    f16x2 h0, h1, l0, l1;
                    split4(e[t_][0], e[t_][1], e[t_][2], e[t_][3], h0, h1, l0, l1);
                    eh[4 * t_] = h0[0]; eh[4 * t_ + 1] = h0[1]; eh[4 * t_ + 2] = h1[0]; eh[4 * t_ + 3] = h1[1];
                    el[4 * t_] = l0[0]; el[4 * t_ + 1] = l0[1]; el[4 * t_ + 2] = l1[0]; el[4 * t_ + 3] = l1[1];
                };
                if constexpr (FUSED) {
                    prodE(3); DSA_SB(); if (vw) vecE4(0); DSA_SB();
                    prodE(4); DSA_SB(); if (vw) { vecC(1); vecD(1); } DSA_SB();
                    prodE(5); DSA_SB();
                    prodE(6); DSA_SB(); if (vw) vecE4(1); DSA_SB();
                    prodE(7); DSA_SB();
                } else {
                    prodE(3); DSA_SB(); if (vw) vecE(0, 0); DSA_SB();
                    prodE(4); DSA_SB(); if (vw) { vecC(1); vecD(1); } DSA_SB();
                    prodE(5); DSA_SB(); if (vw) vecE(0, 2); DSA_SB();
                    prodE(6); DSA_SB(); if (vw) vecE(1, 0); DSA_SB();
                    prodE(7); DSA_SB(); if (vw) vecE(1, 2); DSA_SB();
                }
                prodE(8); DSA_SB();
                eh_p = eh; el_p = el;
#pragma unroll
                for (int it = 0; it < 3; ++it) eah[it] = eah_n[it];
            }
#undef DSA_SB
            DSA_STAMP(9);
            rt48 = __builtin_ldexpf(rt48, back);
#endif

            // ------------- rt and its reflection into this frame's LDS windows -------------
            DSA_STAMP(1);
            {
                // rt[idx], idx = 16 it + 4 g + r, and for idx <= 27 its reflection rr[27 +- idx].  The stores
                // are unconditional: lane group 3 of tile 1 (idx 28..31) is pointed at free slots 55..62 of
                // the window instead of being masked off (a masked store is an exec-mask branch each).
                int g_it = g;
                asm volatile("" : "+v"(g_it));  // keeps the address selects inside the loop (not live across the solve)
                float* rtw = rt_lds + 4 * g_it;
                float* rra = rr_lds + 27 + 4 * g_it;
                float* rrb = rr_lds + 24 - 4 * g_it;            // rr[27 - idx] = rrb[3 - r]: immediate offsets stay non-negative
                float* rra1 = g_it < 3 ? rra + 16 : rr_lds + 55;
                float* rrb1 = g_it < 3 ? rrb - 16 : rr_lds + 59;
                const int bk = back - SE_LOG2;
#ifdef DSA_MCEP_WIN_SCALAR   // (A/B: the 28 four-byte stores of rounds 1-4; lanes (n, g) of one store hit bank 4 (n + g) + r: eight to a bank)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v0 = __builtin_ldexpf(accB[0][r], bk);
                    const float v1 = __builtin_ldexpf(accB[1][r], bk);
                    rtw[r] = v0;
                    rra[r] = v0;
                    rrb[3 - r] = v0;
                    rtw[16 + r] = v1;
                    rra1[r] = v1;
                    rrb1[3 - r] = v1;
                    rtw[32 + r] = __builtin_ldexpf(accB[2][r], bk);
                }
#else
                {   // the lane's four consecutive entries as ONE 16-byte store per window (7 stores instead of 28)
                    f32x4 w0, w1, w2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        w0[r] = __builtin_ldexpf(accB[0][r], bk);
                        w1[r] = __builtin_ldexpf(accB[1][r], bk);
                        w2[r] = __builtin_ldexpf(accB[2][r], bk);
                    }
                    *reinterpret_cast<f32x4*>(rtw) = w0;
                    *reinterpret_cast<f32x4_u4*>(rra) = w0;
                    *reinterpret_cast<f32x4*>(rrb) = __builtin_shufflevector(w0, w0, 3, 2, 1, 0);
                    *reinterpret_cast<f32x4*>(rtw + 16) = w1;
                    *reinterpret_cast<f32x4_u4*>(rra1) = w1;
                    *reinterpret_cast<f32x4_u4*>(rrb1) = __builtin_shufflevector(w1, w1, 3, 2, 1, 0);
                    *reinterpret_cast<f32x4*>(rtw + 32) = w2;
                }
#endif
                rt_lds[48] = rt48;  // same value on the four lanes of a frame
                if (HIST_RT && f_ok) {   // the row as the backward's windows want it: lane (n, g) owns rt[16 it + 4 g + r]
                    float* hr = hist_rt + ((long)iter * F + f) * M2 + 4 * g_it;
#pragma unroll
                    for (int it = 0; it < 3; ++it)
                        *reinterpret_cast<f32x4_u4*>(hr + 16 * it) = f32x4_u4{__builtin_ldexpf(accB[it][0], bk), __builtin_ldexpf(accB[it][1], bk),
                                                                            __builtin_ldexpf(accB[it][2], bk), __builtin_ldexpf(accB[it][3], bk)};
                    if (g_it == 0) hr[48] = rt48;
                }
            }
            __builtin_amdgcn_wave_barrier();
            DSA_STAMP(2);

            // ------------- rows of R + Q, symmetric elimination, back substitution (as v2) -------------
#if DSA_FUSED_DBG & 2   // reduction build: no system, no elimination
            float xq[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#else
#ifdef DSA_MCEP_SOLVE_VALU   // the column-cyclic v_fmac_f32_dpp elimination of rounds 1-2 (A/B builds; bit-identical results)
            float a[colm::TOTAL];
#else
            f32x4 a[blk::NBLK];
#endif
            {
                // slot c = 6 of every row through two per-lane pointers (see col_build_rows_p); re-derived every step so
                // that they do not occupy registers across the chains
                int gsv = gs;
                asm volatile("" : "+v"(gsv));
                const float* pa6 = gsv == 0 ? rt_q + 24 : (gsv == 1 ? rt_q : lds + H_ZERO);
                const float* pb6 = gsv == 0 ? rr_q + 3 : (gsv == 1 ? lds + H_NAV : lds + H_ZERO);
#ifdef DSA_MCEP_SOLVE_VALU
                col_build_rows_p<0>(a, rt_q, rr_q, pa6, pb6, gs);
#else
                blk_build_rows<0>(a, rt_q, rr_q, pa6, pb6, gs);
#endif
            }
            __builtin_amdgcn_wave_barrier();
            DSA_STAMP(3);
            float xq[KS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, keep_if(gq.m[1], -1.f)};
#ifdef DSA_MCEP_SOLVE_VALU
            col_elim_all(a, std::make_integer_sequence<int, M1>{});
            DSA_STAMP(4);
            col_backsub_all(a, xq, gq, std::make_integer_sequence<int, M1>{});
#else
            float ninvs[M1];
            blk_elim_all(a, gq, ninvs, std::make_integer_sequence<int, M1>{});
            DSA_STAMP(4);
#if !(DSA_FUSED_DBG & 4)
            blk_backsub_all(a, xq, gq, ninvs, std::make_integer_sequence<int, blk::NG>{});
#else
            xq[0] = ninvs[0] + a[0][0];
#endif
#endif
#endif
            xq[6] = keep_if(gq.m[0], xq[6]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) rt_q[4 * ks + gs] = xq[ks];
            __builtin_amdgcn_wave_barrier();
            // mc += x  (mcep.py:222); window entries 28.. still hold rt, and 25..27 are zero
            mcv[0] += rt_lds[8 * g];
#pragma unroll
            for (int i = 1; i < 8; ++i) mcv[i] += keep_if(g_lt3, rt_lds[8 * g + i]);
            __builtin_amdgcn_wave_barrier();
            DSA_STAMP(5);
            DSA_STAMPS_FLUSH;
            if (!(DSA_FUSED_DBG & 8) && hist && f_ok) store_mc_row(hist + ((long)(iter + 1) * F + f) * M1, g, mcv);
        }
        DSA_STAMP_T(19);
        if (f_ok) store_mc_row(mc_out + f * M1, g, mcv);
        DSA_STAMP_T(20);
        tile = tile_next;
        DSA_STAMP_T(21);
#ifdef DSA_MCEP_TIMING
        ++tcount;
        ++slot_tiles;
        if (blockIdx.x == 0 && threadIdx.x == 0) { g_mcep_stamps[13] = __builtin_readcyclecounter(); g_mcep_stamps[14] += 1; }
#endif
    }
#ifdef DSA_MCEP_TIMING
    if (lane == 0 && wave_id < 2048) {
        g_mcep_slotlog[3 * wave_id] = slot_t0;
        g_mcep_slotlog[3 * wave_id + 1] = wall_clock64();
        g_mcep_slotlog[3 * wave_id + 2] = (unsigned long long)slot_tiles;
    }
#endif
    // the counters go back to zero with the last wave out (every draw of a wave precedes its own arrival here)
#ifndef DSA_MCEP_NO_EXIT_ATOMIC   // (measurement builds: what the 2048 arrivals on one address cost at the end of the launch)
    if (lane == 0) {
        const unsigned arrived = atomicAdd(queue + 2, 1u);
        if (arrived == gridDim.x * WAVES - 1) {
            queue[0] = 0u;
            queue[1] = 0u;
            queue[2] = 0u;
        }
    }
#endif
}

}  // namespace dsa
