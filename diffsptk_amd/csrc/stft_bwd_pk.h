// Packed-float32 backward of the fused STFT for the BASELINE configuration, and the inverse STFT's synthesis
// (included by stft.hip after stft_pk.h).  Autograd of ShortTimeFourierTransform._forward (stft.py:237-241, power
// format) and the adjoint of the complex format / the inverse transform's irfft + window + overlap-add
// (istft.py:141-146 -> ifftr.py:138, unframe.py:192-205).
//
// Differences to stft512_bwd_kernel (same mathematics, see the derivation there):
//  * every complex value is a register pair and every butterfly / twiddle / Hermitian packing step a v_pk_*_f32
//    (stft_pk.h): the pass of four frames costs ~620 vector instructions instead of ~1500;
//  * NO partial spans in memory and no gather kernel: a wave owns a run of consecutive passes of ONE utterance and
//    carries the L - P samples that the next pass still adds to in registers (P a multiple of 16: the carried
//    samples stay on their lanes), so every waveform cotangent is written once, complete, by one store.  A run that
//    does not start at the utterance's first frame begins with a warm-up pass over the four frames before it (their
//    tail IS the carry; nothing is stored): 6 % more frame work at the bench shape, no communication between waves,
//    and -- the carried sums being the same arithmetic as the previous run's -- results that do not depend on how
//    the utterance is cut into runs.
//  * Per sample the frames are added in increasing frame order (the old pair of kernels added the frames of a pass
//    first): deterministic, last-bit different from before.
#pragma once

// DSA_SBWD_ABL (ablation bit mask for tools/gpu_ab_lib.sh builds only; 0 in the product): 1 no cotangent loads | 2 no
// waveform fetch | 4 no stores | 8 no forward transform | 16 no inverse transform | 32 no overlap-add gather
#ifndef DSA_SBWD_ABL
#define DSA_SBWD_ABL 0
#endif

namespace dsa {

__device__ __forceinline__ v2f pk_add_posi_conj(v2f ab, v2f q)   // conj(ab - i q) = (ab.re + q.im, q.re - ab.im)
{
    v2f r;
#if DSA_PK_CROSSED
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(ab), "v"(q));
#else   // no packed float32 instruction whose low half reads a high source half (pk_math.h, DSA_PK_CROSSED)
    asm("v_add_f32 %0, %1, %2" : "=v"(r.x) : "v"(ab.x), "v"(q.y));
    asm("v_sub_f32 %0, %2, %1" : "=v"(r.y) : "v"(ab.y), "v"(q.x));
#endif
    return r;
}
// conj(t) * a, t = (c, s) in vector registers: (a.re c + a.im s, a.im c - a.re s)
__device__ __forceinline__ v2f pk_cmul_conj(v2f a, v2f t)
{
    v2f t1, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t1) : "v"(a), "v"(t));
#if DSA_PK_CROSSED
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(t), "v"(t1));
#else
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r.x) : "v"(a.y), "v"(t.y), "v"(t1.x));
    asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(r.y) : "v"(a.x), "v"(t.y), "v"(t1.y));
#endif
    return r;
}
__device__ __forceinline__ v2f pk_cmul_conj_s(v2f a, v2f t)   // the same, t uniform in a scalar register pair
{
    v2f t1, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t1) : "v"(a), "s"(t));
#if DSA_PK_CROSSED
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(a), "s"(t), "v"(t1));
#else
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r.x) : "v"(a.y), "s"(t.y), "v"(t1.x));
    asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(r.y) : "v"(a.x), "s"(t.y), "v"(t1.y));
#endif
    return r;
}
__device__ __forceinline__ v2f pk_mul_lo(v2f a, v2f g)   // a * g.x
{
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(g));
    return r;
}
__device__ __forceinline__ v2f pk_mul_hi(v2f a, v2f g)   // a * g.y
{
    v2f r;
#if DSA_PK_CROSSED
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(g));
#else
    asm("v_mul_f32 %0, %1, %2" : "=v"(r.x) : "v"(a.x), "v"(g.y));
    asm("v_mul_f32 %0, %1, %2" : "=v"(r.y) : "v"(a.y), "v"(g.y));
#endif
    return r;
}

// 4-point INVERSE DFT in place (W4 = +i)
__device__ __forceinline__ void pk_idft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3)
{
    const v2f s02 = pk_add(a0, a2), d02 = pk_sub(a0, a2), s13 = pk_add(a1, a3), d13 = pk_sub(a1, a3);
    a0 = pk_add(s02, s13);
    a2 = pk_sub(s02, s13);
    a1 = pk_add_posi(d02, d13);
    a3 = pk_add_negi(d02, d13);
}
// the same with a2 standing for +i a2
__device__ __forceinline__ void pk_idft4_posi2(v2f& a0, v2f& a1, v2f& a2, v2f& a3)
{
    const v2f s02 = pk_add_posi(a0, a2), d02 = pk_add_negi(a0, a2), s13 = pk_add(a1, a3), d13 = pk_sub(a1, a3);
    a0 = pk_add(s02, s13);
    a2 = pk_sub(s02, s13);
    a1 = pk_add_posi(d02, d13);
    a3 = pk_add_negi(d02, d13);
}
// 16-point unnormalised inverse DFT in registers, output order as pk_fft16 (x[n] in v[FFT16_OUT(n)])
__device__ __forceinline__ void pk_ifft16(v2f (&v)[16])
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    pk_idft4(v[0], v[4], v[8], v[12]);
    pk_idft4(v[1], v[5], v[9], v[13]);
    pk_idft4(v[2], v[6], v[10], v[14]);
    pk_idft4(v[3], v[7], v[11], v[15]);
    v[5] = pk_cmul_s(v[5], v2f{C1, S1});
    v[9] = pk_cmul_s(v[9], v2f{R2, R2});
    v[13] = pk_cmul_s(v[13], v2f{S1, C1});
    v[6] = pk_cmul_s(v[6], v2f{R2, R2});
    v[14] = pk_cmul_s(v[14], v2f{-R2, R2});
    v[7] = pk_cmul_s(v[7], v2f{S1, C1});
    v[11] = pk_cmul_s(v[11], v2f{-R2, R2});
    v[15] = pk_cmul_s(v[15], v2f{-C1, -S1});
    pk_idft4(v[0], v[1], v[2], v[3]);
    pk_idft4(v[4], v[5], v[6], v[7]);
    pk_idft4_posi2(v[8], v[9], v[10], v[11]);
    pk_idft4(v[12], v[13], v[14], v[15]);
}

// LC / PC: frame length and period at compile time (LC <= 512, PC a multiple of 16, LC - PC <= 4 PC: one warm-up pass
// covers everything a run inherits; instantiated for 400 / 80 and 400 / 160, see the launcher).
// CPLX: the cotangent is complex (format "complex", or the inverse transform with scale 1/512): the waveform is not
// needed.  `div` != nullptr: the stored value is divided by div[t] + div_eps (Unframe's normalisation,
// unframe.py:203-205).
// MAG (real cotangent only): the cotangent belongs to sqrt(|X|^2 + eps) instead of |X|^2 + eps (spec.py:129, the
// amplitude-domain filter bank / MFCC front end): one more factor 1 / (2 sqrt(.)) per bin.
template <int LC, int PC, bool CPLX, bool MAG = false>
__global__ __launch_bounds__(256, 4) DSA_PK_TARGET void stft512_bwd_pk_kernel(
    const float* __restrict__ x, const float* __restrict__ gy, long Tlen, long N, int left, const float* __restrict__ w,
    const float* __restrict__ twiddle, float cot_scale, float cot_edge, float* __restrict__ gx, long total_items,
    int runs_per_utt, int passes_per_utt, const float* __restrict__ div, float div_eps, float eps)
{
    static_assert(!(CPLX && MAG), "the magnitude factor belongs to a real cotangent");
    constexpr int L = LC, P = PC, K = 257;
    constexpr int NR = (LC + 31) / 32;          // sample pairs of a lane inside the frame
    constexpr int SPAN = 3 * PC + LC;           // samples a pass touches
    constexpr int NI = (SPAN + 63) / 64;        // samples per lane of the span
    constexpr int NS = 4 * PC / 64;             // ... of which the first NS are complete after the pass
    constexpr int TB = 4;                       // twiddle-table reads per batch
    static_assert(PC % 16 == 0 && LC - PC <= 4 * PC && NI - NS <= NS && SPAN <= 2 * kFPW * kZS, "unsupported frame geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    v2f* zbuf = reinterpret_cast<v2f*>(smem_raw) + wv * kFPW * kZS;
    float* io_buf = reinterpret_cast<float*>(zbuf);
    constexpr int WPB = 4;   // waves per workgroup: they share the twiddle and the window table, nothing else (no barriers)
    v2f* t256 = reinterpret_cast<v2f*>(smem_raw) + WPB * kFPW * kZS;
    v2f* wtab = t256 + 256;   // [16][NR] window pairs (in registers they would cost 26 of the 128)
    const long nw = (long)gridDim.x * WPB;
    const long wid = (long)blockIdx.x * WPB + wv;
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, fl = lane >> 4;
    if (wid >= total_items) return;

    {
        // every wave writes the whole (identical) tables before its first use: no workgroup barrier needed
        v2f t4[4], wt[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int i = lane + 64 * q4;
            const int m = 2 * (i & 15) * (i >> 4);
            t4[q4] = *reinterpret_cast<const v2f*>(twiddle + 2 * m);
            const int l = 2 * (i / NR) + 32 * (i % NR);
            wt[q4] = v2f{(i < 16 * NR && l < L) ? w[l < L ? l : 0] : 0.f, (i < 16 * NR && l + 1 < L) ? w[l + 1 < L ? l + 1 : 0] : 0.f};
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            t256[lane + 64 * q4] = t4[q4] * 0.5f;   // halved: the 1/2 of the real-FFT split (exact)
            if (lane + 64 * q4 < 16 * NR) wtab[lane + 64 * q4] = wt[q4];
        }
    }
    v2f twA = v2f{twiddle[2 * lane], twiddle[2 * lane + 1]};                 // W512^k, k = lane
    v2f twB = v2f{twiddle[2 * (lane + 64)], twiddle[2 * (lane + 64) + 1]};   //         k = lane + 64
    asm volatile("" : "+v"(twA), "+v"(twB));
    const v2f* wrow = wtab + j * NR;
    v2f* zf = zbuf + fl * kZS;
    // the inverse runs through the halved table too: every cotangent is doubled on the way in
    const float gsc = CPLX ? 2.f * cot_scale : 2.f;
    const float gsc0 = CPLX ? 2.f * cot_scale * cot_edge : 4.f;   // k = 0 / 256: real-valued bins, full weight

    // ---- the passes of this wave: runs of consecutive passes, one run after the other ----
    struct Cursor {
        long item, b;
        int p, p0, p1;
        bool ok;
    };
    auto start_item = [&](long item) __attribute__((always_inline)) -> Cursor {
        Cursor c;
        c.item = item;
        c.ok = item < total_items;
        const long it = c.ok ? item : 0;
        c.b = (long)((unsigned long)it / (unsigned)runs_per_utt);
        const int run = (int)(it - c.b * runs_per_utt);
        c.p0 = (int)((long)run * passes_per_utt / runs_per_utt);
        c.p1 = (int)((long)(run + 1) * passes_per_utt / runs_per_utt);
        c.p = c.p0 > 0 ? c.p0 - 1 : 0;   // a run inside the utterance warms up on the pass before it
        return c;
    };
    auto next_pass = [&](const Cursor& c) __attribute__((always_inline)) -> Cursor {
        if (c.p + 1 < c.p1) {
            Cursor n = c;
            ++n.p;
            return n;
        }
        return start_item(c.item + nw);
    };
    // cotangent registers of one pass.  power: (g[k], g[256 - k]) for k = lane (gA) / lane + 64 (gB); complex: the bins
    // k (gA, gB) and their partners 256 - k (gA2, gB2); gM: bin 128 of frame `lane` (lanes 0..3)
    v2f gA[kFPW], gB[kFPW], gA2[kFPW], gB2[kFPW], gM;
    auto load_g = [&](const Cursor& c) __attribute__((always_inline)) {
        if (DSA_SBWD_ABL & 1) {
#pragma unroll
            for (int f = 0; f < kFPW; ++f) gA[f] = gB[f] = gA2[f] = gB2[f] = v2f{1.f + c.p, 0.5f};
            gM = v2f{1.f, 1.f};
            return;
        }
        const long frame0 = (long)c.p * kFPW;
        const int nvalid = (int)((N - frame0) < kFPW ? (N - frame0) : kFPW);
        // one 64-bit base per pass; everything else is a 32-bit lane offset plus an immediate
        const float* gp = gy + (c.b * N + frame0) * (CPLX ? 2 * K : K);
        const unsigned up = (unsigned)lane, dn = 256u - (unsigned)lane;
        gM = v2f{0.f, 0.f};
        if constexpr (CPLX) {
            const v2f* g2 = reinterpret_cast<const v2f*>(gp);
            if (nvalid == kFPW) {
#pragma unroll
                for (int f = 0; f < kFPW; ++f) {
                    gA[f] = g2[f * K + up];
                    gA2[f] = g2[f * K + dn];
                    gB[f] = g2[f * K + 64 + up];
                    gB2[f] = g2[f * K - 64 + dn];
                }
            } else {
#pragma unroll
                for (int f = 0; f < kFPW; ++f) {
                    const bool fv = f < nvalid;
                    const int r = fv ? f * K : 0;
                    const v2f z = v2f{0.f, 0.f};
                    gA[f] = fv ? g2[r + up] : z;
                    gA2[f] = fv ? g2[r + dn] : z;
                    gB[f] = fv ? g2[r + 64 + up] : z;
                    gB2[f] = fv ? g2[r - 64 + dn] : z;
                }
            }
            if (lane < nvalid) gM = g2[up * K + 128];
        } else {
            if (nvalid == kFPW) {
#pragma unroll
                for (int f = 0; f < kFPW; ++f) {
                    gA[f] = v2f{gp[f * K + up], gp[f * K + dn]};
                    gB[f] = v2f{gp[f * K + 64 + up], gp[f * K - 64 + dn]};
                }
            } else {
#pragma unroll
                for (int f = 0; f < kFPW; ++f) {
                    const bool fv = f < nvalid;
                    const int r = fv ? f * K : 0;
                    gA[f] = fv ? v2f{gp[r + up], gp[r + dn]} : v2f{0.f, 0.f};
                    gB[f] = fv ? v2f{gp[r + 64 + up], gp[r - 64 + dn]} : v2f{0.f, 0.f};
                }
            }
            if (lane < nvalid) {
                const float g = gp[up * K + 128];
                gM = v2f{g, g};
            }
        }
    };
    // the stretch of samples a pass's four frames share: fetched into registers one pass ahead (interior, aligned
    // stretches), written to the tile when the pass begins
    constexpr int n4 = SPAN >> 2;
    constexpr int NPRE = (n4 + 63) / 64;   // 16-byte fetches per lane and stretch
    static_assert(CPLX || (SPAN & 3) == 0, "the stretch is fetched in 16-byte pieces");
    v4f pre[NPRE];
#pragma unroll
    for (int q = 0; q < NPRE; ++q) pre[q] = v4f{0.f, 0.f, 0.f, 0.f};
    auto prefetch_x = [&](const Cursor& c) __attribute__((always_inline)) -> bool {
        if constexpr (CPLX) return false;
        if (DSA_SBWD_ABL & 2) {
#pragma unroll
            for (int q = 0; q < NPRE; ++q) pre[q] = v4f{0.25f, -0.5f, 1.f, 0.125f};
            return true;
        }
        const long g0 = (long)c.p * kFPW * P - left;
        const float* xs = x + c.b * Tlen + g0;
        if (g0 >= 0 && g0 + SPAN <= Tlen && (((size_t)xs) & 15) == 0) {
            // 32-bit byte offsets from an opaque copy of the lane index, formed here: as a 64-bit index the clamped offset of the last
            // piece was hoisted out of the pass loop, spilled (the kernel sits at 128 registers), and its reload -- a
            // `s_waitcnt vmcnt(0)` in the middle of this fetch -- made every pass wait for the first piece's trip to memory and for the
            // stores issued before it
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const char* src = reinterpret_cast<const char*>(xs);
#pragma unroll
            for (int q = 0; q < NPRE; ++q) {
                const unsigned idx = (unsigned)(ln + 64 * q < n4 ? ln + 64 * q : n4 - 1);
                pre[q] = *reinterpret_cast<const v4f*>(src + idx * 16u);
            }
            return true;
        }
        return false;
    };

    Cursor cur = start_item(wid);
    load_g(cur);
    bool pre_ok = prefetch_x(cur);
    float carry[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) carry[i] = 0.f;
    for (;;) {
        const bool warm = cur.p < cur.p0;
        const long frame0 = (long)cur.p * kFPW;
        const int nvalid = (int)((N - frame0) < kFPW ? (N - frame0) : kFPW);
        const long g0 = frame0 * P - left;
        float* gxb = gx + cur.b * Tlen;
        DSA_WAVE_SYNC();
        v2f v[16];
        if constexpr (!CPLX) {
            // ---- the stretch into the tile ----
            if (pre_ok) {
#pragma unroll
                for (int q = 0; q < NPRE; ++q) asm volatile("" : "+v"(pre[q]) : : "memory");
                v4f* dst4 = reinterpret_cast<v4f*>(io_buf);
#pragma unroll
                for (int q = 0; q < NPRE; ++q)
                    if (lane + 64 * q < n4) dst4[lane + 64 * q] = pre[q];
            } else {
                const float* xb = x + cur.b * Tlen;
                for (int s = lane; s < SPAN; s += 64) io_buf[s] = load_padded(xb, g0 + s, Tlen, (int)DSA_PAD_CONSTANT);
            }
            DSA_WAVE_SYNC();
            {
                const v2f* src = reinterpret_cast<const v2f*>(io_buf + fl * P + 2 * j);
                v2f raw[NR];
#pragma unroll
                for (int m1 = 0; m1 < NR; ++m1) raw[m1] = src[16 * m1];
#pragma unroll
                for (int m1 = 0; m1 < NR; ++m1) {
                    const bool in0 = 32 * m1 + 30 < LC || 32 * m1 + 2 * j < LC;
                    const bool in1 = 32 * m1 + 31 < LC || 32 * m1 + 1 + 2 * j < LC;
                    const v2f r = v2f{in0 ? raw[m1].x : 0.f, in1 ? raw[m1].y : 0.f};
                    v[m1] = pk_mul(r, wrow[m1]);
                }
#pragma unroll
                for (int m1 = NR; m1 < 16; ++m1) v[m1] = v2f{0.f, 0.f};
                if (nvalid < kFPW && fl >= nvalid) {   // frames past the utterance's last: their (zero) cotangent must not meet a non-finite sample
#pragma unroll
                    for (int m1 = 0; m1 < NR; ++m1) v[m1] = v2f{0.f, 0.f};
                }
            }
            DSA_WAVE_SYNC();
            if (!(DSA_SBWD_ABL & 8)) pk_fft16<(NR <= 13)>(v);
            // (the table reads go in batches of TB ahead of the stores they feed: read -> multiply -> store one at a time
            // would wait out an LDS round trip per element -- the compiler cannot move a table read above a tile store)
#pragma unroll
            for (int kb = 0; kb < 16; kb += TB) {
                v2f tw[TB];
#pragma unroll
                for (int q = 0; q < TB; ++q) tw[q] = t256[(kb + q) * 16 + j];
#pragma unroll
                for (int q = 0; q < TB; ++q) zf[(kb + q) * 17 + j] = pk_cmul(v[FFT16_OUT(kb + q)], tw[q]);
            }
            DSA_WAVE_SYNC();
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = zf[j * 17 + i];
            DSA_WAVE_SYNC();
            if (!(DSA_SBWD_ABL & 8)) pk_fft16<false>(v);
#pragma unroll
            for (int k0 = 0; k0 < 16; ++k0) zf[j + 16 * k0] = v[FFT16_OUT(k0)];   // Z[k] / 2, natural order
            DSA_WAVE_SYNC();
        }
        // ---- split -> cotangent of the half spectrum -> Hermitian packing, in place: a lane reads and writes only
        //      its own positions k, 256 - k (k = lane, lane + 64); bin 128 of frame f on lane f ----
        //   S = a + conj(b), Dd = a - conj(b), Pp = W^k Dd:  X[k] = S - i Pp,  conj(X[256-k]) = S + i Pp
        //   A = g1 X[k], B = conj(g2 X[256-k]);  Zin[k] = (A + B) + i Q,  Zin[256-k] = conj((A + B) - i Q),
        //   Q = conj(W^k) (A - B)
        auto pack = [&](v2f a, v2f bq, v2f W, v2f g, v2f g2, v2f& zk, v2f& zm) __attribute__((always_inline)) {
            v2f A, Bv;
            if constexpr (CPLX) {
                A = g;                                                  // S[k]       (already scaled)
                Bv = v2f{g2.x, -g2.y};                                  // conj(S[256-k])
            } else {
                const v2f S = pk_add_conj(a, bq), Dd = pk_sub_conj(a, bq);
                const v2f Pp = pk_cmul(Dd, W);
                const v2f X1 = pk_add_negi(S, Pp), Y2 = pk_add_posi(S, Pp);   // X[k], conj(X[256 - k])
                if constexpr (MAG) {
                    // d sqrt(s) / d s = 1 / (2 sqrt(s)), s = |X|^2 + eps of either bin (v_rsq_f32: 1 ulp)
                    const float s1 = __builtin_fmaf(X1.x, X1.x, __builtin_fmaf(X1.y, X1.y, eps));
                    const float s2 = __builtin_fmaf(Y2.x, Y2.x, __builtin_fmaf(Y2.y, Y2.y, eps));
                    g = v2f{g.x * (0.5f * __builtin_amdgcn_rsqf(s1)), g.y * (0.5f * __builtin_amdgcn_rsqf(s2))};
                }
                A = pk_mul_lo(X1, g);
                Bv = pk_mul_hi(Y2, g);
            }
            const v2f ab = pk_add(A, Bv), amb = pk_sub(A, Bv);
            const v2f Q = pk_cmul_conj(amb, W);
            zk = pk_add_posi(ab, Q);
            zm = pk_add_posi_conj(ab, Q);
        };
        const float s0 = lane == 0 ? gsc0 : gsc;
        v2f za0[kFPW], zb0[kFPW], za1[kFPW], zb1[kFPW], zmid = v2f{0.f, 0.f};   // all reads ahead of the in-place writes (see TB)
        if constexpr (!CPLX) {
#pragma unroll
            for (int f = 0; f < kFPW; ++f) {
                const v2f* z = zbuf + f * kZS;
                za0[f] = z[lane], zb0[f] = z[(256 - lane) & 255], za1[f] = z[lane + 64], zb1[f] = z[192 - lane];
            }
            zmid = zbuf[(lane & 3) * kZS + 128];
        }
#pragma unroll
        for (int f = 0; f < kFPW; ++f) {
            v2f* z = zbuf + f * kZS;
            v2f zk, zm;
            if constexpr (CPLX) {
                // k = 0 pairs with 256: both real-valued (the imaginary parts carry no weight)
                v2f g1 = gA[f] * s0, g2 = gA2[f] * s0;
                if (lane == 0) g1.y = 0.f, g2.y = 0.f;
                pack(v2f{0.f, 0.f}, v2f{0.f, 0.f}, twA, g1, g2, zk, zm);
                z[lane] = zk;
                if (lane != 0) z[256 - lane] = zm;
                pack(v2f{0.f, 0.f}, v2f{0.f, 0.f}, twB, gB[f] * gsc, gB2[f] * gsc, zk, zm);
                z[lane + 64] = zk;
                z[192 - lane] = zm;
            } else {
                const v2f a0 = za0[f], b0 = zb0[f], a1 = za1[f], b1 = zb1[f];
                pack(a0, b0, twA, gA[f] * s0, v2f{0.f, 0.f}, zk, zm);
                z[lane] = zk;
                if (lane != 0) z[256 - lane] = zm;
                pack(a1, b1, twB, gB[f] * gsc, v2f{0.f, 0.f}, zk, zm);
                z[lane + 64] = zk;
                z[192 - lane] = zm;
            }
        }
        {   // bin 128 (W = -i) of frame `lane`
            v2f* z = zbuf + (lane & 3) * kZS;
            v2f zk, zm;
            const v2f a = zmid;
            pack(a, a, v2f{0.f, -1.f}, gM * gsc, gM * gsc, zk, zm);
            if (lane < kFPW) z[128] = zk;
        }
        // ---- the next pass's cotangents and stretch: their registers are free now, and they have the whole inverse
        //      transform (and the next forward transform) to arrive ----
        const Cursor nxt = next_pass(cur);
        if (nxt.ok) {
            load_g(nxt);
            pre_ok = prefetch_x(nxt);
        }
        DSA_WAVE_SYNC();
        // ---- inverse 256-point transform (unnormalised, conjugated twiddles), same data movement ----
#pragma unroll
        for (int m1 = 0; m1 < 16; ++m1) v[m1] = zf[j + 16 * m1];
        DSA_WAVE_SYNC();
        if (!(DSA_SBWD_ABL & 16)) pk_ifft16(v);
#pragma unroll
        for (int kb = 0; kb < 16; kb += TB) {
            v2f tw[TB];
#pragma unroll
            for (int q = 0; q < TB; ++q) tw[q] = t256[(kb + q) * 16 + j];
#pragma unroll
            for (int q = 0; q < TB; ++q) zf[(kb + q) * 17 + j] = pk_cmul_conj(v[FFT16_OUT(kb + q)], tw[q]);
        }
        DSA_WAVE_SYNC();
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = zf[j * 17 + i];
        DSA_WAVE_SYNC();
        if (!(DSA_SBWD_ABL & 16)) pk_ifft16(v);
        // lane j holds time points m = j + 16 k0 = samples 2m, 2m + 1: the forward's register <-> sample map
#pragma unroll
        for (int k0 = 0; k0 < NR; ++k0) zf[j + 16 * k0] = pk_mul(v[FFT16_OUT(k0)], wrow[k0]);
        DSA_WAVE_SYNC();
        // ---- overlap-add.  Sample s = lane + 64 i of the pass's span takes frame f's value at l = s - f P; the
        //      first NS values per lane are complete (stored), the rest is carried to the next pass ----
        float acc[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            acc[i] = i < NS ? carry[i < NS ? i : 0] : 0.f;
#pragma unroll
            for (int f = 0; f < kFPW; ++f) {
                const int lo = 64 * i - P * f;   // l of lane 0
                if (lo + 63 < 0 || lo >= L || ((DSA_SBWD_ABL & 32) && f > 0)) continue;
                const float* gf = reinterpret_cast<const float*>(zbuf + f * kZS);
                if (lo >= 0 && lo + 63 < L) {
                    acc[i] += gf[lo + lane];
                } else {
                    const int l = lo + lane;
                    const bool ok = l >= 0 && l < L;
                    const float t = gf[ok ? l : 0];
                    acc[i] += ok ? t : 0.f;
                }
            }
        }
        if (!warm && (!(DSA_SBWD_ABL & 4) || acc[0] == 123.456f)) {
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const long t = g0 + lane + 64 * i;
                if (t >= 0 && t < Tlen) gxb[t] = div ? acc[i] / (div[t] + div_eps) : acc[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) carry[i] = NS + i < NI ? acc[NS + i < NI ? NS + i : 0] : 0.f;
        if (cur.p + 1 == cur.p1) {   // the run ends
            if (cur.p1 == passes_per_utt) {   // ... with the utterance: what is still carried is complete too
                const long g1 = (long)passes_per_utt * kFPW * P - left;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const long t = g1 + lane + 64 * i;
                    if (t >= 0 && t < Tlen) gxb[t] = div ? carry[i] / (div[t] + div_eps) : carry[i];
                }
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) carry[i] = 0.f;
        }
        if (!nxt.ok) break;
        cur = nxt;
    }
}

}  // namespace dsa
