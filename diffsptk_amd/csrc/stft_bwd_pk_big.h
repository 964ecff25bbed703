// Packed-float32 backward of the STFT for fft_length 1024 and 2048 (included by stft.hip after stft_pk_big.h and stft_bwd_pk.h):
// autograd of ShortTimeFourierTransform._forward (stft.py:237-241), power format, constant padding, at the 44.1 / 48 kHz set-ups of
// diffsptk/utils/public.py:61-104.  Until round 6 a gradient at these geometries ran the generic row-DFT backward (one workgroup per
// frame, a radix-2 LDS transform of fft_length complex points, the framed cotangent (F, L) through memory and a second launch for the
// overlap-add): 2.45 + 0.18 ms per 102 912 frames of 1200 samples, against 0.19 ms for the packed forward.
//
// A pass is the forward's (stft_pk_big.h), run to the split, and then its adjoint, piece by piece in reverse (tools/proto/
// proto_stft_big_bwd.py is the numpy model that pins the formulas and scale factors):
//   forward    four 256-point transforms Y_r of the S = fft_length / 512 decimated subsequences of 4 / S frames, pair reads, combine
//              Z[k' + 256 q] = sum_r W_S^(r q) W_C^(r k') Y_r[k'] in the lane that holds k' and 256 - k' of every slot;
//   pack       per pair (k, C - k): X[k], conj X[C - k] by the real-FFT split, A = 2 gy[k] X[k], B = conj(2 gy[C - k] X[C - k]),
//              Zbar[k] = (A + B) + i Q, Zbar[C - k] = conj((A + B) - i Q), Q = conj(W_2C^k)(A - B)          (stft_bwd_pk.h: pack)
//   combine^H  Ybar_r[k'] = conj(W_C^(r k')) sum_q conj(W_S^(r q)) Zbar[k' + 256 q]: an inverse radix-S butterfly, lane-local;
//   the pairs go back where the forward read them, four inverse 256-point transforms (unnormalised, conjugated twiddles: the 512
//   kernel's), the window, and the overlap-add.
// Overlap-add: a wave owns a RUN of consecutive passes of one utterance and keeps the running sums in an LDS ring of fft_length
// floats; after frame n the samples [n P - left, (n + 1) P - left) are complete: stored once, their slots zeroed.  A run that starts
// inside an utterance warms up on the ceil(L / P) - 1 frames before it (their tails are what the run inherits; nothing is stored), so no
// two waves ever add into one sample and the result does not depend on how the utterance is cut into runs.
#pragma once

namespace dsa {

template <int S>
__device__ __forceinline__ void big_combine_inv(v2f (&t)[S])   // unnormalised inverse DFT of length S in place, natural order
{
    if (S == 2) {
        const v2f a = t[0], b = t[1];
        t[0] = pk_add(a, b);
        t[1] = pk_sub(a, b);
    } else {
        pk_idft4(t[0], t[1], t[2 % S], t[3 % S]);
    }
}

// (Zbar[k], Zbar[C - k]) from a = Z[k], b = Z[C - k] (both halved), W = W_2C^k and the doubled cotangents g = (2 gy[k], 2 gy[C - k])
__device__ __forceinline__ void big_pack(v2f a, v2f b, v2f W, v2f g, v2f& zk, v2f& zm)
{
    const v2f Ss = pk_add_conj(a, b), Dd = pk_sub_conj(a, b);
    const v2f Pp = pk_cmul(Dd, W);
    const v2f X1 = pk_add_negi(Ss, Pp), Y2 = pk_add_posi(Ss, Pp);   // X[k], conj(X[C - k])
    const v2f A = pk_mul_lo(X1, g), Bv = pk_mul_hi(Y2, g);
    const v2f ab = pk_add(A, Bv), amb = pk_sub(A, Bv);
    const v2f Q = pk_cmul_conj(amb, W);
    zk = pk_add_posi(ab, Q);
    zm = pk_add_posi_conj(ab, Q);
}

template <int S, int NR>   // as stft_big_fwd_pk_kernel: S 256-point sub-transforms per frame, NR sample pairs per lane
__global__ __launch_bounds__(256, 2) DSA_PK_TARGET void stft_big_bwd_pk_kernel(
    const float* __restrict__ x, const float* __restrict__ gy, long Tlen, long N, int L, int P, int left, const float* __restrict__ w,
    const float* __restrict__ twiddle, float* __restrict__ gx, long total_items, int runs_per_utt, int passes_per_utt, int warm_passes)
{
    static_assert(S == 2 || S == 4, "fft_length 1024 or 2048");
    constexpr int FPP = 4 / S;     // frames per pass
    constexpr int C = 256 * S;     // complex transform length
    constexpr int K = C + 1;       // bins per frame
    constexpr int WPB = 4;         // waves per workgroup
    constexpr int RING = 2 * C;    // floats of the overlap-add ring (>= L)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, slot = lane >> 4;
    const int phi = slot / S, r = slot % S;   // the slot's frame within the pass and its subsequence
    v2f* zbuf = reinterpret_cast<v2f*>(smem_raw) + wv * 4 * kZS;
    v2f* zf = zbuf + slot * kZS;
    v2f* t256 = reinterpret_cast<v2f*>(smem_raw) + WPB * 4 * kZS;
    float* ring = reinterpret_cast<float*>(t256 + 256) + wv * RING;
    const long nw = (long)gridDim.x * WPB;
    const long wid = (long)blockIdx.x * WPB + wv;
    if (wid >= total_items) return;

    // ---- the fetch of a pass (as the forward's): NR pairs per lane, unconditional loads from clamped addresses ----
    v2f pre[NR];
    auto fetch = [&](long b, int p) __attribute__((always_inline)) {
        const long n = (long)p * FPP + phi;
        const float* xb = x + b * Tlen;
        const long g0 = n * P - left + 2 * r + 2 * S * j;
        const bool fv = n < N;
#pragma unroll
        for (int m1 = 0; m1 < NR; ++m1) {
            const int off = 2 * (S * (j + 16 * m1) + r);
            const long idx = g0 + 32 * S * m1;
            const bool ok = fv && off < L && idx >= 0 && idx < Tlen;
            const v2f val = *reinterpret_cast<const v2f*>(xb + (ok ? idx : 0));
            pre[m1] = v2f{ok ? val.x : 0.f, ok ? val.y : 0.f};
        }
    };
    // ---- the cotangents of a pass: per frame of the pass the lane's bins (k' + 256 q, its mirror), k' = 2 lane + 1, 2 lane + 2, and
    //      the column k' = 0 (bins 256 q and C: uniform loads) ----
    v2f ga[FPP][S], gb[FPP][S];   // ga[f][q] = (gy[kk0 + 256 q], gy[kk1 + 256 q]); gb[f][q] = (gy[C - kk1 - 256 q], gy[C - kk0 - 256 q])
    float g0c[FPP][S + 1];        // gy[256 q], q = 0 .. S (bin C last)
    auto load_g = [&](long b, int p) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < FPP; ++f) {
            const long n = (long)p * FPP + f;
            const bool fv = n < N;
            const float* gr = gy + (b * N + (fv ? n : N - 1)) * K;
#pragma unroll
            for (int q = 0; q < S; ++q) {
                const v2f a = *reinterpret_cast<const v2f_u4*>(gr + 2 * lane + 1 + 256 * q);
                const v2f m = *reinterpret_cast<const v2f_u4*>(gr + C - (2 * lane + 2) - 256 * q);
                ga[f][q] = fv ? a : v2f{0.f, 0.f};
                gb[f][q] = fv ? m : v2f{0.f, 0.f};
            }
#pragma unroll
            for (int q = 0; q <= S; ++q) g0c[f][q] = fv ? gr[256 * q] : 0.f;
        }
    };

    // ---- tables (as the forward's) ----
    v2f wreg[NR];
#pragma unroll
    for (int m1 = 0; m1 < NR; ++m1) {
        const int l = 2 * (S * (j + 16 * m1) + r);
        wreg[m1] = v2f{l < L ? w[l < L ? l : 0] : 0.f, l + 1 < L ? w[l + 1 < L ? l + 1 : 0] : 0.f};
    }
    {
        v2f t4[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int i = lane + 64 * q4;
            const int m = 2 * S * (i & 15) * (i >> 4);
            t4[q4] = *reinterpret_cast<const v2f*>(twiddle + 2 * m);
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) t256[lane + 64 * q4] = t4[q4] * 0.5f;   // halved: the 1/2 of the real-FFT split (exact)
    }
    const int kk[2] = {2 * lane + 1, 2 * lane + 2};
    v2f cwa[2][S - 1], cwb[2][S - 1], sw[2][S];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int rr = 1; rr < S; ++rr) {
            cwa[h][rr - 1] = *reinterpret_cast<const v2f*>(twiddle + 2 * (2 * rr * kk[h]));               // W_C^(rr k')
            cwb[h][rr - 1] = *reinterpret_cast<const v2f*>(twiddle + 2 * (2 * rr * (256 - kk[h])));       // W_C^(rr (256 - k'))
        }
#pragma unroll
        for (int q = 0; q < S; ++q) sw[h][q] = *reinterpret_cast<const v2f*>(twiddle + 2 * (kk[h] + 256 * q));   // W_2C^(k' + 256 q)
    }
#pragma unroll
    for (int m1 = 0; m1 < NR; ++m1) asm volatile("" : "+v"(wreg[m1]));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int rr = 0; rr < S - 1; ++rr) asm volatile("" : "+v"(cwa[h][rr]), "+v"(cwb[h][rr]));
#pragma unroll
        for (int q = 0; q < S; ++q) asm volatile("" : "+v"(sw[h][q]));
    }
    constexpr float R2 = 0.70710678118654752f;

    for (long item = wid; item < total_items; item += nw) {
        const long b = (long)((unsigned long)item / (unsigned)runs_per_utt);
        const int run = (int)(item - b * runs_per_utt);
        const int p0 = (int)((long)run * passes_per_utt / runs_per_utt), p1 = (int)((long)(run + 1) * passes_per_utt / runs_per_utt);
        const int pw = p0 - warm_passes > 0 ? p0 - warm_passes : 0;   // the run warms up on the frames whose tails it inherits
        float* gxb = gx + b * Tlen;
        DSA_WAVE_SYNC();
        for (int s = lane; s < RING; s += 64) ring[s] = 0.f;
        int base = 0;                       // ring position of sample (pw FPP) P - left
        fetch(b, pw);
        load_g(b, pw);
        for (int p = pw; p < p1; ++p) {
            // ---- forward: window, four 256-point transforms, pair reads (stft_pk_big.h) ----
            v2f v[16];
#pragma unroll
            for (int m1 = 0; m1 < NR; ++m1) asm volatile("" : "+v"(pre[m1]) : : "memory");
#pragma unroll
            for (int m1 = 0; m1 < NR; ++m1) v[m1] = pk_mul(pre[m1], wreg[m1]);
#pragma unroll
            for (int m1 = NR; m1 < 16; ++m1) v[m1] = v2f{0.f, 0.f};
            if (p + 1 < p1) fetch(b, p + 1);
            DSA_WAVE_SYNC();
            pk_fft16<(NR <= 13)>(v);
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) zf[k1 * 17 + j] = pk_cmul(v[FFT16_OUT(k1)], t256[k1 * 16 + j]);
            DSA_WAVE_SYNC();
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = zf[j * 17 + i];
            DSA_WAVE_SYNC();
            pk_fft16<false>(v);
#pragma unroll
            for (int k0 = 0; k0 < 16; ++k0) {
                zf[j + 16 * k0 + (k0 < 8 ? 1 : 2)] = v[FFT16_OUT(k0)];
                if (k0 == 8 && j == 0) zf[129] = v[FFT16_OUT(k0)];
            }
            DSA_WAVE_SYNC();
            v2f pa[4][2], pb[4][2], z0[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const v2f* z = zbuf + f * kZS;
                const v4f a2 = *reinterpret_cast<const v4f*>(z + 2 * lane + 2);     // Y[2l+1], Y[2l+2]
                const v4f b2 = *reinterpret_cast<const v4f*>(z + 256 - 2 * lane);   // Y[254-2l], Y[255-2l]
                pa[f][0] = v2f{a2.x, a2.y};
                pa[f][1] = v2f{a2.z, a2.w};
                pb[f][1] = v2f{b2.x, b2.y};   // mirror of k' = 2l+2
                pb[f][0] = v2f{b2.z, b2.w};   // mirror of k' = 2l+1
                z0[f] = z[1];
            }
            DSA_WAVE_SYNC();   // all pairs are read: the cotangent pairs go back into the same places
            // ---- per frame: combine, pack, combine^H; the pairs of Ybar back where Y was read ----
#pragma unroll
            for (int f = 0; f < FPP; ++f) {
                v2f ya[2][S], yb[2][S];   // Ybar_r[k'] / Ybar_r[256 - k'] of k' = kk[h]
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v2f ta[S], tb[S];
                    ta[0] = pa[S * f][h];
                    tb[0] = pb[S * f][h];
#pragma unroll
                    for (int rr = 1; rr < S; ++rr) {
                        ta[rr] = pk_cmul(pa[S * f + rr][h], cwa[h][rr - 1]);
                        tb[rr] = pk_cmul(pb[S * f + rr][h], cwb[h][rr - 1]);
                    }
                    big_combine<S>(ta);   // ta[q] = Z[k' + 256 q]
                    big_combine<S>(tb);   // tb[q] = Z[(256 - k') + 256 q]
                    v2f za[S], zm_[S];
#pragma unroll
                    for (int q = 0; q < S; ++q) {
                        // the pair (k, C - k), k = k' + 256 q: its mirror is element S - 1 - q of the other column
                        const v2f g = v2f{2.f * (h == 0 ? ga[f][q].x : ga[f][q].y), 2.f * (h == 0 ? gb[f][q].y : gb[f][q].x)};
                        big_pack(ta[q], tb[S - 1 - q], sw[h][q], g, za[q], zm_[S - 1 - q]);   // za[q] = Zbar[k' + 256 q], zm_[q'] = Zbar[(256 - k') + 256 q']
                    }
                    big_combine_inv<S>(za);
                    big_combine_inv<S>(zm_);
                    ya[h][0] = za[0];
                    yb[h][0] = zm_[0];
#pragma unroll
                    for (int rr = 1; rr < S; ++rr) {
                        ya[h][rr] = pk_cmul_conj(za[rr], cwa[h][rr - 1]);
                        yb[h][rr] = pk_cmul_conj(zm_[rr], cwb[h][rr - 1]);
                    }
                }
                // the column k' = 0: bins 256 q and C (every lane computes them; lane 0 writes)
                v2f t0[S], zc[S];
#pragma unroll
                for (int rr = 0; rr < S; ++rr) t0[rr] = z0[S * f + rr];
                big_combine<S>(t0);       // t0[q] = Z[256 q]
                {
                    v2f dummy;
                    big_pack(t0[0], t0[0], v2f{1.f, 0.f}, v2f{4.f * g0c[f][0], 4.f * g0c[f][S]}, zc[0], dummy);          // the pair (0, C): real-valued bins, full weight
                    if (S == 2) {
                        big_pack(t0[1], t0[1], v2f{0.f, -1.f}, v2f{2.f * g0c[f][1], 2.f * g0c[f][1]}, zc[1], dummy);     // bin 256 = C / 2: its own mirror
                    } else {
                        big_pack(t0[1], t0[3 % S], v2f{R2, -R2}, v2f{2.f * g0c[f][1], 2.f * g0c[f][3 % (S + 1)]}, zc[1], zc[3 % S]);   // bins 256, 768
                        big_pack(t0[2 % S], t0[2 % S], v2f{0.f, -1.f}, v2f{2.f * g0c[f][2], 2.f * g0c[f][2]}, zc[2 % S], dummy);       // bin 512 = C / 2
                    }
                }
                big_combine_inv<S>(zc);   // zc[r] = Ybar_r[0]
#pragma unroll
                for (int rr = 0; rr < S; ++rr) {
                    v2f* z = zbuf + (S * f + rr) * kZS;
                    *reinterpret_cast<v4f*>(z + 2 * lane + 2) = v4f{ya[0][rr].x, ya[0][rr].y, ya[1][rr].x, ya[1][rr].y};
                    *reinterpret_cast<v4f*>(z + 256 - 2 * lane) = v4f{yb[1][rr].x, yb[1][rr].y, yb[0][rr].x, yb[0][rr].y};
                    if (lane == 0) z[1] = zc[rr];
                }
            }
            // ---- the next pass's cotangents: their registers are free now ----
            if (p + 1 < p1) load_g(b, p + 1);
            DSA_WAVE_SYNC();
            // ---- four inverse 256-point transforms (unnormalised, conjugated twiddles), the window ----
#pragma unroll
            for (int m1 = 0; m1 < 16; ++m1) v[m1] = zf[j + 16 * m1 + (m1 < 8 ? 1 : 2)];
            DSA_WAVE_SYNC();
            pk_ifft16(v);
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) zf[k1 * 17 + j] = pk_cmul_conj(v[FFT16_OUT(k1)], t256[k1 * 16 + j]);
            DSA_WAVE_SYNC();
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = zf[j * 17 + i];
            DSA_WAVE_SYNC();
            pk_ifft16(v);
            // lane (r, j) holds the time points m = j + 16 k0 of subsequence r: the samples 2 (S m + r), + 1 of frame phi -- in natural
            // order the frame's cotangent is one contiguous run of v2f per frame (index S m + r), in the frame's S slots of the tile
            {
                v2f* gfw = zbuf + phi * S * kZS;
#pragma unroll
                for (int k0 = 0; k0 < NR; ++k0) gfw[S * (j + 16 * k0) + r] = pk_mul(v[FFT16_OUT(k0)], wreg[k0]);
            }
            DSA_WAVE_SYNC();
            // ---- overlap-add into the ring; the frame's first P samples are then complete ----
            const bool store_ok = p >= p0;
#pragma unroll
            for (int f = 0; f < FPP; ++f) {
                const long n = (long)p * FPP + f;
                const float* gf = reinterpret_cast<const float*>(zbuf + f * S * kZS);
                // (sample PAIRS: L, P, left and Tlen are even -- the host checks -- so `base`, every frame start and every run of complete
                //  samples are even: 8-byte LDS accesses and stores, half the instructions)
                v2f* ring2 = reinterpret_cast<v2f*>(ring);
                const v2f* gf2 = reinterpret_cast<const v2f*>(gf);
                if (n < N) {
                    for (int l = 2 * lane; l < L; l += 128) {
                        const int pos = ((base + l) & (RING - 1)) >> 1;
                        ring2[pos] = ring2[pos] + gf2[l >> 1];
                    }
                }
                DSA_WAVE_SYNC();
                const long t0s = n * P - left;
                for (int s = 2 * lane; s < P; s += 128) {
                    const int pos = ((base + s) & (RING - 1)) >> 1;
                    const v2f val = ring2[pos];
                    ring2[pos] = v2f{0.f, 0.f};
                    const long t = t0s + s;
                    if (store_ok && t >= 0 && t < Tlen) *reinterpret_cast<v2f*>(gxb + t) = val;
                }
                base = (base + P) & (RING - 1);
                DSA_WAVE_SYNC();
            }
        }
        if (p1 == passes_per_utt) {   // the utterance's last run: what the ring still holds, and zeros for samples no frame reaches
            const long t0s = (long)p1 * FPP * P - left;
            for (long s = lane; t0s + s < Tlen; s += 64) {
                const long t = t0s + s;
                if (t >= 0) gxb[t] = s < RING ? ring[(base + (int)s) & (RING - 1)] : 0.f;
            }
        }
    }
}

}  // namespace dsa
