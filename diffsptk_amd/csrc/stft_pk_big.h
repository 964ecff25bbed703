// Packed-float32 STFT forward for fft_length 1024 and 2048 (included by stft.hip): the 48 kHz / 44.1 kHz set-ups of
// diffsptk/utils/public.py:61-104 (frame_length 800-1200, frame_period 200-256), which rounds 1-5 ran on the generic row-DFT kernel
// at 5.6-6.7 % of the HBM peak (a fifth of the whole 48 kHz analysis).  Semantics: ShortTimeFourierTransform._forward,
// stft.py:237-241 -- Frame (frame.py:130-140, constant padding), Window (window.py:190, zero padding to fft_length on the right),
// rfft, power spectrum + eps (spec.py:173); the other options keep the generic kernel.
//
// The pass of stft512_fwd_pk_kernel (stft_pk.h) computes FOUR 256-point complex FFTs per wave -- four frames of a 512-point real
// transform, 16 lanes each: radix-16 in registers, twiddle + transposition through the wave's own LDS tile, radix-16 again.  Here
// the same four transforms are the S = fft_length / 512 DECIMATED subsequences of 4 / S frames: with the frame's sample pairs
// c[n] = (x[2 n], x[2 n + 1]) as a complex sequence of length C = 256 S, subsequence r is c[S m + r], m = 0 .. 255, and
//     Z[k' + 256 q] = sum_r W_S^(r q) W_C^(r k') Y_r[k']                    (one radix-S butterfly per k', lane-local)
// because the real-FFT split's lane already holds bin k' and its mirror 256 - k' of EVERY slot (the 16-byte pair reads of the
// 512 kernel): Z[k' + 256 q] and Z[C - (k' + 256 q)] = Z[(256 - k') + 256 (S - 1 - q)] meet in one lane, so the combine, the split
// X[k] = (Z[k] + conj Z[C - k]) / 2 - i W_2C^k (Z[k] - conj Z[C - k]) / 2 and |X|^2 + eps need no further exchange.
// Memory side: a lane's samples are 8-byte pairs at a 8 S-byte stride -- the four slots (frame, r) of a wave interleave to 512
// contiguous bytes per load instruction --, fetched straight into registers ONE PASS AHEAD (no staged stretch: consecutive frames
// overlap by L - P samples and find them in cache); a pass stores (4 / S) x (C + 1) floats as 8-byte pairs, 512 contiguous bytes
// per instruction, exactly like the 512 kernel.  Per pass the same FFT work and the same bytes as the 512 kernel's pass.
#pragma once

#include "pk_math.h"

namespace dsa {

template <int S>
__device__ __forceinline__ void big_combine(v2f (&t)[S])   // forward DFT of length S in place, natural order
{
    if (S == 2) {
        const v2f a = t[0], b = t[1];
        t[0] = pk_add(a, b);
        t[1] = pk_sub(a, b);
    } else {
        pk_dft4(t[0], t[1], t[2 % S], t[3 % S]);
    }
}

// |X[k]|^2 + eps and |X[C - k]|^2 + eps (in .x / .y) from a = Z[k], b = Z[C - k] (both halved) and w = W_2C^k
__device__ __forceinline__ v2f big_split_power(v2f a, v2f b, v2f w, v2f eps2)
{
    const v2f Ss = pk_add_conj(a, b), Dd = pk_sub_conj(a, b);
    const v2f Pp = pk_cmul(Dd, w);
    const v2f R = pk_lo_pm_hi(Ss, Pp), I = pk_hi_mp_lo(Ss, Pp);
    v2f s = pk_fma_sc(R, R, eps2);
    return pk_fma(I, I, s);
}

template <int S, int NR>   // S: 256-point sub-transforms per frame (fft_length = 512 S); NR: sample pairs per lane = ceil(L / (32 S))
__global__ __launch_bounds__(256, 2) DSA_PK_TARGET void stft_big_fwd_pk_kernel(
    const float* __restrict__ x, long Tlen, long N, int L, int P, int left, const float* __restrict__ w,
    const float* __restrict__ twiddle, float eps, float* __restrict__ y, long total_chunks, int chunks_per_utt)
{
    static_assert(S == 2 || S == 4, "fft_length 1024 or 2048");
    constexpr int FPP = 4 / S;     // frames per pass
    constexpr int C = 256 * S;     // complex transform length
    constexpr int K = C + 1;       // bins per frame
    constexpr int WPB = 4;         // waves per workgroup: they share the 256-entry twiddle table, nothing else
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, slot = lane >> 4;
    const int phi = slot / S, r = slot % S;   // the slot's frame within the pass and its subsequence
    v2f* zbuf = reinterpret_cast<v2f*>(smem_raw) + wv * 4 * kZS;
    v2f* zf = zbuf + slot * kZS;
    v2f* t256 = reinterpret_cast<v2f*>(smem_raw) + WPB * 4 * kZS;
    const long nw = (long)gridDim.x * WPB;
    const long wid = (long)blockIdx.x * WPB + wv;
    if (wid >= total_chunks) return;

    // ---- the fetch of a pass: NR pairs per lane, unconditional loads from clamped addresses, zeros selected afterwards ----
    v2f pre[NR];
    auto fetch = [&](long c) __attribute__((always_inline)) {
        const long b = (long)((unsigned long)c / (unsigned)chunks_per_utt);
        const int ci = (int)(c - b * chunks_per_utt);
        const long n = (long)ci * FPP + phi;
        const float* xb = x + b * Tlen;
        const long g0 = n * P - left + 2 * r + 2 * S * j;
        const bool fv = n < N;
#pragma unroll
        for (int m1 = 0; m1 < NR; ++m1) {
            const int off = 2 * (S * (j + 16 * m1) + r);           // position inside the frame (even; L is even)
            const long idx = g0 + 32 * S * m1;
            const bool ok = fv && off < L && idx >= 0 && idx < Tlen;   // frame.py:130-140: constant (zero) padding on the fly
            const v2f val = *reinterpret_cast<const v2f*>(xb + (ok ? idx : 0));
            // selected, never multiplied: padding is exact and a non-finite neighbour stays out of frames that do not contain it
            pre[m1] = v2f{ok ? val.x : 0.f, ok ? val.y : 0.f};
        }
    };
    long c = wid;
    fetch(c);

    // ---- tables: window pairs of this lane's samples, the 256-point twiddles (LDS), combine and split twiddles ----
    v2f wreg[NR];
#pragma unroll
    for (int m1 = 0; m1 < NR; ++m1) {
        const int l = 2 * (S * (j + 16 * m1) + r);
        wreg[m1] = v2f{l < L ? w[l < L ? l : 0] : 0.f, l + 1 < L ? w[l + 1 < L ? l + 1 : 0] : 0.f};
    }
    {
        // entry [k1 = i >> 4][jj = i & 15]: W256^(jj k1) = W_nfft^(2 S jj k1), HALVED: the 1/2 of the real-FFT split (exact; the
        // combine and the split are linear).  Every wave writes the whole, identical table and reads it only after its own writes.
        v2f t4[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int i = lane + 64 * q4;
            const int m = 2 * S * (i & 15) * (i >> 4);
            t4[q4] = *reinterpret_cast<const v2f*>(twiddle + 2 * m);
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) t256[lane + 64 * q4] = t4[q4] * 0.5f;
    }
    // the lane's bins: k' = 2 lane + 1, 2 lane + 2 and their mirrors 256 - k' (the pair reads of the split)
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
    const v2f eps2 = v2f{eps, eps};
    // every prologue load is consumed HERE (stft_pk.h: otherwise its wait lands inside the pass loop, on top of the pass's own)
#pragma unroll
    for (int m1 = 0; m1 < NR; ++m1) asm volatile("" : "+v"(wreg[m1]));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int rr = 0; rr < S - 1; ++rr) asm volatile("" : "+v"(cwa[h][rr]), "+v"(cwb[h][rr]));
#pragma unroll
        for (int q = 0; q < S; ++q) asm volatile("" : "+v"(sw[h][q]));
    }

    for (; c < total_chunks; c += nw) {
        const long b = (long)((unsigned long)c / (unsigned)chunks_per_utt);
        const int ci = (int)(c - b * chunks_per_utt);
        const long frame0 = (long)ci * FPP;
        const int nvalid = (int)((N - frame0) < FPP ? (N - frame0) : FPP);
        // ---- window (window.py:190), then the fetch for the NEXT pass goes out: it has this whole pass to arrive ----
        v2f v[16];
#pragma unroll
        for (int m1 = 0; m1 < NR; ++m1) asm volatile("" : "+v"(pre[m1]) : : "memory");
#pragma unroll
        for (int m1 = 0; m1 < NR; ++m1) v[m1] = pk_mul(pre[m1], wreg[m1]);
#pragma unroll
        for (int m1 = NR; m1 < 16; ++m1) v[m1] = v2f{0.f, 0.f};
        if (c + nw < total_chunks) fetch(c + nw);
        // ---- four 256-point complex FFTs: radix 16, twiddle + transposition through the slot's LDS tile, radix 16 ----
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
            // Y[k'], k' = j + 16 k0, at position k' + 1 (k' <= 128) or k' + 2 (k' >= 128; Y[128] at both 129 and 130): the pairs
            // (Y[2l+1], Y[2l+2]) and (Y[254-2l], Y[255-2l]) are ONE 16-byte aligned read each (stft_pk.h)
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
        DSA_WAVE_SYNC();   // all pairs are read: the next pass may overwrite the tile
        // the NEXT pass's fetch is waited for here, BEFORE this pass's stores (vector-memory operations retire in order: a wait
        // placed after the stores would wait for them too)
#pragma unroll
        for (int m1 = 0; m1 < NR; ++m1) asm volatile("" : "+v"(pre[m1]) : : "memory");
        const long row0 = b * N + frame0;
#pragma unroll
        for (int f = 0; f < FPP; ++f) {
            v2f outp[2][S];
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
#pragma unroll
                for (int q = 0; q < S; ++q) outp[h][q] = big_split_power(ta[q], tb[S - 1 - q], sw[h][q], eps2);
            }
            // the bins k = 256 q of the column k' = 0 (every lane computes them; lanes 0 .. S store one each)
            v2f t0[S];
#pragma unroll
            for (int rr = 0; rr < S; ++rr) t0[rr] = z0[S * f + rr];
            big_combine<S>(t0);       // t0[q] = Z[256 q]
            const v2f E = pk_lo_pm_hi(t0[0], t0[0]);                   // X[0] = 2 (re + im), X[C] = 2 (re - im) (Z arrives halved)
            const v2f se = pk_fma_sc(pk_mul_s(E, v2f{4.f, 4.f}), E, eps2);
            constexpr float R2 = 0.70710678118654752f;
            float spc = se.x;                                          // lane 0: bin 0
            if (S == 2) {
                const v2f s1 = big_split_power(t0[1], t0[1], v2f{0.f, -1.f}, eps2);            // W_1024^256 = -i: bin 256 (its own mirror)
                spc = lane == 1 ? s1.x : spc;
                spc = lane == 2 ? se.y : spc;
            } else {
                const v2f s1 = big_split_power(t0[1], t0[3 % S], v2f{R2, -R2}, eps2);           // W_2048^256: bins 256, 768
                const v2f s2 = big_split_power(t0[2 % S], t0[2 % S], v2f{0.f, -1.f}, eps2);     // W_2048^512 = -i: bin 512
                spc = lane == 1 ? s1.x : spc;
                spc = lane == 2 ? s2.x : spc;
                spc = lane == 3 ? s1.y : spc;
                spc = lane == 4 ? se.y : spc;
            }
            if (f < nvalid) {
                float* yr = y + (row0 + f) * K;
#pragma unroll
                for (int q = 0; q < S; ++q) {
                    // bins (2l+1, 2l+2) + 256 q and their mirrors C - k: two 8-byte stores, 512 consecutive bytes per instruction
                    // (lane 63 writes the pair around k' = 128 from both sides, the same values)
                    *reinterpret_cast<v2f_u4*>(yr + kk[0] + 256 * q) = v2f{outp[0][q].x, outp[1][q].x};
                    *reinterpret_cast<v2f_u4*>(yr + C - kk[1] - 256 * q) = v2f{outp[1][q].y, outp[0][q].y};
                }
                if (lane <= S) yr[256 * lane] = spc;
            }
        }
    }
}

}  // namespace dsa
