// EXPERIMENT, not part of the build: one launch for the spectrum arithmetic of a mel-generalized Newton step AND its five
// row products (mgcep.py:199-221).  Correct (matched dsa_mgcep_spectra + five dsa_freqt_fwd to 2e-5 / 1e-11), but measured
// SLOWER than the chain it replaces: 8.6-11 ms per analysis of 51 200 frames against 6.1 ms -- with the matrix entries read
// inside the bin loop every bin costs an L2 round trip, and held in registers (this version) the 104 + 64 values per thread
// spill at two waves per SIMD.  A version with Cr / Ci in LDS and three waves per SIMD was not built.
// The same arithmetic AND the five row products it feeds, in one launch (the spectra never reach memory):
//   pt = pp Pr   qt = qscale (qq (X^2 - Y^2) Qr + qq 2XY Qi)   r = pp X Rr + pp Y Ri          (mgcep.py:210-221)
// with Pr:(K, nP), Qr / Qi:(K, nQ), Rr / Ri:(K, nR) row-major.  Persistent workgroups of eight waves over tiles of eight
// frames.  Phase 1 is mgcep_spectra_kernel's (a thread per bin, its Cr / Ci entries in registers), writing the tile's five
// spectra to LDS.  In phase 2 a thread owns one output column for a fifth of the bins: its matrix entries stay in REGISTERS
// for the whole launch (read once: as loads inside the loop every bin cost an L2 round trip), the bins' values are LDS
// broadcasts shared by the eight frames, and the partial sums of the bin ranges meet in LDS.  As five matrix-core launches
// plus the spectra kernel this step moved 1.3 GB per 51 200 frames and took 0.38 ms.
constexpr int kStFrames = 8, kStThreads = 512, kStPer = 52;   // bins per phase-2 thread (<= kStPer): nP + nQ + nR <= 102 for K = 257
template <typename T, int MT>
__global__ __launch_bounds__(kStThreads, 2) void mgcep_step_kernel(const T* __restrict__ x, const T* __restrict__ b1, long F, int K, int M,
                                                                  const T* __restrict__ Cr, const T* __restrict__ Ci, T gamma,
                                                                  const T* __restrict__ Pr, int nP, const T* __restrict__ Qr,
                                                                  const T* __restrict__ Qi, int nQ, const T* __restrict__ Rr,
                                                                  const T* __restrict__ Ri, int nR, T qscale, T* __restrict__ pt,
                                                                  T* __restrict__ qt, T* __restrict__ r, int G)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ms_smem[];
    T* bs = reinterpret_cast<T*>(ms_smem);                     // [kStFrames][MT]
    T* S = bs + kStFrames * MT;                                // [5][K][kStFrames]
    T* part = S + (size_t)5 * K * kStFrames;                   // [G][kStFrames][NCOL]
    const int NCOL = nP + nQ + nR;
    const T ex = T(-1) / gamma - T(1);
    const int k = threadIdx.x;
    T cr[MT], ci[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        cr[m] = (k < K && m < M) ? Cr[(long)(m + 1) * K + k] : T(0);
        ci[m] = (k < K && m < M) ? Ci[(long)(m + 1) * K + k] : T(0);
    }
    // phase-2 role of this thread: column `col` of the stacked outputs, bins [kb, kb + kStPer) clipped to its range
    const int grp = threadIdx.x / NCOL, col = threadIdx.x - grp * NCOL;
    const bool worker = grp < G;
    const int per = (K + G - 1) / G;
    const int kb = worker ? grp * per : 0, ke = worker ? (kb + per < K ? kb + per : K) : 0;
    const int kind = col < nP ? 0 : (col < nP + nQ ? 1 : 2);
    const int cc = kind == 0 ? col : (kind == 1 ? col - nP : col - nP - nQ);
    const T* A0 = kind == 0 ? Pr : (kind == 1 ? Qr : Rr);
    const T* A1 = kind == 1 ? Qi : Ri;
    const int ld = kind == 0 ? nP : (kind == 1 ? nQ : nR);
    const T* S0 = S + (size_t)(kind == 0 ? 0 : (kind == 1 ? 1 : 3)) * K * kStFrames;
    const T* S1 = S + (size_t)(kind == 1 ? 2 : 4) * K * kStFrames;
    T a0[kStPer], a1[kStPer];
#pragma unroll
    for (int i = 0; i < kStPer; ++i) {
        const bool in = kb + i < ke;
        a0[i] = in ? A0[(long)(kb + i) * ld + cc] : T(0);
        a1[i] = (in && kind != 0) ? A1[(long)(kb + i) * ld + cc] : T(0);
    }
    const long ntile = (F + kStFrames - 1) / kStFrames;
    for (long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long f0 = tile * kStFrames;
        const int nf = (int)((F - f0) < kStFrames ? (F - f0) : kStFrames);
        __syncthreads();   // the previous tile's partial sums have been read
        for (int i = threadIdx.x; i < kStFrames * MT; i += blockDim.x) {
            const int fi = i / MT, m = i - fi * MT;
            bs[i] = (fi < nf && m < M) ? b1[(f0 + fi) * M + m] : T(0);
        }
        __syncthreads();
        if (k < K) {
#pragma unroll
            for (int fi = 0; fi < kStFrames; ++fi) {
                T re = 0, im = 0;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const T b = bs[fi * MT + m];
                    re += b * cr[m];
                    im += b * ci[m];
                }
                const T X = T(1) + gamma * re, Y = gamma * im;
                const T XX = X * X, YY = Y * Y, D = XX + YY;
                T dp;
                if constexpr (sizeof(T) == 4) dp = __builtin_amdgcn_exp2f(ex * __builtin_amdgcn_logf(D));
                else dp = dsa_pow(D, ex);
                const T xv = fi < nf ? x[(f0 + fi) * K + k] : T(0);
                const T pp = xv * dp;
                const T qq = pp / D;
                S[((size_t)0 * K + k) * kStFrames + fi] = pp;
                S[((size_t)1 * K + k) * kStFrames + fi] = qq * (XX - YY);
                S[((size_t)2 * K + k) * kStFrames + fi] = qq * (T(2) * X * Y);
                S[((size_t)3 * K + k) * kStFrames + fi] = pp * X;
                S[((size_t)4 * K + k) * kStFrames + fi] = pp * Y;
            }
        }
        __syncthreads();
        if (worker) {
            T acc[kStFrames];
#pragma unroll
            for (int fi = 0; fi < kStFrames; ++fi) acc[fi] = T(0);
#pragma unroll
            for (int i = 0; i < kStPer; ++i) {
                // (entries past the thread's range are zero; the index is clamped into the tile)
                const int kk = kb + i < K ? kb + i : K - 1;
#pragma unroll
                for (int fi = 0; fi < kStFrames; ++fi)
                    acc[fi] += S0[(size_t)kk * kStFrames + fi] * a0[i] + S1[(size_t)kk * kStFrames + fi] * a1[i];
            }
#pragma unroll
            for (int fi = 0; fi < kStFrames; ++fi) part[((size_t)grp * kStFrames + fi) * NCOL + col] = acc[fi];
        }
        __syncthreads();
        for (int o = threadIdx.x; o < kStFrames * NCOL; o += blockDim.x) {
            const int fi = o / NCOL, c = o - fi * NCOL;
            if (fi >= nf) continue;
            T v = part[(size_t)fi * NCOL + c];
            for (int gi = 1; gi < G; ++gi) v += part[((size_t)gi * kStFrames + fi) * NCOL + c];
            const long f = f0 + fi;
            if (c < nP) pt[f * nP + c] = v;
            else if (c < nP + nQ) qt[f * nQ + (c - nP)] = v * qscale;
            else r[f * nR + (c - nP - nQ)] = v;
        }
    }
}

template <typename T>
static int mgcep_step_launch(const void* x, const void* b1, int64_t F, int K, int M, const void* Cr, const void* Ci, double gamma,
                             const void* Pr, int nP, const void* Qr, const void* Qi, int nQ, const void* Rr, const void* Ri, int nR,
                             double qscale, void* pt, void* qt, void* r, hipStream_t st)
{
    const int NCOL = nP + nQ + nR;
    const int G = NCOL > 0 ? kStThreads / NCOL : 0;
    if (K > kStThreads || G < 1 || (K + G - 1) / G > kStPer || M > 32)
        return fail(DSA_ERR_UNSUPPORTED, "mgcep_step: sizes beyond the fused kernel (fft_length 512, cep_order <= 25)%s");
    const int MTv = M <= 16 ? 16 : 32;
    const size_t lds = sizeof(T) * ((size_t)kStFrames * MTv + (size_t)5 * K * kStFrames + (size_t)G * kStFrames * NCOL);
    const long ntile = (F + kStFrames - 1) / kStFrames;
    const unsigned grid = (unsigned)(ntile < 512 ? ntile : 512);
#define DSA_MST_LAUNCH(MT)                                                                                                \
    do {                                                                                                                  \
        static std::atomic<uint64_t> attr_devices{0};                                                                     \
        if (!ensure_dynamic_lds((const void*)mgcep_step_kernel<T, MT>, 150 * 1024, attr_devices))                          \
            return fail(DSA_ERR_LAUNCH, "mgcep_step: cannot reserve LDS%s");                                               \
        hipLaunchKernelGGL((mgcep_step_kernel<T, MT>), dim3(grid), dim3(kStThreads), lds, st, (const T*)x, (const T*)b1, (long)F, K, M, \
                           (const T*)Cr, (const T*)Ci, (T)gamma, (const T*)Pr, nP, (const T*)Qr, (const T*)Qi, nQ,          \
                           (const T*)Rr, (const T*)Ri, nR, (T)qscale, (T*)pt, (T*)qt, (T*)r, G);                          \
    } while (0)
    if (M <= 16) DSA_MST_LAUNCH(16);
    else DSA_MST_LAUNCH(32);
#undef DSA_MST_LAUNCH
    return check_launch("mgcep_step");
}

