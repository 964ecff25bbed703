// The row-pivoted Toeplitz-plus-Hankel solve of ONE system by ONE wave, rows in registers (shared by csrc/mgc.hip -- the general
// kernel -- and the quad-layout kernels of csrc/mcep_mfma.hip / csrc/thsolve_quad.hip, which fall back to it for the systems
// their unpivoted elimination gives up on).
#pragma once

namespace dsa {

// Register version for n <= NMAX (NMAX = 24 or 32: cep_order 24 is the usual size): lane i holds row i of the system and
// its right-hand side in registers, the pivot row reaches the other lanes through v_readlane (the pivot lane is uniform),
// everything is statically indexed (both loops unrolled).  Same pivot rule as the LDS version below, which remains for
// larger systems: that one spends ~80 cycles per element on dependent LDS round trips (0.53 ms per 51 200 frames of 24 x 24,
// 43 % of a mel-generalized analysis), this one ~6 k cycles per frame.
// Returns in (col, sol): lane i < n was the pivot row of column `col`, and x[col] = sol.
template <typename T>
__device__ __forceinline__ T th_readlane(T v, int src)
{
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
    } else {
        const long long b = __builtin_bit_cast(long long, v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src);
        return __builtin_bit_cast(T, (long long)(((unsigned long long)hi << 32) | lo));
    }
}
template <typename T, int NMAX>
__device__ __forceinline__ void th_solve_reg(const T* ps, const T* qs, T rhs, int n, int lane, int& col, T& sol)
{
    T row[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        const int d = lane > j ? lane - j : j - lane;
        row[j] = (lane < n && j < n) ? ps[d] + qs[lane + j] : T(0);
    }
    if (lane >= n) rhs = T(0);
    bool used = lane >= n;
    T piv = T(1);
    col = 0;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        if (k < n) {   // uniform
            // pivot = the unused row with the largest |a_ik|: one unsigned key per lane (magnitude bits with the lane in the
            // low 6 bits: ties and near-ties go to the lowest lane), maximum over the wave by DPP shifts -- cross-lane
            // shuffles through the LDS crossbar cost 12 dependent round trips per step here
            const float magf = (float)(row[k] < T(0) ? -row[k] : row[k]);
            unsigned key = used ? 0u : ((__builtin_bit_cast(unsigned, magf) & 0xffffffc0u) | (unsigned)(63 - lane));
#define DSA_TH_MAX(CTRL, RM)                                                                                            \
    {                                                                                                                  \
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, CTRL, RM, 0xf, false);                   \
        key = o > key ? o : key;                                                                                       \
    }
            DSA_TH_MAX(0x111, 0xf) DSA_TH_MAX(0x112, 0xf) DSA_TH_MAX(0x114, 0xf) DSA_TH_MAX(0x118, 0xf)   // row_shr:1, 2, 4, 8
            DSA_TH_MAX(0x142, 0xa) DSA_TH_MAX(0x143, 0xc)                                                 // row_bcast:15, :31
#undef DSA_TH_MAX
            const int p = 63 - (int)(__builtin_amdgcn_readlane((int)key, 63) & 63);
            const T pk = th_readlane(row[k], p);
            const T fac = (lane != p) ? row[k] / pk : T(0);
            if (lane == p) {
                used = true;
                col = k;
                piv = row[k];
            }
#pragma unroll
            for (int j = k + 1; j < NMAX; ++j) row[j] -= fac * th_readlane(row[j], p);
            rhs -= fac * th_readlane(rhs, p);
        }
    }
    sol = rhs / piv;
}

}  // namespace dsa
