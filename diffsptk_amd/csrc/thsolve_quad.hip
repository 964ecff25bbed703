// Batched Toeplitz-plus-Hankel solve for orders up to 55 on the float32 matrix instruction -- the order-24 scheme of
// csrc/mcep_mfma.hip (thsolve_quad24_kernel / the elimination inside the tuned mel-cepstral kernels) as a template over the size.
//
//   g:(F,n) = solve(T(p) + H(q), r - sub)  (+ add)          mcep.py:216-222, mgcep.py:226-229 (torch.linalg.solve there)
//
// What it replaces: th_solve_fwd_kernel (csrc/mgc.hip) -- one wave per system, one row per lane, row pivoting, the pivot row broadcast
// lane by lane through scalar registers -- took 174 us per 12 800 systems of order 50 (52 % of a Newton step of the mel-cepstral
// analysis at the 48 kHz set-ups, profiles/r04_48k_kernel_trace_v1.txt).
//
// Mapping (gfx950): the four lanes of a QUAD share a system, a wave solves 16 systems at once.  The augmented matrix
// [A | rhs] lives in registers as 4 x 4 blocks -- block (rg, cg), cg >= rg: register i = row 4 rg + i, lane gs = column 4 cg + gs
// -- upper triangle only (A is symmetric).  Step k: the pivot row scaled by -1 / a_kk is, by symmetry, the column of multipliers,
// already on the lanes that own those ROWS; one v_mfma_f32_4x4x1_16b_f32 (sixteen independent 4 x 4 outer products, one per quad)
// updates block (rg, cg) of all 16 systems: A = slot rg of the multipliers, B = slot cg of the unscaled pivot row.  No broadcast, no
// LDS, no pivot search.  The right-hand side rides as column n.  Back substitution: row sums per lane, two quad rotations, one
// multiply by the pivot's reciprocal.
// No pivoting: sound for the positive definite systems of the analysis; a system whose elimination meets a non-positive or
// non-finite pivot is solved again inside the launch by the whole wave with row pivoting (th_solve_reg.h).
#include <cstdlib>
#include <utility>

#include "common.h"
#include "th_solve_reg.h"
#include "thsolve_tq.h"   // namespace dsa::tq: the device templates and the two kernels

namespace dsa {

template <int NG, int NMIN>
static int thsolve_quadn_launch(const void* p, int ldp, const void* q, int ldq, const void* r, int ldr, const void* sub, const void* add,
                                int64_t F, int n, void* g, hipStream_t st)
{
    constexpr int REC = ((16 * NG + 3 - 4 + 31) / 32) * 32 + 4;   // >= 16 NG + 3 and = 4 (mod 32): see the octet kernel (rounds 3-4: 16 NG + 3, odd)
    const int lds_bytes = 4 * 16 * REC * (int)sizeof(float);
    static std::atomic<uint64_t> attr{0};
    if (lds_bytes > 48 * 1024 && !ensure_dynamic_lds((const void*)tq::thsolve_quadn_kernel<NG, NMIN>, lds_bytes, attr))
        return fail(DSA_ERR_LAUNCH, "thsolve_quad: cannot reserve LDS%s");
    long blocks = ((F + 15) / 16 + 3) / 4;
    if (blocks > 256) blocks = 256;   // one workgroup per CU (one wave per SIMD: the matrix takes up to 420 registers)
    hipLaunchKernelGGL((tq::thsolve_quadn_kernel<NG, NMIN>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, (const float*)p, ldp,
                       (const float*)q, ldq, (const float*)r, ldr, (const float*)sub, (const float*)add, (long)F, n, (float*)g);
    return check_launch("th_solve_quadn_fwd");
}

template <int NG, int NMIN>
static int thsolve_octn_launch(const void* p, int ldp, const void* q, int ldq, const void* r, int ldr, const void* sub, const void* add,
                               int64_t F, int n, void* g, hipStream_t st)
{
#ifdef TQ_OCT_REC_ODD
    constexpr int NCP = (NG + 1) / 2, REC = ((4 * NG + 8 * NCP - 1) + 7 + 8 * NCP + 4 * NG) | 1;
#else
    constexpr int NCP = (NG + 1) / 2, REC = ((((4 * NG + 8 * NCP - 1) + 7 + 8 * NCP + 4 * NG) - 8 + 31) / 32) * 32 + 8;   // as in the kernel
#endif
    const int lds_bytes = 4 * 8 * REC * (int)sizeof(float);
    long blocks = ((F + 7) / 8 + 3) / 4;
    if (blocks > 512) blocks = 512;   // two workgroups per CU
    hipLaunchKernelGGL((tq::thsolve_octn_kernel<NG, NMIN>), dim3((unsigned)blocks), dim3(256), lds_bytes, st, (const float*)p, ldp,
                       (const float*)q, ldq, (const float*)r, ldr, (const float*)sub, (const float*)add, (long)F, n, (float*)g);
    return check_launch("th_solve_octn_fwd");
}

// float32, 2 <= n <= 55.  p:(F, n) row stride ldp, q:(F, 2n-1) stride ldq, r:(F, n) stride ldr; sub: NULL or (n), subtracted from
// every right-hand side; add: NULL or (F, n) contiguous, added to the solution; g:(F, n) contiguous.
int thsolve_quadn_fwd(const void* p, int ldp, const void* q, int ldq, const void* r, int ldr, const void* sub, const void* add, int64_t F,
                      int n, void* g, hipStream_t st)
{
    if (n < 2) return fail(DSA_ERR_UNSUPPORTED, "thsolve_quad: order below 2%s");
    // orders from 36 on: eight systems per wave (an octet each); below: sixteen (a quad each) -- below 36 the octets measured slower
    // (19.9 against 17.2 us at order 35, 15.7 against 13.4 at 25), from 36 on the quads spill (57 / 271 / 419 registers at
    // <11,36> / <13,44> / <14,52>: those instantiations left the build in round 6 together with their A/B switch).  The choice
    // depends on the order alone: a system's rounding never depends on how many systems travel with it.
    if (n >= 36) {
        if (n <= 43) return thsolve_octn_launch<11, 36>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
        if (n <= 51) return thsolve_octn_launch<13, 44>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
        if (n <= 55) return thsolve_octn_launch<14, 52>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
    }
    if (n <= 27) return thsolve_quadn_launch<7, 2>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
    if (n <= 35) return thsolve_quadn_launch<9, 28>(p, ldp, q, ldq, r, ldr, sub, add, F, n, g, st);
    return fail(DSA_ERR_UNSUPPORTED, "thsolve_quad: order above 55%s");
}

}  // namespace dsa
