// Tuned mel-cepstral analysis kernel for gfx950 (float32, f32 MFMA) -- placeholder until the
// MFMA kernel lands; reports "unsupported" so the dispatcher uses the generic kernels.
#include "common.h"

namespace dsa {
int mcep_mfma_supported(int, int, int) { return 0; }
int mcep_mfma_fwd(const void*, int64_t, int, int, int, const void*, const void*, const void*, const void*,
                  void*, void*, hipStream_t)
{
    return fail(DSA_ERR_UNSUPPORTED, "mcep: tuned kernel not built%s");
}
}  // namespace dsa
