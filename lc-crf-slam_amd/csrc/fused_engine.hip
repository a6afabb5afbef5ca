// fused_engine.hip -- one workgroup per frame, lattice values resident in LDS.
// (placeholder until the fused kernel lands: reports "not supported" so the streaming
// engine is used; the streaming engine is complete on its own)
#include "engine.h"

namespace lccrf {

bool fused_supported(const CrfDev &, const KernelDev *, const int *, size_t *lds_bytes)
{
    if (lds_bytes) *lds_bytes = 0;
    return false;
}

void launch_inference_fused(const CrfDev &, const KernelDev *, const int *, int, int, float, size_t, hipStream_t) {}

}  // namespace lccrf
