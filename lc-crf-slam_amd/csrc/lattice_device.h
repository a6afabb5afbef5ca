// lattice_device.h -- device helpers shared by the streaming build and the fused small-frame build.
#pragma once

#include "engine.h"
#include "device_math.h"

namespace lccrf {

// Key of the lattice vertex that entry e = point*(d+1)+remainder touches, recomputed from the
// point record (keys are never stored).  ref: permutohedral_cpu.h:371-374.
template <int D>
__device__ __forceinline__ void load_entry_key(const KernelDev &kd, int f, int e, int16_t (&key)[D])
{
    constexpr int D1 = D + 1;
    const int pt = e / D1, rem = e - pt * D1;
    const int16_t *r0 = kd.rem0 + ((size_t)f * kd.maxNpad + pt) * D;
    const uint8_t *rk = kd.rank + ((size_t)f * kd.maxNpad + pt) * D;
#pragma unroll
    for (int i = 0; i < D; ++i) key[i] = vertex_coord<D>(r0[i], rk[i], rem);
}

}  // namespace lccrf
