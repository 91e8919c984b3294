// apt_kernels_fused_any_256x4.hip — the 256-thread, 4-outputs-per-thread launch shape of k_fused_any (tiles of 1024 work
// samples: input rates whose 2048-sample tile does not fit the LDS).
#include "apt_kernels_fused_any_impl.hpp"

namespace apt::gpu {

void fused_any_launch_256x4(APT_ANY_SHAPE_ARGS)
{
    launch_any_shape<256, 4>(s, call, d_slots, max_w, pcm16, table, h2, h2p, cosphi2, sinphi, inv_sinphi, want_gm, g, lds, prof);
}

}  // namespace apt::gpu
