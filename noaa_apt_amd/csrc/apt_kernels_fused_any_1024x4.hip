// apt_kernels_fused_any_1024x4.hip — the 1024-thread, 4-outputs-per-thread launch shape of k_fused_any.
#include "apt_kernels_fused_any_impl.hpp"

namespace apt::gpu {

void fused_any_launch_1024x4(APT_ANY_SHAPE_ARGS)
{
    launch_any_shape<1024, 4>(s, call, d_slots, max_w, pcm16, table, h2, h2p, cosphi2, sinphi, inv_sinphi, want_gm, g, lds, prof);
}

}  // namespace apt::gpu
