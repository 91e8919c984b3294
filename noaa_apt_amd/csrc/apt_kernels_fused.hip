// apt_kernels_fused.hip — fused, specialised gfx950 kernels.
#include "apt_kernels.hpp"

namespace apt::gpu {
}  // namespace apt::gpu
