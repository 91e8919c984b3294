#!/bin/bash
# tools/probes/mfma_variants.sh name:flags ... — timing variants of the matrix-core front end (kModeMfma): for every
# name:flags build noaa_apt_amd/libaptgpu_mfma_<name>.so = the probe library (`make probe-lib`: APTGPU_DEBUG_SKIP) with the
# 48 kHz f32 MFMA instantiation rebuilt under those flags (e.g. w5:-DAPT_MFMA_WAVES=5).  Run on the GPU box:
#   APTGPU_PROBE_LIB=noaa_apt_amd/libaptgpu_mfma_<name>.so APTGPU_DEBUG_SKIP=7 python tools/sweep.py --configs fast:16:3
cd "$(dirname "$0")/../../noaa_apt_amd/csrc" || exit 1
make -s -j8 probe-lib || exit 1
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wall -Wno-unused-result"
OBJS=$(ls *.o | grep -v -e '^apt_plan.o$' -e '^apt_kernels_fused_48k_mfma_f32.o$' -e '^probe_')
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $f -c apt_kernels_fused_48k_mfma_f32.hip -o /tmp/mfma_$n.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libaptgpu_mfma_$n.so $OBJS /tmp/mfma_$n.o || exit 1
  echo "built noaa_apt_amd/libaptgpu_mfma_$n.so"
done
