#!/bin/bash
# tools/isa.sh <tu-name>... : device-only assembly of csrc/apt_kernels_<name>.hip into /tmp/isa/<name>.s
# and the register / spill figures of every kernel in it.  ISA_FLAGS=-DAPT_FUSED_MARKS=1 adds the stage marks
# tools/isa_budget.py --marks reads.
mkdir -p /tmp/isa
cd /root/repo/noaa_apt_amd/csrc || exit 1
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
    -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize $ISA_FLAGS --cuda-device-only -S -o /tmp/isa/$k.s apt_kernels_$k.hip 2>&1 | grep -E "error" -A5 &
done
wait
for k in "$@"; do
  echo "== $k"
  grep -E "^\s+\.(name|sgpr_spill_count|vgpr_count|vgpr_spill_count|private_segment_fixed_size):" /tmp/isa/$k.s \
    | sed -e 's/^\s*//' | paste - - - - - | sed -e 's/_ZN3apt3gpu12_GLOBAL__N_1//' | cut -c1-60,110-
done
