#!/bin/bash
# tools/probes/halves_variant.sh: noaa_apt_amd/libaptgpu_nohalves.so = the library with the PHASE stage 1 of one branch per
# thread built the old way (-DAPT_PHASE_HALVES=0: sixteen windows' input in LDS at once), for the same-box A/B of
# tools/probes/exp_halves.sh (profiles/r06_phase_halves_ab.txt).  Run after `make -C noaa_apt_amd/csrc`.
set -e
cd "$(dirname "$0")/../../noaa_apt_amd/csrc"
V=/tmp/aptgpu_nohalves; mkdir -p $V
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wall -Wno-unused-result -DAPT_PHASE_HALVES=0"
TUS="apt_kernels_fused apt_kernels_fused_phase_std_f32 apt_kernels_fused_phase_std_i16 apt_kernels_fused_phase_std_fast_f32 apt_kernels_fused_phase_std_fast_i16 apt_kernels_fused_phase_fastp_f32 apt_kernels_fused_phase_fastp_i16 apt_kernels_fused_phase_fastp_fast_f32 apt_kernels_fused_phase_fastp_fast_i16"
for t in $TUS; do /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -c $t.hip -o $V/$t.o & done; wait
OBJS=""
for src in $(grep "^SRCS" Makefile | head -1 | cut -d= -f2); do
  b=${src%.*}; o=$b.o
  case " $TUS " in *" $b "*) OBJS="$OBJS $V/$o";; *) OBJS="$OBJS $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libaptgpu_nohalves.so $OBJS
ls -la ../libaptgpu_nohalves.so
