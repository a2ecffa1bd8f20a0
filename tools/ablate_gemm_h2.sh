#!/bin/bash
# Builds diagnostic copies of libpd_hip.so whose gemm_f16x2.hip is compiled with -DPD_ABL=<bits> (see that file) into build/abl/,
# for tools/ablate_gemm_h2.py (which times them on the GPU box).  Usage (here, before gpurun): tools/ablate_gemm_h2.sh 0 1 3 7 ...
set -eu
cd "$(dirname "$0")/../partdistillation_amd/csrc"
make -s -j8
mkdir -p ../../build/abl
OTHERS=$(ls *.o | grep -v '^gemm_f16x2.o$')
for k in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I../../include -I. -DPD_ABL=${k%%x*} $( [[ $k == *x* ]] && echo -DPD_ROWS_TWO_CHAINS ) -c gemm_f16x2.hip -o ../../build/abl/gemm_f16x2_$k.o &
done
wait
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/abl/libpd_abl_$k.so $OTHERS ../../build/abl/gemm_f16x2_$k.o
  rm ../../build/abl/gemm_f16x2_$k.o
done
ls -la ../../build/abl/
