#!/bin/bash
# Usage: tools/debug/build_prev.sh [rev]: builds the library of <rev> (default HEAD) as partdistillation_amd/libpd_hip_prev.so for same-box A/B runs
#   tools/ab_bench.sh "prev|PD_LIB_PATH=/root/repo/partdistillation_amd/libpd_hip_prev.so" "new|"
REV=${1:-HEAD}
# (the scratch tree lives under gpurun_out/ — outside the package, which tests/test_product_cpu.py walks — only the library lands in the package
#  directory so that it travels to the GPU box; delete it after the A/B)
B=/root/repo/gpurun_out/prev_build
cd /root/repo/partdistillation_amd/csrc
rm -rf $B; mkdir -p $B/csrc $B/include
for f in $(git ls-tree --name-only $REV ./ | grep -v "/$"); do git show $REV:partdistillation_amd/csrc/$f > $B/csrc/$f 2>/dev/null; done
for f in $(git ls-tree --name-only $REV ../../include/); do git show $REV:include/$(basename $f) > $B/include/$(basename $f); done
cd $B/csrc
ls *.hip | xargs -P 8 -I{} sh -c 'hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I../include -I. -c {} -o {}.o 2>/dev/null || echo FAILED {}'
hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/partdistillation_amd/libpd_hip_prev.so *.o && ls -la /root/repo/partdistillation_amd/libpd_hip_prev.so
