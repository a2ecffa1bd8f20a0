#!/bin/bash
# Usage: tools/debug/build_prev.sh [rev]: builds the library of <rev> (default HEAD) as partdistillation_amd/libpd_hip_prev.so for same-box A/B runs
#   tools/ab_bench.sh "prev|PD_LIB_PATH=/root/repo/partdistillation_amd/libpd_hip_prev.so" "new|"
REV=${1:-HEAD}
cd /root/repo/partdistillation_amd/csrc
rm -rf .prev; mkdir -p .prev/csrc
for f in $(git ls-tree --name-only $REV ./ | grep -v "/$"); do git show $REV:partdistillation_amd/csrc/$f > .prev/csrc/$f 2>/dev/null; done
mkdir -p .prev/include; for f in $(git ls-tree --name-only $REV ../../include/); do git show $REV:include/$(basename $f) > .prev/include/$(basename $f); done
cd .prev/csrc
ls *.hip | xargs -P 8 -I{} sh -c 'hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I../include -I. -c {} -o {}.o 2>/dev/null || echo FAILED {}'
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../../libpd_hip_prev.so *.o && ls -la ../../../libpd_hip_prev.so
