#!/bin/bash
# Usage: tools/debug/kregs.sh <file.hip> <name filter>: VGPR / AGPR / scratch / LDS of every kernel of the file whose name contains the filter
cd /root/repo/partdistillation_amd/csrc
mkdir -p /root/repo/gpurun_out/tmp
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I../../include -I. -S --cuda-device-only $1 -o /root/repo/gpurun_out/tmp/k.s 2>/dev/null
python3 - "$2" <<'PY'
import re, sys
s = open('/root/repo/gpurun_out/tmp/k.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    if sys.argv[1] not in name: continue
    g = lambda k: re.search(k + r'\s+(\S+)', body).group(1)
    print(name[:70], 'vgpr', g('next_free_vgpr'), 'accum_off', g('accum_offset'), 'scratch', g('private_segment_fixed_size'))
PY
