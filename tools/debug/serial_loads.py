"""Usage: python tools/debug/serial_loads.py gpurun_out/tmp/asm/*.s — kernels whose loops wait for their vector-memory loads one at a time:
per loop (backward branch), the number of global / buffer loads and of `s_waitcnt vmcnt(0)` between them.  A loop with L >= 3 loads and almost
as many vmcnt(0) waits issues each load behind the previous one's data (the "serialised loads" lens of DESIGN round 6)."""
import re, sys
for path in sys.argv[1:]:
    s = open(path).read()
    for m in re.finditer(r'^(\S+):\s*; @\1\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
        name, body = m.group(1), m.group(2).split('\n')
        labels = {l.split(':')[0]: n for n, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
        loops = []
        for n, l in enumerate(body):
            b = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
            if b and b.group(1) in labels and labels[b.group(1)] < n:
                loops.append((labels[b.group(1)], n))
        worst = None
        for a, b in loops:
            seq = [('L' if re.search(r'\t(global|buffer|flat)_load', x) and 'lds' not in x else 'W') for x in body[a:b + 1]
                   if (re.search(r'\t(global|buffer|flat)_load', x) and 'lds' not in x) or 'vmcnt(0)' in x]
            L = seq.count('L')
            # waits that separate two loads
            sep = sum(1 for i in range(1, len(seq) - 1) if seq[i] == 'W' and 'L' in seq[:i] and 'L' in seq[i + 1:])
            if L >= 3 and sep >= 2 and (worst is None or sep > worst[1]):
                worst = (L, sep, b - a)
        if worst:
            print(f"{path.split('/')[-1][:-6]:16s} {name[:70]:70s} loads {worst[0]:3d}  separating vmcnt(0) {worst[1]:3d}  loop lines {worst[2]}")
