"""Usage: python tools/debug/scratch_where.py <file.s> <kernel-name substring>: every scratch (spill) instruction of the kernel and the innermost loop
(backward branch range, in lines) it sits in — a spill in the epilogue is harmless, one in the main loop is not."""
import re
import sys

s = open(sys.argv[1]).read()
for m in re.finditer(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    body = m.group(2).split('\n')
    labels = {l.split(':')[0]: n for n, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
    loops = []
    for n, l in enumerate(body):
        b = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if b and b.group(1) in labels and labels[b.group(1)] < n:
            loops.append((labels[b.group(1)], n))
    sc = [n for n, l in enumerate(body) if 'scratch_' in l]
    print(m.group(1)[:100], '| lines', len(body), '| scratch ops', len(sc), '| loops', sorted(set(loops))[:12])
    for n in sc:
        inl = [(a, b) for a, b in loops if a <= n <= b]
        inner = min(inl, key=lambda ab: ab[1] - ab[0]) if inl else None
        print('   line', n, body[n].strip()[:80], '| innermost loop', inner)
