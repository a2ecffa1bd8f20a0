"""Usage: python tools/debug/loop_mix.py <file.s> <kernel-name substring>: for every kernel that matches, the largest loop that contains an MFMA —
instructions per iteration, MFMAs, LDS reads, global loads, waits.  (A K-step of 700 instructions around 8 MFMAs is bound by instruction issue.)"""
import collections, re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    body = m.group(2).split('\n')
    labels = {l.split(':')[0]: n for n, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)}
    loops = []
    for n, l in enumerate(body):
        b = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if b and b.group(1) in labels and labels[b.group(1)] < n:
            loops.append((labels[b.group(1)], n))
    cand = [(b - a, a, b) for a, b in loops if any('v_mfma' in x for x in body[a:b])]
    if not cand:
        continue
    cand.sort()
    out = []
    for _, a, b in (cand[0], cand[-1]):
        c = collections.Counter(l.split()[0] for l in body[a:b + 1] if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';')))
        mf = sum(v for k, v in c.items() if k.startswith('v_mfma'))
        out.append(f"[{sum(c.values())} instr, {mf} mfma, {sum(v for k, v in c.items() if k.startswith('ds_read'))} ds_read, "
                   f"{sum(v for k, v in c.items() if 'load' in k and not k.startswith(('ds_', 's_')))} vmem loads, {c['s_waitcnt']} waits, {sum(c.values()) / max(mf, 1):.0f} instr/mfma]")
    print(m.group(1)[:64], 'innermost', out[0], 'outermost', out[1])
