"""The last step of a rocprofv3 --kernel-trace CSV as a timeline: start offset, duration, gap to the previous kernel's end, short name —
to see what a chain of small kernels (the decoder's layers) really costs.  python tools/trace_timeline.py trace.csv --last-ms 23.8 [--grep decoder_head]"""
import argparse, csv, re
ap = argparse.ArgumentParser()
ap.add_argument("trace"); ap.add_argument("--last-ms", type=float, required=True); ap.add_argument("--out", default=None)
a = ap.parse_args()
csv.field_size_limit(1 << 30)
rows = []
for r in csv.DictReader(open(a.trace)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]),
                 r.get("Queue_Id", "?")))
rows.sort()
t_end = max(r[1] for r in rows)
t0 = t_end - int(a.last_ms * 1e6)
rows = [r for r in rows if r[0] >= t0]
out = open(a.out, "w") if a.out else None
prev_end = rows[0][0]
gaps = 0
queues = {q: i for i, q in enumerate(sorted({r[3] for r in rows}))}       # queue ids renumbered 0.. (two HIP streams = two queues)
both = 0
for s, e, n, q in rows:
    line = f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  q{queues[q]}  {n[:90]}"
    both += max(0, min(prev_end, e) - s)
    gaps += max(0, s - prev_end)
    prev_end = max(prev_end, e)
    (out.write(line + "\n") if out else print(line))
print(f"{len(rows)} kernels on {len(queues)} queue(s), idle gaps between them {gaps / 1e3:.1f} us, kernel time overlapping an earlier kernel {both / 1e3:.1f} us")
