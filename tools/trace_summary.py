"""Summarise a rocprofv3 --kernel-trace CSV over the steady-state window only (the last --last-ms milliseconds of
device activity), so MIOpen's warm-up algorithm search does not pollute the per-kernel statistics.
Writes a small CSV (name, calls, total_us, avg_us, pct) and prints the top entries."""
import argparse
import csv
import collections
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(_ZN\w{0,160})", name)
    return name[:200]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--last-ms", type=float, required=True)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--list", default=None, help="regex: list every launch of the LAST step whose kernel name matches, in issue order")
    a = ap.parse_args()
    csv.field_size_limit(1 << 30)
    rows = []
    with open(a.trace) as f:
        rd = csv.DictReader(f)
        for r in rd:
            try:
                wg = 1
                for ax in "XYZ":
                    wg *= max(1, int(r["Grid_Size_" + ax]) // max(1, int(r["Workgroup_Size_" + ax])))
            except (KeyError, ValueError):
                wg = 0
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], wg))
    t_end = max(r[1] for r in rows)
    wgs = collections.defaultdict(int)
    t0 = t_end - int(a.last_ms * 1e6)
    agg = collections.defaultdict(lambda: [0, 0])
    busy = 0
    for s, e, n, wg in rows:
        if s >= t0:
            wgs[n] += wg
            agg[n][0] += 1
            agg[n][1] += e - s
            busy += e - s
    if a.list:
        t1 = t_end - int(a.last_ms * 1e6 / a.steps)
        pat = re.compile(a.list)
        for s, e, n, wg in sorted(rows):
            if s >= t1 and pat.search(n):
                print(f"  +{(s - t1) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  wg {wg:6d}  {short(n)[:90]}")
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    print(f"window {a.last_ms:.1f} ms, kernel-busy {busy / 1e6:.2f} ms ({busy / 1e6 / a.steps:.2f} ms/step over {a.steps} steps), "
          f"{sum(v[0] for v in agg.values())} launches ({sum(v[0] for v in agg.values()) / a.steps:.0f}/step), {len(items)} distinct kernels")
    if a.out:
        with open(a.out, "w") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "calls_per_step", "total_us", "us_per_step", "avg_us", "pct_of_busy", "avg_workgroups"])
            for n, (c, t) in items:
                w.writerow([short(n), c, f"{c / a.steps:.1f}", f"{t / 1e3:.1f}", f"{t / 1e3 / a.steps:.1f}", f"{t / 1e3 / c:.2f}", f"{100 * t / busy:.2f}", f"{wgs[n] / c:.0f}"])
    for n, (c, t) in items[: a.top]:
        print(f"{t / 1e3 / a.steps:9.1f} us/step {100 * t / busy:5.1f}%  x{c / a.steps:6.1f}  avg {t / 1e3 / c:8.1f} us  wg {wgs[n] / c:7.0f}  {short(n)[:120]}")


if __name__ == "__main__":
    main()
