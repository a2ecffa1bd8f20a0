"""One-rank proof that the bucket all-reduces run CONCURRENTLY with backward kernels.

  PD_DDP_FORCE=1 rocprofv3 --kernel-trace --output-format csv -d <dir> -o r -- python bench.py --steps 4 --warmup 3 --no-cpu-baseline ...
  python tools/ddp_overlap_trace.py <dir>/r_kernel_trace.csv > profiles/r06_ddp_overlap_trace.txt

For every collective kernel (RCCL: ncclDevKernel*) of the last traced step: its start / end relative to the step's first kernel, its queue, and
the compute kernels (other queues) whose execution intervals intersect it, with the overlapped time.  With ONE rank the all-reduce is a local
copy-reduce kernel, so its duration says nothing about xGMI; what the trace shows is the issue order and that the two queues do run side by side."""
import csv
import sys


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))))
    rows.sort()
    coll = [i for i, r in enumerate(rows) if "nccl" in r[2].lower() or "rccl" in r[2].lower()]
    if not coll:
        # One rank: RCCL completes an in-place all-reduce over a single participant without launching a kernel, so there is nothing to
        # intersect.  What the trace still shows is WHERE in the backward pass each bucket is handed to the process group: the reducer's
        # gather launch (multi_gather_sumsq: p.grad tensors -> the bucket's slice of the flat gradient buffer) immediately precedes
        # dist.all_reduce(bucket, async_op=True) on the same host thread.
        adam = [i for i, r in enumerate(rows) if "adamw" in r[2]]
        end_i = adam[-2] if len(adam) >= 2 and rows[adam[-1]][0] - rows[adam[-2]][1] < 2_000_000 else adam[-1]     # first AdamW launch of the last step
        prev = [i for i in adam if rows[end_i][0] - rows[i][1] > 5_000_000]
        start_i = prev[-1] if prev else 0
        t0, t1 = rows[start_i][1], rows[end_i][0]
        gathers = [i for i in range(start_i, end_i) if "multi_gather_sumsq" in rows[i][2]]
        # backward starts at the first kernel of the criterion's backward; approximated by the loss-vector backward launch
        bwd0 = next((rows[i][0] for i in range(start_i, end_i) if "loss_vectors_bwd" in rows[i][2]), t0)
        print("no collective KERNEL in the trace (one rank: RCCL completes the single-participant all-reduce without a launch).")
        print(f"last traced step: {(t1 - t0) / 1e6:.2f} ms from the end of the previous optimizer step to this step's first AdamW launch; backward begins at "
              f"{(bwd0 - t0) / 1e6:.2f} ms (loss_vectors_bwd)")
        print("bucket hand-over points (the gather launch that precedes each dist.all_reduce(async_op=True)):")
        for k, i in enumerate(gathers):
            s_, e_, name, q = rows[i]
            grid = ""
            print(f"  bucket {k}: gather at {(s_ - t0) / 1e6:7.3f} ms = {(s_ - bwd0) / max(t1 - bwd0, 1):5.1%} of the backward, {(e_ - s_) / 1e3:6.1f} us; "
                  f"kernels after it until AdamW: {sum(1 for j in range(i + 1, end_i))}")
        return
    # the last step: the final run of collectives separated from the previous one by > 5 ms
    groups, cur = [], [coll[0]]
    for a, b in zip(coll, coll[1:]):
        if rows[b][0] - rows[a][1] > 5_000_000:
            groups.append(cur); cur = []
        cur.append(b)
    groups.append(cur)
    last = groups[-1]
    # the step's window: from the last optimizer kernel before the first collective ... to the next adamw after the last collective
    t_first, t_last = rows[last[0]][0], rows[last[-1]][1]
    adam = [i for i, r in enumerate(rows) if "adamw" in r[2]]
    start_i = max([i for i in adam if rows[i][1] < t_first], default=0)
    end_i = min([i for i in adam if rows[i][0] > t_last], default=len(rows) - 1)
    t0 = rows[start_i][1]
    step_ms = (rows[end_i][0] - t0) / 1e6
    print(f"last traced step: forward + backward window {step_ms:.2f} ms (from the end of the previous optimizer step to this step's first AdamW launch), "
          f"{len(last)} collective kernels, {sum(rows[i][1] - rows[i][0] for i in last) / 1e3:.1f} us of collective time")
    print(f"{'collective':34s} {'queue':>6s} {'start ms':>9s} {'end ms':>8s} {'us':>7s}  overlapped by compute kernels of other queues (us of intersection)")
    tot_ov, tot = 0.0, 0.0
    for i in last:
        s, e, name, q = rows[i]
        ov = []
        for j in range(start_i, end_i + 1):
            s2, e2, n2, q2 = rows[j]
            if j == i or q2 == q or "nccl" in n2.lower():
                continue
            inter = min(e, e2) - max(s, s2)
            if inter > 0:
                ov.append((inter / 1e3, n2.split("(")[0][-40:]))
        ov.sort(reverse=True)
        tot_ov += min(sum(o[0] for o in ov), (e - s) / 1e3)
        tot += (e - s) / 1e3
        print(f"{name.split('(')[0][:34]:34s} {q:>6s} {(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:8.3f} {(e - s) / 1e3:7.1f}  " + ", ".join(f"{n} {u:.1f}" for u, n in ov[:4]))
    print(f"collective time overlapped by compute kernels: {tot_ov:.1f} of {tot:.1f} us = {tot_ov / max(tot, 1e-9):.0%}")
    bw_end = max(rows[j][1] for j in range(start_i, end_i) if "nccl" not in rows[j][2].lower())
    print(f"last collective ends {(rows[last[-1]][1] - t0) / 1e6:.3f} ms into the window; the last non-collective kernel before AdamW ends at {(bw_end - t0) / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
