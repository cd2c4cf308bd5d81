#!/usr/bin/env python
"""Concurrency summary of a rocprofv3 --kernel-trace CSV over the decode loop: how much of the wall span has 0 / 1 / >= 2 kernels
in flight, per hardware queue and per kernel family.  `python scripts/trace_overlap.py <kernel_trace.csv> [label]` prints a few
lines meant to be kept under profiles/ (the raw trace is not)."""
import csv
import sys
from collections import Counter, defaultdict


def family(name):
    for pat, fam in (("decode_attn", "decode_attn"), ("rownorm", "rownorm"), ("EpiQkvDecode", "gemm_qkv_decode"), ("gemm_glds", "gemm"),
                     ("sample_kernel", "sample"), ("ar_advance", "advance")):
        if pat in name:
            return fam
    return "other"


def main():
    path = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else path
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    rows.sort()
    att = [i for i, r in enumerate(rows) if "decode_attn" in r[3]]
    if not att:
        print("trace %s: no decode attention kernels" % label)
        return
    # the LAST contiguous generation of the run (earlier ones are warm-up): the final 60 % of the attention launches
    lo = att[int(len(att) * 0.4)]
    hi = att[-1]
    win = rows[lo:hi + 1]
    t0, t1 = win[0][0], max(r[1] for r in win)
    ev = []
    for s, e, q, n in win:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, hist = 0, t0, Counter()
    for t, d in ev:
        hist[min(depth, 3)] += t - last
        last = t
        depth += d
    span = t1 - t0
    busy = sum(e - s for s, e, _, _ in win)
    per_q = Counter(q for _, _, q, _ in win)
    fam_t, fam_n = defaultdict(int), Counter()
    for s, e, q, n in win:
        fam_t[family(n)] += e - s
        fam_n[family(n)] += 1
    nattn = sum(1 for r in win if "decode_attn" in r[3])
    print("trace %s: %d kernels over %.2f ms (%d attention launches); sum of kernel durations %.2f ms = %.2f x span" %
          (label, len(win), span / 1e6, nattn, busy / 1e6, busy / span))
    print("  in flight: 0 kernels %.1f %%, 1 kernel %.1f %%, 2 kernels %.1f %%, >= 3 kernels %.1f %% of the span" %
          tuple(100.0 * hist[i] / span for i in range(4)))
    print("  queues: " + ", ".join("%s: %d" % kv for kv in sorted(per_q.items())))
    print("  per family: " + ", ".join("%s %d x %.2f us" % (k, fam_n[k], fam_t[k] / fam_n[k] / 1e3) for k in sorted(fam_t)))


if __name__ == "__main__":
    main()
