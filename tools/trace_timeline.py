#!/usr/bin/env python3
"""Reduces a rocprofv3 --kernel-trace CSV of bench.py to a per-step timeline summary: device busy time (union of kernel
intervals), time with two or more kernels running, idle gaps, and per-kernel-family busy time.  Usage:
  python tools/trace_timeline.py <dir with *_kernel_trace.csv> [--list]
--list additionally prints one whole step launch by launch: queue, start (us from the step's first launch), duration, idle time
of that queue in front of the launch, kernel name -- the main stream's chain and the gaps the fork/join events leave in it."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    folder = sys.argv[1]
    files = glob.glob(os.path.join(folder, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    # steps are delimited by the pack_pair kernel (first launch of a train step)
    starts = [i for i, r in enumerate(rows) if "pack_pair" in r[2]]
    if len(starts) < 4:
        raise SystemExit("not enough steps in the trace")
    lo, hi = starts[-4], starts[-1]           # three whole steps near the end (timed region / profile pass)
    seg = rows[lo:hi]
    t0, t1 = seg[0][0], rows[hi][0]
    nsteps = 3
    ev = []
    for s, e, _, _ in seg:
        ev.append((s, 1))
        ev.append((min(e, t1), -1))
    ev.sort()
    busy = over = 0
    depth, last = 0, t0
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            over += t - last
        depth += d
        last = t
    fam = defaultdict(float)
    for s, e, n, _ in seg:
        key = n.split("(")[0].split("<")[0].replace("void ", "")[:40]
        fam[key] += (e - s)
    span = t1 - t0
    print(f"steps {nsteps}: span {span / nsteps / 1e3:.1f} us/step, device busy {busy / nsteps / 1e3:.1f} us/step, "
          f">=2 kernels {over / nsteps / 1e3:.1f} us/step, idle {(span - busy) / nsteps / 1e3:.1f} us/step, "
          f"launches {len(seg) / nsteps:.0f}/step, sum of kernel times {sum(fam.values()) / nsteps / 1e3:.1f} us/step")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:25]:
        print(f"  {v / nsteps / 1e3:8.1f} us/step  {k}")
    if "--list" in sys.argv:
        a, b = starts[-3], starts[-2]
        s0, last_end, gaps = rows[a][0], {}, defaultdict(float)
        print("\none step, launch by launch (queue, start us, +duration us, queue idle before it, kernel):")
        for s, e, n, q in rows[a:b + 1]:
            gap = (s - last_end.get(q, s)) / 1e3
            if q in last_end:
                gaps[q] += max(gap, 0.0)
            print(f"q{q} {(s - s0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {n.replace('void ', '').split('(')[0][:60]}")
            last_end[q] = e
        for q, g in sorted(gaps.items()):
            print(f"queue {q}: idle between its launches {g:.1f} us")


if __name__ == "__main__":
    main()
