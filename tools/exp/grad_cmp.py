#!/usr/bin/env python3
import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
names, offs = list(a["names"]), list(a["offs"])
order = np.argsort(offs)
ends = [offs[order[i + 1]] if i + 1 < len(order) else len(a["G"]) for i in range(len(order))]
print("losses", a["losses"], b["losses"])
for i, o in enumerate(order):
    lo, hi = offs[o], ends[i]
    x, y = a["G"][lo:hi].astype(np.float64), b["G"][lo:hi].astype(np.float64)
    d = np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-30)
    if d > 1e-5:
        print(f"{names[o]:16s} rel L2 diff {d:.3e}  max|x| {np.abs(x).max():.3e}")
print("D", np.linalg.norm(a["D"] - b["D"]) / np.linalg.norm(a["D"]))
