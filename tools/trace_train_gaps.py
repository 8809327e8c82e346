#!/usr/bin/env python3
"""Idle time inside a training step, from a rocprofv3 --kernel-trace CSV of `bench.py --train`: steps are cut at the
weight-packing kernel (first launch of the forward graph)."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in csv.DictReader(open(f))]
rows.sort()
cut = [i for i, r in enumerate(rows) if r[2].startswith("void pack_weights_kernel")]
steps = [rows[a:b] for a, b in zip(cut[:-1], cut[1:])]
steps = steps[-6:]
for s in steps:
    busy = sum(e - b for b, e, _ in s) / 1e3
    span = (s[-1][1] - s[0][0]) / 1e3
    gaps = [(s[i + 1][0] - s[i][1]) / 1e3 for i in range(len(s) - 1)]
    big = sorted(((g, s[i][2][:45], s[i + 1][2][:45]) for i, g in enumerate(gaps)), reverse=True)
    print(f"step: {len(s)} kernels, span {span:.0f} us, busy {busy:.0f} us, idle {span - busy:.0f} us; gaps > 20 us: {sum(1 for g in gaps if g > 20)} "
          f"({sum(g for g in gaps if g > 20):.0f} us), gaps 2..20 us: {sum(1 for g in gaps if 2 < g <= 20)} ({sum(g for g in gaps if 2 < g <= 20):.0f} us), "
          f"median gap {np.median(gaps):.2f} us")
s = steps[-1]
gaps = [(s[i + 1][0] - s[i][1]) / 1e3 for i in range(len(s) - 1)]
for g, a, b in sorted(((g, s[i][2][:50], s[i + 1][2][:50]) for i, g in enumerate(gaps)), reverse=True)[:25]:
    print(f"   {g:8.1f} us  {a} -> {b}")
