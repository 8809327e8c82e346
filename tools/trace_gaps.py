#!/usr/bin/env python3
"""Idle time between consecutive kernels of the replayed forward graph, from a rocprofv3 --kernel-trace CSV."""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(f))]
rows.sort()
per = int(sys.argv[2]) if len(sys.argv) > 2 else 36
# the last steps: take the last 4 * per kernels, split into steps at the stem kernel
tail = rows[-(4 * per + 8):]
starts = [i for i, r in enumerate(tail) if r[2].startswith("void stem_kernel") or "stem" in r[2]]
for a, b in zip(starts[:-1], starts[1:]):
    step = tail[a:b]
    busy = sum(e - s for s, e, _ in step)
    gaps = [step[i + 1][0] - step[i][1] for i in range(len(step) - 1)]
    print(f"step of {len(step)} kernels: span {(step[-1][1] - step[0][0]) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, gaps total {sum(gaps) / 1e3:.1f} us "
          f"(median {np.median(gaps) / 1e3:.2f}, max {max(gaps) / 1e3:.2f} us); to next step {(tail[b][0] - step[-1][1]) / 1e3:.1f} us")
    big = sorted(((g, step[i][2], step[i + 1][2]) for i, g in enumerate(gaps)), reverse=True)[:5]
    for g, x, y in big:
        print(f"    {g / 1e3:6.2f} us between {x[:40]} -> {y[:40]}")
