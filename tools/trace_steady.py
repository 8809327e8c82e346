#!/usr/bin/env python3
"""The clock ramp and the settled per-kernel durations, from a rocprofv3 --kernel-trace CSV of bench.py (inference).
Steps are cut at the stem kernel; prints the step span along the run and, over the steps within 3 % of the shortest span,
the average duration of every kernel name and the idle time between kernels."""
import collections, csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
cut = [i for i, r in enumerate(rows) if r[2].startswith("void stem_kernel")]
steps = [rows[a:b] for a, b in zip(cut[:-1], cut[1:])]
n_k = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
steps = [s for s in steps if len(s) == n_k]                    # graph replays (the eager passes carry extra kernels)
t0 = steps[0][0][0]
span = np.array([(s[-1][1] - s[0][0]) / 1e3 for s in steps])
busy = np.array([sum(e - b for b, e, _ in s) / 1e3 for s in steps])
period = np.array([(steps[i + 1][0][0] - steps[i][0][0]) / 1e3 for i in range(len(steps) - 1)] + [np.nan])
print(f"{len(steps)} steps of {n_k} kernels")
print("step   t since first launch   period us   span us   kernels busy us")
for i in sorted(set([0, 2, 5, 10, 15, 20, 30, 50, 75, 100, 150, 200, 300, 400, 500, 600, 700, len(steps) - 2])):
    if 0 <= i < len(steps):
        print(f"{i:5d}  {(steps[i][0][0] - t0) / 1e6:10.1f} ms        {period[i]:8.1f}   {span[i]:8.1f}   {busy[i]:8.1f}")
ok = np.where((period < np.nanmin(period) * 1.03))[0]
print(f"settled steps: {len(ok)} (period <= 1.03 x {np.nanmin(period):.1f} us); mean period {np.nanmean(period[ok]):.1f} us, kernels busy {busy[ok].mean():.1f} us")
dur = collections.defaultdict(list)
for i in ok:
    for b, e, k in steps[i]:
        dur[k.split("(")[0][:70]].append((e - b) / 1e3)
print("kernel, launches per step, average us (settled), us per step")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:70s} {len(v) / len(ok):5.1f}  {np.mean(v):8.2f}  {sum(v) / len(ok):8.1f}")
cold = [i for i in range(len(steps)) if (steps[i][0][0] - t0) < 45e6 and i >= 5][:20]
if cold:
    dc = collections.defaultdict(list)
    for i in cold:
        for b, e, k in steps[i]:
            dc[k.split("(")[0][:70]].append((e - b) / 1e3)
    print(f"cold region (steps {cold[0]}..{cold[-1]}): mean period {np.nanmean(period[cold]):.1f} us; per kernel name, us per step cold / settled:")
    for k, v in sorted(dc.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {k:70s} {sum(v) / len(cold):8.1f} / {sum(dur[k]) / len(ok):8.1f}")
