#!/usr/bin/env python3
"""Diagnostic: where do the workgroups of one grouped fuse launch spend their cycles?  Builds the library with
-DRSG_CONV_STAMPS (GPU box only), runs the HRNet-W32 plan eagerly up to the chosen launch and prints, per item of
that launch, the workgroup count, the median duration and its phases, plus the per-CU packing."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from rsgnet_amd import build as B  # noqa: E402

so = "/tmp/librsg_cstamps.so"
srcs = sorted(os.path.join(B.CSRC, f) for f in os.listdir(B.CSRC) if f.endswith(".hip"))
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-shared",
                       "-DRSG_CONV_STAMPS", "-mllvm", "-amdgpu-mfma-vgpr-form", *srcs, "-o", so])
os.environ["RSGNET_HIP_LIB"] = so
os.environ["RSGNET_HIP_PLAN"] = "graph=0"
import bench  # noqa: E402
from rsgnet_amd import hip  # noqa: E402

name = os.environ.get("OP", "stage4.0.fuse.level0")
cfg, model, sd, wh, batch = bench.make_model("hrnet_w32_256x192")
model = model.cuda()
x = torch.randn(batch, 3, wh[1], wh[0], device="cuda")
with torch.no_grad():
    model(x)
plan = list(model._plans.values())[0]
ops = plan.ops
k = [n for n, _ in ops].index(name)
entry = [c for c in plan.conv_log if c.get("name") == name][0]
single = "group" not in entry
nblk = entry["tile"][7] if single else entry["blocks"]
members = [name] if single else entry["group"]
tiles = {c["name"]: c for c in plan.conv_log if "tile" in c}
lib = hip.lib()
stamps = torch.zeros(max(nblk, 1 << 16) * 8, dtype=torch.int64, device="cuda")   # every launch before the chosen one stamps too
(lib.rsg_conv_set_stamp_buffer_single if single else lib.rsg_conv_set_stamp_buffer)(C.c_void_p(stamps.data_ptr()))
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    for _, op in ops[:k]:
        op(s)
    torch.cuda.synchronize()
    stamps.zero_()
    ops[k][1](s)
    torch.cuda.synchronize()
raw = stamps.cpu().numpy().reshape(-1, 8)[:nblk]
st = raw.astype(np.float64)
print(f"{name}: {nblk} workgroups, items {len(members)}")
first = 0
for m in members:
    t = tiles.get(m)
    if t is None:                      # a fuse-sum item: no stamps inside
        continue
    n = t["tile"][7] if len(t["tile"]) > 7 else 0
    blk = st[first:first + n]
    d = blk[:, 4] - blk[:, 0]
    ok = blk[:, 4] > 0
    if ok.any():
        b = blk[ok]
        print(f"  {m:26s} cin {t['cin']:3d} cout {t['cout']:3d} k{t['k']} s{t['s']} tile {t['tile'][:6]}: {n:4d} wgs, median {np.median(d[ok]):6.0f} clk = "
              f"args {np.median(b[:, 1] - b[:, 0]):5.0f} + stage {np.median(b[:, 2] - b[:, 1]):5.0f} + mfma {np.median(b[:, 3] - b[:, 2]):5.0f} + store {np.median(b[:, 4] - b[:, 3]):5.0f}")
    first += n
# the launch's timeline: when each item's first workgroup starts and its last one ends, from the launch's first stamp
t0 = st[st[:, 4] > 0, 0].min()
print(f"  launch span {st[:, 4].max() - t0:.0f} clk")
first = 0
for m in members:
    t = tiles.get(m)
    if t is None:
        continue
    n = t["tile"][7] if len(t["tile"]) > 7 else 0
    blk = st[first:first + n]
    blk = blk[blk[:, 4] > 0]
    if len(blk):
        print(f"  {m:26s} first start {blk[:, 0].min() - t0:6.0f}  last start {blk[:, 0].max() - t0:6.0f}  last end {blk[:, 4].max() - t0:6.0f}")
    first += n
xcc = (raw[:, 7] >> 32) & 0xf
hwid = raw[:, 7] & 0xffffffff
cu = ((hwid >> 8) & 0xf) | (((hwid >> 13) & 0x7) << 4) | (xcc << 8)
# cold-start effects: the phases of a workgroup by its rank among the workgroups of the SAME item on its CU (entry order)
first = 0
for m in members:
    t = tiles.get(m)
    if t is None:
        continue
    n = t["tile"][7] if len(t["tile"]) > 7 else 0
    idx = np.arange(first, first + n)
    idx = idx[st[idx, 4] > 0]
    ranks = np.zeros(len(idx), int)
    for c in np.unique(cu[idx]):
        sel = np.where(cu[idx] == c)[0]
        ranks[sel[np.argsort(st[idx[sel], 0])]] = np.arange(len(sel))
    line = []
    for r in (0, 1, 2, 3):
        mm = ranks == r if r < 3 else ranks >= 3
        if mm.any():
            b = st[idx[mm]]
            line.append(f"rank{'>=' if r == 3 else ' '}{r}: n {mm.sum():4d} args {np.median(b[:, 1] - b[:, 0]):5.0f} stage {np.median(b[:, 2] - b[:, 1]):5.0f} "
                        f"mfma {np.median(b[:, 3] - b[:, 2]):5.0f} store {np.median(b[:, 4] - b[:, 3]):5.0f}")
    print(f"  {m:26s} " + " | ".join(line))
    first += n
per = []
for c in np.unique(cu):
    m = cu == c
    valid = m & (st[:, 4] > 0)
    if valid.any():
        per.append((st[valid, 4].max() - st[valid, 0].min(), int(valid.sum()), float((st[valid, 4] - st[valid, 0]).sum())))
per = np.array(per)
print(f"  per CU: {len(per)} CUs, workgroups per CU median {np.median(per[:, 1]):.0f} (max {per[:, 1].max():.0f}), busy span median {np.median(per[:, 0]):.0f} "
      f"max {per[:, 0].max():.0f} clk, sum of workgroup durations / span = {np.median(per[:, 2] / per[:, 0]):.2f} resident on average")
