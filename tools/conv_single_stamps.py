#!/usr/bin/env python3
"""Diagnostic: phase stamps of ONE stand-alone conv launch of the plan (conv_igemm / conv_pair / conv_pair_next kernels
of csrc/conv_igemm.hip).  Builds a -DRSG_CONV_STAMPS library on the GPU box, runs the plan eagerly up to the op, prints
the per-workgroup medians of the phases: 0 entry, 1 set-up done, 2 first staging round in LDS, 3 main k-loop done,
5 tail set-up done, 6 tail loop done, 4 end."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from rsgnet_amd import build as B  # noqa: E402

so = "/tmp/librsg_cstamps.so"
if not os.path.exists(so):
    srcs = sorted(os.path.join(B.CSRC, f) for f in os.listdir(B.CSRC) if f.endswith(".hip"))
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-shared",
                           "-DRSG_CONV_STAMPS", "-mllvm", "-amdgpu-mfma-vgpr-form", *srcs, "-o", so])
os.environ["RSGNET_HIP_LIB"] = so
os.environ["RSGNET_HIP_PLAN"] = "graph=0"
import bench  # noqa: E402
from rsgnet_amd import hip  # noqa: E402

cfg, model, sd, wh, batch = bench.make_model(os.environ.get("WORKLOAD", "hrnet_w32_256x192"))
model = model.cuda()
x = torch.randn(batch, 3, wh[1], wh[0], device="cuda")
with torch.no_grad():
    model(x)
plan = list(model._plans.values())[0]
ops = plan.ops
lib = hip.lib()
stamps = torch.zeros((1 << 16) * 8, dtype=torch.int64, device="cuda")
lib.rsg_conv_set_stamp_buffer_single(C.c_void_p(stamps.data_ptr()))
s = torch.cuda.current_stream().cuda_stream
for name in (sys.argv[1:] or ["layer1.1.conv2+conv3+next.conv1"]):
    k = [n for n, _ in ops].index(name)
    for _ in range(2):
        for _, op in ops[:k]:
            op(s)
        torch.cuda.synchronize()
        stamps.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops[k][1](s)
        e1.record()
        torch.cuda.synchronize()
    raw = stamps.cpu().numpy().reshape(-1, 8).astype(np.float64)
    live = raw[:, 4] > 0
    b = raw[live]
    n = int(live.sum())
    span = b[:, 4].max() - b[:, 0][b[:, 0] > 0].min() if (b[:, 0] > 0).any() else float("nan")
    print(f"{name}: {n} workgroups, launch {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build), first entry -> last end {span:.0f} clk")
    has5 = (b[:, 5] > 0).any()
    seq = [0, 1, 2, 3] + ([5, 6] if has5 else []) + [4]
    lab = {0: "entry", 1: "setup", 2: "stage0", 3: "k-loop", 5: "tail-setup", 6: "tail-loop", 4: "end"}
    parts = []
    for a, c in zip(seq[:-1], seq[1:]):
        ok = (b[:, a] > 0) & (b[:, c] > 0)
        parts.append(f"{lab[a]}->{lab[c]} {np.median(b[ok, c] - b[ok, a]):7.0f}")
    ok = b[:, 0] > 0
    print("   per workgroup (median clk): " + "  ".join(parts) + f"   total {np.median(b[ok, 4] - b[ok, 0]):7.0f}")
    d = np.sort(b[ok, 4] - b[ok, 0])
    print(f"   total duration percentiles 10/50/90/max: {d[len(d) // 10]:.0f} {d[len(d) // 2]:.0f} {d[len(d) * 9 // 10]:.0f} {d[-1]:.0f}")
