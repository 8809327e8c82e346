#!/usr/bin/env python3
"""Diagnostic (GPU box): where does a workgroup of rsg_fuse_band spend its cycles?  Builds the library with
-DRSG_BAND_STAMPS, runs the HRNet-W32 plan (RSGNET_HIP_FUSE_BAND=1) eagerly up to the chosen band launch and prints the
median phase durations."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from rsgnet_amd import build as B  # noqa: E402

so = "/tmp/librsg_bstamps.so"
srcs = sorted(os.path.join(B.CSRC, f) for f in os.listdir(B.CSRC) if f.endswith(".hip"))
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-shared",
                       "-DRSG_BAND_STAMPS", "-mllvm", "-amdgpu-mfma-vgpr-form", *srcs, "-o", so])
os.environ["RSGNET_HIP_LIB"] = so
os.environ["RSGNET_HIP_GRAPH"] = "0"
os.environ["RSGNET_HIP_FUSE_BAND"] = "1"
import bench  # noqa: E402
from rsgnet_amd import hip  # noqa: E402

cfg, model, sd, wh, batch = bench.make_model("hrnet_w32_256x192")
model = model.cuda()
x = torch.randn(batch, 3, wh[1], wh[0], device="cuda")
with torch.no_grad():
    model(x)
plan = list(model._plans.values())[0]
lib = C.CDLL(so)
lib.rsg_band_set_stamp_buffer.argtypes = [C.c_void_p]
s = torch.cuda.current_stream().cuda_stream
for name in sys.argv[1:] or ["stage2.0.fuse.band", "stage3.1.fuse.band", "stage4.0.fuse.band"]:
    k = [n for n, _ in plan.ops].index(name)
    nblk = batch * (wh[1] // 4 // 8)
    stamps = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
    lib.rsg_band_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
    for _ in range(2):
        for _, op in plan.ops[:k]:
            op(s)
        torch.cuda.synchronize()
        stamps.zero_()
        plan.ops[k][1](s)
        torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(-1, 8).astype(np.float64)
    d = [np.median(st[:, i + 1] - st[:, i]) for i in range(4)]
    print(f"{name}: {nblk} workgroups; median clk: stage {d[0]:.0f} | 1x1 convs {d[1]:.0f} | d10 + y1 {d[2]:.0f} | y0 sum {d[3]:.0f} | "
          f"total {np.median(st[:, 4] - st[:, 0]):.0f}; launch span {(st[:, 4].max() - st[:, 0].min()):.0f} clk")
