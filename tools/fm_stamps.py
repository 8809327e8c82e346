#!/usr/bin/env python3
"""Diagnostic: where do the workgroups of the band launches (rsg_fuse_module) spend their cycles?  Builds the library with
-DRSG_FM_STAMPS (GPU box), runs the HRNet-W32 plan eagerly and prints, per band launch, the phases of the median
workgroup (per wave: start, staged, phase A done, y0 sum done, end) and the launch span."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from rsgnet_amd import build as B  # noqa: E402

so = "/tmp/librsg_fmstamps.so"
srcs = sorted(os.path.join(B.CSRC, f) for f in os.listdir(B.CSRC) if f.endswith(".hip"))
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-shared",
                       "-DRSG_FM_STAMPS", "-mllvm", "-amdgpu-mfma-vgpr-form", *srcs, "-o", so])
os.environ["RSGNET_HIP_LIB"] = so
os.environ["RSGNET_HIP_PLAN"] = "graph=0"
import bench  # noqa: E402
from rsgnet_amd import hip  # noqa: E402

cfg, model, sd, wh, batch = bench.make_model("hrnet_w32_256x192")
model = model.cuda()
x = torch.randn(batch, 3, wh[1], wh[0], device="cuda")
with torch.no_grad():
    model(x)
plan = list(model._plans.values())[0]
ops = plan.ops
lib = hip.lib()
nblk = batch * 8
stamps = torch.zeros(nblk * 16 * 8, dtype=torch.int64, device="cuda")
lib.rsg_fuse_module_set_stamp_buffer.argtypes = [C.c_void_p]
lib.rsg_fuse_module_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
s = torch.cuda.current_stream().cuda_stream
for want in os.environ.get("OPS", "stage2.0.fuse.band stage3.1.fuse.band stage4.0.fuse.band").split():
    k = [n for n, _ in ops].index(want)
    for _ in range(2):
        for _, op in ops[:k]:
            op(s)
        torch.cuda.synchronize()
        stamps.zero_()
        ops[k][1](s)
        torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(nblk, 16, 8).astype(np.float64)
    live = st[:, 0, 4] > 0
    st = st[live]
    t0 = st[:, :, 0].min()
    wg0 = st[:, :, 0].min(axis=1)           # workgroup start
    end = st[:, :, 4].max(axis=1)
    print(f"{want}: {int(live.sum())} workgroups; launch span {end.max() - t0:.0f} clk; workgroup starts spread {wg0.max() - t0:.0f}; "
          f"median workgroup life {np.median(end - wg0):.0f}")
    ph = ["staged (barrier 1)", "phase A + barrier 2", "y0 sum", "phase B"]
    for i, nm in enumerate(ph):
        d = st[:, :, i + 1] - st[:, :, i]
        print(f"   {nm:22s} median over waves {np.median(d):7.0f}   min {np.median(d.min(axis=1)):7.0f}   max {np.median(d.max(axis=1)):7.0f}")
    # per wave: phase A own time is (stamp2 - stamp1) incl. waiting at the barrier; phase B per wave
    b = np.median(st[:, :, 4] - st[:, :, 3], axis=0)
    print("   phase B per wave (median over workgroups): " + " ".join(f"{v:.0f}" for v in b))
    print(f"   end of the last wave after its workgroup's start: median {np.median(end - wg0):.0f}, p90 {np.percentile(end - wg0, 90):.0f}")
