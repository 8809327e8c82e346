#!/usr/bin/env python3
"""Timing experiment (GPU box): step time with every chain layer reading layer 0's weights (RSGNET_HIP_TUNE=chain_debug=1, results
are WRONG) against the real forward — how much of the chain kernels' time is the weight stream from L2?"""
import os
import sys
import torch
sys.path.insert(0, ".")
import bench  # noqa: E402

cfg, model, sd, wh, batch = bench.make_model("hrnet_w32_256x192")
model = model.cuda()
x = torch.randn(batch, 3, wh[1], wh[0], device="cuda")
with torch.no_grad():
    for _ in range(5):
        model(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        model(x)
    e1.record()
    torch.cuda.synchronize()
print(f"RSGNET_HIP_TUNE={os.environ.get('RSGNET_HIP_TUNE', '')}: {e0.elapsed_time(e1) / 30:.4f} ms per step")
