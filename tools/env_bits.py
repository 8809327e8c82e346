#!/usr/bin/env python3
"""GPU box: the headline model's output under several values of one environment switch, bit for bit against the first value.
usage: env_bits.py VAR "v0 v1 ..." [workload] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

var, vals = sys.argv[1], sys.argv[2].split()
workload = sys.argv[3] if len(sys.argv) > 3 else "hrnet_w32_256x192"
cfg, model, sd, wh, batch = bench.make_model(workload)
if len(sys.argv) > 4:
    batch = int(sys.argv[4])
from rsgnet_amd.weights import synth_input  # noqa: E402
x = synth_input(batch, wh, 5).cuda()
model = model.cuda()
ref = None
for v in vals:
    os.environ[var] = v
    model.invalidate_plan()
    with torch.no_grad():
        out = model(x)
    outs = [out] if torch.is_tensor(out) else list(out[:3])
    outs = [o.clone() for o in outs]
    torch.cuda.synchronize()
    plan = model.plan_for(x)
    if ref is None:
        ref = outs
        print(f"{var}={v}: {len(plan.ops)} launches (reference)")
    else:
        same = [bool(torch.equal(a, b)) for a, b in zip(ref, outs)]
        print(f"{var}={v}: {len(plan.ops)} launches, bit-identical to {var}={vals[0]}: {same}",
              "" if all(same) else [float((a - b).abs().max()) for a, b in zip(ref, outs)])
        if not all(same):
            sys.exit(1)
