#!/usr/bin/env python3
"""GPU box: per-launch device time of one training step of pose_hrnet (eager launches with an event pair around each),
the 25 longest and the totals by kind."""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import train_step_oracle as tso  # noqa: E402  (targets only)
from rsgnet_amd import make_cfg, models  # noqa: E402
from rsgnet_amd.loss import JointsMSELoss  # noqa: E402
from rsgnet_amd.weights import synth_input, synth_state_dict  # noqa: E402

B = int(os.environ.get("BATCH", "32"))
cfg = make_cfg("pose_hrnet", 32)
model = models.pose_hrnet.get_pose_net(cfg, is_train=False)
bn = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "bnstats_hrnet_w32_256x192.npz")))
model.load_state_dict(synth_state_dict(model.state_dict(), 0, bn))
model = model.cuda().train()
x = synth_input(B, (192, 256), 1).cuda()
t, w = tso.gaussian_targets(B, 17, (64, 48), 2)
for _ in range(2):
    out = model(x)
    JointsMSELoss(True)(out, t.cuda(), w.cuda()).backward()
torch.cuda.synchronize()
plan = next(iter(model._train_plans.values()))
st = torch.cuda.current_stream().cuda_stream
rows = []
for which, ops in (("fwd", plan.fwd), ("bwd", plan.bwd)):
    for rep in range(3):
        evs = []
        for rec in ops:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); rec[1](st); e1.record()
            evs.append((rec[0], e0, e1))
        torch.cuda.synchronize()
    rows += [(e0.elapsed_time(e1) * 1e3, which, nm) for nm, e0, e1 in evs]
if os.environ.get("TRAIN_OPS_JSON"):          # every launch in execution order
    import json
    with open(os.environ["TRAIN_OPS_JSON"], "w") as f:
        json.dump([dict(us=round(us, 2), phase=which, op=nm) for us, which, nm in rows], f)
rows.sort(reverse=True)
for us, which, nm in rows[:25]:
    print(f"{us:8.1f} us  {which}  {nm}")
kinds = collections.Counter()
for us, which, nm in rows:
    k = "wgrad" if ".wgrad" in nm else "dgrad" if ".dgrad" in nm or ".spread" in nm else "bn" if ".bn" in nm else "add" if nm.endswith(".add") else "conv/other"
    kinds[f"{which}:{k}"] += us
print({k: round(v) for k, v in kinds.items()}, "total", round(sum(kinds.values())))
