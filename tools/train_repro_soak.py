#!/usr/bin/env python3
"""Bit-reproducibility of the training step with the weight gradients running beside the data-gradient chain: the same N steps
from the same state, twice (and once with everything on one stream); the parameters must agree bit for bit after every step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rsgnet_amd.loss import JointsMSELoss
from rsgnet_amd.optim import FusedAdam
from oracle.train_step_oracle import gaussian_targets          # targets (data) only

N = int(sys.argv[1]) if len(sys.argv) > 1 else 25


def run(env):
    for k, v in env.items():
        os.environ[k] = v
    torch.manual_seed(0)
    cfg, model, sd, wh, _ = bench.make_model("hrnet_w32_256x192")
    model = model.cuda().train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    crit = JointsMSELoss(True)
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(32, 3, wh[1], wh[0], generator=g).cuda() for _ in range(3)]
    t, w = gaussian_targets(32, 17, (64, 48), 0)
    t, w = t.cuda(), w.cuda()
    snaps = []
    for it in range(N):
        out = model(xs[it % 3])
        loss = crit(out, t, w)
        opt.zero_grad()
        loss.backward()
        opt.step()
        snaps.append(opt.flat_param.clone())
    torch.cuda.synchronize()
    for k in env:
        os.environ.pop(k)
    return snaps, float(loss)


a, la = run({})
b, lb = run({})
c, lc = run({"RSGNET_HIP_TRAIN": "wgrad_stream=0"})
first_ab = next((i for i in range(N) if not torch.equal(a[i], b[i])), None)
first_ac = next((i for i in range(N) if not torch.equal(a[i], c[i])), None)
print(f"{N} training steps of 32 crops, parameters after every step: default vs default again: first difference at step {first_ab}; "
      f"default (weight gradients on the side stream) vs one stream: first difference at step {first_ac}; final losses {la:.6f} {lb:.6f} {lc:.6f}")
