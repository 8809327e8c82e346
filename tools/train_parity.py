#!/usr/bin/env python3
"""GPU box: one training step of pose_hrnet on the HIP path against the fp64 oracle; prints the error distribution of
the gradients by parameter kind (what tests/test_train_step.py bounds)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_train_step as t  # noqa: E402
from rsgnet_amd.loss import JointsMSELoss  # noqa: E402

cfg, model, sd, x, target, tw, gold = t._case()
loss_o, heat_o, grads_o, stats_o = t._oracle()
model.load_state_dict(sd)
model = model.cuda().train()
out = model(x.cuda())
loss = JointsMSELoss(True)(out, target.cuda(), tw.cuda())
loss.backward()
torch.cuda.synchronize()
print("loss", loss.item(), "oracle", loss_o.item(), "heat max-abs", (out.detach().cpu().double() - heat_o).abs().max().item())
noise = dict(zip([str(n) for n in gold["grad_names"]], gold["ref_fp32_relL2"]))
rows = {"conv": [], "bn.weight": [], "bn.bias": [], "other": []}
for k, p in model.named_parameters():
    g, o = p.grad.detach().cpu().double(), grads_o[k].double()
    rel = (g - o).norm().item() / (o.norm().item() + 1e-30)
    cos = (g * o).sum().item() / (g.norm().item() * o.norm().item() + 1e-30)
    kind = "conv" if g.dim() == 4 and not k.startswith("final") else ("bn.weight" if k.endswith("weight") and g.dim() == 1 else
                                                                        ("bn.bias" if k.endswith("bias") and not k.startswith("final") else "other"))
    rows[kind].append((rel, cos, noise[k], o.norm().item(), k))
for kind, r in rows.items():
    rel = np.array([a[0] for a in r]); cos = np.array([a[1] for a in r]); nz = np.array([a[2] for a in r])
    print(f"{kind:10s} n={len(r):4d} relL2 median {np.median(rel):.3f} p90 {np.quantile(rel, .9):.3f} max {rel.max():.3f} | cos min {cos.min():.4f} "
          f"median {np.median(cos):.5f} | reference fp32-vs-fp64 relL2 median {np.median(nz):.2e} max {nz.max():.2e}")
    for a in sorted(r, reverse=True)[:3]:
        print("     worst: relL2 %.3f cos %.4f ref-noise %.2e |g| %.2e %s" % a)
