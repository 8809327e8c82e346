#!/usr/bin/env python3
"""GPU box probe: training steps of pose_rsgnet (FusedAdam through DataParallelStep, as bench.py --train) with the head
eager and with the head replayed as HIP graphs (RSGNET_HIP_TRAIN="head_graph=1"), step by step from the same state: loss,
largest gradient and parameter differences, and the first tensors that differ."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rsgnet_amd import make_cfg, models                                        # noqa: E402
from rsgnet_amd.loss import JointsMSELoss, relation_target_vector, rsgnet_loss  # noqa: E402
from rsgnet_amd.optim import FusedAdam                                         # noqa: E402
from rsgnet_amd.parallel import DataParallelStep                               # noqa: E402
from rsgnet_amd.weights import gaussian_targets, limb_targets, synth_input, synth_state_dict  # noqa: E402

B = int(os.environ.get("PROBE_BATCH", "8"))
STEPS = int(os.environ.get("PROBE_STEPS", "6"))
dev = torch.device("cuda:0")


def run(mode):
    os.environ["RSGNET_HIP_TRAIN"] = f"head_graph={mode}"
    cfg = make_cfg("pose_rsgnet", 32, (192, 256), 17)
    model = models.pose_rsgnet.get_pose_net(cfg, is_train=False)
    bn = dict(np.load(os.path.join(ROOT, "tests", "golden", "bnstats_rsgnet_w32_256x192_coco.npz")))
    model.load_state_dict(synth_state_dict(model.state_dict(), 0, bn))
    model = model.to(dev).train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    crit = JointsMSELoss(True)
    step_fn = DataParallelStep(model, lambda outs, *t: rsgnet_loss(outs, *t, crit), opt)
    x = synth_input(B, (192, 256), seed=100).to(dev)
    target, tw = gaussian_targets(B, 17, (128, 96), 200)
    all_t, all_w = gaussian_targets(B, 17, (128, 96), 300)
    limbs_t = limb_targets(B, 18, (128, 96), 400)
    target, tw, all_t, all_w, limbs_t = (t.to(dev) for t in (target, tw, all_t, all_w, limbs_t))
    trace = []
    for it in range(STEPS):
        loss, outs = step_fn(x, target, tw, all_t, all_w, limbs_t, relation_target=relation_target_vector(target))
        torch.cuda.synchronize()
        trace.append((float(loss), [o.detach().clone() for o in outs], opt.flat_grad.clone(), opt.flat_param.clone(),
                      {k: b.detach().clone() for k, b in model.named_buffers()}))
    names = [(n, opt.flat_offset_of(p), p.numel()) for n, p in model.named_parameters() if opt.flat_offset_of(p) is not None]
    return trace, names


eager, names = run("0")
graph, _ = run("1")
for it, ((le, oe, ge, pe, be), (lg, og, gg, pg, bg)) in enumerate(zip(eager, graph)):
    dout = [float((a.float() - b.float()).abs().max()) for a, b in zip(oe, og)]
    print(f"step {it}: loss eager {le:.6f} graph {lg:.6f}; outputs max diff {['%.2e' % d for d in dout]}; "
          f"gradient max diff {float((ge - gg).abs().max()):.3e} (|g| max {float(ge.abs().max()):.3e}); "
          f"parameters after the step max diff {float((pe - pg).abs().max()):.3e}", flush=True)
    bad = [(n, float((ge[o:o + k] - gg[o:o + k]).abs().max())) for n, o, k in names if not torch.equal(ge[o:o + k], gg[o:o + k])]
    if bad:
        print(f"   {len(bad)} of {len(names)} gradients differ; first: {bad[:6]}")
    badb = [k for k in be if not torch.equal(be[k], bg[k])]
    if badb:
        print(f"   {len(badb)} buffers differ; first: {badb[:6]}")
    if bad or badb:
        break
