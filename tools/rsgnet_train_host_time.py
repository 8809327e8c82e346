#!/usr/bin/env python3
"""Is the RSGNet training step host-bound?  Host time to ENQUEUE forward / loss / backward / optimizer against the
step's wall time with a synchronise at the end."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench  # noqa: E402
from rsgnet_amd.loss import JointsMSELoss, relation_target_vector, rsgnet_loss  # noqa: E402
from rsgnet_amd.optim import FusedAdam  # noqa: E402
from oracle.train_step_oracle import gaussian_targets, limb_targets  # noqa: E402  (data only)

cfg, model, sd, wh, _ = bench.make_model("rsgnet_w32_256x192_coco")
B = 32
model = model.cuda().train()
opt = FusedAdam(model.parameters(), lr=1e-3)
x = torch.randn(B, 3, wh[1], wh[0], device="cuda")
t, w = gaussian_targets(B, 17, (128, 96), 0)
t, w = t.cuda(), w.cuda()
lt = limb_targets(B, cfg.MODEL.NUM_LIMBS, (128, 96), 1).cuda()
crit = JointsMSELoss(True)
p = relation_target_vector(t)
rows = []
for it in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = model(x, relation_target=p)
    t1 = time.perf_counter()
    loss = rsgnet_loss(outs, t, w, t, w, lt, crit)
    opt.zero_grad()
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    rows.append([(b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t0, t5))])
for r in rows[3:]:
    print("host ms: forward %.2f  loss+zero_grad %.2f  backward %.2f  step %.2f  | wait for the GPU %.2f  | total %.2f" % tuple(r))
