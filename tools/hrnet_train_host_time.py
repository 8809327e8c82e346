#!/usr/bin/env python3
"""Is the HRNet training step host-bound?  Host time to ENQUEUE forward / loss / backward / optimizer against the step's
wall time with a synchronize at the end; then a run of steps without synchronizes."""
import sys, time
import torch
sys.path.insert(0, ".")
import bench  # noqa: E402
from rsgnet_amd.loss import JointsMSELoss  # noqa: E402
from rsgnet_amd.optim import FusedAdam  # noqa: E402
from oracle.train_step_oracle import gaussian_targets  # noqa: E402  (data only)

cfg, model, sd, wh, _ = bench.make_model("hrnet_w32_256x192")
B = 32
model = model.cuda().train()
opt = FusedAdam(model.parameters(), lr=1e-3)
x = torch.randn(B, 3, wh[1], wh[0], device="cuda")
t, w = gaussian_targets(B, 17, (64, 48), 0)
t, w = t.cuda(), w.cuda()
crit = JointsMSELoss(True)
rows = []
for it in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model(x)
    t1 = time.perf_counter()
    loss = crit(out, t, w)
    opt.zero_grad()
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    rows.append([(b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t0, t5))])
for r in rows[5:]:
    print("host ms: forward %.2f  loss+zero_grad %.2f  backward %.2f  step %.2f  | wait for the GPU %.2f  | total %.2f" % tuple(r))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    out = model(x)
    loss = crit(out, t, w)
    opt.zero_grad()
    loss.backward()
    opt.step()
th = time.perf_counter()
torch.cuda.synchronize()
t1 = time.perf_counter()
print("30 steps without synchronizes: host done after %.2f ms per step, GPU after %.2f ms per step" % ((th - t0) / 30 * 1e3, (t1 - t0) / 30 * 1e3))
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    out = model(x)
    loss = crit(out, t, w)
    opt.zero_grad()
    loss.backward()
    opt.step()
pr.disable()
torch.cuda.synchronize()
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(22)
print("\n".join(l[:160] for l in buf.getvalue().splitlines()[:45]))
