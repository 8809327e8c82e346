#!/usr/bin/env python3
"""How long does one launch of the plan take (a) in the running forward, (b) alone after a pause, (c) repeated back to
back, (d) after the forward up to it with a sync in front, (e) after a cache-flushing fill?  (GPU box)"""
import os
import sys
import time

import torch

os.environ["RSGNET_HIP_PLAN"] = "graph=0"
sys.path.insert(0, ".")
import bench  # noqa: E402

name = os.environ.get("OP", "stage4.0.chains")
cfg, model, sd, wh, batch = bench.make_model("hrnet_w32_256x192")
model = model.cuda().half() if False else model.cuda()
x = torch.randn(batch, 3, wh[1], wh[0], device="cuda")
with torch.no_grad():
    model(x)
plan = list(model._plans.values())[0]
ops = plan.ops
k = [n for n, _ in ops].index(name)
s = torch.cuda.current_stream().cuda_stream


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(fn_before, reps=6):
    out = []
    for _ in range(reps):
        fn_before()
        e0, e1 = ev(), ev()
        e0.record()
        ops[k][1](s)
        e1.record()
        for _, op in ops[k + 1:]:
            op(s)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3)
    return sorted(out)[len(out) // 2]


def prefix():
    for _, op in ops[:k]:
        op(s)


def prefix_sync():
    prefix()
    torch.cuda.synchronize()


def prefix_sync_sleep():
    prefix_sync()
    time.sleep(0.05)


big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")          # 1 GiB


def prefix_flush():
    prefix()
    big.fill_(1.0)


print(f"{name}: in the running forward {timed(prefix):.1f} us; forward, sync, launch {timed(prefix_sync):.1f} us; "
      f"forward, sync, 50 ms idle, launch {timed(prefix_sync_sleep):.1f} us; forward, 1 GiB fill, launch {timed(prefix_flush):.1f} us")
e0, e1 = ev(), ev()
prefix_sync()
ops[k][1](s)
e0.record()
for _ in range(4):
    ops[k][1](s)
e1.record()
torch.cuda.synchronize()
print(f"   repeated back to back: {e0.elapsed_time(e1) * 1e3 / 4:.1f} us")

# which cold buffer costs the time?  flush everything, then warm one class of buffers before the launch
wts = [t for t in plan._keep if torch.is_tensor(t) and t.numel() >= 1024]
wsl = list(getattr(plan, "_chain_ws", {}).values())
acts = list({id(v.storage): v.storage for v in plan._vts if getattr(v, "storage", None) is not None}.values())
print(f"   weights {sum(t.numel() * t.element_size() for t in wts) / 1e6:.0f} MB in {len(wts)} tensors, workspaces "
      f"{sum(t.numel() for t in wsl) / 1e6:.0f} MB, activations {sum(t.numel() * t.element_size() for t in acts) / 1e6:.0f} MB")


def warm(tensors):
    def f():
        prefix()
        big.fill_(1.0)
        for t in tensors:
            t.view(torch.uint8).view(-1)[::64].sum() if t.dtype != torch.uint8 else t[::64].sum()
    return f


print(f"   after flush, warmed: nothing {timed(prefix_flush):.1f}; weights {timed(warm(wts)):.1f}; workspaces {timed(warm(wsl)):.1f}; "
      f"activations {timed(warm(acts)):.1f}; weights+workspaces {timed(warm(wts + wsl)):.1f} us")
