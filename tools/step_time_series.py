#!/usr/bin/env python3
"""How the headline step's time moves along a run: 20-step windows with a host synchronize after each, the same windows
cut by events only (no idle time between them), and long regions.  One process, the bench's model and input."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rsgnet_amd.weights import synth_input

dt = sys.argv[1] if len(sys.argv) > 1 else "fp16"
cfg, model, sd, wh, batch = bench.make_model("hrnet_w32_256x192")
dev = torch.device("cuda", 0)
model = model.to(dev)
model.compute_dtype = dt
x = synth_input(batch, wh, seed=0).to(dev)


def step():
    with torch.no_grad():
        return model(x)


def windows_sync(n, k=20):
    out = []
    for _ in range(n):
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / k * 1e3)
    return out


def windows_events(n, k=20):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        for _ in range(k):
            step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) / k for i in range(n)]


def fmt(v):
    return " ".join(f"{a:.3f}" for a in v)


for _ in range(3):
    step()
torch.cuda.synchronize()
print(dt, "20-step windows, host synchronize after each (ms per step):")
print(fmt(windows_sync(40)), flush=True)
print("20-step windows cut by events, no host synchronize (device ms per step):")
print(fmt(windows_events(60)), flush=True)
print("windows with synchronize again:")
print(fmt(windows_sync(10)), flush=True)
t0 = time.perf_counter()
for _ in range(1000):
    step()
torch.cuda.synchronize()
print("1000 steps, one region:", f"{(time.perf_counter() - t0):.4f} s = {(time.perf_counter() - t0):.4f} ms per step")
time.sleep(3.0)
print("after 3 s of idle: windows with synchronize:")
print(fmt(windows_sync(20)), flush=True)
print("5-step windows with synchronize:")
print(fmt(windows_sync(20, 5)), flush=True)
print("100-step windows with synchronize:")
print(fmt(windows_sync(8, 100)), flush=True)
