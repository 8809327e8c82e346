#!/usr/bin/env python3
"""What makes bench.py's 20-step timed region slower than a bare 20-step window?  Variants of what runs in front of it."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rsgnet_amd.weights import synth_input

cfg, model, sd, wh, batch = bench.make_model("hrnet_w32_256x192")
dev = torch.device("cuda", 0)
model = model.to(dev)
x = synth_input(batch, wh, seed=0).to(dev)


def step():
    with torch.no_grad():
        return model(x)


def region(k=20, events=False):
    torch.cuda.synchronize()
    if events:
        st = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if events:
        e0.record(st)
    for _ in range(k):
        step()
    if events:
        e1.record(st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


for _ in range(200):
    step()
torch.cuda.synchronize()


def show(name, f, n=6):
    print(f"{name:70s}", " ".join(f"{f():.3f}" for _ in range(n)), flush=True)


show("bare region", region)
show("region with timing events", lambda: region(events=True))


def with_check():
    out = step()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    return region()


show("isfinite(out).all() in front", with_check)


def with_item():
    out = step()
    v = float(out[0, 0, 0, 0])
    return region()


show("one element read back (.item()) in front", with_item)


def with_hold():
    out = step()
    torch.cuda.synchronize()
    r = region()
    del out
    return r


show("last output kept alive across the region", with_hold)


def with_sleep(ms):
    def f():
        torch.cuda.synchronize()
        time.sleep(ms / 1e3)
        return region()
    return f


for ms in (1, 5, 20, 100):
    show(f"{ms} ms of idle time in front", with_sleep(ms))
show("timed_steps (bench.py)", lambda: bench.timed_steps(step, 20, None, dev)[0] / 20 * 1e3)
show("bare region again", region)
