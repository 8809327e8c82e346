#!/usr/bin/env python3
"""Does the phase offset between the two streams matter?  200 forwards on two alternating streams, stream B's first forward
held back by a spin of 0 .. 1 forward's length."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rsgnet_amd.weights import synth_input

cfg, model, sd, wh, batch = bench.make_model("hrnet_w32_256x192")
dev = torch.device("cuda", 0)
model = model.to(dev)
x = synth_input(batch, wh, seed=0).to(dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run(n, offset_ms):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        if offset_ms > 0:
            with torch.cuda.stream(sb):
                torch.cuda._sleep(int(offset_ms * 1e-3 * 2.2e9))      # _sleep(1e8) = 45 ms on this stack
        for k in range(n):
            with torch.cuda.stream(sa if k % 2 == 0 else sb):
                model(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0 - offset_ms * 1e-3 * 0) / n * 1e3


for _ in range(3):
    run(100, 0.0)
# calibrate _sleep
torch.cuda.synchronize(); t0 = time.perf_counter(); torch.cuda._sleep(100000000); torch.cuda.synchronize()
print(f"_sleep(1e8) = {(time.perf_counter() - t0) * 1e3:.1f} ms")
for rep in range(3):
    print("offset ms -> ms per batch:", "  ".join(f"{o:.2f}: {run(400, o):.4f}" for o in (0.0, 0.35, 0.7, 1.05, 1.4)), flush=True)
