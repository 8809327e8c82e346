#!/usr/bin/env python3
"""Would a plain sequential caller gain from model(x) running the two halves of its batch beside each other on two
internal streams (as forward_flip_test does with its two forwards)?  HRNet-W32 256x192, batch 64."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rsgnet_amd.weights import synth_input

wl = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w32_256x192"
cfg, model, sd, wh, batch = bench.make_model(wl)
dev = torch.device("cuda", 0)
model = model.to(dev)
x = synth_input(batch, wh, seed=0).to(dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def whole():
    with torch.no_grad():
        return model(x)


def split(parts=2):
    cur = torch.cuda.current_stream()
    h = batch // parts
    outs = []
    with torch.no_grad():
        for k in range(parts):
            st = (s1, s2)[k % 2]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                o = model(x[k * h:(k + 1) * h])
            o.record_stream(cur)
            outs.append(o)
        cur.wait_stream(s1)
        cur.wait_stream(s2)
        return torch.cat(outs)


def timeit(f, n=300):
    for _ in range(60):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a = whole()
b = split()
print("bit-identical:", torch.equal(a, b))
for rep in range(3):
    print(f"one forward of {batch}: {timeit(whole):.4f} ms   two halves beside each other: {timeit(split):.4f} ms   four quarters on two streams: {timeit(lambda: split(4)):.4f} ms", flush=True)
