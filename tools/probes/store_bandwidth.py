#!/usr/bin/env python3
"""How fast does this GPU take a store-only stream, and a read-only one?  (the launches whose output is large --
layer1's pair launches, the fp32 heat-map writers -- sit at 2.6 - 2.8 TB/s of stores.)  torch.fill_ / sum / copy_ on buffers of
64 MB .. 1 GB, best of 5 after a warm-up, HIP events."""
import torch

dev = torch.device("cuda:0")
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e-3
for mb in (64, 128, 256, 512, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty(n, dtype=torch.float32, device=dev)
    t_fill = timed(lambda: a.fill_(1.0))
    t_sum = timed(lambda: a.sum())
    t_copy = timed(lambda: b.copy_(a))
    gb = mb / 1024
    print(f"{mb:5d} MB: fill {gb / t_fill / 1e3 * 1024:6.2f} TB/s ({t_fill * 1e6:7.1f} us)   sum {gb / t_sum / 1e3 * 1024:6.2f} TB/s   "
          f"copy {2 * gb / t_copy / 1e3 * 1024:6.2f} TB/s read+write ({t_copy * 1e6:7.1f} us)", flush=True)
