#!/usr/bin/env python3
"""Bit-reproducibility soak on the GPU box: N replays of the captured forward on one input must all give the first
replay's bits (hand-off flags, LDS reuse in the fused launches, grouped grids)."""
import sys
import torch
sys.path.insert(0, ".")
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for wl, dt in (("hrnet_w32_256x192", "fp16"), ("hrnet_w32_256x192", "bf16"), ("rsgnet_w32_256x192_coco", "fp16"), ("hrnet_w48_384x288", "fp16")):
    cfg, model, sd, wh, batch = bench.make_model(wl)
    model.compute_dtype = dt
    model = model.cuda()
    x = torch.randn(batch, 3, wh[1], wh[0], device="cuda")
    with torch.no_grad():
        first = model(x)
        first = [t.clone() for t in (first if isinstance(first, tuple) else (first,)) if torch.is_tensor(t)]
        bad = 0
        for it in range(n):
            out = model(x)
            out = [t for t in (out if isinstance(out, tuple) else (out,)) if torch.is_tensor(t)]
            if not all(torch.equal(a, b) for a, b in zip(first, out)):
                bad += 1
    torch.cuda.synchronize()
    print(f"{wl} {dt}: {n} replays, {bad} differ from the first; finite {all(bool(torch.isfinite(t).all()) for t in first)}")
    assert bad == 0
    del model
    torch.cuda.empty_cache()
