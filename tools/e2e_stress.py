#!/usr/bin/env python3
"""Determinism stress of the whole forward (GPU box): the same input must give the same bits every time."""
import os
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from util import load_case  # noqa: E402

tag = os.environ.get("TAG", "hrnet_w32_256x192")
Bn = int(os.environ.get("BATCH", "4"))
cfg, model, sd, x, gold = load_case(tag)
model = model.cuda()
xx = torch.cat([x, x.flip(3)] * (Bn // 2), 0).cuda()
first, nd = None, 0
with torch.no_grad():
    for rep in range(int(os.environ.get("REPS", "30"))):
        y = model(xx)
        y = y[1] if isinstance(y, tuple) else y
        a_ = torch.randn(4096, 4096, device="cuda", dtype=torch.half) * 100
        junk = a_ @ a_                                   # fresh garbage in every CU's LDS and caches
        torch.cuda.synchronize()
        if first is None:
            first = y.clone()
        elif not torch.equal(y, first):
            nd += 1
            d = (y - first).abs()
            bad = torch.nonzero(d.amax(dim=(1, 2, 3)) > 0).flatten().tolist()
            print(f"  rep {rep}: max diff {float(d.max()):.3g} in images {bad}")
print(f"{tag} B={Bn} env={ {k: v for k, v in os.environ.items() if k.startswith('RSGNET_HIP')} }: {nd} repeats differ")
