#!/usr/bin/env python3
"""Determinism stress of the chain kernel (GPU box): repeated launches of one chain must agree bit for bit."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from rsgnet_amd import hip  # noqa: E402

lib = hip.lib()
bad = 0
for Cc, H, W, L, Bn in ((32, 64, 48, 4, 4), (32, 64, 48, 4, 64), (64, 32, 24, 8, 4), (64, 32, 24, 8, 64), (128, 16, 12, 8, 4),
                        (256, 8, 6, 8, 4), (256, 8, 6, 8, 64)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(Bn, H, W, Cc, generator=g).half().cuda()
    keep, wps, bps = [], [], []
    for l in range(L):
        w = (np.random.RandomState(l).randn(Cc, Cc, 3, 3) * (2.0 / (Cc * 9)) ** 0.5).astype(np.float32)
        wp, bo = hip.pack_chain_layer(w, None, None, 0)      # the pack the chain kernels of this width read
        wd, bd = torch.from_numpy(wp).cuda(), torch.from_numpy(bo).cuda()
        keep += [wd, bd]
        wps.append(wd.data_ptr())
        bps.append(bd.data_ptr())
    wsp = torch.empty(lib.rsg_basic_chain_workspace_bytes(Cc, H, W, L, Bn), dtype=torch.uint8, device="cuda")
    first, ndiff = None, 0
    for rep in range(40):
        out = torch.full_like(x, float("nan"))
        a_ = torch.randn(4096, 4096, device="cuda", dtype=torch.half) * 100
        other = a_ @ a_                                      # fresh garbage in every CU's LDS and caches
        rc = lib.rsg_basic_chain(x.data_ptr(), out.data_ptr(), (C.c_void_p * L)(*wps), (C.c_void_p * L)(*bps), L, Bn, H, W, Cc, 0,
                                 wsp.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.rsg_last_error()
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
            assert torch.isfinite(first.float()).all()
        elif not torch.equal(out, first):
            ndiff += 1
            d = (out.float() - first.float()).abs()
            idx = torch.nonzero(d.view(Bn, H, W, Cc).amax(3) > 0)
            print(f"  rep {rep}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3g}; pixels (b,y,x) e.g. {idx[:6].tolist()} ... rows {sorted(set(idx[:, 1].tolist()))[:12]}")
    print(f"C={Cc} {H}x{W} L={L} B={Bn}: {ndiff} of 39 repeats differ")
    bad += ndiff
sys.exit(1 if bad else 0)
