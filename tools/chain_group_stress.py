#!/usr/bin/env python3
"""Determinism stress of grouped chain launches (GPU box): which combination of branches disagrees with itself?"""
import ctypes as C
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from rsgnet_amd import hip  # noqa: E402

lib = hip.lib()
dbg = None
if hasattr(lib, 'rsg_chain_set_dbg_buffer'):
    dbg = torch.zeros(4 * 4096, dtype=torch.int32, device='cuda')
    lib.rsg_chain_set_dbg_buffer(C.c_void_p(dbg.data_ptr()))
Bn = int(os.environ.get("BATCH", "8"))
ALL = {"c256": (256, 8, 6, 8), "c128": (128, 16, 12, 8), "c64": (64, 32, 24, 8), "c32": (32, 64, 48, 4)}
keep = []


def make(Cc, H, W, L):
    g = torch.Generator().manual_seed(Cc)
    x = torch.randn(Bn, H, W, Cc, generator=g).half().cuda()
    wps, bps = [], []
    for l in range(L):
        w = (np.random.RandomState(l).randn(Cc, Cc, 3, 3) * (2.0 / (Cc * 9)) ** 0.5).astype(np.float32)
        wp, bo = hip.pack_chain_layer(w, None, None, 0)      # the pack the chain kernels of this width read
        wd, bd = torch.from_numpy(wp).cuda(), torch.from_numpy(bo).cuda()
        keep.extend([wd, bd])
        wps.append(wd.data_ptr())
        bps.append(bd.data_ptr())
    wsp = torch.empty(lib.rsg_basic_chain_workspace_bytes(Cc, H, W, L, Bn), dtype=torch.uint8, device="cuda")
    wa, ba = (C.c_void_p * L)(*wps), (C.c_void_p * L)(*bps)
    keep.extend([x, wsp, wa, ba])
    return x, wa, ba, wsp


data = {k: make(*v) for k, v in ALL.items()}
names = list(ALL)
for n in range(4, 5):
    for combo in list(itertools.combinations(names, n)) * 6:
        items = (hip.ChainItem * len(combo))()
        outs = []
        for it, k in zip(items, combo):
            Cc, H, W, L = ALL[k]
            x, wa, ba, wsp = data[k]
            out = torch.empty_like(x)
            outs.append(out)
            it.inp, it.out = x.data_ptr(), out.data_ptr()
            it.w_packed, it.bias = C.cast(wa, C.POINTER(C.c_void_p)), C.cast(ba, C.POINTER(C.c_void_p))
            it.nlayers, it.batch, it.h, it.w, it.c = L, Bn, H, W, Cc
            it.workspace = wsp.data_ptr()
        first, nd = None, [0] * len(combo)
        for rep in range(20):
            for o in outs:
                o.fill_(float("nan"))
            a_ = torch.randn(4096, 4096, device="cuda", dtype=torch.half) * 100
            junk = a_ @ a_
            assert lib.rsg_basic_chain_group(items, len(combo), 0, torch.cuda.current_stream().cuda_stream) == 0, lib.rsg_last_error()
            torch.cuda.synchronize()
            if dbg is not None and int(dbg.abs().sum()):
                d4 = dbg.view(-1, 4).cpu()
                for b in torch.nonzero(d4[:, 0]).flatten().tolist()[:4]:
                    print(f'     rep {rep}: block {b}: {int(d4[b, 0])} dirty border chunks, first at LDS offset {int(d4[b, 1])} value {int(d4[b, 2]) & 0xffffffff:#x} (C*1e5+R = {int(d4[b, 3])})')
                dbg.zero_()
            if first is None:
                first = [o.clone() for o in outs]
                prev = first
            else:
                for j, (o, f) in enumerate(zip(outs, first)):
                    if not torch.equal(o, f):
                        nd[j] += 1
                        if nd[j] <= 3:
                            d = (o.float() - f.float()).abs()
                            px = torch.nonzero(d.amax(3) > 0)
                            print(f"     item {j} rep {rep}: {int((d > 0).sum())} elements, max {float(d.max()):.3g}, images {sorted(set(px[:, 0].tolist()))}, "
                                  f"rows {sorted(set(px[:, 1].tolist()))[:8]}.., cols {sorted(set(px[:, 2].tolist()))[:8]}.., "
                                  f"channels {sorted(set(torch.nonzero(d.amax((0, 1, 2)) > 0).flatten().tolist()))[:10]}.., same as prev rep: {torch.equal(o, prev[j])}")
                prev = [o.clone() for o in outs]
        print(f"{'+'.join(combo):28s} repeats that differ per item: {nd}", flush=True)
