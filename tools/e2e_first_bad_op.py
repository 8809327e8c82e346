#!/usr/bin/env python3
"""Which op first gives different bits on identical inputs?  Runs the first k ops of the plan (eager, back to back,
one sync at the end) repeatedly, with a GEMM on random data in between (fresh garbage in every CU's LDS and caches),
and compares every activation buffer between repeats; bisects over k."""
import os
import sys

import torch

os.environ["RSGNET_HIP_PLAN"] = "graph=0"
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from util import load_case  # noqa: E402

cfg, model, sd, x, gold = load_case(os.environ.get("TAG", "hrnet_w32_256x192"))
model = model.cuda()
xx = torch.cat([x, x.flip(3)] * 2, 0).cuda()
with torch.no_grad():
    model(xx)
plan = list(model._plans.values())[0]
ops = plan.ops
stores = {id(v.storage): v.storage for v in plan._vts if getattr(v, "storage", None) is not None}
print(f"{len(ops)} ops, {len(stores)} buffers")
plan.x_static.copy_(xx)
s = torch.cuda.current_stream().cuda_stream


def differs(k, reps=15):
    ref, nd = None, 0
    for _ in range(reps):
        a = torch.randn(4096, 4096, device="cuda", dtype=torch.half) * 100
        junk = a @ a
        for _, op in ops[:k]:
            op(s)
        torch.cuda.synchronize()
        snap = [t.clone() for t in stores.values()]
        if ref is None:
            ref = snap
        else:
            nd += int(any(not torch.equal(p, q) for p, q in zip(snap, ref)))
    return nd


lo, hi = 0, len(ops)
d = differs(hi)
print(f"whole plan: {d} repeats differ")
if d:
    while hi - lo > 1:
        mid = (lo + hi) // 2
        d = differs(mid)
        print(f"  first {mid:3d} ops ({ops[mid - 1][0]}): {d} repeats differ", flush=True)
        if d:
            hi = mid
        else:
            lo = mid
    print(f"first nondeterministic op: #{hi} {ops[hi - 1][0]}; the op before it: {ops[hi - 2][0] if hi > 1 else None}")
    print(f"again: first {hi - 1} ops: {differs(hi - 1, 30)} of 29 differ; first {hi} ops: {differs(hi, 30)} of 29 differ")
