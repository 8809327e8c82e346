#!/usr/bin/env python3
"""Which launch of the RSGNet plan goes wrong when another stream's forwards share the GPU?  The launch list of one plan
runs eagerly with every 16-bit map cloned behind its writer (Plan.run_with_taps), alone and under contention."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import load_case

tag = sys.argv[1] if len(sys.argv) > 1 else "rsgnet_w32_256x192_coco"
cfg, model, sd, x, gold = load_case(tag)
model = model.cuda().eval()
xc = x.cuda()
with torch.no_grad():
    model(xc)
    torch.cuda.synchronize()
    plan = model.plan_for(xc)
    order, seen = [], set()
    for w in plan._op_writes:
        for v in w:
            if v.name and v.name not in seen:
                seen.add(v.name)
                order.append(v.name)
    base = plan.run_with_taps(xc, order)
    extra0 = {k: getattr(plan, k).clone() for k in ("e32", "multi_lo", "limbs_lo", "kpt", "multi_up", "limbs", "tail_ws") if hasattr(plan, k)}
    torch.cuda.synchronize()
    again = plan.run_with_taps(xc, order)
    torch.cuda.synchronize()
    print("quiet repeat differs at:", [n for n in order if not torch.equal(base[n], again[n])])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    x2 = torch.flip(xc, dims=[3]).contiguous()
    for rep in range(40):
        with torch.cuda.stream(side):
            for _ in range(6):
                model(x2)                      # the other stream's plan: contention only
        got = plan.run_with_taps(xc, order)
        extra = {k: getattr(plan, k).clone() for k in extra0}
        torch.cuda.synchronize()
        bad = [n for n in order if not torch.equal(base[n], got[n])]
        badx = [k for k in extra0 if not torch.equal(extra0[k], extra[k])]
        print(f"round {rep}: maps that differ (in launch order): {bad[:8]}; fp32 buffers: {badx}")
        if "tail_ws" in badx:
            ny = 2 * 3072 * 32
            dd = (extra0["tail_ws"] != extra["tail_ws"]).nonzero().flatten()
            print("    tail_ws: y rows that differ:", sorted({int(i) // 32 for i in dd.tolist() if i < ny})[:20], "; partial floats that differ:", [int(i) - ny for i in dd.tolist() if i >= ny][:40])
            for i in [int(i) for i in dd.tolist() if i >= ny][:8]:
                print("       part", i - ny, float(extra["tail_ws"][i]), "was", float(extra0["tail_ws"][i]))
        if bad:
            n = bad[0]
            d = (base[n].float() - got[n].float()).abs()
            idx = (d > 0).nonzero()
            print(f"    first: {n} shape {tuple(d.shape)}: {idx.shape[0]} elements, images {idx[:, 0].unique().tolist()}, max {float(d.max()):.3e}; "
                  f"rows {int(idx[:, 1].min())}..{int(idx[:, 1].max())} cols {int(idx[:, 2].min())}..{int(idx[:, 2].max())} ch {idx[:, 3].unique().tolist()[:12]}")
