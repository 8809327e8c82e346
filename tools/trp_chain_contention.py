#!/usr/bin/env python3
"""The TRP launch sequence (project_g -> attention -> W + GroupNorm) through the C ABI on one stream with a fixed input,
while the other stream replays a model's forward graph: which buffer deviates first?  Run twice: a synchronize between
the launches of the sequence or none."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsgnet_amd import hip
from tests.util import load_case

lib = hip.lib()
B, S, C = 2, 3072, 32
g = torch.Generator().manual_seed(0)
x16 = (torch.randn(B, S, C, generator=g) * 0.5).half().cuda()
wg = (torch.randn(C, C, generator=g) * 0.2).cuda()
bg = torch.randn(C, generator=g).cuda()
w = (torch.randn(C, C, generator=g) * 0.3).cuda()
bias = torch.randn(C, generator=g).cuda()
gam = (1 + 0.1 * torch.randn(C, generator=g)).cuda()
bet = torch.randn(C, generator=g).cuda()
gT = torch.zeros(B, C, S, dtype=torch.float16, device="cuda")
e32 = torch.zeros(B, S, C, device="cuda")
out = torch.zeros(B, S, C, dtype=torch.float16, device="cuda")
ws = torch.zeros(lib.rsg_trp_tail_workspace_bytes(B, S, C) // 4, device="cuda")
ny = B * S * C
F16 = hip.DTYPES["fp16"]


def run():
    st = torch.cuda.current_stream().cuda_stream
    hip.check(lib.rsg_trp_project_g(x16.data_ptr(), C, wg.data_ptr(), bg.data_ptr(), gT.data_ptr(), None, B, S, C, F16, st), "g")
    g1 = gT.clone()
    hip.check(lib.rsg_trp_attention(x16.data_ptr(), C, gT.data_ptr(), None, e32.data_ptr(), B, S, C, F16, st), "attn")
    e1 = e32.clone()
    hip.check(lib.rsg_trp_tail(e32.data_ptr(), w.data_ptr(), bias.data_ptr(), gam.data_ptr(), bet.data_ptr(), out.data_ptr(), C, B, S, C, F16,
                               ws.data_ptr(), st), "tail")
    return {"gT": g1, "e32 (right behind attention)": e1, "e32 (after the tail)": e32.clone(), "y": ws[:ny].clone(), "partials": ws[ny:].clone(),
            "out": out.clone()}


cfg, model, sd, x, gold = load_case(sys.argv[1] if len(sys.argv) > 1 else "rsgnet_w32_256x192_coco")
model = model.cuda().eval()
xc = torch.cat([x] * 4).cuda()
side = torch.cuda.Stream()
with torch.no_grad():
    with torch.cuda.stream(side):
        model(xc)
        model(xc)
    torch.cuda.synchronize()
    ref = run()
    torch.cuda.synchronize()
    for k in range(10):
        r = run()
    torch.cuda.synchronize()
    print("quiet:", [n for n in ref if not torch.equal(ref[n], r[n])])
    bad = 0
    for rep in range(1200):
        if rep % 4 == 0:
            with torch.cuda.stream(side):
                for _ in range(3):
                    model(xc)
        r = run()
        if rep % 4 == 3:
            torch.cuda.synchronize()
        d = [n for n in ref if not torch.equal(ref[n], r[n])]
        if d:
            torch.cuda.synchronize()
            bad += 1
            if bad <= 8:
                n0 = d[0]
                dd = (ref[n0] != r[n0]).nonzero()
                print(f"rep {rep}: differ {d}; first {n0}: {dd.shape[0]} elements; index ranges {[(int(dd[:, i].min()), int(dd[:, i].max())) for i in range(dd.shape[1])]}")
                if n0 == "y":
                    idx = dd.flatten().tolist()
                    print("      (row, channel):", [(i // C, i % C) for i in idx[:40]])
                    ee = ref["e32 (after the tail)"].reshape(-1, C)
                    for i in idx[:3]:
                        row, co = i // C, i % C
                        terms = (w[co] * ee[row]).tolist()
                        delta = float(r["y"][i] - ref["y"][i])
                        near = min(range(C), key=lambda ci: abs(abs(delta) - abs(terms[ci])))
                        print(f"      y[{row},{co}] = {float(r['y'][i]):.6f}, right {float(ref['y'][i]):.6f}, delta {delta:.6f}; nearest single term w[{co},{near}]*e = {terms[near]:.6f}; bias {float(bias[co]):.6f}")
    print("contended runs that differ:", bad, "of 1200")
