#!/usr/bin/env python3
"""rsg_trp_tail (W 1x1 + GroupNorm) alone on one stream, fixed input, while another stream keeps the GPU busy: do the
projected rows, the statistics partials or only the output change?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsgnet_amd import hip

lib = hip.lib()
B, S, C = 2, 3072, 32
g = torch.Generator().manual_seed(0)
e = torch.randn(B, S, C, generator=g).cuda()
w = (torch.randn(C, C, generator=g) * 0.3).cuda()
bias = torch.randn(C, generator=g).cuda()
gam = (1 + 0.1 * torch.randn(C, generator=g)).cuda()
bet = torch.randn(C, generator=g).cuda()
out = torch.zeros(B, S, C, dtype=torch.float16, device="cuda")
ws = torch.zeros(lib.rsg_trp_tail_workspace_bytes(B, S, C) // 4, device="cuda")
ny = B * S * C


def run():
    st = torch.cuda.current_stream().cuda_stream
    hip.check(lib.rsg_trp_tail(e.data_ptr(), w.data_ptr(), bias.data_ptr(), gam.data_ptr(), bet.data_ptr(), out.data_ptr(), C, B, S, C,
                               hip.DTYPES["fp16"], ws.data_ptr(), st), "trp_tail")
    return out.clone(), ws[:ny].clone(), ws[ny:].clone()


o0, y0, p0 = run()
torch.cuda.synchronize()
for _ in range(20):
    o, y, p = run()
torch.cuda.synchronize()
print("quiet: equal", torch.equal(o, o0), torch.equal(y, y0), torch.equal(p, p0))
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda", dtype=torch.half)
# the other stream's work: the SAME kernels on buffers of their own (what a second plan of the model runs)
e2 = torch.randn(B, S, C, generator=g).cuda()
out2 = torch.zeros(B, S, C, dtype=torch.float16, device="cuda")
ws2 = torch.zeros_like(ws)
mode = sys.argv[1] if len(sys.argv) > 1 else "gemm"


def other():
    if mode == "gemm":
        a @ a
    else:
        hip.check(lib.rsg_trp_tail(e2.data_ptr(), w.data_ptr(), bias.data_ptr(), gam.data_ptr(), bet.data_ptr(), out2.data_ptr(), C, B, S, C,
                                   hip.DTYPES["fp16"], ws2.data_ptr(), torch.cuda.current_stream().cuda_stream), "trp_tail")
bad = 0
for rep in range(300):
    if rep % 10 == 0:
        with torch.cuda.stream(side):
            for _ in range(4 if mode == "gemm" else 40):
                other()
    o, y, p = run()
    if rep % 10 == 9:
        torch.cuda.synchronize()
    if not (torch.equal(o, o0) and torch.equal(y, y0) and torch.equal(p, p0)):
        torch.cuda.synchronize()
        bad += 1
        if bad <= 6:
            dp = (p != p0).nonzero().flatten().tolist()
            do = (o != o0).nonzero()
            print(f"rep {rep}: y equal {torch.equal(y, y0)}; partials that differ (float index; 16 per workgroup record): {dp[:24]}; "
                  f"out: {do.shape[0]} elements, images {do[:, 0].unique().tolist() if do.numel() else []}, channels {do[:, 2].unique().tolist()[:16] if do.numel() else []}")
            if dp:
                i = dp[0]
                print("      ", i, float(p[i]), float(p0[i]))
print("contended runs that differ:", bad, "of 300")
