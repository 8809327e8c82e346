#!/usr/bin/env python3
"""GPU box probe: does torch.cuda.graph capture of a fork/join over side streams work with librsgnet_hip launches on
this ROCm?  (Round 1 saw hipStreamEndCapture crash on the inference plan's multi-stream order.)"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from rsgnet_amd import hip  # noqa: E402

lib = hip.lib()
dev = torch.device("cuda:0")
BF = hip.DTYPES["bf16"]
n = 4
maps = [torch.randn(32, 64 >> k, 48 >> k, 32 << k, device=dev).to(torch.bfloat16) for k in range(n)]
outs = [torch.empty_like(m) for m in maps]
ws = [torch.empty(lib.rsg_bn_workspace_bytes(256), dtype=torch.uint8, device=dev) for _ in range(n)]
mean = [torch.zeros(m.shape[-1], device=dev) for m in maps]
inv = [torch.zeros(m.shape[-1], device=dev) for m in maps]
gam = [torch.ones(m.shape[-1], device=dev) for m in maps]
bet = [torch.zeros(m.shape[-1], device=dev) for m in maps]


def unit(k, s):
    m = maps[k]
    px, Cc = m.shape[0] * m.shape[1] * m.shape[2], m.shape[3]
    for _ in range(8):
        hip.check(lib.rsg_bn_stats(m.data_ptr(), px, Cc, BF, ws[k].data_ptr(), mean[k].data_ptr(), inv[k].data_ptr(), None, None, 0.1, 1e-5, s))
        hip.check(lib.rsg_bn_apply(m.data_ptr(), mean[k].data_ptr(), inv[k].data_ptr(), gam[k].data_ptr(), bet[k].data_ptr(), None, 1,
                                   outs[k].data_ptr(), px, Cc, BF, s))


def serial():
    s = torch.cuda.current_stream().cuda_stream
    for k in range(n):
        unit(k, s)


side = [torch.cuda.Stream() for _ in range(n - 1)]


def forked():
    main = torch.cuda.current_stream()
    for st in side:
        st.wait_stream(main)
    unit(0, main.cuda_stream)
    for k, st in enumerate(side, 1):
        with torch.cuda.stream(st):
            unit(k, st.cuda_stream)
    for st in side:
        main.wait_stream(st)


for name, fn in (("serial", serial), ("forked", forked)):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    print(f"{name}: captured and replayed, {(time.time() - t0) / 50 * 1e6:.1f} us per replay (96 launches)", flush=True)
    ref = [o.clone() for o in outs]
print("ok")
