#!/usr/bin/env python3
"""Sweep rsg_conv_igemm tile hints for the bandwidth-bound convs of the stem / layer1 (GPU box only)."""
import ctypes as C
import itertools
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from rsgnet_amd import hip  # noqa: E402

lib = hip.lib()
B = 64
SHAPES = [  # name, cin, cout, H, W (input), k, stride, residual
    ("layer1.conv1 256->64 1x1", 256, 64, 64, 48, 1, 1, False),
    ("layer1.conv3 64->256 1x1 +res", 64, 256, 64, 48, 1, 1, True),
    ("layer1.conv2 64->64 3x3", 64, 64, 64, 48, 3, 1, False),
    ("layer1.0.conv3+ds 128->256 1x1", 128, 256, 64, 48, 1, 1, False),
    ("stem.conv1 16->64 3x3 s2", 16, 64, 256, 192, 3, 2, False),
    ("stem.conv2 64->64 3x3 s2", 64, 64, 128, 96, 3, 2, False),
    ("transition1.0 256->32 3x3", 256, 32, 64, 48, 3, 1, False),
    ("transition1.1 256->64 3x3 s2", 256, 64, 64, 48, 3, 2, False),
]


if len(sys.argv) > 1 and sys.argv[1] == "head":      # the 3x3 convs of RSGNet-W32's head (pose_rsgnet.py:666-750)
    SHAPES = [
        ("contact_conv 96->96 3x3", 96, 96, 64, 48, 3, 1, False),
        ("predict_contact_net 96->32 3x3", 96, 32, 64, 48, 3, 1, False),
        ("kpt_net 64->32 3x3", 64, 32, 64, 48, 3, 1, False),
        ("vis_conv / limbs_net 32->32 3x3", 32, 32, 64, 48, 3, 1, False),
        ("type_conv(folded) 32->32 3x3 (17 of 32 in)", 32, 32, 64, 48, 3, 1, False),
        ("predict_net 32->32 3x3 128x96", 32, 32, 128, 96, 3, 1, False),
    ]


def bench(desc, reps=5):
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        if lib.rsg_conv_igemm(C.byref(desc), st):
            return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.rsg_conv_igemm(C.byref(desc), st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, cin, cout, H, W, k, s, res in SHAPES:
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(B, H, W, cin, device="cuda").half()
    out = torch.empty(B, Ho, Wo, cout, device="cuda", dtype=torch.float16)
    r = torch.randn(B, Ho, Wo, cout, device="cuda").half() if res else None
    w = np.random.randn(cout, cin, k, k).astype(np.float32) * 0.05
    wp = np.zeros(lib.rsg_conv_packed_bytes(cin, cout, k, k), np.uint8)
    bo = np.zeros((cout + 31) // 32 * 32, np.float32)
    lib.rsg_pack_conv_weights_host(w.ctypes.data, None, None, None, cin, cout, k, k, 0, wp.ctypes.data, bo.ctypes.data)
    wd, bd = torch.from_numpy(wp).cuda(), torch.from_numpy(bo).cuda()
    d = hip.ConvDesc()
    d.inp, d.w_packed, d.bias, d.out16 = x.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr()
    d.residual = r.data_ptr() if res else None
    d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = B, H, W, cin, cin
    d.out_h, d.out_w, d.out_c, d.out_ld, d.res_ld = Ho, Wo, cout, cout, cout
    d.kh = d.kw = k
    d.stride, d.pad_y, d.pad_x = s, pad, pad
    d.up, d.phys_h, d.phys_w, d.relu, d.dtype = 1, Ho, Wo, 1, 0
    mb = (x.numel() + out.numel() + (r.numel() if res else 0)) * 2 / 1e6
    d.tile_hint = 0
    base = bench(d)
    info = (C.c_int32 * 8)()
    lib.rsg_conv_plan(C.byref(d), info)
    results = []
    cpad = (cout + 31) // 32 * 32
    for wm in (1, 2, 4):
        if cpad % (32 * wm):
            continue
        wn = 4 // wm
        for nt in ((3, 1) if wm < 4 else (2,)):
            cap = 32 * nt * wn
            for tw in (Wo, 48, 32, 24, 16, 12, 8):
                if tw > Wo or tw > cap:
                    continue
                for th in sorted({cap // tw, max(1, cap // tw // 2)}):
                    if th < 1 or th > Ho or th > 255:
                        continue
                    d.tile_hint = wm | (nt << 4) | (th << 8) | (tw << 16)
                    t = bench(d, 3)
                    if t:
                        results.append((t, wm, nt, th, tw))
    results.sort()
    best = results[:3]
    print(f"{name:34s} auto {base:6.1f} us ({mb / base:5.2f} TB/s) tile{list(info)[:6]} | best " +
          "  ".join(f"{t:5.1f}us WM{wm} NT{nt} {th}x{tw}" for t, wm, nt, th, tw in best), flush=True)
