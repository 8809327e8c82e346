#!/usr/bin/env python3
"""Diagnostic: where does a chain workgroup spend its cycles?  Builds chain.hip with -DRSG_CHAIN_STAMPS into a
scratch library (GPU box only), runs one chain per HRNet-W32 branch and prints the phase shares."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from rsgnet_amd import build as B  # noqa: E402
from rsgnet_amd import hip  # noqa: E402

so = "/tmp/librsg_stamps.so"
srcs = [os.path.join(B.CSRC, f) for f in ("chain.hip", "graph.hip", "conv_igemm.hip")]
cmd = [B._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DRSG_CHAIN_STAMPS",
       *os.environ.get("CHAIN_STAMP_FLAGS", "-mllvm -amdgpu-mfma-vgpr-form").split(), *srcs, "-o", so]
subprocess.check_call(cmd)
lib = C.CDLL(so)
lib.rsg_last_error.restype = C.c_char_p
Bn = int(os.environ.get("CHAIN_STAMP_BATCH", "64"))
SHAPES = ((32, 64, 48, 4), (64, 32, 24, 4), (128, 16, 12, 4), (256, 8, 6, 4), (64, 32, 24, 8), (32, 64, 48, 8), (128, 16, 12, 8), (256, 8, 6, 8))
if os.environ.get("CHAIN_STAMP_SHAPES"):          # "C,H,W,L;C,H,W,L": e.g. HRNet-W48's 48,96,72,2;96,48,36,2;192,24,18,2;384,12,9,2
    SHAPES = tuple(tuple(int(v) for v in sh.split(",")) for sh in os.environ["CHAIN_STAMP_SHAPES"].split(";"))
for Cc, H, W, L in SHAPES:
    x = (torch.zeros if os.environ.get("STAMP_ZERO") else torch.randn)(Bn, H, W, Cc, device="cuda").half()
    out = torch.empty_like(x)
    keep, wps, bps = [], [], []
    for l in range(L):
        w = (np.random.randn(Cc, Cc, 3, 3) * (2.0 / (Cc * 9)) ** 0.5).astype(np.float32)
        wp, bo = hip.pack_chain_layer(w, None, None, 0)      # the pack the chain kernels of this width read
        wd, bd = torch.from_numpy(wp).cuda(), torch.from_numpy(bo).cuda()
        keep += [wd, bd]
        wps.append(wd.data_ptr())
        bps.append(bd.data_ptr())
    info = (C.c_int32 * 4)()
    lib.rsg_basic_chain_supported(Cc, H, W, L, Bn, info)
    nblk = info[3]
    stamps = torch.zeros(nblk * 16 * 64, dtype=torch.int64, device="cuda")
    lib.rsg_chain_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
    lib.rsg_basic_chain_workspace_bytes.restype = C.c_size_t
    wsp = torch.empty(max(lib.rsg_basic_chain_workspace_bytes(Cc, H, W, L, Bn), 16), dtype=torch.uint8, device="cuda")
    args = (C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), (C.c_void_p * L)(*wps), (C.c_void_p * L)(*bps), L, Bn, H, W, Cc, 0,
            C.c_void_p(wsp.data_ptr()), None)
    for _ in range(3):
        rc = lib.rsg_basic_chain(*args)
        assert rc == 0, lib.rsg_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.rsg_basic_chain(*args)
    e1.record()
    torch.cuda.synchronize()
    sw = stamps.cpu().numpy().reshape(nblk, 16, 64).astype(np.float64)
    nw = int(np.count_nonzero(sw[nblk // 2, :, 0]))
    blk = nblk // 2
    t0 = sw[blk, 0, 0]
    print('   k-loop cycles per layer, wave 0 / last wave: ' + '  '.join(f'{sw[blk, 0, 5 + 4 * l] - sw[blk, 0, 3 + 4 * l]:.0f}/{sw[blk, nw - 1, 5 + 4 * l] - sw[blk, nw - 1, 3 + 4 * l]:.0f}' for l in range(L)))
    for l in range(min(L, 2)):
        print(f"   layer {l}: per-wave [k-loop start, k-loop end, after barriers] rel. to WG start: " + "  ".join(f"w{w}:{sw[blk, w, 3 + 4 * l] - t0:.0f}/{sw[blk, w, 5 + 4 * l] - t0:.0f}/{sw[blk, w, 6 + 4 * l] - t0:.0f}" for w in range(nw)))
    st = sw[:, 0, :]
    d = lambda a, b: np.median(st[:, b] - st[:, a])
    tot = d(0, 62)
    parts = [("zero-fill", d(0, 1)), ("stage input", d(1, 2))]
    kl = np.median(sum(st[:, 5 + 4 * l] - st[:, 3 + 4 * l] for l in range(L)))
    pro = np.median(sum(st[:, 3 + 4 * l] - (st[:, 2] if l == 0 else st[:, 6 + 4 * (l - 1)]) for l in range(L)))
    epi = np.median(sum(st[:, 6 + 4 * l] - st[:, 5 + 4 * l] for l in range(L)))
    parts += [("layer prologue (bias/res/B preload)", pro), ("k-loops", kl), ("epilogue+barriers+write", epi),
              ("final store", d(6 + 4 * (L - 1), 62))]
    print(f"C={Cc} {H}x{W} L={L} blocks={nblk} waves={nw} tile_rows={info[0]} kernel {e0.elapsed_time(e1) * 1e3:.1f} us; "
          f"median WG {tot:.0f} shader clk: " +
          ", ".join(f"{n} {v / tot * 100:.0f}%" for n, v in parts), flush=True)
