#!/usr/bin/env python3
"""Diagnostic: when does every workgroup of a grouped chain launch start and end?  Builds chain.hip with
-DRSG_CHAIN_STAMPS (GPU box only), launches the four branches of a stage-4 HRNet-W32 module in one grid and
prints, per branch, the start/end spread of its workgroups (shader-clock cycles from the first start)."""
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
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-DRSG_CHAIN_STAMPS",
                       "-mllvm", "-amdgpu-mfma-vgpr-form", *srcs, "-o", so])
lib = C.CDLL(so)
lib.rsg_last_error.restype = C.c_char_p
lib.rsg_basic_chain_workspace_bytes.restype = C.c_size_t
Bn = int(os.environ.get("BATCH", "64"))
L = int(os.environ.get("LAYERS", "4"))
ni = [int(v) for v in os.environ.get("NI", "0,0,0,0").split(",")]
branches = [(256, 8, 6), (128, 16, 12), (64, 32, 24), (32, 64, 48)][:int(os.environ.get("BRANCHES", "4"))]
keep = []
items = (hip.ChainItem * len(branches))()
nblk = []
for k, (Cc, H, W) in enumerate(branches):
    x = torch.randn(Bn, H, W, Cc, device="cuda").half()
    out = torch.empty_like(x)
    wps, bps = [], []
    for l in range(L):
        w = (np.random.randn(Cc, Cc, 3, 3) * (2.0 / (Cc * 9)) ** 0.5).astype(np.float32)
        wp = np.zeros(Cc * Cc * 9 * 2, np.uint8)
        bo = np.zeros(Cc, np.float32)
        lib.rsg_pack_conv_weights_host(C.c_void_p(w.ctypes.data), None, None, None, Cc, Cc, 3, 3, 0,
                                       C.c_void_p(wp.ctypes.data), C.c_void_p(bo.ctypes.data))
        wd, bd = torch.from_numpy(wp).cuda(), torch.from_numpy(bo).cuda()
        keep += [wd, bd]
        wps.append(wd.data_ptr())
        bps.append(bd.data_ptr())
    wsp = torch.empty(lib.rsg_basic_chain_workspace_bytes(Cc, H, W, L, Bn), dtype=torch.uint8, device="cuda")
    wa, ba = (C.c_void_p * L)(*wps), (C.c_void_p * L)(*bps)
    keep += [x, out, wsp, wa, ba]
    it = items[k]
    it.inp, it.out = x.data_ptr(), out.data_ptr()
    it.w_packed, it.bias = C.cast(wa, C.POINTER(C.c_void_p)), C.cast(ba, C.POINTER(C.c_void_p))
    it.nlayers, it.batch, it.h, it.w, it.c = L, Bn, H, W, Cc
    it.images_per_workgroup = ni[k]
    it.workspace = wsp.data_ptr()
    info = (C.c_int32 * 4)()
    lib.rsg_basic_chain_supported(Cc, H, W, L, Bn, info)
    nblk.append(info[3] // max(ni[k], 1) if info[0] == H else info[3])
total = sum(nblk)
stamps = torch.zeros(total * 8 * 32, dtype=torch.int64, device="cuda")
lib.rsg_chain_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
for _ in range(3):
    assert lib.rsg_basic_chain_group(items, len(branches), 0, None) == 0, lib.rsg_last_error()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
lib.rsg_basic_chain_group(items, len(branches), 0, None)
e1.record()
torch.cuda.synchronize()
sw = stamps.cpu().numpy().reshape(total, 8, 32).astype(np.float64)
t0 = sw[:, 0, 0].min()
print(f"grouped launch: {total} workgroups, {e0.elapsed_time(e1) * 1e3:.1f} us by events; "
      f"last end at {(sw[:, 0, 30].max() - t0):.0f} cycles")
first = 0
for (Cc, H, W), n in zip(branches, nblk):
    st, en = sw[first:first + n, 0, 0] - t0, sw[first:first + n, 0, 30] - t0
    d = en - st
    late = int((st > 0.2 * en.max()).sum())
    print(f"  C={Cc:3d} {H}x{W}: {n:3d} workgroups, start {st.min():.0f}..{st.max():.0f}, duration median {np.median(d):.0f} "
          f"(min {d.min():.0f} max {d.max():.0f}), end max {en.max():.0f}; {late} started late")
    first += n
