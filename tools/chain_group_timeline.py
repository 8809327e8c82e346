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
ni = [int(v) for v in os.environ.get("NI", "0,0,0,0,0").split(",")]
nb = int(os.environ.get("BRANCHES", "4"))
if os.environ.get("HANDOFF", "1") != "0":      # the module as the plan launches it: 8 layers, branch 0 as two halves
    branches = [(256, 8, 6, 8, None), (128, 16, 12, 8, None), (64, 32, 24, 8, None)][3 - (nb - 1):] + \
               [(32, 64, 48, 4, None), (32, 64, 48, 4, nb - 1)]
else:
    L = int(os.environ.get("LAYERS", "4"))
    branches = [(256, 8, 6, L, None), (128, 16, 12, L, None), (64, 32, 24, L, None), (32, 64, 48, L, None)][4 - nb:]
keep = []
items = (hip.ChainItem * len(branches))()
nblk = []
outs = []
for k, (Cc, H, W, L, after) in enumerate(branches):
    x = torch.randn(Bn, H, W, Cc, device="cuda").half() if after is None else outs[after]
    out = torch.empty_like(x)
    outs.append(out)
    wps, bps = [], []
    for l in range(L):
        w = (np.random.randn(Cc, Cc, 3, 3) * (2.0 / (Cc * 9)) ** 0.5).astype(np.float32)
        wp, bo = hip.pack_chain_layer(w, None, None, 0)      # the pack the chain kernels of this width read
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
    if after is not None:
        fl = torch.zeros(3 * info[3], dtype=torch.int32, device="cuda")
        keep.append(fl)
        items[after].done_flags = fl.data_ptr()
        it.wait_flags = fl.data_ptr()
total = sum(nblk)
stamps = torch.zeros(total * 16 * 64, dtype=torch.int64, device="cuda")
lib.rsg_chain_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
for _ in range(3):
    assert lib.rsg_basic_chain_group(items, len(branches), 0, None) == 0, lib.rsg_last_error()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
lib.rsg_basic_chain_group(items, len(branches), 0, None)
e1.record()
torch.cuda.synchronize()
raw = stamps.cpu().numpy().reshape(total, 16, 64)
sw = raw.astype(np.float64)
# every XCD has a clock of its own: times are taken from the first start seen on the workgroup's XCD
xcc = (raw[:, 0, 63] >> 32) & 0xf
hwid = raw[:, 0, 63] & 0xffffffff
cu = ((hwid >> 8) & 0xf) | (((hwid >> 13) & 0x7) << 4) | (xcc << 8)      # cu_id, se_id (gfx9 HW_ID), XCD
for x in np.unique(xcc):
    m = xcc == x
    sw[m, :, :63] -= sw[m, 0, 0].min()
print(f"XCDs seen: {sorted(int(v) for v in np.unique(xcc))}, distinct CUs: {len(np.unique(cu))}")
print(f"grouped launch: {total} workgroups, {e0.elapsed_time(e1) * 1e3:.1f} us by events; "
      f"last end at {sw[:, 0, 62].max():.0f} cycles")
first = 0
for (Cc, H, W, L, after), n in zip(branches, nblk):
    st, en = sw[first:first + n, 0, 0], sw[first:first + n, 0, 62]
    d = en - st
    late = int((st > 20000).sum())
    print(f"  C={Cc:3d} {H}x{W} L={L}: {n:3d} workgroups, start {st.min():.0f}..{st.max():.0f}, duration median {np.median(d):.0f} "
          f"(min {d.min():.0f} max {d.max():.0f}), end median {np.median(en):.0f} max {en.max():.0f}; {late} started after 20k cycles "
          f"(their median start {np.median(st[st > 20000]) if late else 0:.0f})")
    first += n

# per-CU view (clock offsets between CUs do not matter here): which workgroups ran on each CU, back to back?
item_of = np.concatenate([np.full(n, k) for k, n in enumerate(nblk)])
from collections import Counter
pat, spans, idle = Counter(), [], []
for c in np.unique(cu):
    m = np.where(cu == c)[0]
    m = m[np.argsort(raw[m, 0, 0])]
    s0 = raw[m, 0, 0].astype(np.float64)
    e0_ = raw[m, 0, 62].astype(np.float64)
    pat[",".join(str(item_of[i]) for i in m)] += 1
    spans.append(e0_.max() - s0.min())
    idle.append(float(sum(max(0.0, s0[j + 1] - e0_[j]) for j in range(len(m) - 1))))
spans = np.array(spans)
print(f"per-CU busy span (first start to last end): median {np.median(spans):.0f} max {spans.max():.0f} cycles; "
      f"gaps between consecutive workgroups on a CU: median {np.median(idle):.0f} max {max(idle):.0f}")
for k, v in pat.most_common(12):
    print(f"   {v:3d} CUs ran items [{k}]")
