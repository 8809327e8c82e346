#!/usr/bin/env python3
"""Diagnostic: per-layer phase attribution of every chain of a grouped launch AS THE PLAN LAUNCHES IT (stage-4 module of
HRNet-W32: 8-layer whole-image chains + branch 0 as two 4-layer row-tile chains with the in-launch hand-off).  Builds
chain.hip with -DRSG_CHAIN_STAMPS into a scratch library (GPU box only).  Stamp slots per wave (64): 0 start, 1 borders
zeroed, 2 staged, per layer l: 3+4l k-loop start, 4+4l k-loop end, 5+4l epilogue VALU done (before the first barrier),
6+4l layer end (after write-back and barrier); 62 end; 63 XCC/HW id.
ALONE=1 launches every item in a launch of its own (no other workgroups on the chip) as well."""
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
                       "-mllvm", "-amdgpu-mfma-vgpr-form", *os.environ.get("CHAIN_STAMP_FLAGS", "").split(), *srcs, "-o", so])
lib = C.CDLL(so)
lib.rsg_last_error.restype = C.c_char_p
lib.rsg_basic_chain_workspace_bytes.restype = C.c_size_t
Bn = int(os.environ.get("BATCH", "64"))
SL = 64
branches = [(256, 8, 6, 8, None), (128, 16, 12, 8, None), (64, 32, 24, 8, None), (32, 64, 48, 4, None), (32, 64, 48, 4, 3)]
keep, outs, nblk = [], [], []
items = (hip.ChainItem * len(branches))()
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
    it.workspace = wsp.data_ptr()
    info = (C.c_int32 * 4)()
    lib.rsg_basic_chain_supported(Cc, H, W, L, Bn, info)
    nblk.append(info[3])
    if after is not None:
        fl = torch.zeros(3 * info[3], dtype=torch.int32, device="cuda")
        keep.append(fl)
        items[after].done_flags = fl.data_ptr()
        it.wait_flags = fl.data_ptr()


def report(tag, sw, L, mfma_cycles_per_layer):
    """sw: [workgroups, 16 waves, 64] float; prints medians over workgroups"""
    nw = int(np.count_nonzero(sw[len(sw) // 2, :, 0]))
    tot = np.median(sw[:, 0, 62] - sw[:, 0, 0])
    first = sw[:, :nw, :]
    line = []
    ksum = esum = wsum = psum = 0.0
    for l in range(L):
        ks, ke, ee, le = (first[:, :, 3 + 4 * l + j] for j in range(4))
        prev = first[:, :, 2] if l == 0 else first[:, :, 6 + 4 * (l - 1)]
        # the workgroup's phase boundaries: first wave in, last wave out
        k0, k1 = ks.min(1), ke.max(1)
        pro = np.median(k0 - prev.max(1))
        kl = np.median(k1 - k0)
        kfirst, klast = np.median((ke - ks).min(1)), np.median((ke - ks).max(1))
        epi = np.median(ee.max(1) - k1)
        wb = np.median(le.max(1) - ee.max(1))
        ksum += kl; esum += epi; wsum += wb; psum += pro
        line.append(f"    layer {l}: prologue {pro:5.0f}  k-loop {kl:6.0f} (per wave {kfirst:.0f}..{klast:.0f}; matrix-pipe time of "
                    f"the SIMD's waves {mfma_cycles_per_layer:.0f} = {mfma_cycles_per_layer / kl * 100:.0f} %)  epilogue VALU {epi:5.0f}  "
                    f"barrier+write-back+barrier {wb:5.0f}")
    st = np.median(sw[:, 0, 2] - sw[:, 0, 0])
    fin = np.median(sw[:, 0, 62] - first[:, :, 6 + 4 * (L - 1)].max(1))
    print(f"  {tag}: {len(sw)} workgroups x {nw} waves, median workgroup {tot:.0f} cycles: wait+zero+stage {st:.0f} ({st / tot * 100:.0f} %), "
          f"prologues {psum:.0f} ({psum / tot * 100:.0f} %), k-loops {ksum:.0f} ({ksum / tot * 100:.0f} %), epilogue VALU {esum:.0f} "
          f"({esum / tot * 100:.0f} %), barriers+write-back {wsum:.0f} ({wsum / tot * 100:.0f} %), final store {fin:.0f} ({fin / tot * 100:.0f} %); "
          f"matrix pipe busy {L * mfma_cycles_per_layer / tot * 100:.0f} % of the workgroup's life")
    print("\n".join(line), flush=True)


def mfma_per_layer(Cc, H, W, L, lib):
    """Matrix-pipe cycles one SIMD spends per layer: 16-pixel groups x 16-channel tiles x (32-channel block, tap) steps x
    16 cycles / 4 SIMDs (32-pixel groups x 32-channel tiles x 16-k steps x 32 cycles for a -DRSG_CHAIN16=0 build)"""
    info = (C.c_int32 * 4)()
    lib.rsg_basic_chain_supported(Cc, H, W, L, Bn, info)
    th, ni = info[0], info[1]
    rows = H if th == H else (th + L)          # two tiles: halo on one side only
    g = 16 if hip.lib().rsg_chain_pack_kind(Cc) == 16 else 32
    groups = (ni * rows * W + g - 1) // g
    ksteps, cyc = (Cc * 9 // 32, 16) if g == 16 else (Cc * 9 // 16, 32)
    return groups * (Cc // g) * ksteps * cyc / 4.0


def run(its, n, blocks, what):
    total = sum(blocks)
    stamps = torch.zeros(total * 16 * SL, dtype=torch.int64, device="cuda")
    lib.rsg_chain_set_stamp_buffer(C.c_void_p(stamps.data_ptr()))
    for _ in range(3):
        assert lib.rsg_basic_chain_group(its, n, 0, None) == 0, lib.rsg_last_error()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.rsg_basic_chain_group(its, n, 0, None)
    e1.record()
    torch.cuda.synchronize()
    print(f"{what}: {total} workgroups, {e0.elapsed_time(e1) * 1e3:.1f} us by events")
    return stamps.cpu().numpy().reshape(total, 16, SL).astype(np.float64)


sw = run(items, len(branches), nblk, "grouped launch (stage-4 module)")
first = 0
for (Cc, H, W, L, after), n in zip(branches, nblk):
    report(f"C={Cc} {H}x{W} L={L}{' (waits on hand-off)' if after is not None else ''}", sw[first:first + n], L, mfma_per_layer(Cc, H, W, L, lib))
    first += n
if os.environ.get("ALONE", "1") != "0":
    for k, (Cc, H, W, L, after) in enumerate(branches[:4]):
        one = (hip.ChainItem * 1)()
        C.memmove(one, C.byref(items[k]), C.sizeof(hip.ChainItem))
        one[0].done_flags = None
        one[0].wait_flags = None
        sw1 = run(one, 1, [nblk[k]], f"alone: C={Cc}")
        report(f"C={Cc} {H}x{W} L={L} alone", sw1, L, mfma_per_layer(Cc, H, W, L, lib))
