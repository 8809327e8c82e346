#!/usr/bin/env python3
"""GPU box only: builds tools/probes/mfma_shape_probe.hip and times the C = 32 k-loop on both MFMA shapes (see the .hip)."""
import ctypes as C
import subprocess
import sys

import numpy as np
import torch

so = "/tmp/mfma_shape_probe.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-mllvm",
                       "-amdgpu-mfma-vgpr-form", "tools/probes/mfma_shape_probe.hip", "-o", so])
lib = C.CDLL(so)
blocks = 256
zero = len(sys.argv) > 1 and sys.argv[1] == "zero"
w = (torch.zeros if zero else torch.randn)(16 * 18 * 64 * 8, device="cuda").half() * 0.05
rnd = (torch.zeros if zero else torch.randn)(4096 * 8, device="cuda").half()
out = torch.zeros(blocks * 512, device="cuda")
clk = torch.zeros(blocks * 2, dtype=torch.int64, device="cuda")
names = {0: "32x32x16 NT=7, 8 waves", 1: "16x16x32 NG=14, 8 waves, ring 3 taps", 2: "32x32x16 NT=7, 4 waves",
         3: "16x16x32 NG=14, 4 waves", 4: "16x16x32 NG=14, 8 waves, ring 2 taps", 5: "32x32x16 NT=6, 8 waves", 6: "16x16x32 NG=12, 8 waves", 7: "16x16x32 NG=14, 8 waves, B ring 7", 8: "16x16x32 NG=14, 8 waves, B ring 10"}
layers = 4000
for rep in range(2):
    for v in (0, 1, 7, 8, 4, 5, 6, 2, 3):
        for _ in range(2):
            assert lib.probe_run(v, C.c_void_p(w.data_ptr()), C.c_void_p(rnd.data_ptr()), C.c_void_p(out.data_ptr()), layers,
                                 C.c_void_p(clk.data_ptr()), blocks) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.probe_run(v, C.c_void_p(w.data_ptr()), C.c_void_p(rnd.data_ptr()), C.c_void_p(out.data_ptr()), layers, C.c_void_p(clk.data_ptr()), blocks)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        c = clk.cpu().numpy().reshape(blocks, 2).astype(np.float64)
        nw = 8 if v in (0, 1, 4, 5, 6, 7, 8) else 4
        px = {0: 7 * 32, 1: 14 * 16, 2: 7 * 32, 3: 14 * 16, 4: 14 * 16, 5: 6 * 32, 6: 12 * 16, 7: 14 * 16, 8: 14 * 16}[v]
        flop = blocks * nw * px * 32 * 32 * 9 * 2.0 * layers
        cyc = np.median(c[:, 0])
        ideal = px / 32 * 18 * 32 * (nw / 4) * layers          # MFMA cycles per SIMD
        print(f"{names[v]:42s}: {ms:8.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s  clock {np.median(c[:, 0] / c[:, 1]) * 100:6.0f} MHz  "
              f"wave-0 cycles {cyc:.3e}; matrix pipe busy {flop / 32768 * 32 / 1024 / (ms * 1e-3 * np.median(c[:, 0] / c[:, 1]) * 1e8) * 100:.1f} % of the held clock  out finite {bool(torch.isfinite(out).all())}", flush=True)
