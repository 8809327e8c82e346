#!/usr/bin/env python3
"""(binaries under tools/_build/: `bash tools/build_reproducers.sh` rebuilds them from the sources; none is committed)
The ONE discriminating build DESIGN 9b' / VERDICT round 3 asked for: the TRP projection kernel as it stood (-O3 assembly of
tools/trp_old_projection.hip) with `s_nop 3` in front of EVERY v_pk_fma_f32, i.e. four wait states between the v_mov_b32 that
assembles a 64-bit operand and the packed FMA that reads it -- and nothing else changed (same registers, same loads, same
order).  If the dropped FMA term is a read-after-write window between those two, this build cannot fail; if it still fails
beside the model graph at the plain build's rate, the window is not the cause.  Both code objects are loaded through
hipModuleLoad (tools/_build/trp_old_{plain,nop_before_pk_fma}.co; recipe: hipcc -S --cuda-device-only, the edit, clang -x
assembler -target amdgcn-amd-amdhsa -mcpu=gfx950, ld.lld -shared).
Round 5: `bash tools/build_reproducers.sh && TRP_OLD_VARIANTS=plain,war_separated,plain python tools/trp_old_projection_nop.py`
runs the write-after-read build of tools/war_separate.py the same way (profiles/r05_trp_war_discriminator.txt)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import load_case

here = os.path.dirname(os.path.abspath(__file__))
hip = C.CDLL("libamdhip64.so")
cfg, model, sd, x, gold = load_case("rsgnet_w32_256x192_coco")
model = model.cuda().eval()
xc = torch.cat([x] * 4).cuda()
side = torch.cuda.Stream()
B, S, Cc = 2, 3072, 32
g = torch.Generator().manual_seed(0)
e = torch.randn(B, S, Cc, generator=g).cuda()
w = (torch.randn(Cc, Cc, generator=g) * 0.3).cuda()
bias = torch.randn(Cc, generator=g).cuda()
y = torch.zeros(B, S, Cc, device="cuda")
part = torch.zeros(B * (S // 256 + 1) * 16, device="cuda")
NAME = b"_Z20trp_w_project_kernelILi32EfEvPKviPKfS3_PfS4_i"
with torch.no_grad():
    with torch.cuda.stream(side):
        model(xc); model(xc)
    torch.cuda.synchronize()
    for variant in (os.environ.get("TRP_OLD_VARIANTS") or "plain,nop_before_pk_fma,plain").split(","):
        mod, fn = C.c_void_p(), C.c_void_p()
        assert hip.hipModuleLoad(C.byref(mod), os.path.join(here, "_build", f"trp_old_{variant}.co").encode()) == 0
        assert hip.hipModuleGetFunction(C.byref(fn), mod, NAME) == 0
        a = [C.c_void_p(e.data_ptr()), C.c_int(0), C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(y.data_ptr()),
             C.c_void_p(part.data_ptr()), C.c_int(S)]
        params = (C.c_void_p * len(a))(*[C.cast(C.pointer(v), C.c_void_p) for v in a])

        def run():
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = hip.hipModuleLaunchKernel(fn, S // 256, B, 1, 256, 1, 1, 0, st, params, None)
            assert rc == 0, rc
            return y.clone()

        ref = run()
        torch.cuda.synchronize()
        for contended in (False, True):
            bad = 0
            for rep in range(600):
                if contended and rep % 4 == 0:
                    with torch.cuda.stream(side):
                        for _ in range(3):
                            model(xc)
                out = run()
                if rep % 4 == 3:
                    torch.cuda.synchronize()
                if not torch.equal(out, ref):
                    bad += 1
            torch.cuda.synchronize()
            print(f"old projection kernel, {variant}, {'beside the model graph' if contended else 'alone'}: {bad} of 600 runs differ", flush=True)
        hip.hipModuleUnload(mod)
