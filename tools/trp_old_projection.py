#!/usr/bin/env python3
"""(binaries under tools/_build/: `bash tools/build_reproducers.sh` rebuilds them from the sources; none is committed)
The TRP projection kernel as it stood before the fix (tools/trp_old_projection.hip), alone through its own .so, beside a
model's forward graph on another stream; variants: the plain build and one with every s_waitcnt forced to zero
(-mllvm -amdgpu-waitcnt-forcezero).  Build:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast [-mllvm -amdgpu-waitcnt-forcezero] -shared -fPIC \\
        -o tools/_build/libtrp_old_{plain,forcezero}.so tools/trp_old_projection.hip"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import load_case

here = os.path.dirname(os.path.abspath(__file__))
cfg, model, sd, x, gold = load_case("rsgnet_w32_256x192_coco")
model = model.cuda().eval()
xc = torch.cat([x] * 4).cuda()
side = torch.cuda.Stream()
B, S, C = 2, 3072, 32
g = torch.Generator().manual_seed(0)
e = torch.randn(B, S, C, generator=g).cuda()
w = (torch.randn(C, C, generator=g) * 0.3).cuda()
bias = torch.randn(C, generator=g).cuda()
y = torch.zeros(B, S, C, device="cuda")
part = torch.zeros(B * (S // 256 + 1) * 16, device="cuda")
with torch.no_grad():
    with torch.cuda.stream(side):
        model(xc); model(xc)
    torch.cuda.synchronize()
    for variant in (sys.argv[1:] or ["plain", "forcezero", "plain"]):
        lib = ctypes.CDLL(os.path.join(here, "_build", f"libtrp_old_{variant}.so"))
        lib.old_project_launch.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2 + [ctypes.c_void_p]

        def run():
            rc = lib.old_project_launch(e.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), part.data_ptr(), S, B, torch.cuda.current_stream().cuda_stream)
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
            print(f"old projection kernel, {variant} build, {'beside the model graph' if contended else 'alone'}: {bad} of 600 runs differ", flush=True)
