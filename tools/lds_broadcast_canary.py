#!/usr/bin/env python3
"""(binaries under tools/_build/: `bash tools/build_reproducers.sh` rebuilds them from the sources; none is committed)
tools/lds_broadcast_canary.hip beside a model's forward graph on another stream: does the reduced form of the TRP
projection's access pattern (LDS table, uniform-address reads feeding an FMA chain) go wrong the way the kernel did?
(hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -shared -fPIC -o tools/_build/liblds_broadcast_canary.so tools/lds_broadcast_canary.hip)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import load_case

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liblds_broadcast_canary.so"))
lib.bcast_launch.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
cfg, model, sd, x, gold = load_case(sys.argv[1] if len(sys.argv) > 1 else "rsgnet_w32_256x192_coco")
model = model.cuda().eval()
xc = torch.cat([x] * 4).cuda()
side = torch.cuda.Stream()
blocks = 24                                    # the TRP projection at 2 crops: 24 workgroups of 256 positions
g = torch.Generator().manual_seed(0)
w = (torch.randn(32 * 32 + 32, generator=g) * 0.3).cuda()
xs = torch.randn(blocks * 256, 32, generator=g).cuda()
report = torch.zeros(256, dtype=torch.int32, device="cuda")
with torch.no_grad():
    with torch.cuda.stream(side):
        model(xc); model(xc)
    torch.cuda.synchronize()
    for mode in (0, 1):
        for contended in (False, True):
            report.zero_()
            torch.cuda.synchronize()
            for rep in range(400):
                if contended and rep % 4 == 0:
                    with torch.cuda.stream(side):
                        for _ in range(3):
                            model(xc)
                rc = lib.bcast_launch(w.data_ptr(), xs.data_ptr(), report.data_ptr(), blocks, 1, mode, torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
                if rep % 4 == 3:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            r = report.cpu().numpy().astype("uint32")
            print(f"{'scalar-cache' if mode else 'LDS broadcast'} form, {'beside the model graph' if contended else 'alone'}: 400 launches, wrong results {r[0]}")
            for s_ in range(min(int(r[0]), 8)):
                b, t, co, it = r[4 + 4 * s_: 8 + 4 * s_]
                print(f"     block {b} thread {t} (lane {t % 64}) output {co}")
