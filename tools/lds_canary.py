#!/usr/bin/env python3
"""(binaries under tools/_build/: `bash tools/build_reproducers.sh` rebuilds them from the sources; none is committed)
tools/lds_canary.hip beside a model's forward graph on another stream: does anything write into the canary's LDS or
damage its registers?  (hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/_build/liblds_canary.so tools/lds_canary.hip)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import load_case

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liblds_canary.so"))
lib.canary_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
tag = sys.argv[1] if len(sys.argv) > 1 else "rsgnet_w32_256x192_coco"
cfg, model, sd, x, gold = load_case(tag)
model = model.cuda().eval()
xc = torch.cat([x] * 4).cuda()
side = torch.cuda.Stream()
report = torch.zeros(512, dtype=torch.int32, device="cuda")
with torch.no_grad():
    with torch.cuda.stream(side):
        model(xc); model(xc)
    torch.cuda.synchronize()
    for lds_bytes in (4096, 16384, 65536):
        for contended in (False, True):
            report.zero_()
            torch.cuda.synchronize()
            for rep in range(60):
                if contended and rep % 2 == 0:
                    with torch.cuda.stream(side):
                        for _ in range(3):
                            model(xc)
                rc = lib.canary_launch(report.data_ptr(), 1024, lds_bytes, 40, torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
                if rep % 4 == 3:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            r = report.cpu().numpy().astype("uint32")
            print(f"LDS {lds_bytes} B, {'beside the model graph' if contended else 'alone'}: LDS words changed {r[0]}, registers changed {r[1]}")
            for s in range(min(int(r[0]), 12)):
                b, i, v, it = r[4 + 4 * s: 8 + 4 * s]
                print(f"     block {b} word {i} (byte {4 * i}) = 0x{v:08x} (expected 0x{(0xC0DE0000 ^ (b << 20) ^ i) & 0xffffffff:08x}) round {it}")
            for s in range(min(int(r[1]), 8)):
                b, t, k, v = r[300 + 4 * s: 304 + 4 * s]
                print(f"     block {b} thread {t} register {k} = 0x{v:08x}")
