"""Generates tests/golden/oks_nms.npz from the reference's own lib/nms/nms.py (imported from /root/reference in
THIS container only).  nms.py begins with `from .cpu_nms import cpu_nms` / `from .gpu_nms import gpu_nms`, Cython
extensions shipped as cpython-36/37 binaries that no interpreter here can load and that `oks_iou`, `oks_nms` and
`soft_oks_nms` never call; two empty placeholder modules take those names so that the file imports.  Inputs are
seeded; the fixture holds inputs and the reference's outputs only."""
import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
pkg = types.ModuleType("refnms")
pkg.__path__ = []
sys.modules["refnms"] = pkg
for name in ("cpu_nms", "gpu_nms"):
    m = types.ModuleType("refnms." + name)
    setattr(m, name, None)
    sys.modules["refnms." + name] = m
spec = importlib.util.spec_from_file_location("refnms.nms", "/root/reference/lib/nms/nms.py")
ref = importlib.util.module_from_spec(spec)
sys.modules["refnms.nms"] = ref
spec.loader.exec_module(ref)

rs = np.random.RandomState(77)
out = {}
cases = []
for case, (n, J, spread, vis) in enumerate([(12, 17, 6.0, None), (25, 17, 15.0, 0.2), (1, 17, 1.0, None),
                                            (30, 17, 3.0, 0.5), (8, 14, 10.0, None)]):
    base = rs.uniform(50, 400, size=(max(1, n // 3), J, 2))
    who = rs.randint(0, base.shape[0], size=n)
    xy = base[who] + rs.normal(0, spread, size=(n, J, 2))
    conf = rs.uniform(0, 1, size=(n, J, 1))
    kpts = np.concatenate([xy, conf], axis=2)
    scores = rs.uniform(0.1, 1.0, size=n)
    areas = rs.uniform(2000, 40000, size=n)
    sig = None if J == 17 else rs.uniform(0.025, 0.1, size=J)
    db = [{"score": scores[i], "keypoints": kpts[i], "area": areas[i]} for i in range(n)]
    thr = 0.9 if case % 2 == 0 else 0.5
    keep = ref.oks_nms(db, thr, sig, vis)
    soft = ref.soft_oks_nms(db, thr, sig, vis)
    iou = ref.oks_iou(kpts[0].flatten(), kpts.reshape(n, -1), areas[0], areas, sig, vis)
    out.update({f"c{case}_kpts": kpts, f"c{case}_scores": scores, f"c{case}_areas": areas,
                f"c{case}_sigmas": np.zeros(0) if sig is None else sig, f"c{case}_thresh": thr,
                f"c{case}_vis": np.nan if vis is None else vis, f"c{case}_keep": np.asarray(keep, np.int64),
                f"c{case}_soft": np.asarray(soft, np.int64), f"c{case}_iou0": iou})
    cases.append(case)
out["n_cases"] = len(cases)
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "oks_nms.npz")
np.savez_compressed(dst, **out)
print("wrote", os.path.normpath(dst), {k: v.shape for k, v in out.items() if k.endswith("keep")})
