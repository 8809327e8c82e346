"""Generates tests/golden/decode_flip.npz from the reference's own numpy code (imported from /root/reference in THIS
container only): `flip_back` (lib/utils/transforms.py:23-37), `get_max_preds` (lib/core/inference.py:21-49) and the
numpy-only helpers of the crop transform (`get_dir`, `get_3rd_point`, `affine_transform`, transforms.py:103-128).

Both files begin with `import cv2`, which this image does not have; none of the functions pinned here calls it, so an
EMPTY placeholder module takes the name for the import (the recipe of make_crop_goldens.py).  What does call OpenCV
(`get_affine_transform` -> cv2.getAffineTransform, `transform_preds`, cv2.warpAffine) stays unpinned and is said so in
DESIGN.md section 3.  Inputs are seeded; the fixture holds inputs and the reference's outputs only."""
import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/lib"
sys.modules.setdefault("cv2", types.ModuleType("cv2"))          # empty: any call into it raises AttributeError
pkg = types.ModuleType("utils")
pkg.__path__ = []
sys.modules["utils"] = pkg


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


tr = load("utils.transforms", os.path.join(REF, "utils", "transforms.py"))
inf = load("refcore_inference", os.path.join(REF, "core", "inference.py"))

rs = np.random.RandomState(123)
out = {}
# ---- flip_back: COCO and CrowdPose pairs (lib/dataset/coco.py:92, crowdpose.py:988-989) -----------------------
PAIRS = {"coco": [[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14], [15, 16]],
         "crowdpose": [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9], [10, 11]]}
for tag, J in (("coco", 17), ("crowdpose", 14)):
    hm = rs.normal(size=(3, J, 16, 12)).astype(np.float32)
    out[f"flip_{tag}_in"] = hm
    out[f"flip_{tag}_out"] = np.ascontiguousarray(tr.flip_back(hm.copy(), PAIRS[tag]))
    # the whole merge of lib/core/function.py:139-146 on the reference's flip_back: shift by one pixel, average
    a = rs.normal(size=(3, J, 16, 12)).astype(np.float32)
    fl = np.ascontiguousarray(tr.flip_back(hm.copy(), PAIRS[tag]))
    sh = fl.copy()
    sh[:, :, :, 1:] = fl.copy()[:, :, :, 0:-1]
    out[f"merge_{tag}_a"] = a
    out[f"merge_{tag}_shift"] = (a + sh) * 0.5
    out[f"merge_{tag}_noshift"] = (a + fl) * 0.5
# ---- get_max_preds: random maps plus the corner cases (ties -> first index, max <= 0 -> zeroed coordinates) ------
hm = rs.normal(size=(4, 17, 64, 48)).astype(np.float32)
hm[0, 0] = -1.0
hm[0, 1] = 0.0
hm[0, 1, 10, 7] = 2.0
hm[0, 1, 30, 20] = 2.0
hm[0, 2, 0, 5] = 9.0
hm[0, 3, 63, 47] = 9.0
hm[1, 4] = 0.0                                  # all-zero map: argmax 0, max 0 -> masked
preds, maxvals = inf.get_max_preds(hm)
out["maxpreds_in"], out["maxpreds_xy"], out["maxpreds_val"] = hm, preds, maxvals
# ---- numpy-only transform helpers ------------------------------------------------------------------------------
pts = rs.uniform(-50, 300, size=(16, 2)).astype(np.float32)
rots = rs.uniform(-np.pi, np.pi, size=16)
out["dir_in"], out["dir_rot"] = pts, rots
out["dir_out"] = np.stack([np.asarray(tr.get_dir(p, r), np.float64) for p, r in zip(pts, rots)])
out["third_out"] = np.stack([tr.get_3rd_point(pts[i], pts[(i + 1) % 16]) for i in range(16)])
mats = rs.normal(size=(16, 2, 3))
out["aff_mats"] = mats
out["aff_out"] = np.stack([tr.affine_transform(pts[i], mats[i]) for i in range(16)])
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "decode_flip.npz")
np.savez_compressed(dst, **out)
print("wrote", os.path.normpath(dst), {k: v.shape for k, v in out.items()})
