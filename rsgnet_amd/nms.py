"""OKS-NMS and rescoring of the evaluation tail: lib/nms/nms.py:75-124,127-180 and
lib/dataset/coco.py:322-357.  A few dozen candidates per picture: host logic, vectorised over candidates (the
reference loops over them in Python)."""
from __future__ import annotations

from typing import Optional

import numpy as np

COCO_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0


def oks_iou(g, d, a_g, a_d, sigmas=None, in_vis_thre: Optional[float] = None):
    """nms.py:75-96: similarity of pose ``g`` [3J] with every row of ``d`` [n,3J]; areas a_g, a_d [n].
    As in the reference, ``in_vis_thre`` selects joints by the CANDIDATE's confidences only."""
    g = np.asarray(g, np.float64).reshape(-1, 3)
    d = np.asarray(d, np.float64).reshape(len(a_d), g.shape[0], 3)
    sig = COCO_SIGMAS if not isinstance(sigmas, np.ndarray) else sigmas
    var = (sig * 2) ** 2
    e = ((d[:, :, 0] - g[:, 0]) ** 2 + (d[:, :, 1] - g[:, 1]) ** 2) / var
    e = e / ((a_g + np.asarray(a_d, np.float64)) / 2 + np.spacing(1))[:, None] / 2
    if in_vis_thre is None:
        return np.exp(-e).sum(axis=1) / e.shape[1] if e.shape[1] else np.zeros(len(a_d))
    use = d[:, :, 2] > in_vis_thre
    cnt = use.sum(axis=1)
    total = np.where(use, np.exp(-e), 0.0).sum(axis=1)
    return np.where(cnt > 0, total / np.maximum(cnt, 1), 0.0)


def _arrays(kpts_db):
    scores = np.array([k["score"] for k in kpts_db])
    kpts = np.array([np.asarray(k["keypoints"]).flatten() for k in kpts_db])
    areas = np.array([k["area"] for k in kpts_db])
    return scores, kpts, areas


def oks_nms(kpts_db, thresh, sigmas=None, in_vis_thre=None):
    """nms.py:99-124: indices kept, best score first; ``kpts_db`` is the reference's list of dicts."""
    if len(kpts_db) == 0:
        return []
    scores, kpts, areas = _arrays(kpts_db)
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i, rest = order[0], order[1:]
        keep.append(i)
        ovr = oks_iou(kpts[i], kpts[rest], areas[i], areas[rest], sigmas, in_vis_thre)
        order = rest[ovr <= thresh]
    return keep


def soft_oks_nms(kpts_db, thresh, sigmas=None, in_vis_thre=None, max_dets=20):
    """nms.py:138-180 with the gaussian rescoring of nms.py:127-135."""
    if len(kpts_db) == 0:
        return []
    scores, kpts, areas = _arrays(kpts_db)
    order = scores.argsort()[::-1]
    scores = scores[order]
    keep = np.zeros(max_dets, dtype=np.intp)
    cnt = 0
    while order.size > 0 and cnt < max_dets:
        i, order = order[0], order[1:]
        ovr = oks_iou(kpts[i], kpts[order], areas[i], areas[order], sigmas, in_vis_thre)
        scores = scores[1:] * np.exp(-ovr ** 2 / thresh)
        by = scores.argsort()[::-1]
        order, scores = order[by], scores[by]
        keep[cnt] = i
        cnt += 1
    return keep[:cnt]


def rescore_and_nms(img_kpts, num_joints, in_vis_thre, oks_thre, soft_nms=False):
    """coco.py:329-357 for one picture: rescoring in place (box score x mean confident-joint score), then
    (soft) OKS-NMS; returns the surviving entries (all of them when NMS keeps none, as the reference)."""
    for p in img_kpts:
        conf = np.asarray(p["keypoints"])[:num_joints, 2]
        sel = conf[conf > in_vis_thre]
        mean_conf = sel.sum() / len(sel) if len(sel) else 0
        p["score"] = mean_conf * p["score"]
    keep = (soft_oks_nms if soft_nms else oks_nms)(list(img_kpts), oks_thre)
    return list(img_kpts) if len(keep) == 0 else [img_kpts[k] for k in keep]
