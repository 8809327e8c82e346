"""Shared helpers for the tests (CPU and GPU)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

TAGS = {
    "hrnet_w32_256x192": ("pose_hrnet", 32, (192, 256), 17),
    "rsgnet_w32_256x192_coco": ("pose_rsgnet", 32, (192, 256), 17),
    "rsgnet_w32_256x192_crowdpose": ("pose_rsgnet", 32, (192, 256), 14),
    "hrnet_w48_384x288": ("pose_hrnet", 48, (288, 384), 17),
    "rsgnet_w48_384x288_coco": ("pose_rsgnet", 48, (288, 384), 17),
    # experiments/crowdpose/hrnet/"rsgnet_w48_256x192_adam_lr1e-3 .yaml": the shipped RELATION_SUB_SAMPLE: true config
    "rsgnet_w48_256x192_crowdpose_sub": ("pose_rsgnet", 48, (192, 256), 14, dict(sub_sample=True)),
}


def case_cfg(tag):
    from rsgnet_amd import make_cfg
    name, width, wh, joints = TAGS[tag][:4]
    return make_cfg(name, width, wh, joints, **(TAGS[tag][4] if len(TAGS[tag]) > 4 else {}))


def load_case(tag):
    """(cfg, model, state_dict, input, golden dict) for a golden tag; weights by the synthetic rule."""
    from rsgnet_amd import make_cfg, models
    from rsgnet_amd.weights import synth_input, synth_state_dict
    name, width, wh, joints = TAGS[tag][:4]
    cfg = case_cfg(tag)
    model = getattr(models, name).get_pose_net(cfg, is_train=False)
    bn = dict(np.load(os.path.join(GOLD, f"bnstats_{tag}.npz")))
    sd = synth_state_dict(model.state_dict(), 0, bn)
    model.load_state_dict(sd)
    model.eval()
    gold = dict(np.load(os.path.join(GOLD, f"golden_{tag}.npz")))
    x = synth_input(2, wh, 0)
    return cfg, model, sd, x, gold


def manifest(tag):
    with open(os.path.join(GOLD, f"manifest_{tag}.json")) as f:
        return json.load(f)


def maxabs(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max())


def like_gold(t, g):
    """Fixtures of the 384x288 RSGNet keep every 2nd pixel of the big maps (make_goldens.py `big`)."""
    t = t.detach().cpu() if torch.is_tensor(t) else torch.as_tensor(t)
    if tuple(t.shape) == tuple(g.shape):
        return t
    t2 = t[..., ::2, ::2]
    assert tuple(t2.shape) == tuple(g.shape), (tuple(t.shape), tuple(g.shape))
    return t2
