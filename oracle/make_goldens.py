#!/usr/bin/env python3
"""Generate tests/golden/* by running the REFERENCE's lib/models on CPU (build container only).

    python oracle/make_goldens.py [tag ...]

For each config tag it
  1. builds the reference model (oracle/reference_import.py) and fills it with the name-keyed
     synthetic weights (rsgnet_amd/weights.py),
  2. calibrates BatchNorm running statistics with one training-mode pass of the reference
     (momentum 1) over a seeded batch                       -> bnstats_<tag>.npz,
  3. runs the reference in eval mode on the seeded input    -> golden_<tag>.npz (outputs + taps
     captured with forward hooks on the reference's own modules),
  4. cross-checks oracle/pose_oracle.py (exact mode, and its calibrate mode) against the reference,
  5. writes the state_dict key/shape manifest               -> manifest_<tag>.json.

The fixtures are data (inputs are regenerated from seeds; expected outputs are stored).  Nothing of
the reference's source is copied.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pose_oracle as po          # noqa: E402
from oracle import reference_import as ri     # noqa: E402
from rsgnet_amd.config import make_cfg        # noqa: E402
from rsgnet_amd.weights import synth_input, synth_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

CONFIGS = {
    # tag: (name, width, image_wh, joints)
    "hrnet_w32_256x192": ("pose_hrnet", 32, (192, 256), 17),
    "rsgnet_w32_256x192_coco": ("pose_rsgnet", 32, (192, 256), 17),
    "rsgnet_w32_256x192_crowdpose": ("pose_rsgnet", 32, (192, 256), 14),
    "hrnet_w48_384x288": ("pose_hrnet", 48, (288, 384), 17),
    "rsgnet_w48_384x288_coco": ("pose_rsgnet", 48, (288, 384), 17),
    # experiments/crowdpose/hrnet/"rsgnet_w48_256x192_adam_lr1e-3 .yaml": RELATION_SUB_SAMPLE: true (line 42)
    "rsgnet_w48_256x192_crowdpose_sub": ("pose_rsgnet", 48, (192, 256), 14, dict(sub_sample=True)),
}
CAL_BATCH, CAL_SEED = 4, 100
EVAL_BATCH, EVAL_SEED = 2, 0
WEIGHT_SEED = 0
N_SAMPLES = 1024


def sample_index(S: int) -> np.ndarray:
    r = np.random.RandomState(7)
    return np.stack([r.randint(0, S, N_SAMPLES), r.randint(0, S, N_SAMPLES)], 1)


def _summ(t: torch.Tensor):
    """per-(b,channel) max value and flat argmax: a cheap checksum for crops stored only in part."""
    b, c = t.shape[:2]
    flat = t.reshape(b, c, -1)
    v, i = flat.max(dim=2)
    return v.numpy(), i.numpy().astype(np.int32)


def generate(tag: str) -> None:
    name, width, wh, joints = CONFIGS[tag][:4]
    cfg = make_cfg(name, width, wh, joints, **(CONFIGS[tag][4] if len(CONFIGS[tag]) > 4 else {}))
    is_rsg = name == "pose_rsgnet"
    t0 = time.time()
    ref = ri.build(cfg)
    template = {k: v.clone() for k, v in ref.state_dict().items()}

    # ---- 2. calibration with the reference itself -------------------------------------------
    ref.load_state_dict(synth_state_dict(template, WEIGHT_SEED))
    for m in ref.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.momentum = 1.0
    ref.train()
    xc = synth_input(CAL_BATCH, wh, CAL_SEED)
    with torch.no_grad():
        ref(xc)
    ref.eval()
    bn_stats = {k: v.numpy().copy() for k, v in ref.state_dict().items()
                if k.endswith("running_mean") or k.endswith("running_var")}
    np.savez(os.path.join(GOLD, f"bnstats_{tag}.npz"), **bn_stats)

    # oracle's calibrate mode must give the same statistics
    sd_cal = synth_state_dict(template, WEIGHT_SEED)
    ctx = po.Ctx(sd_cal, calibrate=True)
    (po.rsgnet_forward if is_rsg else po.hrnet_forward)(sd_cal, cfg, xc, ctx=ctx)
    worst = max(float(np.abs(sd_cal[k].numpy() - v).max() / (np.abs(v).max() + 1e-6))
                for k, v in bn_stats.items())
    print(f"[{tag}] oracle calibrate vs reference train-mode stats: worst rel diff {worst:.2e}")
    assert worst < 1e-3, worst

    # ---- 3. reference eval forward with hooks -------------------------------------------------
    sd = synth_state_dict(template, WEIGHT_SEED, bn_stats)
    ref.load_state_dict(sd)
    taps = {}

    def hook(key):
        def fn(_m, _inp, out):
            taps[key] = out
        return fn

    hs = [ref.layer1.register_forward_hook(hook("layer1"))]
    for s in (2, 3, 4):
        hs.append(getattr(ref, f"stage{s}").register_forward_hook(hook(f"stage{s}")))
    if is_rsg:
        hs.append(ref.predict_contact_net.register_forward_hook(hook("final_vis")))
        hs.append(ref.relation_head.register_forward_hook(hook("relation_head")))
        hs.append(ref.kt_machine.register_forward_hook(hook("limb_weights")))
        hs.append(ref.limbs_net.register_forward_hook(hook("limbs_features")))
    x = synth_input(EVAL_BATCH, wh, EVAL_SEED)
    with torch.no_grad():
        out = ref(x)
    for h in hs:
        h.remove()

    gold = {}
    if is_rsg:
        multi, kpt, limbs, rel = out
        outs = {"multi_kpt_scores": multi, "kpt_scores": kpt, "limbs_scores": limbs}
    else:
        outs = {"heatmaps": out}
    big = wh[0] * wh[1] > 192 * 256 and is_rsg      # 384x288 RSGNet: keep fixtures small, store every 2nd pixel
    for k, v in outs.items():
        gold[k + "_crop0"] = v[0, :, ::2, ::2].numpy() if big else v[0].numpy()
        gold[k + "_max"], gold[k + "_argmax"] = _summ(v)
        gold[k + "_crop1_sub"] = v[1, :, ::4, ::4].numpy()
    gold["layer1_crop0_sub"] = taps["layer1"][0, ::8, ::2, ::2].numpy()
    for s in (2, 3, 4):
        gold[f"stage{s}_br0_crop0"] = taps[f"stage{s}"][0][0].numpy()
    if is_rsg:
        S = rel.shape[1]
        idx = sample_index(S)
        gold["final_vis_crop0"] = taps["final_vis"][0, :, ::2, ::2].numpy() if big else taps["final_vis"][0].numpy()
        rf = taps["relation_head"][0][0]
        gold["relation_feats_crop0"] = rf[:, ::2, ::2].numpy() if big else rf.numpy()
        gold["relation_scores_rowsum"] = rel.sum(dim=2).numpy()
        gold["relation_scores_samples"] = rel[:, idx[:, 0], idx[:, 1]].numpy()
        gold["limb_weights"] = taps["limb_weights"].numpy()
        lfe = taps["limbs_features"][0]
        gold["limbs_features_crop0"] = lfe[:, ::2, ::2].numpy() if big else lfe.numpy()
        # training-tail variant: relation_target given -> per-sample MSE [B] (pose_rsgnet.py:1014-1018)
        r = np.random.RandomState(11)
        p = torch.from_numpy(r.uniform(0, 1, (EVAL_BATCH, S)).astype(np.float32))
        with torch.no_grad():
            mse = ref(x, relation_target=p[:, :, None] * p[:, None, :])[3]
        gold["relation_target_vec"] = p.numpy()
        gold["relation_mse"] = mse.numpy()
    np.savez(os.path.join(GOLD, f"golden_{tag}.npz"), **gold)

    # ---- 4. oracle (exact mode) against the reference ------------------------------------------
    ctx = po.Ctx(sd)
    if is_rsg:
        o = po.rsgnet_forward(sd, cfg, x, ctx=ctx)
        errs = {k: float((a - b).abs().max()) for k, a, b in
                zip(("multi", "kpt", "limbs", "rel"), o, out)}
        errs["final_vis"] = float((ctx.taps["final_vis"] - taps["final_vis"]).abs().max())
        errs["relation_feats"] = float((ctx.taps["relation_feats"] - taps["relation_head"][0]).abs().max())
        errs["limb_weights"] = float((ctx.taps["limb_weights"] - taps["limb_weights"]).abs().max())
    else:
        o = po.hrnet_forward(sd, cfg, x, ctx=ctx)
        errs = {"heatmaps": float((o - out).abs().max())}
    for s in (2, 3, 4):
        errs[f"stage{s}"] = float((ctx.taps[f"stage{s}"] - taps[f"stage{s}"][0]).abs().max())
    print(f"[{tag}] oracle vs reference max-abs: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    # fp32 summation-order noise only: the two agree to ~1e-6 of the tensor's scale
    scale = {"heatmaps": float(out.abs().max())} if not is_rsg else {}
    for s in (2, 3, 4):
        scale[f"stage{s}"] = float(taps[f"stage{s}"][0].abs().max())
    for k, v in errs.items():
        assert v < 2e-5 * max(1.0, scale.get(k, 1.0)), (k, v)

    mags = {k: float(v.abs().max()) for k, v in outs.items()}
    print(f"[{tag}] output |max|: {mags}; stage4 br0 |max| {float(taps['stage4'][0].abs().max()):.2f}")
    if is_rsg:
        print(f"[{tag}] relation score mean {float(rel.mean()):.3f} min {float(rel.min()):.3f} "
              f"max {float(rel.max()):.4f}; relation_feats |max| "
              f"{float(taps['relation_head'][0].abs().max()):.3f}")

    # ---- 5. manifest ------------------------------------------------------------------------------
    manifest = {k: list(v.shape) for k, v in template.items()}
    with open(os.path.join(GOLD, f"manifest_{tag}.json"), "w") as f:
        json.dump({"tag": tag, "config": list(CONFIGS[tag]), "weight_seed": WEIGHT_SEED,
                   "cal": [CAL_BATCH, CAL_SEED], "eval": [EVAL_BATCH, EVAL_SEED],
                   "torch": torch.__version__, "state_dict": manifest}, f)
    print(f"[{tag}] done in {time.time() - t0:.1f}s")


if __name__ == "__main__":
    if not ri.available():
        sys.exit("the reference is not mounted; goldens can only be regenerated in the build container")
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    for t in (sys.argv[1:] or list(CONFIGS)[:3]):
        generate(t)
