"""The CPU oracle against the fixtures captured from the reference's own lib/models (make_goldens.py)."""
import numpy as np
import pytest
import torch

from oracle import pose_oracle as po
from tests.util import TAGS, like_gold, load_case, maxabs
from oracle.make_goldens import sample_index

TOL = 2e-5   # fp32 summation-order noise on O(1) outputs


@pytest.mark.parametrize("tag", list(TAGS))
def test_oracle_matches_reference_goldens(tag):
    cfg, _model, sd, x, gold = load_case(tag)
    ctx = po.Ctx(sd)
    if TAGS[tag][0] == "pose_hrnet":
        out = {"heatmaps": po.hrnet_forward(sd, cfg, x, ctx=ctx)}
    else:
        multi, kpt, limbs, rel = po.rsgnet_forward(sd, cfg, x, ctx=ctx)
        out = {"multi_kpt_scores": multi, "kpt_scores": kpt, "limbs_scores": limbs}
        idx = sample_index(rel.shape[1])
        assert maxabs(rel[:, idx[:, 0], idx[:, 1]], gold["relation_scores_samples"]) < TOL
        assert maxabs(rel.sum(dim=2), gold["relation_scores_rowsum"]) < 5e-2   # sums of 3072 / 6912 terms
        assert maxabs(like_gold(ctx.taps["final_vis"][0], gold["final_vis_crop0"]), gold["final_vis_crop0"]) < 1e-4
        assert maxabs(like_gold(ctx.taps["relation_feats"][0], gold["relation_feats_crop0"]),
                      gold["relation_feats_crop0"]) < 1e-4
        assert maxabs(ctx.taps["limb_weights"], gold["limb_weights"]) < 1e-6
        p = torch.from_numpy(gold["relation_target_vec"])
        mse = po.rsgnet_forward(sd, cfg, x, relation_target=p[:, :, None] * p[:, None, :])[3]
        assert maxabs(mse, gold["relation_mse"]) < 1e-6
    for k, v in out.items():
        assert maxabs(like_gold(v[0], gold[k + "_crop0"]), gold[k + "_crop0"]) < TOL, k
        assert maxabs(v[1, :, ::4, ::4], gold[k + "_crop1_sub"]) < TOL, k
        b, c = v.shape[:2]
        assert np.array_equal(v.reshape(b, c, -1).argmax(2).numpy(), gold[k + "_argmax"]), k
    for s in (2, 3, 4):
        a = ctx.taps[f"stage{s}"][0]
        assert maxabs(a, gold[f"stage{s}_br0_crop0"]) < 2e-5 * max(1.0, float(np.abs(gold[f'stage{s}_br0_crop0']).max()))


def test_fp16_emulation_noise_floor():
    """Documented expectation: 16-bit MFMA operands put the deviation from the
    fp32 reference at ~3e-2 of the output's std (max-abs); the emulation must stay in that band."""
    cfg, _m, sd, x, gold = load_case("hrnet_w32_256x192")
    ref = torch.from_numpy(gold["heatmaps_crop0"])
    emu = po.hrnet_forward(sd, cfg, x[:1], emulate="fp16")[0]
    err = maxabs(emu, ref)
    assert err < 0.06 * float(ref.std()), err
