#!/usr/bin/env python3
"""Generate tests/golden/train_step_rsgnet_w32_256x192_coco.npz by running the REFERENCE's pose_rsgnet in train() mode
with the losses of its own training loop (`rsgnet_train`, /root/reference/lib/core/function.py:239-323: the reference's
JointsMSELoss, torch.nn.BCELoss, the relation target built from the target heat maps) and torch autograd on the CPU
(build container only; the fixture holds seeds and expected outputs, nothing of the reference's source).

    python oracle/make_rsgnet_train_goldens.py

Stored: loss, the four outputs (sub-sampled), for EVERY parameter the gradient's sum and sum of squares, a dozen
gradients in full, a few running statistics.  Cross-checks oracle/train_step_oracle.rsgnet_train_step against it."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import reference_import as ri          # noqa: E402
from oracle import train_step_oracle as tso        # noqa: E402
from rsgnet_amd.config import make_cfg             # noqa: E402
from rsgnet_amd.weights import synth_input, synth_state_dict  # noqa: E402

TAG = "rsgnet_w32_256x192_coco"
BATCH, X_SEED, T_SEED, A_SEED, L_SEED, W_SEED = 2, 31, 32, 33, 34, 0
FULL = ["conv1.weight", "layer1.0.conv1.weight", "stage3.1.branches.1.2.bn2.bias", "stage4.2.branches.0.3.conv2.weight",
        "multi_final_layer.weight", "multi_final_layer.bias", "vis_conv.0.weight", "type_features", "type_fc.0.weight",
        "type_fc.1.weight", "type_conv.0.weight", "loc_conv.0.weight", "contact_conv.1.bias", "predict_contact_net.0.weight",
        "relation_head.g.weight", "relation_head.g.bias", "relation_head.W.0.weight", "relation_head.W.1.weight",
        "kpt_net.0.weight", "predict_convtranspose.0.weight", "predict_net.1.weight", "final_layer.weight", "final_layer.bias",
        "kt_machine.matrix_limb", "kt_machine.kpt_transformer.0.weight", "kt_machine.kpt_transformer.2.bias", "limbs_net.0.weight"]
STATS = ["bn1.running_mean", "stage4.2.branches.0.3.bn2.running_var", "vis_conv.1.running_mean", "type_fc.1.running_var",
         "type_conv.1.running_mean", "loc_conv.1.running_var", "predict_convtranspose.1.running_mean", "limbs_net.1.running_var"]


def inputs():
    x = synth_input(BATCH, (192, 256), X_SEED)
    target, tw = tso.gaussian_targets(BATCH, 17, (128, 96), T_SEED)
    all_t, all_w = tso.gaussian_targets(BATCH, 17, (128, 96), A_SEED)
    limbs_t = tso.limb_targets(BATCH, 18, (128, 96), L_SEED)
    return x, target, tw, all_t, all_w, limbs_t


def main():
    if not ri.available():
        sys.exit("the reference is not mounted; goldens can only be regenerated in the build container")
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ref_loss", "/root/reference/lib/core/loss.py")
    ref_loss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_loss)
    cfg = make_cfg("pose_rsgnet", 32, (192, 256), 17)
    ref = ri.build(cfg)
    template = {k: v.clone() for k, v in ref.state_dict().items()}
    bn = dict(np.load(os.path.join(ROOT, "tests", "golden", f"bnstats_{TAG}.npz")))
    sd = synth_state_dict(template, W_SEED, bn)
    x, target, tw, all_t, all_w, limbs_t = inputs()

    def run(dtype):
        ref.to(dtype)
        ref.load_state_dict(sd)
        ref.zero_grad()
        ref.train()
        # function.py:261-270
        person, _ = torch.max(target.to(dtype), dim=1)
        b, h, w = person.size()
        person = torch.nn.functional.interpolate(person.reshape(b, 1, h, w), scale_factor=1 / 2, mode="bilinear", align_corners=True)
        person = torch.squeeze(person)
        b, h, w = person.size()
        person = person.reshape(b, 1, h * w)
        rel_t = torch.matmul(person.permute(0, 2, 1), person)
        multi, out, limbs, rel = ref(x.to(dtype), rel_t)
        crit = ref_loss.JointsMSELoss(use_target_weight=True)
        loss = crit(multi, all_t.to(dtype), all_w.to(dtype)) + crit(out, target.to(dtype), tw.to(dtype)) \
            + 0.01 * torch.nn.BCELoss()(limbs, limbs_t.to(dtype)) + 0.001 * torch.mean(rel)       # :272-299
        loss.backward()
        grads = {k: p.grad.detach().double().clone() for k, p in ref.named_parameters() if p.grad is not None}
        stats = {k: v.detach().double().clone() for k, v in ref.state_dict().items() if k in STATS}
        return loss.item(), [t.detach().double().clone() for t in (multi, out, limbs, rel)], grads, stats

    l32, o32, g32, _ = run(torch.float32)
    l64, o64, g64, s64 = run(torch.float64)
    gold = {"loss": np.float64(l64), "loss_fp32": np.float64(l32), "multi_sub": o64[0][:, :, ::4, ::4].float().numpy(),
            "kpt_sub": o64[1][:, :, ::4, ::4].float().numpy(), "limbs_sub": o64[2][:, :, ::4, ::4].float().numpy(),
            "relation_mse": o64[3].numpy()}
    names = list(g64)
    gold["grad_names"] = np.array(names)
    gold["grad_sum"] = np.array([g64[k].sum().item() for k in names])
    gold["grad_sqsum"] = np.array([(g64[k] ** 2).sum().item() for k in names])
    gold["ref_fp32_relL2"] = np.array([((g32[k] - g64[k]).norm() / (g64[k].norm() + 1e-300)).item() for k in names])
    for k in FULL:
        gold["grad:" + k] = g64[k].float().numpy()
    for k in STATS:
        gold["stat:" + k] = s64[k].float().numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"train_step_{TAG}.npz"), **gold)
    # ---- the oracle (fp64) against the reference (fp64)
    oloss, outs, ograds, ostats = tso.rsgnet_train_step(sd, cfg, x, target, tw, all_t, all_w, limbs_t)
    print(f"loss: reference fp64 {l64:.10e}, fp32 {l32:.10e}, oracle {oloss.item():.10e}")
    for nm, a, b in zip(("multi", "kpt", "limbs", "relation_mse"), outs, o64):
        print(nm, "max-abs oracle vs reference:", float((a - b).abs().max()))
    missing = [k for k in names if ograds.get(k) is None]
    assert not missing, missing
    assert set(names) == {k for k, v in ograds.items() if v is not None}, set(names) ^ set(ograds)
    worst = max((float((ograds[k] - g64[k]).abs().max()) / (float(g64[k].abs().max()) + 1e-300), k) for k in names)
    print("worst relative gradient difference oracle vs reference (fp64) over", len(names), "parameters:", worst)
    print("reference fp32 vs its own fp64 run, worst relative L2:", float(gold["ref_fp32_relL2"].max()))
    for k in STATS:
        print(k, float((ostats[k] - s64[k]).abs().max()))
    assert worst[0] < 1e-8 and abs(oloss.item() - l64) < 1e-11


if __name__ == "__main__":
    main()
