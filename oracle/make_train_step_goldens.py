#!/usr/bin/env python3
"""Generate tests/golden/train_step_hrnet_w32_256x192.npz by running the REFERENCE's pose_hrnet in train() mode with
the reference's own JointsMSELoss and torch autograd on CPU (build container only; nothing of the reference's source
is copied -- the fixture holds inputs' seeds and expected outputs).

    python oracle/make_train_step_goldens.py

Stored: the loss, the heat maps, for EVERY parameter the gradient's sum and sum of squares, a handful of gradients in
full, a handful of running statistics after the step.  Also cross-checks oracle/train_step_oracle.py against it."""
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

TAG = "hrnet_w32_256x192"
BATCH, X_SEED, T_SEED, W_SEED = 2, 21, 22, 0
FULL = ["conv1.weight", "bn1.weight", "bn1.bias", "layer1.0.conv1.weight", "layer1.0.downsample.1.weight",
        "stage2.0.fuse_layers.0.1.0.weight", "stage3.0.fuse_layers.2.0.0.0.weight", "stage3.1.branches.1.2.bn2.bias",
        "stage4.2.branches.0.3.conv2.weight", "transition3.3.0.1.weight", "final_layer.weight", "final_layer.bias"]
STATS = ["bn1.running_mean", "bn1.running_var", "layer1.3.bn3.running_var", "stage3.2.branches.2.1.bn1.running_mean",
         "stage4.1.fuse_layers.3.0.2.1.running_var", "stage4.2.branches.3.3.bn2.running_mean"]


def main():
    if not ri.available():
        sys.exit("the reference is not mounted; goldens can only be regenerated in the build container")
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ref_loss", "/root/reference/lib/core/loss.py")
    ref_loss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_loss)
    cfg = make_cfg("pose_hrnet", 32, (192, 256), 17)
    ref = ri.build(cfg)
    template = {k: v.clone() for k, v in ref.state_dict().items()}
    bn = dict(np.load(os.path.join(ROOT, "tests", "golden", f"bnstats_{TAG}.npz")))
    sd = synth_state_dict(template, W_SEED, bn)
    x = synth_input(BATCH, (192, 256), X_SEED)
    target, tw = tso.gaussian_targets(BATCH, 17, (64, 48), T_SEED)

    def run(dtype):
        """forward + JointsMSELoss + backward of the reference in train() mode, in the given precision"""
        ref.to(dtype)
        ref.load_state_dict(sd)
        ref.zero_grad()
        ref.train()
        out = ref(x.to(dtype))
        loss = ref_loss.JointsMSELoss(use_target_weight=True)(out, target.to(dtype), tw.to(dtype))
        loss.backward()
        grads = {k: p.grad.detach().double().clone() for k, p in ref.named_parameters()}
        stats = {k: v.detach().double().clone() for k, v in ref.state_dict().items() if k in STATS}
        return loss.item(), out.detach().double().clone(), grads, stats

    # The reference trains in fp32, and in fp32 its own gradients scatter by up to several per cent of a tensor's
    # max-abs around the fp64 run of the same code (ReLU decisions near zero, batch of 2).  The fixture therefore
    # stores the fp64 run as the truth and the fp32 run's deviation from it as the reference's own noise floor.
    l32, h32, g32, _ = run(torch.float32)
    l64, h64, g64, s64 = run(torch.float64)
    gold = {"loss": np.float64(l64), "loss_fp32": np.float64(l32), "heatmaps": h64.float().numpy()}
    names = list(g64)
    gold["grad_names"] = np.array(names)
    gold["grad_sum"] = np.array([g64[k].sum().item() for k in names])
    gold["grad_sqsum"] = np.array([(g64[k] ** 2).sum().item() for k in names])
    gold["grad_absmax"] = np.array([g64[k].abs().max().item() for k in names])
    gold["ref_fp32_noise"] = np.array([((g32[k] - g64[k]).abs().max() / (g64[k].abs().max() + 1e-300)).item() for k in names])
    gold["ref_fp32_relL2"] = np.array([((g32[k] - g64[k]).norm() / (g64[k].norm() + 1e-300)).item() for k in names])
    for k in FULL:
        gold["grad:" + k] = g64[k].float().numpy()
    for k in STATS:
        gold["stat:" + k] = s64[k].float().numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"train_step_{TAG}.npz"), **gold)
    # ---- the oracle (fp64) against the reference (fp64)
    oloss, oheat, ograds, ostats = tso.hrnet_train_step(sd, cfg, x, target, tw)
    print(f"loss: reference fp64 {l64:.10e}, reference fp32 {l32:.10e}, oracle {oloss.item():.10e}")
    print("heat max-abs diff oracle vs reference (fp64):", float((oheat - h64).abs().max()))
    worst = max(float((ograds[k] - g64[k]).abs().max()) / (float(g64[k].abs().max()) + 1e-300) for k in names)
    print("worst relative gradient difference oracle vs reference (fp64) over", len(names), "parameters:", worst)
    print("reference fp32 vs its own fp64 run: worst max-abs/max-abs", float(gold["ref_fp32_noise"].max()),
          "worst relative L2", float(gold["ref_fp32_relL2"].max()))
    for k in STATS:
        print(k, float((ostats[k] - s64[k]).abs().max()))
    assert worst < 1e-9 and abs(oloss.item() - l64) < 1e-12


if __name__ == "__main__":
    main()
