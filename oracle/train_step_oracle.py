"""ORACLE -- test infrastructure, NOT product code (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import it).

CPU restatement of ONE training step of pose_hrnet as the reference runs it:
  * `outputs = model(input)` with the model in train() mode       /root/reference/lib/core/function.py:27-52
      = pose_hrnet.py:428-463 with every BatchNorm2d normalising by the batch statistics (biased variance) and
        updating its running statistics with momentum 0.1 (unbiased variance)   pose_hrnet.py:19, SURVEY App. D
  * `loss = criterion(outputs, target, target_weight)`            lib/core/loss.py:14-38 (JointsMSELoss)
  * `loss.backward()`                                             function.py:62-64 -- torch autograd
  * `optimizer.step()`                                            lib/utils/utils.py:70-74 (Adam): oracle/train_oracle.adam_step
It walks the same functional graph as oracle/pose_oracle.py (its `calibrate` mode IS training-mode BatchNorm) with
autograd enabled, in fp64 by default.  Pinned by tests/golden/train_step_hrnet_w32_256x192.npz, produced by
oracle/make_train_step_goldens.py from the imported reference model and loss (forward, loss, gradients, running stats).
"""
from __future__ import annotations

from typing import Dict, Mapping, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import pose_oracle as po

BN_MOMENTUM = 0.1   # pose_hrnet.py:19


def is_buffer(name: str) -> bool:
    return name.endswith(("running_mean", "running_var", "num_batches_tracked"))


def joints_mse_loss(output: torch.Tensor, target: torch.Tensor, target_weight: torch.Tensor, use_target_weight: bool = True):
    """lib/core/loss.py:20-38, in torch ops (differentiable): mean over joints of 0.5 * MSE(pred_j * w_j, gt_j * w_j)."""
    B, J = output.shape[:2]
    pred = output.reshape(B, J, -1)
    gt = target.reshape(B, J, -1).to(output.dtype)
    loss = output.new_zeros(())
    for j in range(J):
        p, g = pred[:, j], gt[:, j]
        if use_target_weight:
            w = target_weight.reshape(B, J, 1)[:, j].to(output.dtype)
            p, g = p * w, g * w
        loss = loss + 0.5 * F.mse_loss(p, g, reduction="mean")
    return loss / J


from rsgnet_amd.weights import gaussian_targets, limb_targets  # noqa: E402,F401  (synthetic data, shared with bench.py --train)


def hrnet_train_step(sd: Mapping[str, torch.Tensor], cfg, x: torch.Tensor, target: torch.Tensor, target_weight: torch.Tensor,
                     use_target_weight: bool = True, dtype=torch.float64, relu_masks=None, drop_grad=()):
    """One forward/backward of pose_hrnet in train() mode.
    drop_grad: fault injection for the sensitivity test of the parity bars (pose_oracle.Ctx.drop_grad).
    Returns (loss, heatmaps, {param name: gradient}, {buffer name: running statistic after the step}).
    relu_masks: teacher forcing (pose_oracle.Ctx.relu_masks): every ReLU listed there takes its 0/1 decisions from the
    given map, so the gradient is that of the SAME piecewise-linear function another implementation differentiated."""
    work: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if not is_buffer(k):
            t = t.to(dtype).requires_grad_(True)
        work[k] = t
    old = {k: v.detach().clone().double() for k, v in sd.items() if k.endswith(("running_mean", "running_var"))}
    ctx = po.Ctx(work, dtype=dtype, calibrate=True, relu_masks=relu_masks)   # calibrate = batch statistics (training-mode BatchNorm)
    ctx.drop_grad = set(drop_grad)
    ctx.sd = dict(work)                                       # the walk overwrites running stats; keep `work` intact
    ys = po.backbone(ctx, x.to(dtype), cfg)
    w, b = ctx.p("final_layer.weight"), ctx.p("final_layer.bias")
    heat = F.conv2d(ys[0], w, b, padding=(w.shape[-1] - 1) // 2)
    loss = joints_mse_loss(heat, target, target_weight, use_target_weight)
    names = [k for k in work if not is_buffer(k)]
    grads = torch.autograd.grad(loss, [work[k] for k in names], allow_unused=True)
    gd = {k: (g.detach() if g is not None else None) for k, g in zip(names, grads)}
    # calibrate mode stored (batch mean, unbiased batch variance): fold them in with the reference's momentum
    stats = {k: ((1 - BN_MOMENTUM) * old[k] + BN_MOMENTUM * ctx.sd[k].detach().double()) for k in old}
    return loss.detach(), heat.detach(), gd, stats


# ------------------------------------------------------------------------------------------------------------------
# RSGNet: one step of `rsgnet_train`, /root/reference/lib/core/function.py:239-323
# ------------------------------------------------------------------------------------------------------------------
def relation_target_factor(target: torch.Tensor) -> torch.Tensor:
    """function.py:261-268: person_target = max over joints, bilinear x1/2 (align_corners=True), flattened: the factor p
    [B,S] of the relation target T = p^T p the reference materialises ([B,S,S])."""
    person, _ = torch.max(target, dim=1)
    b, h, w = person.shape
    person = F.interpolate(person.reshape(b, 1, h, w), scale_factor=1 / 2, mode="bilinear", align_corners=True)
    return person.reshape(b, -1)


def rsgnet_train_step(sd: Mapping[str, torch.Tensor], cfg, x: torch.Tensor, target, target_weight, all_ins_target,
                      all_ins_target_weight, target_limbs, dtype=torch.float64, relu_masks=None):
    """One forward/backward of pose_rsgnet in train() mode with the loss of rsgnet_train (function.py:272-299):
    JointsMSELoss(kpt, target) + JointsMSELoss(multi, all_ins_target) + 0.01 * BCE(limbs, target_limbs)
    + 0.001 * mean_b(mean_ij (T - R)^2).  Returns (loss, (multi, kpt, limbs, relation_mse), {name: gradient},
    {running statistic after the step})."""
    work: Dict[str, torch.Tensor] = {}
    frozen = ("loc_features", "kt_machine.real_matrix_limb")            # requires_grad False in the reference (:691, :574)
    for k, v in sd.items():
        t = v.detach().clone()
        if not is_buffer(k):
            t = t.to(dtype)
            if k not in frozen:
                t.requires_grad_(True)
        work[k] = t
    old = {k: v.detach().clone().double() for k, v in sd.items() if k.endswith(("running_mean", "running_var"))}
    ctx = po.Ctx(work, dtype=dtype, calibrate=True, relu_masks=relu_masks)
    ctx.sd = dict(work)
    p = relation_target_factor(target.to(dtype))
    rel_t = p[:, :, None] * p[:, None, :]
    multi, kpt, limbs, rel = po.rsgnet_forward(work, cfg, x.to(dtype), relation_target=rel_t, ctx=ctx, grad=True)
    loss = joints_mse_loss(kpt, target, target_weight) + joints_mse_loss(multi, all_ins_target, all_ins_target_weight) \
        + 0.01 * F.binary_cross_entropy(limbs, target_limbs.to(dtype)) + 0.001 * torch.mean(rel)
    names = [k for k in work if not is_buffer(k) and k not in frozen]
    grads = torch.autograd.grad(loss, [work[k] for k in names], allow_unused=True)
    gd = {k: (g.detach() if g is not None else None) for k, g in zip(names, grads)}
    stats = {k: ((1 - BN_MOMENTUM) * old[k] + BN_MOMENTUM * ctx.sd[k].detach().double()) for k in old}
    return loss.detach(), tuple(t.detach() for t in (multi, kpt, limbs, rel)), gd, stats
