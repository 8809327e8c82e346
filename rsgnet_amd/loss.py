"""Host mirror of the reference's training losses on the HIP kernels of csrc/train_tail.hip.

Same names, arguments and results as the reference (`lib/core/loss.py:14-38` JointsMSELoss; the RSGNet step of
`lib/core/function.py:246-299`), but each loss is ONE fused pass that produces the value and the gradient, wrapped
in a torch.autograd.Function.  No CPU fallback: tensors must live on the GPU and the HIP library must be built."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import hip


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("rsgnet_amd losses run on the GPU only (HIP kernels, no CPU fallback)")


class _JointsMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, gt, weight):
        _need_gpu(pred, gt, weight)
        lib = hip.lib()
        B, J = pred.shape[:2]
        P = pred[0, 0].numel()
        p32 = pred.detach().float().contiguous()
        g32 = gt.detach().float().contiguous()
        w32 = None if weight is None else weight.detach().float().reshape(B, J).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(p32)
        ws = torch.empty(lib.rsg_loss_workspace_bytes(B * J), dtype=torch.uint8, device=pred.device)
        hip.check(lib.rsg_joints_mse_loss(p32.data_ptr(), g32.data_ptr(), None if w32 is None else w32.data_ptr(), B, J, P,
                                          1.0, loss.data_ptr(), 0, grad.data_ptr(), ws.data_ptr(), _stream(pred)),
                  "joints_mse_loss")
        ctx.save_for_backward(grad)
        ctx.dtype = pred.dtype
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return (grad * gout).to(ctx.dtype), None, None


class JointsMSELoss(nn.Module):
    """lib/core/loss.py:14-38: mean over joints of 0.5 * MSE(pred_j * w_j, gt_j * w_j)."""

    def __init__(self, use_target_weight: bool):
        super().__init__()
        self.use_target_weight = use_target_weight

    def forward(self, output, target, target_weight):
        return _JointsMSE.apply(output, target, target_weight if self.use_target_weight else None)


class _BCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob, target, scale):
        _need_gpu(prob, target)
        lib = hip.lib()
        p32 = prob.detach().float().contiguous()
        t32 = target.detach().float().contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=prob.device)
        grad = torch.empty_like(p32)
        ws = torch.empty(lib.rsg_loss_workspace_bytes(1024), dtype=torch.uint8, device=prob.device)
        hip.check(lib.rsg_bce_loss(p32.data_ptr(), t32.data_ptr(), p32.numel(), float(scale), 1.0, loss.data_ptr(), 0,
                                   grad.data_ptr(), ws.data_ptr(), _stream(prob)), "bce_loss")
        ctx.save_for_backward(grad)
        ctx.dtype = prob.dtype
        return loss[0]

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return (grad * gout).to(ctx.dtype), None, None


def limbs_bce_loss(limbs_scores, target_limbs, scale: float = 0.01):
    """`0.01 * torch.nn.BCELoss()(limbs_outputs, target_limbs)` (function.py:240,293)."""
    return _BCE.apply(limbs_scores, target_limbs, scale)


def relation_target_vector(target: torch.Tensor) -> torch.Tensor:
    """The factor p of the reference's relation target p p^T (function.py:261-268) as [B, S]; pass it to
    `model(x, relation_target=p)`: the [B,S,S] outer product (9.7 GB at batch 256) is never built."""
    _need_gpu(target)
    lib = hip.lib()
    B, J, H, W = target.shape
    t32 = target.detach().float().contiguous()
    out = torch.empty(B, (H // 2) * (W // 2), dtype=torch.float32, device=target.device)
    hip.check(lib.rsg_relation_target_vec(t32.data_ptr(), out.data_ptr(), B, J, H, W, _stream(target)), "relation_target_vec")
    return out


def rsgnet_loss(outputs, target, target_weight, all_ins_target, all_ins_target_weight, target_limbs,
                criterion: JointsMSELoss):
    """The loss of `rsgnet_train` (function.py:272-299) from the model's 4-tuple."""
    multi, kpt, limbs, rel = outputs
    loss = criterion(kpt, target, target_weight)
    if multi is not None:
        loss = loss + criterion(multi, all_ins_target, all_ins_target_weight)
    if limbs is not None:
        loss = loss + limbs_bce_loss(limbs, target_limbs, 0.01)
    return loss + 0.001 * torch.mean(rel)
