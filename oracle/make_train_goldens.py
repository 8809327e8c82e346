"""Generates tests/golden/train_tail.npz from the reference's own loss module (lib/core/loss.py, imported from
/root/reference in THIS container only) and torch.optim.Adam / torch.nn.BCELoss / F.interpolate, the operators the
reference's training loop calls (lib/core/function.py:246-305, lib/utils/utils.py:70-74).  Inputs are seeded."""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

spec = importlib.util.spec_from_file_location("ref_loss", "/root/reference/lib/core/loss.py")
ref_loss = importlib.util.module_from_spec(spec)
sys.dont_write_bytecode = True
spec.loader.exec_module(ref_loss)

g = torch.Generator().manual_seed(2024)
B, J, L, H, W = 3, 17, 18, 32, 24
out = {}
# every input is drawn in fp32 (what the fixture stores) and evaluated in fp64
pred = torch.rand(B, J, H, W, generator=g).double().requires_grad_(True)
gt = torch.rand(B, J, H, W, generator=g).double()
w = ((torch.rand(B, J, 1, generator=g) > 0.3).float() * torch.rand(B, J, 1, generator=g)).double()
loss = ref_loss.JointsMSELoss(use_target_weight=True)(pred, gt, w)
loss.backward()
out.update(mse_pred=pred.detach().numpy(), mse_gt=gt.numpy(), mse_w=w.numpy(), mse_loss=loss.item(), mse_grad=pred.grad.numpy())
pred2 = pred.detach().clone().requires_grad_(True)
loss2 = ref_loss.JointsMSELoss(use_target_weight=False)(pred2, gt, w)
loss2.backward()
out.update(mse_loss_now=loss2.item(), mse_grad_now=pred2.grad.numpy())

prob = torch.sigmoid(torch.randn(B, L, H, W, generator=g) * 3).double().requires_grad_(True)
tl = (torch.rand(B, L, H, W, generator=g) > 0.8).double()
bl = torch.nn.BCELoss()(prob, tl)
bl.backward()
out.update(bce_prob=prob.detach().numpy(), bce_t=tl.numpy(), bce_loss=bl.item(), bce_grad=prob.grad.numpy())

# relation target vector (function.py:261-268) on a heatmap-sized target
person, _ = torch.max(gt, dim=1)
half = F.interpolate(person.reshape(B, 1, H, W), scale_factor=1 / 2, mode="bilinear", align_corners=True)
out.update(rel_p=torch.squeeze(half).reshape(B, -1).numpy())

# three Adam steps on a small parameter vector with given gradients
p = torch.randn(1000, generator=g).double().requires_grad_(True)
opt = torch.optim.Adam([p], lr=1e-3)
grads = torch.randn(3, 1000, generator=g).double()
out.update(adam_p0=p.detach().numpy().copy(), adam_grads=grads.numpy())
for k in range(3):
    opt.zero_grad()
    p.grad = grads[k].clone()
    opt.step()
    out[f"adam_p{k + 1}"] = p.detach().numpy().copy()
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "train_tail.npz")
f32 = ("mse_pred", "mse_gt", "mse_w", "bce_prob", "bce_t", "adam_p0", "adam_grads", "mse_grad", "mse_grad_now", "bce_grad")
np.savez_compressed(path, **{k: (np.asarray(v, np.float32) if k in f32 else np.asarray(v)) for k, v in out.items()})
print("wrote", path, {k: np.asarray(v).shape for k, v in out.items()})
