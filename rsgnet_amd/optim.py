"""`get_optimizer` of the reference (lib/utils/utils.py:57-74) on a fused HIP Adam.

`FusedAdam` keeps parameters' gradients and both moments in ONE flat fp32 buffer each (29.3 M elements for
RSGNet-W32: 117 MB per buffer, trivially resident in 288 GB), so a step is one launch of `rsg_adam_step` over
the whole model instead of ~1 800 per-tensor updates, and the data-parallel gradient exchange
(rsgnet_amd/parallel.py) all-reduces slices of the same flat gradient buffer with no packing copies."""
from __future__ import annotations

from typing import Iterable

import torch

from . import hip


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr) semantics (betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("optimizer got an empty parameter list")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam runs on the GPU only (HIP kernel, no CPU fallback)")
        self._params = params
        self._numel = sum(p.numel() for p in params)
        pad = (-self._numel) % 4
        n = self._numel + pad
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.grad_scale = 1.0          # 1/world_size when the gradients are all-reduce SUMS
        off = 0
        for p in params:               # parameters and gradients become views of the flat buffers
            k = p.numel()
            self.flat_param[off:off + k].copy_(p.data.reshape(-1).float())
            p.data = self.flat_param[off:off + k].view_as(p)
            p.grad = self.flat_grad[off:off + k].view_as(p)
            off += k

    def zero_grad(self, set_to_none: bool = False):       # gradients stay views of the flat buffer
        self.flat_grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        off = 0
        for p in self._params:         # a backward may have replaced .grad by a fresh tensor: fold it back in
            k = p.numel()
            view = self.flat_grad[off:off + k]
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad.reshape(-1))
                p.grad = view.view_as(p)
            off += k
        self.step_count += 1
        lib = hip.lib()
        hip.check(lib.rsg_adam_step(self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                    self.exp_avg_sq.data_ptr(), self.flat_param.numel(), float(g["lr"]), float(g["betas"][0]),
                                    float(g["betas"][1]), float(g["eps"]), self.step_count, float(self.grad_scale),
                                    torch.cuda.current_stream(self.flat_param.device).cuda_stream), "adam_step")
        return loss


def get_optimizer(cfg, model):
    """lib/utils/utils.py:57-74: TRAIN.OPTIMIZER 'adam' -> Adam(lr=TRAIN.LR); 'sgd' is not on this path."""
    name = cfg["TRAIN"]["OPTIMIZER"] if isinstance(cfg, dict) else cfg.TRAIN.OPTIMIZER
    lr = cfg["TRAIN"]["LR"] if isinstance(cfg, dict) else cfg.TRAIN.LR
    if name != "adam":
        raise ValueError(f"TRAIN.OPTIMIZER={name!r}: only 'adam' (the reference's default and every experiment yaml) is built")
    return FusedAdam(model.parameters(), lr=lr)
