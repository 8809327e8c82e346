"""`get_optimizer` of the reference (lib/utils/utils.py:57-74) on a fused HIP Adam.

`FusedAdam` keeps parameters, gradients and both moments in ONE flat fp32 buffer each (29.3 M elements for
RSGNet-W32: 117 MB per buffer, trivially resident in 288 GB), so a step is one launch of `rsg_adam_step` over
the whole model instead of ~1 800 per-tensor updates, and the data-parallel gradient exchange
(rsgnet_amd/parallel.py) all-reduces slices of the same flat gradient buffer with no packing copies.

Checkpoints: `state_dict()` / `load_state_dict()` speak torch.optim.Adam's per-parameter layout
({'step', 'exp_avg', 'exp_avg_sq'} per parameter index), so the reference trainers' `'optimizer':
optimizer.state_dict()` (tools/train.py:223) and resume (`optimizer.load_state_dict`, lib/core/function.py /
rsgnet_train_.py:187) work in both directions, including checkpoints written by torch.optim.Adam itself."""
from __future__ import annotations

from typing import Dict, Iterable, List, Tuple

import torch

from . import hip


#: bumped by every FusedAdam.step(): its kernel updates parameters through raw pointers, which torch's per-tensor
#: version counters do not see; models/_base.py adds this to its "were the parameters written?" signature
PARAM_EPOCH = 0


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr) semantics (betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad):
    a parameter whose `.grad` is None is skipped (moments, step count and value untouched), as torch does."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        every = list(params)
        if not every:
            raise ValueError("optimizer got an empty parameter list")
        super().__init__(every, dict(lr=lr, betas=betas, eps=eps))
        # checkpoint indices count EVERY parameter handed in, as torch.optim.Adam(model.parameters()) does (RSGNet's
        # frozen loc_features / real_matrix_limb included: they never get a gradient, so never any state); only the
        # trainable ones live in the flat buffers
        self._index = [i for i, p in enumerate(every) if p.requires_grad]
        self._n_all = len(every)
        params = [every[i] for i in self._index]
        if not params:
            raise ValueError("optimizer got no trainable parameter")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam runs on the GPU only (HIP kernel, no CPU fallback)")
        self._params = params
        # every parameter starts on a 16-byte boundary of the flat buffers (the kernel moves float4s, and a run of
        # parameters can then be stepped on its own)
        self._offsets: List[int] = []
        off = 0
        for p in params:
            self._offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self._numel = off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self._steps: List[int] = [0] * len(params)     # per parameter, as torch.optim.Adam counts them
        self.grad_scale = 1.0          # 1/world_size when the gradients are all-reduce SUMS
        for p, o in zip(params, self._offsets):        # parameters and gradients become views of the flat buffers
            k = p.numel()
            self.flat_param[o:o + k].copy_(p.data.reshape(-1).float())
            p.data = self.flat_param[o:o + k].view_as(p)
            p.grad = self.flat_grad[o:o + k].view_as(p)

    # ------------------------------------------------------------------ views
    def _view(self, flat: torch.Tensor, i: int) -> torch.Tensor:
        p, o = self._params[i], self._offsets[i]
        return flat[o:o + p.numel()].view_as(p)

    def flat_offset_of(self, p: torch.nn.Parameter):
        """Element offset of a parameter in the flat buffers (None for a frozen one)."""
        m = getattr(self, "_offset_map", None)
        if m is None:
            m = self._offset_map = {id(q): o for q, o in zip(self._params, self._offsets)}
        return m.get(id(p))

    @property
    def step_count(self) -> int:
        """Steps taken by the most-stepped parameter (all parameters, unless some had no gradient)."""
        return max(self._steps)

    def zero_grad(self, set_to_none: bool = False):       # gradients stay views of the flat buffer
        self.flat_grad.zero_()
        for i, p in enumerate(self._params):
            if p.grad is None:
                p.grad = self._view(self.flat_grad, i)

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        active: List[bool] = []
        ptrs = getattr(self, "_grad_ptrs", None)
        if ptrs is None or self._grad_ptrs_base != self.flat_grad.data_ptr():      # address of every parameter's slice of the
            base = self.flat_grad.data_ptr()                                        # flat gradient buffer, computed once
            ptrs = self._grad_ptrs = [base + 4 * o for o in self._offsets]
            self._grad_ptrs_base = base
        for i, p in enumerate(self._params):
            gr = p.grad
            if gr is None:                              # torch.optim.Adam skips it: so do we
                active.append(False)
                continue
            if gr.data_ptr() != ptrs[i]:                # a backward replaced .grad by a fresh tensor: fold it back in
                view = self._view(self.flat_grad, i)
                view.copy_(gr)
                p.grad = view
            active.append(True)
        lib = hip.lib()
        stream = torch.cuda.current_stream(self.flat_param.device).cuda_stream
        # one launch per run of consecutive active parameters with the same step count (normally: one launch)
        i, n = 0, len(self._params)
        while i < n:
            if not active[i]:
                i += 1
                continue
            j = i
            while j + 1 < n and active[j + 1] and self._steps[j + 1] == self._steps[i]:
                j += 1
            lo = self._offsets[i]
            hi = self._offsets[j + 1] if j + 1 < n else self._numel
            step = self._steps[i] + 1
            hip.check(lib.rsg_adam_step(self.flat_param.data_ptr() + 4 * lo, self.flat_grad.data_ptr() + 4 * lo,
                                        self.exp_avg.data_ptr() + 4 * lo, self.exp_avg_sq.data_ptr() + 4 * lo, hi - lo,
                                        float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), step,
                                        float(self.grad_scale), stream), "adam_step")
            for k in range(i, j + 1):
                self._steps[k] = step
            i = j + 1
        global PARAM_EPOCH
        PARAM_EPOCH += 1
        return loss

    # ------------------------------------------------------------------ checkpoints (torch.optim.Adam layout)
    def state_dict(self) -> Dict:
        """{'state': {index: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} exactly as
        torch.optim.Adam(model.parameters()).state_dict() lays it out: loadable by either optimizer."""
        state = {}
        for i in range(len(self._params)):
            if self._steps[i] == 0:
                continue                                 # torch creates a parameter's state at its first step
            state[self._index[i]] = {"step": torch.tensor(float(self._steps[i])),
                                     "exp_avg": self._view(self.exp_avg, i).clone(),
                                     "exp_avg_sq": self._view(self.exp_avg_sq, i).clone()}
        # the keys torch.optim.Adam's own param_groups carry, so that its load_state_dict/step accept the dict as is
        group = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                     differentiable=False, fused=None, decoupled_weight_decay=False)
        group.update({k: v for k, v in self.param_groups[0].items() if k != "params"})
        group["params"] = list(range(self._n_all))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, state_dict: Dict) -> None:
        groups = state_dict["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != self._n_all:
            raise ValueError(f"loaded state dict has {len(ids)} parameters, the optimizer has {self._n_all}")
        flat_of = {a: k for k, a in enumerate(self._index)}
        where: Dict[int, int] = {}
        for k, pid in enumerate(ids):
            if k in flat_of:
                where[pid] = flat_of[k]
        hyper = {k: v for k, v in groups[0].items() if k != "params"}
        if hyper.get("weight_decay", 0) or hyper.get("amsgrad", False) or hyper.get("maximize", False):
            raise ValueError("FusedAdam implements Adam without weight decay / amsgrad (lib/utils/utils.py:70-74)")
        # every key of the saved group comes back, as torch.optim.Optimizer.load_state_dict does it: a scheduler's
        # `initial_lr` must survive a resume (tools/cp_train.py:184-193 builds MultiStepLR(last_epoch=checkpoint epoch)
        # right after optimizer.load_state_dict, and torch raises KeyError without it)
        self.param_groups[0].update(hyper)
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self._steps = [0] * len(self._params)
        for pid, st in state_dict["state"].items():
            if pid not in where:
                raise ValueError(f"optimizer state for parameter {pid}, which is frozen (requires_grad=False) here")
            i = where[pid]
            p = self._params[i]
            for name, flat in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
                t = st[name]
                if tuple(t.shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {name} of parameter {pid} has shape {tuple(t.shape)}, "
                                     f"parameter has {tuple(p.shape)}")
                self._view(flat, i).copy_(t.to(flat.device, torch.float32))
            step = st["step"]
            self._steps[i] = int(step.item() if torch.is_tensor(step) else step)


def get_optimizer(cfg, model):
    """lib/utils/utils.py:57-74: TRAIN.OPTIMIZER 'adam' -> Adam(lr=TRAIN.LR); 'sgd' is not on this path."""
    name = cfg["TRAIN"]["OPTIMIZER"] if isinstance(cfg, dict) else cfg.TRAIN.OPTIMIZER
    lr = cfg["TRAIN"]["LR"] if isinstance(cfg, dict) else cfg.TRAIN.LR
    if name != "adam":
        raise ValueError(f"TRAIN.OPTIMIZER={name!r}: only 'adam' (the reference's default and every experiment yaml) is built")
    return FusedAdam(model.parameters(), lr=lr)
