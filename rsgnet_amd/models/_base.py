"""Shared behaviour of the two drop-in models: plan cache, loud failure off-GPU, init_weights."""
from __future__ import annotations

import logging
import operator
import os

import torch
import torch.nn as nn

logger = logging.getLogger(__name__)

# Bumped whenever any module of the process registers a parameter, a buffer or a sub-module (assignments go through
# register_parameter / register_buffer / add_module): HipPoseModule._param_signature rebuilds its flat tensor list then.
_REGISTRATIONS = [0]
_VERSION_OF = operator.attrgetter("_version")


def _registered(*_args, **_kwargs):
    _REGISTRATIONS[0] += 1


nn.modules.module.register_module_parameter_registration_hook(_registered)
nn.modules.module.register_module_buffer_registration_hook(_registered)
nn.modules.module.register_module_module_registration_hook(_registered)


class HipPoseModule(nn.Module):
    """nn.Module facade over a compiled HIP execution plan.

    The module owns reference-named parameters (``_tree.py``); ``forward`` hands the input to a plan
    (``rsgnet_amd.plan``) that was built from those parameters: BatchNorm folded into fp16 weights,
    weights packed into MFMA fragment order, a static NHWC activation arena, and the launch list.
    The plan is rebuilt whenever the parameters may have changed (``load_state_dict``, ``train()``,
    ``.to()/.cuda()``, any in-place update torch's version counters or ``FusedAdam`` record — ``_param_signature``;
    ``invalidate_plan()`` is the explicit form for writes through raw pointers) or the batch size / device changes.

    There is deliberately NO CPU or eager fallback: off the GPU, or without the HIP library,
    ``forward`` raises.
    """

    #: storage/MFMA input type of the plan; accumulation is always fp32
    compute_dtype = "fp16"
    #: plans kept for streams other than the default one, per (batch, device, dtype)
    MAX_STREAM_PLANS = 4

    def __init__(self):
        super().__init__()
        self._plans = {}
        self._train_plans = {}
        #: callables(pending_parameters) the training backward pass calls after each of its segments
        #: (rsgnet_amd.parallel.DataParallelStep starts the all-reduce of the finished tail of the gradient buffer)
        self.grad_ready_hooks = []

    # -- plan lifecycle ---------------------------------------------------------------------
    def invalidate_plan(self) -> None:
        self._plans = {}
        self._train_plans = {}
        self.__dict__.pop("_sig_tensors", None)

    def _apply(self, fn, *a, **kw):
        self.invalidate_plan()
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self.invalidate_plan()
        return super().load_state_dict(*a, **kw)

    def train(self, mode: bool = True):
        if mode != self.training:
            self._plans = {}                  # eval plans fold the running statistics; training plans read everything live
        return super().train(mode)

    #: models whose training-mode forward/backward is built (rsgnet_amd/train.py) set this
    supports_training = False

    def train_forward(self, x: torch.Tensor) -> torch.Tensor:
        """`model(x)` under `model.train()`: batch-statistics BatchNorm, one autograd node whose backward runs the HIP
        backward kernels and writes the parameters' `.grad` (rsgnet_amd/train.py)."""
        if x.device.type != "cuda":
            raise RuntimeError(f"{type(self).__name__} runs only on an AMD GPU through librsgnet_hip.so; got a "
                               f"{x.device.type} tensor. There is no CPU fallback.")
        if not self.supports_training:
            raise NotImplementedError(
                f"training-mode forward of {type(self).__name__} is not built; call model.eval()")
        from ..train import train_forward
        return train_forward(self, x)

    def _build_plan(self, batch: int, device: torch.device):  # pragma: no cover - overridden
        raise NotImplementedError

    def _param_signature(self):
        """Changes whenever a parameter or buffer was written in place (torch bumps `_version` on every in-place op:
        `p.data.copy_`, torch optimizers' updates), was replaced by another tensor object, or FusedAdam stepped (its kernel
        writes through raw pointers and bumps `optim.PARAM_EPOCH` instead): the folded/packed weights of a cached plan are
        then stale.  Runs in front of EVERY forward, so it is one C-level pass over a cached flat list of the model's
        tensors (0.05 ms for the 1 754 of HRNet-W32); the list is rebuilt when ANY module of the process registered a
        parameter, buffer or sub-module since (`_REGISTRATIONS`, torch's global registration hooks: `m.weight =
        nn.Parameter(...)` goes through them), after `.to()` / `load_state_dict` / `invalidate_plan()`.  History: the first
        form, `self.parameters()` + `self.buffers()` (two generator walks over the ~1 000 modules of the tree), took 1.4 ms
        of host time per forward - as long as the forward itself takes on the GPU (profiles/r03m_host_time_per_forward.txt);
        the second walked the modules' `_parameters` / `_buffers` dicts (0.2 ms)."""
        from .. import optim
        ts = self.__dict__.get("_sig_tensors")
        if ts is None or self.__dict__.get("_sig_epoch") != _REGISTRATIONS[0]:
            ts = [t for m in self.modules() for d in (m._parameters, m._buffers) for t in d.values() if t is not None]
            self.__dict__["_sig_tensors"] = ts
            self.__dict__["_sig_epoch"] = _REGISTRATIONS[0]
            self.__dict__["_sig_ids"] = hash(tuple(map(id, ts)))
        return optim.PARAM_EPOCH + sum(map(_VERSION_OF, ts)), self.__dict__["_sig_ids"]

    def run_plan(self, x: torch.Tensor):
        """forward's way into the plan: everything (plan build, launches, output allocation) happens with the INPUT's
        device current, whatever the caller's current device is (DataParallel replicas, a model on cuda:1)."""
        from ..plan import HandoffTimeout
        if x.device.type != "cuda":
            self.plan_for(x)                  # raises: there is no CPU path
        with torch.cuda.device(x.device):
            plan = self.plan_for(x)
            try:
                return plan, plan.run(x)
            except HandoffTimeout:
                # an in-launch producer->consumer hand-off gave up: the flags were re-zeroed by the plan; from now on
                # this model runs the dependent chains as separate launches (no hand-off), and the caller is told
                self._no_handoff = True
                self.invalidate_plan()
                raise

    def plan_for(self, x: torch.Tensor):
        if x.device.type != "cuda":
            raise RuntimeError(
                f"{type(self).__name__} runs only on an AMD GPU through librsgnet_hip.so; got a "
                f"{x.device.type} tensor. There is no CPU fallback (the CPU oracle lives in oracle/ "
                "and is test infrastructure).")
        if self.training:
            raise NotImplementedError(
                f"{type(self).__name__}: the inference plan serves model.eval() only; training-mode forward goes "
                "through the training plan (rsgnet_amd/train.py, train_rsgnet.py).")
        sig = self._param_signature()
        if sig != getattr(self, "_plan_sig", None):
            self._plans = {}                  # parameters were updated in place since the plans were built
            self._plan_sig = sig
        # one plan (activation arena, captured graph) per STREAM: forwards issued on different streams may overlap on the
        # GPU (torch.cuda.stream(s): two batches in flight fill each other's idle CUs), forwards on one stream are ordered
        # and share theirs.  At most MAX_STREAM_PLANS side-stream plans per (batch, device, dtype); the oldest goes first.
        base = (int(x.shape[0]), x.device.index, self.compute_dtype)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        key = base + (stream,)
        plan = self._plans.get(key)
        if plan is None:
            mine = [k for k in self._plans if k[:3] == base and k[3] != 0]
            if stream != 0 and len(mine) >= self.MAX_STREAM_PLANS:
                torch.cuda.synchronize(x.device)  # its last forward may still be running on its stream (rare path)
                del self._plans[mine[0]]          # dicts keep insertion order
            self._plan_for_side_stream = stream != 0     # read by Plan.__init__ (throughput tuning of overlapping forwards)
            try:
                with torch.cuda.device(x.device):
                    plan = self._build_plan(int(x.shape[0]), x.device)
            finally:
                self._plan_for_side_stream = False
            self._plans[key] = plan
        return plan

    # -- reference-compatible initialisation --------------------------------------------------
    def init_weights(self, pretrained: str = "") -> None:
        """pose_hrnet.py:465-495 / pose_rsgnet.py:1023-1063: N(0, 1e-3) convs/linears/deconvs, zero
        biases, identity BatchNorm affine, then a partial load filtered by PRETRAINED_LAYERS."""
        logger.info("=> init weights from normal distribution")
        linear_too = getattr(self, "_init_linear", False)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) or (linear_too and isinstance(m, nn.Linear)):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if os.path.isfile(pretrained):
            sd = torch.load(pretrained, map_location="cpu")
            logger.info("=> loading pretrained model %s", pretrained)
            layers = list(self.pretrained_layers)
            keep = {k: v for k, v in sd.items() if k.split(".")[0] in layers or layers[0] == "*"}
            self.load_state_dict(keep, strict=False)
        elif pretrained:
            logger.error("=> please download pre-trained models first!")
            raise ValueError("{} is not exist!".format(pretrained))
        self.invalidate_plan()
