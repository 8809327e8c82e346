"""Training-mode execution of pose_hrnet on the HIP kernels: forward with batch-statistics BatchNorm, backward, all
gradients written straight into the parameters' fp32 `.grad` tensors.

What it replaces: `outputs = model(input)` ... `loss.backward()` of the reference's training loop
(/root/reference/lib/core/function.py:27-80), i.e. torch autograd over pose_hrnet.py:428-463 in `model.train()` mode.
The model's forward returns the fp32 NCHW heat maps as the output of ONE autograd node (`_TrainFn`); its backward takes
the heat-map gradient that the loss hands back and runs the recorded backward launch list.

Numerics: activations, their gradients and the packed weights are bf16 (same exponent range as fp32: no loss scaling),
every contraction accumulates in fp32, BatchNorm statistics and all parameter gradients are fp32, master weights stay
the module's fp32 parameters (re-packed on the device at every forward: one launch).  All reductions have a fixed
order, so a step is bit-reproducible.

Recording is two-phase: the forward walk creates op objects (inputs, outputs, forward launcher); the backward launch
list is then generated from the ops in reverse, which is when gradient buffers are allocated and the "first writer
writes, later writers add" decisions are taken in execution order.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import hip
from . import switches as sw

BF16 = hip.DTYPES["bf16"]


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


class T:
    """A dense NHWC bf16 map [B,H,W,C] of the training graph and, during backward recording, its gradient."""

    __slots__ = ("H", "W", "C", "data", "grad", "grad_shared", "needs_grad", "name")

    def __init__(self, plan: "TrainPlan", H: int, W: int, C_: int, name: str = "", needs_grad: bool = True):
        self.H, self.W, self.C, self.name = H, W, C_, name
        self.data = torch.zeros(plan.batch, H, W, C_, dtype=torch.bfloat16, device=plan.device)
        self.grad: Optional[torch.Tensor] = None
        self.grad_shared = False
        self.needs_grad = needs_grad

    @property
    def pixels(self) -> int:
        return self.data.shape[0] * self.H * self.W


class _WgradLaunch:
    """One grouped weight-gradient launch (rsg_conv_wgrad_group_ex).  Linked in execution order by TrainPlan._link_wgrads:
    a launch defers the reduction of its partial tiles to the NEXT weight-gradient launch, which runs it in its first
    workgroups next to its own main kernel (workspaces alternate between two sets, `par`); the last one of a backward
    segment reduces its own."""

    def __init__(self, plan: "TrainPlan", makers: list):
        self.plan, self.makers, self.n = plan, makers, len(makers)
        self.items = (hip.WgradItem * self.n)()
        self.prev: Optional["_WgradLaunch"] = None
        self.prev_items = None
        self.par, self.defer = 0, False

    def link(self, prev: "_WgradLaunch"):
        self.prev, prev.defer = prev, True
        self.prev_items = (hip.WgradItem * prev.n)()

    def __call__(self, s: int):
        for k, mk in enumerate(self.makers):     # addresses are read at launch time (a re-seated parameter re-captures)
            mk(self.items[k], self.par)
        n_prev = 0
        if self.prev is not None:
            for k, mk in enumerate(self.prev.makers):
                mk(self.prev_items[k], self.prev.par)
            n_prev = self.prev.n
        self.plan._chk(self.plan.lib.rsg_conv_wgrad_group_ex(self.items, self.n, self.prev_items, n_prev, int(self.defer), BF16, s),
                       "conv_wgrad_group")


class TrainPlan:
    """Forward and backward launch lists of pose_hrnet in training mode for one (batch, device)."""

    #: top-level children of the HRNet backbone shared by pose_hrnet and pose_rsgnet (pose_rsgnet.py:921-953)
    BACKBONE = ("conv1", "bn1", "conv2", "bn2", "layer1", "transition1", "stage2", "transition2", "stage3", "transition3", "stage4")

    def __init__(self, model: nn.Module, batch: int, device: torch.device, head: bool = True):
        """head=True: pose_hrnet (backbone + final_layer -> heat maps).  head=False: the backbone alone -- its output map
        leaves as fp32 NCHW and its gradient comes back the same way (pose_rsgnet: rsgnet_amd/train_rsgnet.py runs the
        head with torch operators on top of it)."""
        self.lib = hip.lib()
        self.with_head = head
        self.port16 = (not head) and bool(getattr(model, "train_feat_nhwc16", False))   # backbone output / gradient as NHWC bf16
        self.model, self.batch, self.device = model, batch, device
        self.fwd: List[tuple] = []      # (name, launch(stream), group kind or None, group item maker or None)
        self.bwd: List[tuple] = []
        self._ops: List[dict] = []
        self._keep: List[object] = []
        self._packs: List[dict] = []            # one entry per packed weight (forward and data-gradient forms)
        self._bn_ws = torch.empty(self.lib.rsg_bn_workspace_bytes(256), dtype=torch.uint8, device=device)
        self._wgrad_lane_need: Dict[int, int] = {}
        self._wgrad_lane_ws: Dict[int, torch.Tensor] = {}
        self._bwd_lane = 0
        self._zero_bias = torch.zeros(256, dtype=torch.float32, device=device)
        self._param_ptrs: Optional[Tuple[int, ...]] = None
        self._gbuf: Dict[int, torch.Tensor] = {}
        self._sink: Dict[str, Optional[list]] = {"fwd": None, "bwd": None}
        self._region: Optional[Tuple[int, int]] = None
        self._rid = 0
        self._shared_rids: set = set()
        self._defer: Optional[list] = None
        self._bn_momentum, self._bn_eps = 0.1, 1e-5
        # the parameters whose gradients THIS plan writes (everything of pose_hrnet; the backbone of pose_rsgnet)
        self._params = [p for n, p in model.named_parameters()
                        if p.requires_grad and (head or n.split(".")[0] in self.BACKBONE)]
        self._sig: Optional[Tuple[int, ...]] = None
        self._graphs: Optional[tuple] = None         # (forward graph, [one graph per backward segment])
        self._warm = 0
        import os
        self.use_graph = sw.train_on("graph")
        self.use_lanes = sw.train_on("lanes")
        # RSGNET_HIP_TRAIN wgrad_stream (profiles/r03o_train_wgrad_stream.txt; gradients bit-identical in every form):
        #   0  every launch of the backward pass on one stream (rounds 2-3: 13.11 ms per step of 32)
        #   1  the weight-gradient launches on a stream of their own INSIDE the segment's graph (events as graph edges):
        #      slower, 13.75 ms -- branches inside one graph cost more than they hide
        #   2  (default) the weight gradients of a backward segment as a GRAPH OF THEIR OWN, replayed on the side stream once
        #      the segment's data-gradient graph has been issued: it runs beside the NEXT segment's chain -- overlap between
        #      graphs, the form that paid for inference.  12.47 ms with the three data-parallel segments, 11.95 ms with one
        #      segment per HighResolutionModule (RSGNET_HIP_TRAIN segments=module), 11.86 ms cutting every 40 launches
        #      (every:40, default)
        mode = sw.train("wgrad_stream")
        self.wgrad_stream_on = mode == "1"
        self.wgrad_split = mode == "2"
        self._wgrad_stream: Optional[torch.cuda.Stream] = None
        self.flops_fwd = 0
        W, H = model.cfg.MODEL.IMAGE_SIZE
        self.H, self.W, self.J = H, W, model.cfg.MODEL.NUM_JOINTS
        self.x_f32 = torch.zeros(batch, 3, H, W, dtype=torch.float32, device=device)
        self.heat = torch.zeros(batch, self.J, H // 4, W // 4, dtype=torch.float32, device=device)
        self.dheat = torch.zeros_like(self.heat)
        self._record(model)
        self._record_backward()
        self._wgrad_lane_ws = {(par, k): torch.empty(max(v, 16), dtype=torch.uint8, device=device)
                               for k, v in self._wgrad_lane_need.items() for par in (0, 1)}
        self._link_wgrads()
        self._build_pack_table()
        self._bns = [m for n, m in model.named_modules() if isinstance(m, nn.BatchNorm2d)
                     and (head or n.split(".")[0] in self.BACKBONE)]
        self._nbt = [m.num_batches_tracked for m in self._bns if m.num_batches_tracked is not None]

    # ------------------------------------------------------------------ helpers
    def _chk(self, rc: int, what: str):
        if rc:
            hip.check(rc, what)

    # ---- launch lists: entries are (name, fn) with optional grouping info; independent lanes (the branches of a
    # HighResolutionModule) are recorded side by side and zipped position by position, so that the k-th BatchNorm of
    # every branch lands next to its siblings and the run becomes ONE grouped launch (rsg_bn_*_group): a replayed graph
    # pays ~3.2 us per kernel whatever its size, and most of these kernels are smaller than that
    def _emit(self, which: str, name: str, fn: Callable[[int], None], gkind: Optional[str] = None, item=None, params=()):
        rec = (name, fn, gkind, item, tuple(params))    # params: the parameters whose gradient this launch finalises
        sink = self._sink.get(which)
        (sink if sink is not None else getattr(self, which)).append(rec)

    def _emit_op(self, op: dict):
        if self._region is not None:
            rid, lane = self._region
            self._region_ops[lane].append(op)
            op["region"] = (rid, lane)
        else:
            self._ops.append(op)

    def _begin_lanes(self, n: int, shared: bool = False):
        """shared: the lanes read common inputs (the fuse paths of a module all start from the branch outputs), so
        their gradient contributions meet; the backward pass then adds them after the lanes' launches (_record_backward)."""
        self._rid += 1
        if shared:
            self._shared_rids.add(self._rid)
        self._region = (self._rid, 0)
        self._region_ops = [[] for _ in range(n)]
        self._region_fwd = [[] for _ in range(n)]

    def _lane(self, k: int):
        self._region = (self._region[0], k)
        self._sink["fwd"] = self._region_fwd[k]

    def _end_lanes(self):
        lanes_ops, lanes_fwd = self._region_ops, self._region_fwd
        self._region, self._sink["fwd"] = None, None
        for pos in range(max(len(l) for l in lanes_ops)):
            for lane in lanes_ops:
                if pos < len(lane):
                    lane[pos]["region"] = lane[pos]["region"] + (pos,)
                    self._ops.append(lane[pos])
        self.fwd.extend(self._merge(self._zip(lanes_fwd)))

    @staticmethod
    def _zip(lanes: List[list]) -> list:
        out = []
        for pos in range(max((len(l) for l in lanes), default=0)):
            for lane in lanes:
                if pos < len(lane):
                    out.append(lane[pos])
        return out

    def _merge(self, recs: list) -> list:
        """Runs of adjacent groupable launches of one kind -> one grouped launch each."""
        out, i = [], 0
        while i < len(recs):
            name, fn, gkind, item = recs[i][:4]
            j = i + 1
            if gkind is not None:
                cap = 4 if gkind.startswith("wgrad") else {"conv": 12}.get(gkind, 8)
                while j < len(recs) and recs[j][2] == gkind and j - i < cap:
                    j += 1
            if j - i >= 2:
                out.append((f"{name}(+{j - i - 1})", self._group_launch(gkind, [r[3] for r in recs[i:j]]), None, None,
                            tuple(q for r in recs[i:j] for q in r[4])))
            elif gkind is not None and gkind.startswith("wgrad"):
                out.append((name, _WgradLaunch(self, [item]), None, None, recs[i][4]))
            else:
                out.append(recs[i])
                j = i + 1
            i = j
        return out

    def _group_launch(self, gkind: str, makers: list) -> Callable[[int], None]:
        lib, n = self.lib, len(makers)
        if gkind == "conv":        # rsg_conv_fuse_group: every address in the descriptors is static, plan once
            import numpy as np
            cd = (hip.ConvDesc * n)(*makers)
            fi = (hip.FuseItem * 1)()
            table = np.zeros(lib.rsg_group_table_bytes(), np.uint8)
            launch2 = (C.c_int32 * 2)()
            hip.check(lib.rsg_conv_fuse_group_plan(cd, n, fi, 0, BF16, table.ctypes.data, launch2), "conv group plan")
            self._keep += [cd, table]
            tptr, blocks, lds = table.ctypes.data, launch2[0], launch2[1]
            return lambda s: self._chk(lib.rsg_conv_fuse_group_run(tptr, blocks, lds, BF16, s), "conv group")
        if gkind.startswith("wgrad"):
            return _WgradLaunch(self, makers)
        arr = (hip.BnItem * n)()
        self._keep.append(arr)

        def launch(s):
            for k, mk in enumerate(makers):      # addresses are read at launch time (a re-seated parameter re-captures)
                mk(arr[k])
            if gkind == "bn_fwd":
                self._chk(lib.rsg_bn_forward_group(arr, n, self._bn_momentum, self._bn_eps, BF16, s), "bn_forward_group")
            else:
                self._chk(lib.rsg_bn_backward_group(arr, n, BF16, s), "bn_backward_group")
        return launch

    def _grad_of(self, p: torch.nn.Parameter) -> torch.Tensor:
        """The fp32 buffer the kernels write this parameter's gradient into.  It is the parameter's own `.grad` when
        that exists (FusedAdam: a view of its flat buffer, so the gradient lands where the optimizer and the all-reduce
        read it), else a buffer of the plan's that `_publish_grads` hands to `p.grad` after every backward (torch
        optimizers' `zero_grad(set_to_none=True)`).  Addresses are stable from step to step: the captured graphs hold them."""
        g = self._gbuf.get(id(p))
        if g is None or (p.grad is not None and p.grad.data_ptr() != g.data_ptr()):
            g = p.grad if p.grad is not None else torch.zeros_like(p, dtype=torch.float32)
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise ValueError("training needs contiguous fp32 gradients")
            self._gbuf[id(p)] = g
        return g

    def _publish_grads(self):
        for p in self._params:
            g = self._gbuf.get(id(p))
            if g is not None and p.grad is None:
                p.grad = g

    def _signature(self) -> Tuple[int, ...]:
        """Every address a captured graph froze: parameters (BatchNorm affine terms are kernel arguments, conv weights
        sit in the pack table) and gradient buffers.  An optimizer that re-seats them (FusedAdam's flat buffers)
        changes it, and the graphs are captured again."""
        sig = []
        for p in self._params:
            sig.append(p.data_ptr())
            sig.append(self._grad_of(p).data_ptr())
        return tuple(sig)

    def _packed(self, w: torch.nn.Parameter, transpose: bool) -> torch.Tensor:
        O, I, kh, kw = w.shape
        cin, cout = (O, I) if transpose else (I, O)
        dst = torch.zeros(self.lib.rsg_conv_packed_bytes(cin, cout, kh, kw), dtype=torch.uint8, device=self.device)
        self._packs.append(dict(w=w, dst=dst, transpose=transpose, blocks=self.lib.rsg_pack_item_blocks(cin, cout, kh, kw)))
        return dst

    def _build_pack_table(self):
        """(Re)build the device table of the weight packer: it holds the parameters' CURRENT addresses (an optimizer
        such as FusedAdam re-seats `.data` into its flat buffer)."""
        items = (hip.PackItem * len(self._packs))()
        blocks = 0
        for it, p in zip(items, self._packs):
            w = p["w"]
            if w.dtype != torch.float32 or not w.is_contiguous():
                raise ValueError("training needs contiguous fp32 master weights")
            O, I, kh, kw = w.shape
            it.w, it.dst, it.cin, it.cout, it.kh, it.kw = w.data_ptr(), p["dst"].data_ptr(), I, O, kh, kw
            it.transpose, it.first_block = int(p["transpose"]), blocks
            blocks += p["blocks"]
        self._pack_items = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(self.device)
        self._pack_blocks = blocks
        self._param_ptrs = tuple(p["w"].data_ptr() for p in self._packs)

    def _conv_desc(self, x: torch.Tensor, wp: torch.Tensor, bias: torch.Tensor, out16: Optional[torch.Tensor],
                   out_f32: Optional[torch.Tensor], H: int, W: int, cin_pad: int, cout: int, k: int, stride: int) -> "hip.ConvDesc":
        pad = (k - 1) // 2
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        d = hip.ConvDesc()
        d.inp, d.w_packed, d.bias = x.data_ptr(), wp.data_ptr(), bias.data_ptr()
        d.out16 = out16.data_ptr() if out16 is not None else None
        d.out_nchw_f32 = out_f32.data_ptr() if out_f32 is not None else None
        d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = self.batch, H, W, cin_pad, cin_pad
        d.out_h, d.out_w, d.out_c, d.out_ld = Ho, Wo, cout, _round_up(cout, 32)
        d.kh, d.kw, d.stride, d.pad_y, d.pad_x = k, k, stride, pad, pad
        d.up, d.up_y, d.up_x, d.phys_h, d.phys_w = 1, 0, 0, Ho, Wo
        d.relu, d.dtype = 0, BF16
        self._keep.append(d)
        return d

    # ------------------------------------------------------------------ forward recording
    def conv(self, x: T, m: nn.Conv2d, name: str) -> T:
        """y = conv(x) (no bias, no activation; BatchNorm follows as its own op)."""
        O, I, k, _ = m.weight.shape
        stride = m.stride[0]
        cin_pad = self.lib.rsg_conv_padded_cin(I)
        if x.C != cin_pad or m.bias is not None or O % 32:
            raise NotImplementedError(f"{name}: conv {I}->{O} on a {x.C}-channel map")
        pad = (k - 1) // 2
        y = T(self, (x.H + 2 * pad - k) // stride + 1, (x.W + 2 * pad - k) // stride + 1, O, name)
        wp = self._packed(m.weight, False)
        d = self._conv_desc(x.data, wp, self._zero_bias, y.data, None, x.H, x.W, cin_pad, O, k, stride)
        fn, ref = self.lib.rsg_conv_igemm, C.byref(d)
        gk = "conv" if self.lib.rsg_conv_groupable(ref) else None
        self._emit("fwd", name, lambda s, fn=fn, ref=ref, nm=name: self._chk(fn(ref, s), nm), gk, d)
        self._emit_op(dict(kind="conv", x=x, y=y, m=m, name=name, cin_pad=cin_pad))
        self.flops_fwd += 2 * self.batch * y.H * y.W * O * I * k * k
        return y

    def bn(self, y: T, m: nn.BatchNorm2d, relu: bool, res: Optional[T], name: str) -> T:
        """z = [ReLU](BatchNorm_train(y) [+ res]); running statistics updated in place (momentum m.momentum)."""
        z = T(self, y.H, y.W, y.C, name)
        mean = torch.zeros(y.C, dtype=torch.float32, device=self.device)
        invstd = torch.zeros(y.C, dtype=torch.float32, device=self.device)
        lib, px, Cc = self.lib, y.pixels, y.C
        if m.momentum is None or not m.track_running_stats or m.running_mean is None or not m.affine:
            raise NotImplementedError(f"{name}: BatchNorm2d without momentum / running statistics / affine terms "
                                      "(the reference builds every BatchNorm2d with momentum 0.1 and the defaults)")
        mom, eps = float(m.momentum), float(m.eps)

        ws = torch.empty(lib.rsg_bn_workspace_bytes(Cc), dtype=torch.uint8, device=self.device)   # own: BatchNorms may share a launch

        def item(it):
            it.y, it.out, it.residual = y.data.data_ptr(), z.data.data_ptr(), (res.data.data_ptr() if res is not None else None)
            it.gamma, it.beta, it.mean, it.invstd = m.weight.data_ptr(), m.bias.data_ptr(), mean.data_ptr(), invstd.data_ptr()
            it.running_mean, it.running_var, it.workspace = m.running_mean.data_ptr(), m.running_var.data_ptr(), ws.data_ptr()
            it.pixels, it.c, it.relu = px, Cc, int(relu)

        def launch(s):
            self._chk(lib.rsg_bn_stats(y.data.data_ptr(), px, Cc, BF16, ws.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                       m.running_mean.data_ptr(), m.running_var.data_ptr(), mom, eps, s), name)
            self._chk(lib.rsg_bn_apply(y.data.data_ptr(), mean.data_ptr(), invstd.data_ptr(), m.weight.data_ptr(),
                                       m.bias.data_ptr(), res.data.data_ptr() if res is not None else None, int(relu),
                                       z.data.data_ptr(), px, Cc, BF16, s), name)

        if (mom, eps) != (self._bn_momentum, self._bn_eps):
            raise NotImplementedError(f"{name}: BatchNorm momentum/eps {mom}/{eps} (the reference uses 0.1 / 1e-5 everywhere)")
        self._emit("fwd", name, launch, "bn_fwd", item)
        self._emit_op(dict(kind="bn", y=y, z=z, m=m, relu=relu, res=res, mean=mean, invstd=invstd, name=name, ws=ws))
        return z

    def unit(self, x: T, conv: nn.Conv2d, bn: nn.BatchNorm2d, relu: bool, name: str, res: Optional[T] = None) -> T:
        return self.bn(self.conv(x, conv, name), bn, relu, res, name + ".bn")

    def fuse_sum(self, same: Sequence[T], low: Sequence[Tuple[T, int]], name: str) -> T:
        """out = ReLU(sum same + sum nearest_up(low)) (pose_hrnet.py:259-266)."""
        ref = same[0]
        out = T(self, ref.H, ref.W, ref.C, name)
        sp = (C.c_void_p * len(same))(*[t.data.data_ptr() for t in same])
        lp = (C.c_void_p * max(len(low), 1))(*[t.data.data_ptr() for t, _ in low])
        ls = (C.c_int32 * max(len(low), 1))(*[s for _, s in low])
        self._keep += [sp, lp, ls]
        lib, B = self.lib, self.batch
        self._emit("fwd", name, lambda s: self._chk(lib.rsg_hr_fuse_sum(sp, len(same), lp, ls, len(low), out.data.data_ptr(), B,
                                                                     out.H, out.W, out.C, BF16, s), name))
        self._emit_op(dict(kind="fuse", same=list(same), low=list(low), out=out, name=name))
        return out

    def _record(self, m: nn.Module):
        lib, B, H, W = self.lib, self.batch, self.H, self.W
        x16 = T(self, H, W, 16, "x_nhwc16", needs_grad=False)
        self._emit("fwd", "input", lambda s: self._chk(lib.rsg_nchw_f32_to_nhwc16(self.x_f32.data_ptr(), x16.data.data_ptr(), B, 3,
                                                                              H, W, 16, BF16, s), "input"))
        x = self.unit(x16, m.conv1, m.bn1, True, "conv1")
        x = self.unit(x, m.conv2, m.bn2, True, "conv2")
        for k, blk in enumerate(m.layer1):
            x = self._block(x, blk, f"layer1.{k}")
        ys = [x]
        for s in (2, 3, 4):
            trans = getattr(m, f"transition{s - 1}")
            xs = []
            for i, tr in enumerate(trans):
                if i < len(ys):
                    xs.append(ys[i] if tr is None else self.unit(ys[i], tr[0], tr[1], True, f"transition{s - 1}.{i}"))
                else:
                    t = ys[-1]
                    for j, step in enumerate(tr):
                        t = self.unit(t, step[0], step[1], True, f"transition{s - 1}.{i}.{j}")
                    xs.append(t)
            for k, mod in enumerate(getattr(m, f"stage{s}")):
                xs = self._hr_module(xs, mod, f"stage{s}.{k}")
            ys = xs
        if not self.with_head:
            x0 = ys[0]
            if x0.C != m.final_layer.in_channels:
                raise NotImplementedError(f"backbone output with {x0.C} channels in a padded buffer (HRNet-W48 training is not built)")
            self.feat_src = x0
            self.dfeat16 = torch.zeros_like(x0.data)      # the incoming gradient, NHWC bf16 (TrainPlan.backward fills it)
            self._emit_op(dict(kind="feat", x=x0, name="backbone_output"))
            return
        # classifier: 1x1 conv + bias -> fp32 NCHW heat maps (pose_hrnet.py:326-332,461)
        fl = m.final_layer
        if fl.kernel_size != (1, 1):
            raise NotImplementedError("FINAL_CONV_KERNEL=3")
        x0 = ys[0]
        J = self.J
        wp = self._packed(fl.weight, False)
        bias32 = torch.zeros(_round_up(J, 32), dtype=torch.float32, device=self.device)
        d = self._conv_desc(x0.data, wp, bias32, None, self.heat, x0.H, x0.W, x0.C, J, 1, 1)
        ref = C.byref(d)

        def launch(s):
            bias32[:J].copy_(fl.bias.detach())
            self._chk(lib.rsg_conv_igemm(ref, s), "final_layer")

        self._emit("fwd", "final_layer", launch)
        self._emit_op(dict(kind="final", x=x0, m=fl, name="final_layer"))
        self.flops_fwd += 2 * B * x0.H * x0.W * J * x0.C

    def _block(self, x: T, blk, name: str) -> T:
        """BasicBlock / Bottleneck in training mode (pose_hrnet.py:30-101)."""
        res = x
        if blk.downsample is not None:
            res = self.unit(x, blk.downsample[0], blk.downsample[1], False, name + ".downsample")
        y = self.unit(x, blk.conv1, blk.bn1, True, name + ".conv1")
        if hasattr(blk, "conv3"):
            y = self.unit(y, blk.conv2, blk.bn2, True, name + ".conv2")
            return self.unit(y, blk.conv3, blk.bn3, True, name + ".conv3", res=res)
        return self.unit(y, blk.conv2, blk.bn2, True, name + ".conv2", res=res)

    def _hr_module(self, xs: List[T], mod, name: str) -> List[T]:
        """HighResolutionModule.forward (pose_hrnet.py:250-268), training mode."""
        xs = list(xs)
        same_shape = len({len(br) for br in mod.branches}) == 1 and len(mod.branches) > 1 and self.use_lanes
        if same_shape:
            self._begin_lanes(len(mod.branches))
        for b, branch in enumerate(mod.branches):
            if same_shape:
                self._lane(b)
            for k, blk in enumerate(branch):
                xs[b] = self._block(xs[b], blk, f"{name}.branches.{b}.{k}")
        if same_shape:
            self._end_lanes()
        if mod.fuse_layers is None:
            return xs
        # the fuse paths (i <- j) are independent of each other: recorded as lanes, so that the k-th conv / BatchNorm of
        # every path (and, in the backward pass, their weight / data gradients) share grouped launches
        paths = [(i, j) for i, row in enumerate(mod.fuse_layers) for j in range(len(row)) if j != i]
        lanes = self.use_lanes and len(paths) > 1 and sw.train_on("fuse_lanes")
        done: Dict[Tuple[int, int], T] = {}
        if lanes:
            self._begin_lanes(len(paths), shared=True)
        for k, (i, j) in enumerate(paths):
            if lanes:
                self._lane(k)
            fl = mod.fuse_layers[i][j]
            if j > i:
                done[(i, j)] = self.unit(xs[j], fl[0], fl[1], False, f"{name}.fuse.{i}.{j}")
            else:
                t = xs[j]
                for q, step in enumerate(fl):
                    t = self.unit(t, step[0], step[1], q != len(fl) - 1, f"{name}.fuse.{i}.{j}.{q}")
                done[(i, j)] = t
        if lanes:
            self._end_lanes()
        outs = []
        for i, row in enumerate(mod.fuse_layers):
            same = [done[(i, j)] for j in range(i)] + [xs[i]]
            low = [(done[(i, j)], j - i) for j in range(i + 1, len(row))]
            outs.append(self.fuse_sum(same, low, f"{name}.fuse.{i}"))
        return outs

    # ------------------------------------------------------------------ backward recording
    def _contribute(self, t: T, src: torch.Tensor, shared: bool, name: str):
        """`src` (a gradient of t) joins t.grad: the first one becomes it, later ones are added.  A buffer that several
        maps alias (`shared`) is never written in place."""
        if not t.needs_grad:
            return
        if self._defer is not None:              # lanes with common inputs: summed after the lanes' launches
            self._defer.append((t, src, shared, name))
            return
        if t.grad is None:
            t.grad, t.grad_shared = src, shared
            return
        lib, n = self.lib, src.numel()
        a = t.grad
        dst = torch.empty_like(a) if t.grad_shared else a
        self._emit("bwd", name + ".add", lambda s: self._chk(lib.rsg_add16(a.data_ptr(), src.data_ptr(), dst.data_ptr(), n, BF16, s), name))
        t.grad, t.grad_shared = dst, False

    def _new_like(self, t: T) -> torch.Tensor:
        return torch.zeros_like(t.data)

    def _record_backward(self):
        lib, B = self.lib, self.batch
        bucket: Dict[int, list] = {}             # lane -> backward launches of the lanes' ops at the current position
        cur_key = None

        pending: list = []                        # gradient contributions of lanes with common inputs (see _contribute)

        def rank(rec):
            # spread (zero insertion), weight gradients, data gradients / BatchNorm, the rest (adds): every lane's own
            # order is a subsequence of this one, and launches of one kind end up adjacent, where _merge groups them
            name, gkind = rec[0], rec[2] or ""
            return 0 if name.endswith(".spread") else 1 if gkind.startswith("wgrad") else \
                2 if gkind in ("conv", "bn_bwd") or name.endswith(".dgrad") else 3

        def flush():
            nonlocal bucket
            self._sink["bwd"], self._defer = None, None
            if bucket:
                recs = self._zip([bucket[k] for k in sorted(bucket, reverse=True)])
                self.bwd.extend(self._merge(sorted(recs, key=rank)))
                bucket = {}
            for args in pending:
                self._contribute(*args)
            pending.clear()

        for op in reversed(self._ops):
            kind, name = op["kind"], op["name"]
            reg = op.get("region")
            key = (reg[0], reg[2]) if reg is not None else None
            if key != cur_key:
                flush()
                cur_key = key
            self._sink["bwd"] = bucket.setdefault(reg[1], []) if reg is not None else None
            self._defer = pending if reg is not None and reg[0] in self._shared_rids else None
            self._bwd_lane = reg[1] if reg is not None else 0
            if kind == "feat":
                self._contribute(op["x"], self.dfeat16, False, name)
            elif kind == "final":
                x, fl = op["x"], op["m"]
                J = self.J
                d16 = torch.zeros(B, x.H, x.W, 32, dtype=torch.bfloat16, device=self.device)
                self._emit("bwd", name + ".dheat", lambda s, d16=d16, x=x: self._chk(
                    lib.rsg_nchw_f32_to_nhwc16(self.dheat.data_ptr(), d16.data_ptr(), B, J, x.H, x.W, 32, BF16, s), "dheat"))
                csw = torch.empty(lib.rsg_channel_sum_workspace_bytes(J), dtype=torch.uint8, device=self.device)
                self._emit("bwd", name + ".dbias", lambda s, fl=fl, x=x, csw=csw: self._chk(
                    lib.rsg_channel_sum_nchw_f32(self.dheat.data_ptr(), self._grad_of(fl.bias).data_ptr(), csw.data_ptr(), B, J, x.H * x.W, s), "dbias"),
                    params=(fl.bias,))
                self._wgrad(x.data, d16, fl.weight, x.H, x.W, x.C, x.C, x.H, x.W, J, 32, 1, 1, name)
                self._dgrad(x, d16, fl.weight, 32, 1, 1, x.H, x.W, name)
            elif kind == "fuse":
                out = op["out"]
                if out.grad is None:
                    continue                              # an output nobody consumed (does not occur in pose_hrnet)
                g = torch.zeros_like(out.data)
                dout, n = out.grad, out.data.numel()
                self._emit("bwd", name + ".relu", lambda s, dout=dout, out=out, g=g, n=n: self._chk(
                    lib.rsg_relu_backward(dout.data_ptr(), out.data.data_ptr(), g.data_ptr(), n, BF16, s), "relu_backward"))
                for t in op["same"]:
                    self._contribute(t, g, True, name)
                for t, sh in op["low"]:
                    lo = torch.zeros_like(t.data)
                    self._emit("bwd", name + ".up", lambda s, g=g, lo=lo, t=t, sh=sh: self._chk(
                        lib.rsg_upsample_nearest_backward(g.data_ptr(), lo.data_ptr(), B, t.H, t.W, t.C, sh, BF16, s), "upsample_bwd"))
                    self._contribute(t, lo, False, name)
            elif kind == "bn":
                y, z, m, res = op["y"], op["z"], op["m"], op["res"]
                if z.grad is None:
                    raise RuntimeError(f"{name}: output has no gradient (dead branch)")
                dz = z.grad
                y.grad = torch.zeros_like(y.data)
                dres = torch.zeros_like(z.data) if res is not None and res.needs_grad else None
                mean, invstd, relu, ws, px, Cc = op["mean"], op["invstd"], op["relu"], op["ws"], y.pixels, y.C

                def item(it, dz=dz, y=y, z=z, m=m, dres=dres, mean=mean, invstd=invstd, relu=relu, px=px, Cc=Cc, ws=ws):
                    it.y, it.dz, it.z = y.data.data_ptr(), dz.data_ptr(), (z.data.data_ptr() if relu else None)
                    it.out, it.dres = y.grad.data_ptr(), (dres.data_ptr() if dres is not None else None)
                    it.gamma, it.mean, it.invstd = m.weight.data_ptr(), mean.data_ptr(), invstd.data_ptr()
                    it.dgamma, it.dbeta, it.workspace = self._grad_of(m.weight).data_ptr(), self._grad_of(m.bias).data_ptr(), ws.data_ptr()
                    it.pixels, it.c, it.relu = px, Cc, int(relu)

                def launch(s, dz=dz, y=y, z=z, m=m, dres=dres, mean=mean, invstd=invstd, relu=relu, px=px, Cc=Cc, ws=ws):
                    self._chk(lib.rsg_bn_backward(dz.data_ptr(), z.data.data_ptr() if relu else None, y.data.data_ptr(),
                                                  mean.data_ptr(), invstd.data_ptr(), m.weight.data_ptr(), y.grad.data_ptr(),
                                                  dres.data_ptr() if dres is not None else None,
                                                  self._grad_of(m.weight).data_ptr(), self._grad_of(m.bias).data_ptr(),
                                                  ws.data_ptr(), px, Cc, BF16, s), "bn_backward")

                self._emit("bwd", name, launch, "bn_bwd", item, params=(m.weight, m.bias))
                if dres is not None:
                    self._contribute(res, dres, False, name)
            elif kind == "conv":
                x, y, m = op["x"], op["y"], op["m"]
                O, I, k, _ = m.weight.shape
                stride = m.stride[0]
                self._wgrad(x.data, y.grad, m.weight, x.H, x.W, I, op["cin_pad"], y.H, y.W, O, O, k, stride, name)
                if x.needs_grad:
                    self._dgrad(x, y.grad, m.weight, O, k, stride, y.H, y.W, name)
        flush()
        self._sink["bwd"], self._defer = None, None
        # Segments of the backward list for the data-parallel overlap: cut where the pass leaves stage 4 and stage 3 (the
        # two stages hold 93 % of the parameters, and the flat gradient buffer keeps parameters in forward order, so
        # what is final after a segment is a tail of that buffer).  After each segment the model's grad_ready_hooks
        # are told which parameters are still pending.
        def top(nm):
            return nm.split(".")[0]

        def module_of(nm):                       # "stage4.2.fuse.1.0" -> "stage4.2"; everything else by its top-level name
            parts = nm.split(".")
            return ".".join(parts[:2]) if parts[0].startswith("stage") and len(parts) > 1 else parts[0]
        cuts = []
        seg_mode = sw.train("segments") if self.wgrad_split else "stage"
        by_module = seg_mode == "module"
        every = int(seg_mode.split(":")[1]) if seg_mode.startswith("every:") else 0
        for k in range(1, len(self.bwd)):
            if every:
                if k % every == 0:
                    cuts.append(k)
                continue
            if by_module:
                # with the weight gradients of a segment running beside the NEXT segment's chain (wgrad_split), finer segments
                # overlap more of them: one per HighResolutionModule, layer1, the stem
                if module_of(self.bwd[k - 1][0]) != module_of(self.bwd[k][0]):
                    cuts.append(k)
                continue
            a, b = top(self.bwd[k - 1][0]), top(self.bwd[k][0])
            if a != b and a in ("stage4", "stage3") and b not in ("stage4", "stage3"):
                cuts.append(k)
        bounds = [0] + cuts + [len(self.bwd)]
        self._bwd_segments = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1) if bounds[i] < bounds[i + 1]]
        done = set()
        self._pending_after = []
        for lo, hi in self._bwd_segments:
            for rec in self.bwd[lo:hi]:
                done.update(id(q) for q in rec[4])
            self._pending_after.append([q for q in self._params if id(q) not in done])

    def _link_wgrads(self):
        """Chain the weight-gradient launches of every backward segment: each defers its partial-tile reduction to the next
        one (which runs it beside its own main kernel), the last of a segment reduces its own -- gradients are final where
        the segment ends, which is where grad_ready_hooks and the optimizer look."""
        # Alone it is neutral (13.40 vs 13.37 ms: the two kernels compete for the same memory system); it pays together
        # with main kernels that leave a quarter of the CUs to the reduction blocks (csrc/wgrad.hip, wgrad_group_tables):
        # 13.40 -> 13.07 ms per step
        if not sw.train_on("wgrad_pipe"):
            return
        for lo, hi in self._bwd_segments:
            chain = [rec[1] for rec in self.bwd[lo:hi] if isinstance(rec[1], _WgradLaunch)]
            for k, wl in enumerate(chain):
                wl.par = k & 1
                if k:
                    wl.link(chain[k - 1])

    def _wgrad(self, x: torch.Tensor, dy: torch.Tensor, w: torch.nn.Parameter, H, W, cin, cin_pad, Ho, Wo, cout, cout_pad, k, stride, name):
        lib, B, pad = self.lib, self.batch, (k - 1) // 2
        need = int(lib.rsg_conv_wgrad_workspace_bytes(B, H, W, cin_pad, Ho, Wo, cout_pad, k, k, stride, pad))
        if not need:
            raise NotImplementedError(f"{name}: weight gradient of a {k}x{k} stride-{stride} conv")

        lane = self._bwd_lane                    # grouped launches run several weight gradients at once: a workspace per lane
        self._wgrad_lane_need[lane] = max(self._wgrad_lane_need.get(lane, 0), need)

        def item(it, par=0):
            it.x, it.dy, it.dw, it.workspace = x.data_ptr(), dy.data_ptr(), self._grad_of(w).data_ptr(), self._wgrad_lane_ws[(par, lane)].data_ptr()
            it.batch, it.in_h, it.in_w, it.cin, it.cin_pad = B, H, W, cin, cin_pad
            it.out_h, it.out_w, it.cout, it.cout_pad, it.kh, it.kw, it.stride, it.pad = Ho, Wo, cout, cout_pad, k, k, stride, pad

        launch = _WgradLaunch(self, [item])      # replaced by a grouped launch where _merge finds neighbours of its kind
        self._emit("bwd", name + ".wgrad", launch, f"wgrad{k}", item, params=(w,))

    def _dgrad(self, x: T, dy: torch.Tensor, w: torch.nn.Parameter, dy_c: int, k: int, stride: int, Ho: int, Wo: int, name: str):
        """dx = conv_transpose(dy, w): a stride-1 conv of dy (stride 2: of dy spread onto the input grid with zeros
        between) with the transposed, flipped weights (rsg_pack_conv_weights_device, transpose = 1)."""
        lib, B = self.lib, self.batch
        wp = self._packed(w, True)
        src, sh, swd = dy, Ho, Wo
        if stride == 2:
            if (2 * Ho, 2 * Wo) != (x.H, x.W):
                raise NotImplementedError(f"{name}: stride-2 data gradient on an odd map")
            src = torch.zeros(B, 2 * Ho, 2 * Wo, dy_c, dtype=torch.bfloat16, device=self.device)
            self._emit("bwd", name + ".spread", lambda s: self._chk(lib.rsg_zero_insert2x(dy.data_ptr(), src.data_ptr(), B, Ho, Wo, dy_c, s), name))
            sh, swd = 2 * Ho, 2 * Wo
        # x already has a gradient of its own (a BasicBlock's input: the residual branch of bn2 came first): the conv adds
        # it in its epilogue and writes in place -- one rounding instead of two, and no separate add launch (104 per step)
        fold = self._defer is None and x.needs_grad and x.grad is not None and not x.grad_shared and \
            sw.train_on("fold_add")
        dx = x.grad if fold else torch.zeros_like(x.data)
        cin_real = w.shape[1]
        d = self._conv_desc(src, wp, self._zero_bias, dx, None, sh, swd, dy_c, cin_real, k, 1)
        if _round_up(cin_real, 32) != x.C:
            raise NotImplementedError(f"{name}: data gradient into a {x.C}-channel map")
        if fold:
            d.residual, d.res_ld = dx.data_ptr(), x.C
        ref = C.byref(d)
        self._emit("bwd", name + ".dgrad", lambda s: self._chk(lib.rsg_conv_igemm(ref, s), name + ".dgrad"),
                   "conv" if lib.rsg_conv_groupable(ref) else None, d)
        if not fold:
            self._contribute(x, dx, False, name)

    # ------------------------------------------------------------------ execution
    def _run_forward(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        self._chk(self.lib.rsg_pack_conv_weights_device(self._pack_items.data_ptr(), len(self._packs), self._pack_blocks, BF16, s),
                  "pack_weights")
        for rec in self.fwd:
            rec[1](s)
        torch._foreach_add_(self._nbt, 1)            # num_batches_tracked of every BatchNorm, one launch

    def _run_backward(self, seg: Optional[int] = None, which: Optional[str] = None):
        """which: None = every launch of the segment; "main" / "wgrad" = all but / only the weight-gradient launches
        (wgrad_mode 2: two graphs per segment, the second replayed on the side stream behind the first)."""
        main = torch.cuda.current_stream(self.device)
        s = main.cuda_stream
        lo, hi = (0, len(self.bwd)) if seg is None else self._bwd_segments[seg]
        if which is not None:
            for rec in self.bwd[lo:hi]:
                if isinstance(rec[1], _WgradLaunch) == (which == "wgrad"):
                    rec[1](s)
            return
        if not self.wgrad_stream_on:
            for rec in self.bwd[lo:hi]:
                rec[1](s)
            return
        # The weight-gradient launches on a stream of their own, beside the data-gradient / BatchNorm chain: a weight
        # gradient reads a forward activation and the gradient its BatchNorm's backward wrote (both final by then and not
        # written again in this pass) and writes only its parameter's gradient and the weight-gradient workspaces, whose
        # order the side stream keeps.  It waits for the main stream's position it was recorded at; the main stream waits
        # for the side stream where the segment ends (where grad_ready_hooks and the optimizer look).  Inside a graph
        # capture the events become edges of the graph.
        if self._wgrad_stream is None:
            self._wgrad_stream = torch.cuda.Stream(self.device)
        side, keep, behind = self._wgrad_stream, [], False
        self._bwd_events = keep                     # events of a capture must outlive it
        for rec in self.bwd[lo:hi]:
            if isinstance(rec[1], _WgradLaunch):
                if not behind:                      # one wait per run of weight-gradient launches
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                    keep.append(ev)
                    behind = True
                rec[1](side.cuda_stream)
            else:
                rec[1](s)
                behind = False
        ev = torch.cuda.Event()
        ev.record(side)
        main.wait_event(ev)
        keep.append(ev)

    def _prepare(self):
        """Before a forward: follow re-seated parameters / gradients, and (re)capture the two HIP graphs once the
        addresses have been stable for a step (the ~3 000 launches of a step are host-bound when enqueued one by one)."""
        if tuple(p["w"].data_ptr() for p in self._packs) != self._param_ptrs:
            self._build_pack_table()
        sig = self._signature()
        if sig != self._sig:
            self._sig, self._graphs, self._warm = sig, None, 0

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if tuple(x.shape) != tuple(self.x_f32.shape):
            raise ValueError(f"input {tuple(x.shape)} != plan {tuple(self.x_f32.shape)}")
        self._prepare()
        # ONE static activation/gradient arena per plan: this forward overwrites what an earlier, not yet
        # back-propagated forward left there.  The generation number lets backward() refuse a stale node.
        self.generation = getattr(self, "generation", 0) + 1
        self._backward_done = False
        self.x_f32.copy_(x)
        if self.use_graph and self._graphs is None and self._warm >= 1:
            torch.cuda.synchronize(self.device)
            gf = torch.cuda.CUDAGraph()
            gb = [torch.cuda.CUDAGraph() for _ in self._bwd_segments]
            nbt = [t.clone() for t in self._nbt]
            stats = [(m.running_mean.clone(), m.running_var.clone()) for m in self._bns]
            with torch.cuda.graph(gf):
                self._run_forward()
            gw = [torch.cuda.CUDAGraph() for _ in self._bwd_segments] if self.wgrad_split else None
            for k, g in enumerate(gb):
                with torch.cuda.graph(g):
                    self._run_backward(k, "main" if self.wgrad_split else None)
                if gw is not None:
                    with torch.cuda.graph(gw[k]):
                        self._run_backward(k, "wgrad")
            self._wgrad_graphs = gw
            # capture does not execute, but be explicit that nothing observable moved
            for t, c in zip(self._nbt, nbt):
                t.copy_(c)
            for m, (rm, rv) in zip(self._bns, stats):
                m.running_mean.copy_(rm)
                m.running_var.copy_(rv)
            self._graphs = (gf, gb)
        if self._graphs is not None:
            self._graphs[0].replay()
        else:
            self._run_forward()
        if not self.with_head:
            if self.port16:                               # the head's HIP operators take the plan's own format
                return self.feat_src.data.clone()
            return self.feat_src.data.permute(0, 3, 1, 2).float().contiguous()
        return self.heat.clone()

    def backward(self, dheat: torch.Tensor, generation: Optional[int] = None) -> None:
        """Parameter gradients are OVERWRITTEN (the state after `optimizer.zero_grad()`, which the reference's loop
        calls before every backward, lib/core/function.py:60-62), not accumulated into."""
        if generation is not None and generation != self.generation:
            raise RuntimeError(
                "backward of a training-mode forward whose activations are gone: the model ran another training-mode "
                f"forward (generation {self.generation}) on the same (batch, device) before this one (generation "
                f"{generation}) was back-propagated. The HIP training plan keeps ONE activation arena per batch size; "
                "back-propagate each forward before the next one (micro-batches: loss.backward() per micro-batch and "
                "accumulate outside, or use different batch sizes).")
        if self._backward_done:
            raise RuntimeError("second backward through the same training-mode forward: its buffers were reused "
                               "(run the forward again; retain_graph is not supported by the HIP training plan)")
        self._backward_done = True
        if self.with_head:
            self.dheat.copy_(dheat)
        elif self.port16:
            self.dfeat16.copy_(dheat)                              # already NHWC (bf16, or fp32 from an accumulating autograd)
        else:
            self.dfeat16.copy_(dheat.permute(0, 2, 3, 1))          # fp32 NCHW -> bf16 NHWC (layout and type in one copy)
        hooks = getattr(self.model, "grad_ready_hooks", ())
        if self.wgrad_split:
            main = torch.cuda.current_stream(self.device)
            if self._wgrad_stream is None:
                self._wgrad_stream = torch.cuda.Stream(self.device)
            side = self._wgrad_stream
            for k in range(len(self._bwd_segments)):
                if self._graphs is not None:
                    self._graphs[1][k].replay()
                else:
                    self._run_backward(k, "main")
                side.wait_stream(main)               # the segment's gradient maps are final where the main stream stands now
                with torch.cuda.stream(side):
                    if self._graphs is not None:
                        self._wgrad_graphs[k].replay()
                    else:
                        self._run_backward(k, "wgrad")
                    if k + 1 < len(self._bwd_segments):
                        for cb in hooks:             # the segment's parameter gradients are final on the SIDE stream
                            cb(self._pending_after[k])
            main.wait_stream(side)                   # the optimizer, and the next forward's writes to the activations
        else:
            for k in range(len(self._bwd_segments)):
                if self._graphs is not None:
                    self._graphs[1][k].replay()
                else:
                    self._run_backward(k)
                if k + 1 < len(self._bwd_segments):
                    for cb in hooks:                 # e.g. DataParallelStep: all-reduce the finished tail under the rest
                        cb(self._pending_after[k])
        if self._graphs is None:
            self._warm += 1
        self._publish_grads()


class _TrainFn(torch.autograd.Function):
    """The whole training-mode forward as one autograd node: backward = the HIP backward launch list, parameter
    gradients are written into `p.grad` by the kernels (overwritten, as after `optimizer.zero_grad()`)."""

    @staticmethod
    def forward(ctx, hook: torch.Tensor, plan: TrainPlan, x: torch.Tensor):
        ctx.plan = plan
        with torch.cuda.device(x.device):
            out = plan.forward(x)
        ctx.generation = plan.generation
        return out

    @staticmethod
    def backward(ctx, dheat: torch.Tensor):
        plan: TrainPlan = ctx.plan
        with torch.cuda.device(dheat.device):
            plan.backward(dheat.contiguous() if plan.port16 else dheat.contiguous().float(), ctx.generation)
        return None, None, None


def train_forward(model: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """`model(x)` in training mode (models/_base.py calls this): the heat maps of pose_hrnet, or -- for a model whose
    head runs elsewhere (`model.train_head_on_torch`, pose_rsgnet) -- the backbone's output map [B,C0,H/4,W/4] fp32."""
    key = (int(x.shape[0]), x.device.index)
    plan = model._train_plans.get(key)
    if plan is None:
        with torch.cuda.device(x.device):
            plan = TrainPlan(model, int(x.shape[0]), x.device, head=not getattr(model, "train_head_on_torch", False))
        model._train_plans[key] = plan
    hook = getattr(model, "_train_hook", None)
    if hook is None or hook.device != x.device:
        hook = model._train_hook = torch.zeros(1, device=x.device, requires_grad=True)
    return _TrainFn.apply(hook, plan, x)
