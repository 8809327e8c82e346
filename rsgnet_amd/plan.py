"""Execution plans: the reference's nn.Module graphs lowered to a flat list of HIP kernel launches.

A plan is built once per (model weights, batch size, device):
  1. *record*  — walk the model (same order as /root/reference/lib/models/pose_hrnet.py:428-463 and
     pose_rsgnet.py:921-1021) and record ops on virtual NHWC 16-bit tensors; fold every BatchNorm
     into the preceding conv and pack the weights into MFMA fragment order (C packer);
  2. *allocate* — linear-scan the op list and give virtual tensors physical buffers, reusing a buffer
     as soon as its last reader has run (keeps the working set inside the 256 MB Infinity Cache);
  3. *lower*   — fill one ``rsg_conv_desc`` (or argument tuple) per op with raw device pointers;
  4. *run*     — enqueue the launches on the caller's stream, or replay them from a captured HIP graph.

Only pointers and sizes cross into librsgnet_hip.so; torch is used for memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import hip
from . import switches as sw

BN_EPS = 1e-5


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


class HandoffTimeout(RuntimeError):
    """A consumer chain of a grouped launch gave up waiting for its producer (csrc/chain.hip): the forwards since
    the last clean poll may hold wrong pixels.  The plan has re-zeroed its flag words; models/_base.py rebuilds
    without the hand-off."""


class VT:
    """Virtual NHWC 16-bit activation [B,H,W,ld]; ``view(c0, c)`` addresses a channel slice of it."""

    __slots__ = ("H", "W", "C", "ld", "coff", "base", "pinned", "first", "last", "storage", "name", "planes")

    def __init__(self, H, W, C_, ld=None, pinned=False, name="", planes=False):
        self.H, self.W, self.C = H, W, C_
        self.ld = ld if ld is not None else C_
        # planes: the map is stored as channel-tile planes [C/32][B*H*W][32] (include/rsgnet_hip.h, RSG_LAYOUT_*): only
        # conv_pair tails write / add it and conv_dual reads it; `ld` keeps the dense value (it sizes the buffer)
        self.planes = planes
        self.coff = 0
        self.base: "VT" = self
        self.pinned = pinned
        self.first = self.last = -1
        self.storage: Optional[torch.Tensor] = None
        self.name = name

    def view(self, c0: int, c: int) -> "VT":
        v = VT(self.H, self.W, c, self.ld, name=self.name + f"[{c0}:{c0 + c}]")
        v.coff = self.coff + c0
        v.base = self.base
        return v

    def ptr(self) -> int:
        return self.base.storage.data_ptr() + self.coff * 2

    def nbytes(self, batch: int) -> int:
        return batch * self.H * self.W * self.ld * 2

    def tensor(self, batch: int) -> torch.Tensor:
        """[B,H,W,C] view of the live buffer (debug / tests)."""
        if self.planes:
            t = self.base.storage[: batch * self.H * self.W * self.C].view(self.C // 32, batch, self.H, self.W, 32)
            return t.permute(1, 2, 3, 0, 4).reshape(batch, self.H, self.W, self.C)
        t = self.base.storage[: batch * self.H * self.W * self.ld].view(batch, self.H, self.W, self.ld)
        return t[..., self.coff: self.coff + self.C]


class OutSlot:
    """An fp32 output of the plan whose address may change from call to call: the ops that write it run eagerly
    behind the captured graph (`Plan.n_tail`), so a forward can hand out fresh tensors instead of clones of static
    buffers (RSGNet: three 53 MB maps per batch of 64)."""

    def __init__(self, static: torch.Tensor):
        self.static = static
        self.ptr = static.data_ptr()

    def fresh(self) -> torch.Tensor:
        t = torch.empty_like(self.static)
        self.ptr = t.data_ptr()
        return t

    def use_static(self) -> torch.Tensor:
        self.ptr = self.static.data_ptr()
        return self.static


class Plan:
    def __init__(self, model: nn.Module, batch: int, device: torch.device, dtype: str):
        if dtype not in hip.DTYPES:
            raise ValueError(f"compute dtype {dtype!r}; expected one of {list(hip.DTYPES)}")
        self.lib = hip.lib()
        self.batch = batch
        self.device = device
        self.dtype_name = dtype
        self.dtype = hip.DTYPES[dtype]
        self.tdtype = torch.float16 if dtype == "fp16" else torch.bfloat16
        self.ops: List[Tuple[str, Callable[[int], None]]] = []
        self.n_tail = 0            # trailing ops that run eagerly behind the captured graph (they write OutSlots)
        self._rec: List[dict] = []          # recorded ops before lowering
        self._keep: List[object] = []       # ctypes structs / tensors that must outlive the plan
        self._vts: List[VT] = []
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.graph_mode = sw.plan("graph")
        self.use_graph = self.graph_mode != "0"
        self.use_chain = sw.plan_on("chain")
        self.use_group = sw.plan_on("group")
        self.conv_log: List[dict] = []
        self.flops = 0
        # lanes = HIP streams the launches are spread over (independent HRNet branches run concurrently);
        # cross-lane ordering comes from events derived from each op's buffer reads/writes
        # a plan built for a side stream expects to share the GPU with another stream's forward: tuned for CU-time, not makespan
        self.throughput = bool(getattr(model, "_plan_for_side_stream", False))
        self.n_lanes = sw.plan_int("lanes")
        self._lane = 0
        self._streams: List[torch.cuda.Stream] = []
        # in-launch hand-off (chain_group items that wait on an earlier item): off after a failure (models/_base.py)
        self.allow_handoff = sw.plan_on("chain_handoff") and \
            not getattr(model, "_no_handoff", False)
        self._ho_flags: List[torch.Tensor] = []       # every flag buffer of this plan (re-zeroed after a timeout)
        self._ho_seen: Optional[int] = None           # error-word value the last time it was read
        self._ho_pinned: Optional[torch.Tensor] = None
        self._ho_pending: Optional[torch.cuda.Event] = None
        self._runs = 0
        # forwards between two asynchronous reads of the error word: every forward enqueues the 4-byte copy behind its
        # launches and looks at the previous forward's copy first, so at most ONE batch of outputs can leave a plan
        # whose hand-off gave up before HandoffTimeout is raised (advisor, round 2: was 32)
        self.handoff_poll_every = sw.plan_int("handoff_poll_every")

    class _LaneCtx:
        def __init__(self, plan, lane):
            self.plan, self.lane, self.prev = plan, lane, 0

        def __enter__(self):
            self.prev = self.plan._lane
            self.plan._lane = self.lane % self.plan.n_lanes
            return self

        def __exit__(self, *a):
            self.plan._lane = self.prev

    def on_lane(self, lane: int) -> "Plan._LaneCtx":
        return Plan._LaneCtx(self, lane)

    def begin_group(self):
        """Ops recorded until end_group() are INDEPENDENT of each other and go into one launch."""
        self._group = [] if self.use_group else None

    def end_group(self, name=""):
        subs, self._group = self._group, None
        if not subs:
            return
        convs = [r for r in subs if r["kind"] == "conv"]
        fuses = [r for r in subs if r["kind"] == "fuse"]
        while convs or fuses:               # respect the per-launch limits of rsg_conv_fuse_group
            c, convs = convs[:12], convs[12:]
            f, fuses = fuses[:4], fuses[4:]
            part = c + f
            if len(part) == 1:
                self._record(part[0], part[0]["reads"], part[0]["writes"])
            else:
                reads = [v for r in part for v in r["reads"]]
                writes = [v for r in part for v in r["writes"]]
                self._record(dict(kind="group", subs=part, name=name), reads, writes)

    def _groupable(self, rec: dict) -> bool:
        if rec["kind"] == "fuse":
            return True
        if rec["kind"] != "conv":
            return False
        d = self._conv_desc(rec, fake_ptrs=True)
        return bool(self.lib.rsg_conv_groupable(C.byref(d)))

    def _record(self, rec: dict, reads, writes) -> int:
        rec["reads"] = [v for v in reads if v is not None]
        rec["writes"] = [v for v in writes if v is not None]
        if getattr(self, "_group", None) is not None and "lane" not in rec and self._groupable(rec):
            rec["lane"] = 0
            self._group.append(rec)
            return -1
        idx = len(self._rec)
        rec.setdefault("lane", self._lane)
        self._touch(idx, *[v for v in rec["reads"] + rec["writes"] if isinstance(v, VT)])
        self._rec.append(rec)
        return idx

    # ------------------------------------------------------------------ recording
    def new(self, H, W, C_, ld=None, pinned=False, name="", planes=False) -> VT:
        vt = VT(H, W, C_, ld, pinned, name, planes)
        self._vts.append(vt)
        return vt

    def _touch(self, idx: int, *vts: Optional[VT]):
        for v in vts:
            if v is None:
                continue
            b = v.base
            if b.first < 0:
                b.first = idx
            b.last = max(b.last, idx)

    def _dev(self, arr: np.ndarray) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
        self._keep.append(t)
        return t

    def pack_conv(self, w: torch.Tensor, scale, shift, bias) -> Tuple[torch.Tensor, torch.Tensor, int, int]:
        """w: fp32 [O,I,kh,kw] (CPU).  Returns (packed weights u8 on device, bias f32 on device)."""
        w = w.detach().to("cpu", torch.float32).contiguous()
        O, I, kh, kw = w.shape
        nbytes = self.lib.rsg_conv_packed_bytes(I, O, kh, kw)
        wp = np.empty(nbytes, dtype=np.uint8)
        bo = np.empty(_round_up(O, 32), dtype=np.float32)

        def p(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a.detach().to("cpu", torch.float32).numpy() if torch.is_tensor(a) else a,
                                     dtype=np.float32)
            self._tmp.append(a)
            return a.ctypes.data

        self._tmp = []
        wn = w.numpy()
        hip.check(self.lib.rsg_pack_conv_weights_host(wn.ctypes.data, p(scale), p(shift), p(bias), I, O, kh, kw,
                                                      self.dtype, wp.ctypes.data, bo.ctypes.data), "pack")
        self._tmp = []
        return self._dev(wp), self._dev(bo), I, O

    def pack_chain(self, conv: nn.Module, scale, shift) -> Tuple[torch.Tensor, torch.Tensor]:
        """One layer of a BasicBlock chain in the pack the chain kernels of its width read (hip.pack_chain_layer:
        csrc/chain16.h for HRNet-W32's widths, csrc/chain48.h for 48 channels, else the conv pack) -> (weights, bias) on the device."""
        if conv.bias is not None:
            raise ValueError("pack_chain: BasicBlock convs carry no bias (pose_hrnet.py:19-22)")

        def arr(a):
            return None if a is None else (a.detach().to("cpu", torch.float32).numpy() if torch.is_tensor(a) else a)

        wp, bo = hip.pack_chain_layer(conv.weight.detach().to("cpu", torch.float32).numpy(), arr(scale), arr(shift), self.dtype)
        return self._dev(wp), self._dev(bo)

    @staticmethod
    def bn_fold(bn: Optional[nn.Module]):
        """eval-mode BatchNorm as (scale, shift) in fp64->fp32."""
        if bn is None:
            return None, None
        g = bn.weight.detach().double().cpu()
        b = bn.bias.detach().double().cpu()
        m = bn.running_mean.detach().double().cpu()
        v = bn.running_var.detach().double().cpu()
        s = g / torch.sqrt(v + bn.eps)
        return s.float(), (b - m * s).float()

    def conv(self, x: VT, weight: torch.Tensor, bn: Optional[nn.Module] = None, bias=None, *, stride=1,
             relu=False, residual: Optional[VT] = None, out: Optional[VT] = None,
             out_f32: Optional[torch.Tensor] = None, want16=True, pad=None, up=(1, 0, 0),
             out_hw=None, phys_hw=None, name="", extra_shift=None, scale_shift=None,
             relu_mask=0, more_residuals: Sequence[VT] = ()) -> Optional[VT]:
        """Record conv(+BN)(+bias)(+residual)(+ReLU).  ``x.C`` must be the padded input channels."""
        scale, shift = self.bn_fold(bn) if scale_shift is None else scale_shift
        if extra_shift is not None:
            shift = extra_shift if shift is None else shift + extra_shift
        wp, bo, I, O = self.pack_conv(weight, scale, shift, bias)
        kh, kw = int(weight.shape[2]), int(weight.shape[3])
        cin_pad = self.lib.rsg_conv_padded_cin(I)
        if x.C > cin_pad and x.C == _round_up(I, 32):
            # e.g. a 48-channel map lives in a 64-channel buffer (outputs are padded to 32-channel tiles,
            # the padding holds zeros): read the first padded_cin channels of it
            x = x.view(0, cin_pad)
        if x.C != cin_pad:
            raise ValueError(f"{name}: input view has {x.C} channels, conv needs padded cin {cin_pad}")
        if pad is None:
            pad = ((kh - 1) // 2, (kw - 1) // 2)
        if out_hw is None:
            out_hw = ((x.H + 2 * pad[0] - kh) // stride + 1, (x.W + 2 * pad[1] - kw) // stride + 1)
        if phys_hw is None:
            phys_hw = (out_hw[0] * up[0], out_hw[1] * up[0])
        cpad = _round_up(O, 32)
        if want16 and out is None:
            out = self.new(phys_hw[0], phys_hw[1], cpad, name=name)
        if out is not None and (out.H, out.W) != tuple(phys_hw):
            raise ValueError(f"{name}: output buffer is {out.H}x{out.W}, conv writes {phys_hw}")
        store = 0
        if out is not None and out.C < cpad:
            # the destination is a channel slice narrower than the 32-channel output tile (e.g. 48 of 64):
            # write only the slice, the neighbouring channels belong to someone else
            if out.C % 8 or out.C < O:
                raise ValueError(f"{name}: output slice of {out.C} channels cannot hold cout {O}")
            store = out.C
        out_slot = None
        if isinstance(out_f32, OutSlot):
            out_slot, out_f32 = out_f32, out_f32.static
        if len(more_residuals) > 2 or (more_residuals and residual is None):
            raise ValueError(f"{name}: at most three residual operands")
        self._record(dict(kind="conv", x=x, wp=wp, bias=bo, res=residual, out=out, out_f32=out_f32, out_slot=out_slot,
                          I=cin_pad, O=O, kh=kh, kw=kw, stride=stride, pad=pad, up=up, out_hw=out_hw,
                          phys_hw=phys_hw, relu=relu, relu_mask=relu_mask, store=store, name=name,
                          res_more=list(more_residuals)),
                     [x, residual, *more_residuals], [out, out_f32])
        self.flops += 2 * self.batch * out_hw[0] * out_hw[1] * O * I * kh * kw
        return out

    def conv_tail1x1(self, x: VT, conv: nn.Conv2d, bn, tail_weight: torch.Tensor, tail_bias, out_f32, name="") -> None:
        """Record a 3x3 stride-1 conv 32 -> 32 (+BN, +ReLU) and the 1x1 conv (+bias, fp32 NCHW output) that follows it as ONE
        launch (rsg_conv_tail1x1: the first conv's accumulators are the second's MFMA operand; the 32-channel map between them
        is never stored).  out_f32: the output tensor or its OutSlot."""
        scale, shift = self.bn_fold(bn)
        wp, bo, I, O = self.pack_conv(conv.weight, scale, shift, conv.bias)
        J = int(tail_weight.shape[0])
        if (I, O) != (32, 32) or x.C != 32 or conv.kernel_size != (3, 3) or conv.stride != (1, 1) or J > 32 or \
                tuple(tail_weight.shape[1:]) != (32, 1, 1):
            raise ValueError(f"{name}: conv_tail1x1 takes a 3x3 stride-1 conv 32 -> 32 followed by a 1x1 conv 32 -> <= 32")
        wp2, bo2 = self.pack_conv(tail_weight, None, None, tail_bias)[:2]
        out_slot = None
        if isinstance(out_f32, OutSlot):
            out_slot, out_f32 = out_f32, out_f32.static
        self._record(dict(kind="conv_tail", x=x, wp=wp, bias=bo, res=None, out=None, out_f32=out_f32, out_slot=out_slot, I=32, O=32,
                          kh=3, kw=3, stride=1, pad=(1, 1), up=(1, 0, 0), out_hw=(x.H, x.W), phys_hw=(x.H, x.W), relu=True,
                          wp2=wp2, bias2=bo2, J=J, name=name), [x], [out_f32])
        self.flops += 2 * self.batch * x.H * x.W * (32 * 32 * 9 + J * 32)

    def conv_pair(self, x: VT, conv_a, bn_a, conv_b, bn_b, residual: Optional[VT], name="", next_unit=None,
                  tail=None, cat_x: Optional[VT] = None, planes: bool = False):
        """Record Bottleneck conv2 (3x3, 64 outputs, BN, ReLU) + conv3 (1x1, BN, + residual, ReLU) as ONE launch
        (rsg_conv_pair): the 64-channel map between them stays in LDS.  next_unit = (conv1, bn1) of the FOLLOWING
        Bottleneck: its 1x1 conv runs in the same launch on the block output (rsg_conv_pair_next) and the call returns
        (block output, that conv1's 64-channel output)."""
        sa, ha = self.bn_fold(bn_a)
        wpa, boa, Ia, Oa = self.pack_conv(conv_a.weight, sa, ha, conv_a.bias)
        if tail is not None:        # (weight [O, 128, 1, 1] already scaled, shift): the K-concat tail of the first Bottleneck
            wpb, bob, Ib, Ob = self.pack_conv(tail[0], None, tail[1], None)
        else:
            sb, hb = self.bn_fold(bn_b)
            wpb, bob, Ib, Ob = self.pack_conv(conv_b.weight, sb, hb, conv_b.bias)
        out = self.new(x.H, x.W, _round_up(Ob, 32), name=name, planes=planes)     # planes: the block output as [C/32][pixel][32]
        rec = dict(kind="conv_pair", x=x, wp=wpa, bias=boa, wp2=wpb, bias2=bob, res=residual, out=out, cat_x=cat_x,
                   out_f32=None, I=self.lib.rsg_conv_padded_cin(Ia), O=Oa, O2=Ob, kh=3, kw=3, stride=1, pad=(1, 1),
                   up=(1, 0, 0), out_hw=(x.H, x.W), phys_hw=(x.H, x.W), relu=True, name=name)
        self.flops += 2 * self.batch * x.H * x.W * (Oa * Ia * 9 + Ob * Ib)
        if next_unit is None:
            self._record(rec, [x, residual], [out])
            return out
        conv_n, bn_n = next_unit
        perm = [self.lib.rsg_conv_pair_next_perm(l) for l in range(Ob)]
        sn, hn = self.bn_fold(bn_n)
        wpn, bon, In, On = self.pack_conv(conv_n.weight.detach().float().cpu()[:, perm].contiguous(), sn, hn, conv_n.bias)
        t_next = self.new(x.H, x.W, On, name=name + "+next.conv1")
        rec.update(wp3=wpn, bias3=bon, out3=t_next, O3=On)
        self._record(rec, [x, residual, cat_x], [out, t_next])
        self.flops += 2 * self.batch * x.H * x.W * On * In
        return out, t_next

    def bottleneck(self, x: VT, blk, name="") -> VT:
        """Record a whole Bottleneck without downsample branch (pose_hrnet.py:85-101) as ONE launch
        (rsg_bottleneck_fused): conv1 is evaluated per tile into LDS, conv2 reads it there, conv3 adds x."""
        s1, h1 = self.bn_fold(blk.bn1)
        perm = [self.lib.rsg_bottleneck_head_perm(l) for l in range(256)]
        w1 = blk.conv1.weight.detach().float().cpu()[:, perm].contiguous()
        wp0, bo0, _, _ = self.pack_conv(w1, s1, h1, blk.conv1.bias)
        sa, ha = self.bn_fold(blk.bn2)
        wpa, boa, Ia, Oa = self.pack_conv(blk.conv2.weight, sa, ha, blk.conv2.bias)
        sb, hb = self.bn_fold(blk.bn3)
        wpb, bob, Ib, Ob = self.pack_conv(blk.conv3.weight, sb, hb, blk.conv3.bias)
        out = self.new(x.H, x.W, 256, name=name)
        self._record(dict(kind="bottleneck", x=x, wp0=wp0, bias0=bo0, wp=wpa, bias=boa, wp2=wpb, bias2=bob, res=None,
                          out=out, out_f32=None, I=64, O=64, O2=256, kh=3, kw=3, stride=1, pad=(1, 1), up=(1, 0, 0),
                          out_hw=(x.H, x.W), phys_hw=(x.H, x.W), relu=True, name=name), [x], [out])
        self.flops += 2 * self.batch * x.H * x.W * (64 * 256 + 64 * 64 * 9 + 256 * 64)
        return out

    def fuse(self, same: Sequence[VT], low: Sequence[Tuple[VT, int]], name="") -> VT:
        ref = same[0]
        out = self.new(ref.H, ref.W, ref.C, name=name)
        self._record(dict(kind="fuse", same=list(same), low=list(low), out=out, name=name),
                     list(same) + [v for v, _ in low], [out])
        return out

    def chain(self, x: VT, blocks, name="") -> VT:
        """Record a fused BasicBlock chain (rsg_basic_chain): every conv of ``blocks`` in one launch."""
        layers = []
        for blk in blocks:
            for conv, bn in ((blk.conv1, blk.bn1), (blk.conv2, blk.bn2)):
                scale, shift = self.bn_fold(bn)
                layers.append(self.pack_chain(conv, scale, shift))
        out = self.new(x.H, x.W, x.C, name=name)
        # block-input scratch: a standalone chain owns its buffer (chains of different branches run on different
        # lanes / graph branches, and the scheduler orders ops by the buffers they name: a shared one would be a race)
        ws = self._chain_workspace(None, x, len(layers))
        self._record(dict(kind="chain", x=x, out=out, layers=layers, name=name, ws=ws), [x], [out, ws])
        self.flops += 2 * self.batch * x.H * x.W * x.C * x.C * 9 * len(layers)
        return out

    def chain_group(self, items, name="") -> List[VT]:
        """Record ONE launch that runs several chains (rsg_basic_chain_group).  items: [(x, blocks)]; x may be the
        INDEX of an earlier item: that chain then reads the earlier one's output and the two hand over inside the
        launch (done/wait flags, include/rsgnet_hip.h)."""
        subs, outs = [], []
        for item in items:
            x, blocks = item[0], item[1]
            after = None
            if isinstance(x, int):
                after, x = x, outs[x]
            layers = []
            for blk in blocks:
                for conv, bn in ((blk.conv1, blk.bn1), (blk.conv2, blk.bn2)):
                    scale, shift = self.bn_fold(bn)
                    layers.append(self.pack_chain(conv, scale, shift))
            # the layers of one chain back to back in memory: the grouped launch prefetches them as one range
            allw = torch.cat([w for w, _ in layers])
            self._keep.append(allw)
            n0 = layers[0][0].numel()
            layers = [(allw[k * n0:(k + 1) * n0], b) for k, (_, b) in enumerate(layers)]
            out = self.new(x.H, x.W, x.C, name=name)
            split = bool(item[2]) if len(item) > 2 else False
            exch = bool(item[3]) if len(item) > 3 else False     # halo exchange inside the launch (rsg_chain_item.exchange)
            cc = blocks[0].conv1.in_channels             # 48 of a 64-channel buffer for HRNet-W48's first branch
            # position k of every grouped launch shares one scratch buffer per size; it is named among the op's
            # writes, so every scheduler (lanes, explicit graph) keeps the sharers in order
            if exch:
                need = int(self.lib.rsg_basic_chain_exchange_bytes(cc, x.H, x.W, len(layers), self.batch, None))
                if not need:
                    raise ValueError(f"{name}: a {cc}-channel {x.H}x{x.W} chain of {len(layers)} layers cannot trade its halo inside the launch")
                pool = self.__dict__.setdefault("_chain_ws", {})
                if ("x", len(subs), need) not in pool:
                    pool[("x", len(subs), need)] = torch.empty(need, dtype=torch.uint8, device=self.device)
                ws = pool[("x", len(subs), need)]
            else:
                ws = self._chain_workspace(len(subs), x, len(layers), cc)
            subs.append(dict(x=x, out=out, layers=layers, after=after, ws=ws, c=cc, split=split, exch=exch))
            outs.append(out)
            self.flops += 2 * self.batch * x.H * x.W * cc * cc * 9 * len(layers)
        self._record(dict(kind="chain_group", subs=subs, name=name),
                     [d["x"] for d in subs if d["after"] is None],
                     outs + [d["ws"] for d in subs])
        return outs

    def chain_mode(self, C_, H, W, nblocks) -> int:
        """How many BasicBlocks one rsg_basic_chain launch should take for this map: all of them when
        whole images fit in LDS, one when the map needs row tiles (halo grows with the chain), 0 = no."""
        if not self.use_chain:
            return 0
        info = (C.c_int32 * 4)()
        if C_ in (48, 96, 192, 384):
            # HRNet-W48 widths: at 384x288 only the smallest map fits LDS whole, and these maps are MFMA-heavy enough
            # that a deeper halo costs more than a launch: ONE BasicBlock of every branch per grouped launch
            return 1 if self.lib.rsg_basic_chain_supported(C_, H, W, 2, self.batch, info) and \
                sw.plan_on("wide_chains") else 0
        if self.lib.rsg_basic_chain_supported(C_, H, W, 2 * nblocks, self.batch, info) and info[0] == H:
            # whole images stay in LDS.  RSGNET_HIP_PLAN chain_slices=2 cuts the chain in two launches so that
            # both grouped launches of a module hold a slice of EVERY branch (better balanced grids)
            slices = sw.plan_int("chain_slices")
            return max(1, nblocks // slices) if nblocks % slices == 0 else nblocks
        # row tiles: the MFMAs are far from busy on these wide, thin maps, so halo recompute is cheap;
        # RSGNET_HIP_PLAN rowtile_blocks blocks per launch (default 2: measured best; 4 makes the 2.6x halo
        # recompute and the second wave of workgroups cost more than the saved launch), then 1
        want = sw.plan_int("rowtile_blocks")
        for nb in (want, 2):
            if 0 < nb <= nblocks and nblocks % nb == 0 and \
                    self.lib.rsg_basic_chain_supported(C_, H, W, 2 * nb, self.batch, info) and info[0] >= 8:
                return nb
        if self.lib.rsg_basic_chain_supported(C_, H, W, 2, self.batch, info):
            return 1
        return 0

    def custom(self, name: str, reads: Sequence[VT], writes: Sequence[VT], make: Callable[[], Callable[[int], None]]):
        """Record an op whose launcher is created by ``make()`` after buffers have addresses."""
        self._record(dict(kind="custom", make=make, name=name), list(reads), list(writes))

    # ------------------------------------------------------------------ allocation + lowering
    def finalize(self):
        B = self.batch
        free: Dict[int, List[torch.Tensor]] = {}
        expiring: Dict[int, List[VT]] = {}
        bases = [v for v in self._vts if v.base is v and v.first >= 0]
        by_first: Dict[int, List[VT]] = {}
        for v in bases:
            by_first.setdefault(v.first, []).append(v)
        total = 0
        for idx in range(len(self._rec)):
            for v in by_first.get(idx, []):
                n = v.nbytes(B) // 2
                pool = free.get(n)
                if pool and not v.pinned:
                    v.storage = pool.pop()
                else:
                    v.storage = torch.zeros(n, dtype=self.tdtype, device=self.device)
                    total += n * 2
                if not v.pinned:
                    expiring.setdefault(v.last, []).append(v)
            # outputs of op idx may not alias its inputs: release only AFTER the op
            for v in expiring.get(idx, []):
                free.setdefault(v.nbytes(B) // 2, []).append(v.storage)
        self.arena_bytes = total
        self._derive_dependencies()
        self._op_writes: List[List[VT]] = []          # per lowered op: the 16-bit maps it writes (run_with_taps)
        for r in self._rec:
            n0 = len(self.ops)
            getattr(self, "_lower_" + r["kind"])(r)
            self._op_writes += [[v for v in r["writes"] if isinstance(v, VT)]] * (len(self.ops) - n0)
        self._rec = []

    def _derive_dependencies(self):
        """Per op: which earlier ops on OTHER lanes it must wait for.  Tracked per physical buffer, so
        read-after-write, write-after-read and the allocator's buffer reuse are all covered."""
        def key(v):
            return v.base.storage.data_ptr() if isinstance(v, VT) else v.data_ptr()

        last_writer: Dict[int, int] = {}
        readers: Dict[int, List[int]] = {}
        self._lanes: List[int] = []
        self._waits: List[List[int]] = []
        self._all_deps: List[List[int]] = []
        for i, r in enumerate(self._rec):
            deps = set()
            for v in r["reads"]:
                k = key(v)
                if k in last_writer:
                    deps.add(last_writer[k])
            for v in r["writes"]:
                k = key(v)
                if k in last_writer:
                    deps.add(last_writer[k])
                deps.update(readers.get(k, ()))
            for v in r["writes"]:
                k = key(v)
                last_writer[k] = i
                readers[k] = []
            for v in r["reads"]:
                readers.setdefault(key(v), []).append(i)
            deps.discard(i)
            lane = r["lane"]
            self._lanes.append(lane)
            self._all_deps.append(sorted(deps))
            # same-lane order is implicit; keep only the latest dependency per foreign lane
            latest: Dict[int, int] = {}
            for d in deps:
                ld = self._lanes[d]
                if ld != lane:
                    latest[ld] = max(latest.get(ld, -1), d)
            self._waits.append(sorted(latest.values()))

    def _conv_desc(self, r, fake_ptrs=False) -> "hip.ConvDesc":
        d = hip.ConvDesc()
        x: VT = r["x"]
        for v in (x, r.get("res"), r.get("out"), *(r.get("res_more") or [])):
            if v is not None and v.planes and r["kind"] != "conv_pair":
                raise RuntimeError(f"{r.get('name')}: {v.name} is stored as channel-tile planes; only conv_pair and conv_dual take it")
        fake = 0x10000
        d.inp = fake if fake_ptrs else x.ptr()
        d.w_packed = r["wp"].data_ptr()
        d.bias = r["bias"].data_ptr()
        d.residual = (fake if fake_ptrs else r["res"].ptr()) if r["res"] is not None else None
        d.out16 = (fake if fake_ptrs else r["out"].ptr()) if r["out"] is not None else None
        d.out_nchw_f32 = r["out_f32"].data_ptr() if r["out_f32"] is not None else None
        d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = self.batch, x.H, x.W, r["I"], x.ld
        d.out_h, d.out_w, d.out_c = r["out_hw"][0], r["out_hw"][1], r["O"]
        d.out_ld = r["out"].ld if r["out"] is not None else _round_up(r["O"], 32)
        d.res_ld = r["res"].ld if r["res"] is not None else 0
        d.kh, d.kw, d.stride = r["kh"], r["kw"], r["stride"]
        d.pad_y, d.pad_x = r["pad"]
        d.up, d.up_y, d.up_x = r["up"]
        d.phys_h, d.phys_w = r["phys_hw"]
        d.relu = 1 if r["relu"] else 0
        d.relu_tile_mask = r.get("relu_mask", 0)
        d.out_c_store = r.get("store", 0)
        d.dtype = self.dtype
        d.tile_hint = 0
        more = r.get("res_more") or []
        if len(more) > 0:
            d.residual2, d.res2_ld = (0x10000 if fake_ptrs else more[0].ptr()), more[0].ld
        if len(more) > 1:
            d.residual3, d.res3_ld = (0x10000 if fake_ptrs else more[1].ptr()), more[1].ld
        return d

    def _log_conv(self, r, d):
        info = (C.c_int32 * 8)()
        hip.check(self.lib.rsg_conv_plan(C.byref(d), info), r["name"])
        self.conv_log.append(dict(name=r["name"], cin=r["I"], cout=r["O"], k=r["kh"], s=r["stride"],
                                  hw=r["out_hw"], tile=list(info)))

    def _lower_conv(self, r):
        d = self._conv_desc(r)
        self._keep.append(d)
        self._log_conv(r, d)
        fn, ref = self.lib.rsg_conv_igemm, C.byref(d)
        nm = r["name"]

        slot = r.get("out_slot")

        def launch(stream, fn=fn, ref=ref, nm=nm, d=d, slot=slot):
            if slot is not None:
                d.out_nchw_f32 = slot.ptr
            rc = fn(ref, stream)
            if rc:
                hip.check(rc, nm)

        self.ops.append((r["name"], launch))

    def _lower_conv_tail(self, r):
        d = self._conv_desc(r)
        self._keep.append(d)
        nm, slot, J = r["name"], r.get("out_slot"), r["J"]
        self.conv_log.append(dict(name=nm, flops=float(2 * self.batch * r["out_hw"][0] * r["out_hw"][1] * (32 * 32 * 9 + J * 32))))
        ref, w2, b2, static = C.byref(d), r["wp2"].data_ptr(), r["bias2"].data_ptr(), r["out_f32"].data_ptr()
        fn = self.lib.rsg_conv_tail1x1

        def launch(stream):
            rc = fn(ref, w2, b2, J, slot.ptr if slot is not None else static, stream)
            if rc:
                hip.check(rc, nm)

        self.ops.append((nm, launch))

    def _lower_conv_pair(self, r):
        rr = dict(r)
        rr["out"] = None                      # the inner descriptor carries no output of its own
        rr["res"] = None
        d = self._conv_desc(rr, fake_ptrs=False)
        d.out16 = r["out"].ptr()              # placeholder (validated, then replaced by the tail's output)
        d.out_ld = 64
        self._keep.append(d)
        out, res = r["out"], r["res"]
        d.layout = (hip.RSG_LAYOUT_OUT_PLANES if out.planes else 0) | (hip.RSG_LAYOUT_RES_PLANES if res is not None and res.planes else 0)
        old, rld = (32 if out.planes else out.ld), (0 if res is None else 32 if res.planes else res.ld)
        args = (C.byref(d), r["wp2"].data_ptr(), r["bias2"].data_ptr(), r["O2"], res.ptr() if res is not None else None,
                rld, out.ptr(), old, 1)
        fn, nm = self.lib.rsg_conv_pair, r["name"]
        entry = dict(name=nm, cin=r["I"], cout=r["O2"], k=3, s=1, hw=r["out_hw"], pair=True)
        if r.get("out3") is not None:          # + the next Bottleneck's conv1 in the same launch
            t3 = r["out3"]
            fn = self.lib.rsg_conv_pair_next
            cx = r.get("cat_x")
            args = (C.byref(d), r["wp2"].data_ptr(), r["bias2"].data_ptr(), res.ptr() if res is not None else None,
                    rld, out.ptr(), old, r["wp3"].data_ptr(), r["bias3"].data_ptr(), t3.ptr(), t3.ld,
                    cx.ptr() if cx is not None else None, cx.ld if cx is not None else 0)
            entry = dict(name=nm, flops=float(2 * self.batch * r["out_hw"][0] * r["out_hw"][1] *
                                             (r["I"] * 64 * 9 + (128 if cx is not None else 64) * r["O2"] + r["O2"] * r["O3"])))
        self.conv_log.append(entry)

        def launch(stream):
            rc = fn(*args, stream)
            if rc:
                hip.check(rc, nm)

        self.ops.append((nm, launch))

    def _lower_bottleneck(self, r):
        x, out = r["x"], r["out"]
        d = hip.ConvDesc()
        d.inp, d.w_packed, d.bias, d.out16 = x.ptr(), r["wp"].data_ptr(), r["bias"].data_ptr(), out.ptr()
        d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = self.batch, x.H, x.W, 64, 64
        d.out_h, d.out_w, d.out_c, d.out_ld = x.H, x.W, 64, 64
        d.kh, d.kw, d.stride, d.pad_y, d.pad_x = 3, 3, 1, 1, 1
        d.up, d.up_y, d.up_x, d.phys_h, d.phys_w = 1, 0, 0, x.H, x.W
        d.relu, d.dtype = 1, self.dtype
        self._keep.append(d)
        args = (C.byref(d), x.ptr(), x.ld, r["wp0"].data_ptr(), r["bias0"].data_ptr(), r["wp2"].data_ptr(),
                r["bias2"].data_ptr(), out.ptr(), out.ld)
        fn, nm = self.lib.rsg_bottleneck_fused, r["name"]
        self.conv_log.append(dict(name=nm, cin=256, cout=256, k=3, s=1, hw=r["out_hw"], pair=True,
                                  flops=float(2 * self.batch * x.H * x.W * (64 * 256 + 64 * 64 * 9 + 256 * 64))))

        def launch(stream):
            rc = fn(*args, stream)
            if rc:
                hip.check(rc, nm)

        self.ops.append((nm, launch))

    def _fuse_item(self, r) -> "hip.FuseItem":
        same, low, out = r["same"], r["low"], r["out"]
        if out.ld != out.C or out.coff:
            raise ValueError("fuse output must be a dense tensor")
        it = hip.FuseItem()
        for k, v in enumerate(same):
            it.same[k] = v.ptr()
            it.same_ld[k] = v.ld
        for k, (v, sh) in enumerate(low):
            it.low[k] = v.ptr()
            it.low_shift[k] = sh
            it.low_ld[k] = v.ld
        it.n_same, it.n_low, it.out = len(same), len(low), out.ptr()
        it.batch, it.h, it.w, it.c = self.batch, out.H, out.W, out.C
        return it

    def _lower_group(self, r):
        convs = [x for x in r["subs"] if x["kind"] == "conv"]
        fuses = [x for x in r["subs"] if x["kind"] == "fuse"]
        cd = (hip.ConvDesc * max(len(convs), 1))()
        for i, x in enumerate(convs):
            d = self._conv_desc(x)
            self._log_conv(x, d)
            cd[i] = d
        fi = (hip.FuseItem * max(len(fuses), 1))()
        for i, x in enumerate(fuses):
            fi[i] = self._fuse_item(x)
        nm, dt = r["name"], self.dtype
        table = np.zeros(self.lib.rsg_group_table_bytes(), np.uint8)
        launch2 = (C.c_int32 * 2)()
        hip.check(self.lib.rsg_conv_fuse_group_plan(cd, len(convs), fi, len(fuses), dt, table.ctypes.data, launch2), nm)
        self._keep.append(table)               # the table is the kernel ARGUMENT: read from host memory at every launch
        fn, tptr, blocks, lds = self.lib.rsg_conv_fuse_group_run, table.ctypes.data, launch2[0], launch2[1]
        self.conv_log.append(dict(name=nm, group=[x["name"] for x in r["subs"]], blocks=blocks))

        def launch(stream):
            rc = fn(tptr, blocks, lds, dt, stream)
            if rc:
                hip.check(rc, nm)

        self.ops.append((nm, launch))

    def _lower_fuse(self, r):
        # a lone fuse-sum is a group of one (the grouped entry point takes strided operands)
        self._lower_group(dict(subs=[r], name=r["name"]))

    def _chain_workspace(self, slot: Optional[int], x: VT, nlayers: int, c: Optional[int] = None) -> Optional[torch.Tensor]:
        """Block-input scratch of a chain.  slot None: a buffer of the op's own.  slot k: the chain in position k of a
        grouped launch; position k of every grouped launch shares one buffer per distinct size (the op names it among
        its writes, so the sharers stay ordered)."""
        need = int(self.lib.rsg_basic_chain_workspace_bytes(c or x.C, x.H, x.W, nlayers, self.batch))
        if not need:
            return None
        if slot is None:
            t = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._keep.append(t)
            return t
        pool = self.__dict__.setdefault("_chain_ws", {})
        if (slot, need) not in pool:
            pool[(slot, need)] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return pool[(slot, need)]

    def _lower_chain(self, r):
        x, out, layers = r["x"], r["out"], r["layers"]
        if x.ld != x.C or x.coff or out.ld != out.C or out.coff:
            raise ValueError("chain operands must be dense tensors")
        n = len(layers)
        wps = (C.c_void_p * n)(*[w.data_ptr() for w, _ in layers])
        bps = (C.c_void_p * n)(*[b.data_ptr() for _, b in layers])
        self._keep += [wps, bps]
        ws = r["ws"].data_ptr() if r["ws"] is not None else None
        args = (x.ptr(), out.ptr(), wps, bps, n, self.batch, x.H, x.W, x.C, self.dtype, ws)
        fn, nm = self.lib.rsg_basic_chain, r["name"]
        info = (C.c_int32 * 4)()
        self.lib.rsg_basic_chain_supported(x.C, x.H, x.W, n, self.batch, info)
        self.conv_log.append(dict(name=nm, cin=x.C, cout=x.C, k=3, s=1, hw=(x.H, x.W), layers=n,
                                  chain=list(info)))

        def launch(stream):
            rc = fn(*args, stream)
            if rc:
                hip.check(rc, nm)

        self.ops.append((nm, launch))

    def _lower_chain_group(self, r):
        subs = r["subs"]
        items = (hip.ChainItem * len(subs))()
        log = []
        for it, d in zip(items, subs):
            x, out, layers = d["x"], d["out"], d["layers"]
            cc = d.get("c", x.C)
            if x.coff or out.coff or ((x.ld != cc or out.ld != cc) and cc != 48):
                raise ValueError("chain operands must be dense tensors (only the 48-channel chain reads padded buffers)")
            n = len(layers)
            wps = (C.c_void_p * n)(*[w.data_ptr() for w, _ in layers])
            bps = (C.c_void_p * n)(*[b.data_ptr() for _, b in layers])
            self._keep += [wps, bps]
            it.inp, it.out = x.ptr(), out.ptr()
            it.in_ld, it.out_ld = x.ld, out.ld
            it.w_packed = C.cast(wps, C.POINTER(C.c_void_p))
            it.bias = C.cast(bps, C.POINTER(C.c_void_p))
            it.nlayers, it.batch, it.h, it.w, it.c = n, self.batch, x.H, x.W, cc
            it.workspace = d["ws"].data_ptr() if d["ws"] is not None else None
            it.split_rows = 1 if d.get("split") else 0
            if d.get("exch"):
                info = (C.c_int32 * 4)()
                self.lib.rsg_basic_chain_exchange_bytes(cc, x.H, x.W, n, self.batch, info)
                flags = torch.zeros(3 * info[3], dtype=torch.int32, device=self.device)     # zeroed once, self-clearing
                self._keep.append(flags)
                self._ho_flags.append(flags)          # ... and again by _handoff_failed after a poll that gave up
                it.exchange, it.done_flags = 1, flags.data_ptr()
            log.append((cc, x.H, x.W, n))
        for k, d in enumerate(subs):
            if d.get("after") is not None:
                prod = items[d["after"]]
                info = (C.c_int32 * 4)()
                self.lib.rsg_basic_chain_supported(prod.c, prod.h, prod.w, prod.nlayers, prod.batch, info)
                flags = torch.zeros(3 * info[3], dtype=torch.int32, device=self.device)   # zeroed once, self-clearing
                self._keep.append(flags)
                self._ho_flags.append(flags)          # ... and again by _handoff_failed after a poll that gave up
                prod.done_flags = flags.data_ptr()
                items[k].wait_flags = flags.data_ptr()
        # When the grid has more workgroups than CUs, let the smallest whole-image maps take two images per workgroup until
        # it fits in one round.  Fewer, twice as long workgroups: a launch running ALONE gets slower (the longest workgroup
        # sets its makespan: chain family 826 -> 913 us, step +4 %), a launch that shares the GPU with another stream's
        # forward gets cheaper in CU-time (two streams: 1.340 -> 1.316, 1.342 -> 1.314 ms per batch of 64,
        # profiles/r03n_ab_chain_pack_two_streams.txt).  RSGNET_HIP_PLAN chain_pack = auto (default): on for the plans of side
        # streams (self.throughput, models/_base.py plan_for), off for the default stream's; 0 / 1 force it.
        pack = sw.plan("chain_pack")
        if pack == "1" or (pack == "auto" and self.throughput):
            cus = torch.cuda.get_device_properties(self.device).multi_processor_count
            info = (C.c_int32 * 4)()
            blocks = []
            for it in items:
                if it.exchange:
                    self.lib.rsg_basic_chain_exchange_bytes(it.c, it.h, it.w, it.nlayers, it.batch, info)
                else:
                    (self.lib.rsg_basic_chain_supported_split if it.split_rows else self.lib.rsg_basic_chain_supported)(
                        it.c, it.h, it.w, it.nlayers, it.batch, info)
                blocks.append((info[3], info[0] == it.h))
            total = sum(b for b, _ in blocks)
            for k in sorted(range(len(subs)), key=lambda k: items[k].h * items[k].w):
                if total <= cus:
                    break
                if blocks[k][1] and items[k].c >= 128:
                    items[k].images_per_workgroup = 2
                    total -= blocks[k][0] // 2
        self._keep.append(items)
        fn, nm, n_items, dt = self.lib.rsg_basic_chain_group, r["name"], len(subs), self.dtype
        self.conv_log.append(dict(name=nm, group=log))

        def launch(stream):
            rc = fn(items, n_items, dt, stream)
            if rc:
                hip.check(rc, nm)

        self.ops.append((nm, launch))

    def _lower_custom(self, r):
        self.ops.append((r["name"], r["make"]()))

    # ------------------------------------------------------------------ execution
    def launch_all(self):
        """Enqueue every op: lane 0 is the caller's current stream, lanes 1.. are side streams that fork
        from it and join it again (so the whole thing can be captured into one HIP graph)."""
        main = torch.cuda.current_stream(self.device)
        if self.n_lanes == 1 or max(self._lanes, default=0) == 0:
            s = main.cuda_stream
            for _, op in self.ops:
                op(s)
            return
        while len(self._streams) < self.n_lanes - 1:
            self._streams.append(torch.cuda.Stream(self.device))
        streams = [main] + self._streams
        need_event = set(d for w in self._waits for d in w)
        keep: List[torch.cuda.Event] = []       # events must outlive a stream capture that uses them
        self._events_keep = keep
        fork = torch.cuda.Event()
        keep.append(fork)
        fork.record(main)
        for st in self._streams:
            st.wait_event(fork)
        events: Dict[int, torch.cuda.Event] = {}
        last_on_lane: Dict[int, int] = {}
        for i, (_, op) in enumerate(self.ops):
            st = streams[self._lanes[i]]
            for d in self._waits[i]:
                st.wait_event(events[d])
            op(st.cuda_stream)
            if i in need_event:
                ev = torch.cuda.Event()
                keep.append(ev)
                ev.record(st)
                events[i] = ev
            last_on_lane[self._lanes[i]] = i
        for lane, st in enumerate(streams):
            if lane:                         # join every side stream (also the ones that only forked)
                ev = torch.cuda.Event()
                keep.append(ev)
                ev.record(st)
                main.wait_event(ev)

    def run_with_taps(self, x: torch.Tensor, names: Sequence[str]) -> Dict[str, torch.Tensor]:
        """Tests / triage: one eager pass over the launch list; after the op that writes a 16-bit map whose recorded
        name is in `names`, that map is cloned as [B,H,W,C] (buffers are reused by later ops, so it has to be taken
        on the spot).  Returns {name: tensor}; the plan's fp32 outputs hold this pass's results afterwards."""
        if self.graph_mode == "dag":
            raise RuntimeError("run_with_taps needs RSGNET_HIP_PLAN graph != dag")
        want, got = set(names), {}
        self._bind_input(x)
        for v in vars(self).values():          # the output ops write where the LAST forward's tensors were: those may have
            if isinstance(v, OutSlot):         # been freed and handed to someone else (e.g. to a tap clone below)
                v.use_static()
        st = torch.cuda.current_stream(self.device).cuda_stream
        for (_, op), writes in zip(self.ops, self._op_writes):
            op(st)
            for v in writes:
                if v.name in want:
                    got[v.name] = v.tensor(self.batch).clone()
        missing = want - set(got)
        if missing:
            raise KeyError(f"no op of this plan writes {sorted(missing)}; maps: {sorted({v.name for w in self._op_writes for v in w})[:60]}")
        return got

    def _build_hip_graph(self):
        """One kernel node per op, edges = the data dependencies (librsgnet_hip's explicit graph API)."""
        g = C.c_void_p()
        hip.check(self.lib.rsg_graph_begin(C.byref(g)), "graph_begin")
        try:
            last_node: List[int] = []                   # op index -> id of the op's last kernel node
            for i, (name, op) in enumerate(self.ops):
                deps = [last_node[d] for d in self._all_deps[i]]
                arr = (C.c_int32 * max(len(deps), 1))(*deps)
                hip.check(self.lib.rsg_graph_next_deps(g, arr, len(deps)), "graph_next_deps")
                before = self.lib.rsg_graph_node_count(g)
                op(None)
                after = self.lib.rsg_graph_node_count(g)
                if after <= before:
                    raise RuntimeError(f"op {name} added no graph node")
                last_node.append(after - 1)
            hip.check(self.lib.rsg_graph_end(g), "graph_end")
        except Exception:
            self.lib.rsg_graph_abort(g)
            self.lib.rsg_graph_destroy(g)
            raise
        return g

    def __del__(self):
        g = getattr(self, "_hipgraph", None)
        if g is not None:
            try:
                self.lib.rsg_graph_destroy(g)
            except Exception:
                pass

    def _bind_input(self, x: torch.Tensor):
        """The first launch (layout conversion) reads the caller's fp32 NCHW tensor in place; it runs eagerly in
        front of the captured graph, whose kernel arguments are frozen.  Anything else (other dtype, strides, the
        explicit-DAG graph mode) goes through the static staging tensor as before."""
        direct = x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and self.graph_mode != "dag" \
            and x.device == self.x_static.device
        if direct:
            self._x_keep = x                   # alive until the next call
            self._x_src = x.data_ptr()
        else:
            self.x_static.copy_(x)
            self._x_src = self.x_static.data_ptr()

    def _replay(self):
        """RSGNET_HIP_PLAN graph = 'capture' (default): stream capture of the single-stream launch order (the
        grouped launches already overlap independent work inside each grid); 'dag': explicit HIP graph with
        data-dependency edges (librsgnet_hip's graph API); '0': eager launches, lanes = streams."""
        self._replay_launch()
        self._runs += 1
        if self._ho_flags:
            self._poll_handoff()

    # ---- the hand-off's error word on the production path --------------------------------------------------
    def _handoff_count(self) -> int:
        v = C.c_uint32(0)
        hip.check(self.lib.rsg_chain_handoff_timeouts(C.byref(v)), "chain_handoff_timeouts")   # synchronises
        return int(v.value)

    def _handoff_failed(self, count: int):
        for f in self._ho_flags:       # a flag nobody saw may still be (or become) 1: start the next launch clean
            f.zero_()
        torch.cuda.synchronize(self.device)
        was, self._ho_seen = self._ho_seen, count
        raise HandoffTimeout(
            f"{count - was} hand-off poll(s) of a grouped chain launch gave up (csrc/chain.hip): forwards since the last "
            f"clean check (at most {self.handoff_poll_every} of this plan) may hold wrong pixels. Flags re-zeroed; the "
            "model now rebuilds its plan without the in-launch hand-off.")

    def check_handoff(self):
        """Blocking read of the error word (plan warm-up, bench end, tests); raises HandoffTimeout on an increase."""
        if not self._ho_flags:
            return
        self._ho_pending = None
        count = self._handoff_count()
        if self._ho_seen is None:
            self._ho_seen = count
        elif count != self._ho_seen:
            self._handoff_failed(count)

    def _poll_handoff(self):
        """Every `handoff_poll_every` forwards: a 4-byte asynchronous copy of the error word into pinned memory behind
        the launches; its value is looked at on a later call, once the copy's event has completed (no stall)."""
        if self._ho_seen is None:
            return                                    # the warm-up run has not set the baseline yet
        if self._ho_pending is not None and self._ho_pending.query():
            self._ho_pending = None
            count = int(self._ho_pinned[0].item()) & 0xFFFFFFFF
            if count != self._ho_seen:
                self._handoff_failed(count)
        if self._ho_pending is None and self._runs % self.handoff_poll_every == 0:
            if self._ho_pinned is None:
                self._ho_pinned = torch.zeros(1, dtype=torch.int32).pin_memory()
            st = torch.cuda.current_stream(self.device)
            hip.check(self.lib.rsg_chain_handoff_timeouts_async(self._ho_pinned.data_ptr(), st.cuda_stream),
                      "chain_handoff_timeouts_async")
            ev = torch.cuda.Event()
            ev.record(st)
            self._ho_pending = ev

    def _replay_launch(self):
        mode = self.graph_mode
        if self._ho_flags and self._ho_seen is None:
            # warm-up of a plan with hand-offs: one eager pass, then a blocking look at the error word (the baseline
            # for the polls, and the first place a hand-off that cannot work on this device shows up)
            self._ho_seen = self._handoff_count()
            lanes, self._lanes = self._lanes, [0] * len(self._lanes)
            try:
                self.launch_all()
            finally:
                self._lanes = lanes
            self.check_handoff()
        if mode == "0":
            self.launch_all()
            return
        if mode == "capture":
            if self.graph is None:
                lanes, self._lanes = self._lanes, [0] * len(self._lanes)   # capture needs one stream
                self.launch_all()
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                allops = self.ops
                self.ops = allops[1:len(allops) - self.n_tail]           # input conversion and output ops stay outside
                try:
                    with torch.cuda.graph(g):
                        self.launch_all()
                finally:
                    self.ops = allops
                self._lanes = lanes
                self.graph = g
            st = torch.cuda.current_stream(self.device).cuda_stream
            self.ops[0][1](st)
            self.graph.replay()
            for _, op in self.ops[len(self.ops) - self.n_tail:]:
                op(st)
            return
        if getattr(self, "_hipgraph", None) is None:
            self.launch_all()                      # first touch outside the graph (module loading)
            torch.cuda.synchronize(self.device)
            self._hipgraph = self._build_hip_graph()
        hip.check(self.lib.rsg_graph_launch(self._hipgraph, torch.cuda.current_stream(self.device).cuda_stream),
                  "graph_launch")


# ====================================================================================================
# HRNet backbone recorder (shared by both models)
# ====================================================================================================

class _BackboneMixin:
    def _record_input(self: Plan, H: int, W: int):
        self.x_static = torch.zeros(self.batch, 3, H, W, dtype=torch.float32, device=self.device)
        x16 = self.new(H, W, 16, name="x_nhwc16")
        lib, B, dt = self.lib, self.batch, self.dtype

        def make():
            dst = x16.ptr()

            def launch(stream):       # reads the CALLER's tensor (self._x_src): no staging copy of the input
                hip.check(lib.rsg_nchw_f32_to_nhwc16(self._x_src, dst, B, 3, H, W, 16, dt, stream), "nchw_to_nhwc16")
            return launch

        self.custom("input_to_nhwc16", [], [x16], make)
        return x16

    def _record_stem_s2d(self: Plan, m, H: int, W: int) -> VT:
        """conv1 (3x3, stride 2, pad 1, 3 -> 64) as a 2x2 stride-1 conv over the space-to-depth input.

        out[y,x] = sum w[ky,kx,c] in[2y+ky-1, 2x+kx-1, c]; the half-resolution pixel (Y,X) holds the four
        phases in[2Y+py, 2X+px], so tap (a,b) in {0,1}^2 of the 2x2 conv reads pixel (y+a-1, x+b-1) and the
        phase (py,px) of it carries w[2(a-1)+py+1, 2(b-1)+px+1] when that index is in [0,2], else 0."""
        self.x_static = torch.zeros(self.batch, 3, H, W, dtype=torch.float32, device=self.device)
        fused = sw.plan_on("stem") and m.conv1.out_channels == 64 and \
            m.conv2.in_channels == 64 and m.conv2.out_channels == 64 and m.conv2.kernel_size == (3, 3) and \
            m.conv2.stride == (2, 2) and H % 4 == 0 and W % 4 == 0
        direct = fused and sw.plan_on("stem_direct")   # the stem reads the picture itself
        xs = None
        lib, B, dt = self.lib, self.batch, self.dtype
        if not direct:
            xs = self.new(H // 2, W // 2, 16, name="x_s2d16")

            def make():
                dst = xs.ptr()

                def launch(stream):       # reads the CALLER's tensor (self._x_src): no staging copy of the input
                    hip.check(lib.rsg_nchw_f32_to_s2d_nhwc16(self._x_src, dst, B, 3, H, W, dt, stream), "nchw_to_s2d")
                return launch

            self.custom("input_to_s2d_nhwc16", [], [xs], make)
        w = m.conv1.weight.detach().float().cpu()                  # [64,3,3,3]
        O = w.shape[0]
        w2 = torch.zeros(O, 16, 2, 2)
        for a in range(2):
            for py in range(2):
                ky = 2 * (a - 1) + py + 1
                if not 0 <= ky <= 2:
                    continue
                for b in range(2):
                    for px in range(2):
                        kx = 2 * (b - 1) + px + 1
                        if 0 <= kx <= 2:
                            w2[:, (py * 2 + px) * 3:(py * 2 + px) * 3 + 3, a, b] = w[:, :, ky, kx]
        if fused:      # conv1 and conv2 in one launch: the caller names the destination (self.stem_to)
            s1, h1 = self.bn_fold(m.bn1)
            self._stem = (xs, self.pack_conv(w2, s1, h1, m.conv1.bias)[:2], H, W)
            return None
        return self.conv(xs, w2, m.bn1, m.conv1.bias, relu=True, pad=(1, 1), out_hw=(H // 2, W // 2),
                         name="conv1(s2d 2x2)")

    def stem_to(self: Plan, m, out: Optional[VT], tail=None):
        """Record the fused stem (rsg_stem_fused) writing conv2's output into ``out`` (a 64-channel view) or a new map.
        tail = (conv, bn): the first Bottleneck's 1x1 conv1 (64 -> 64, BN, ReLU) on that output in the same launch
        (rsg_stem_fused_conv1); returns (stem output, conv1 output) then."""
        xs, (wp1, b1), H, W = self._stem
        s2, h2 = self.bn_fold(m.bn2)
        wp2, b2 = self.pack_conv(m.conv2.weight, s2, h2, m.conv2.bias)[:2]
        if out is None:
            out = self.new(H // 4, W // 4, 64, name="stem")
        lib, B, dt = self.lib, self.batch, self.dtype
        t3 = None
        if tail is not None:
            s3, h3 = self.bn_fold(tail[1])
            wp3, b3 = self.pack_conv(tail[0].weight, s3, h3, tail[0].bias)[:2]
            t3 = self.new(H // 4, W // 4, 64, name="layer1.0.conv1")
        name = "stem(conv1+conv2)" + ("+layer1.0.conv1" if tail is not None else "")

        def make():
            args = (wp1.data_ptr(), b1.data_ptr(), wp2.data_ptr(), b2.data_ptr(), out.ptr(), out.ld)
            if t3 is not None:
                args += (wp3.data_ptr(), b3.data_ptr(), t3.ptr(), t3.ld)
            args += (B, H, W, dt)
            fn = lib.rsg_stem_fused_conv1 if t3 is not None else lib.rsg_stem_fused
            src = xs.ptr() if xs is not None else None

            def launch(stream):       # without an s2d map the kernel reads the CALLER's picture (self._x_src)
                hip.check(fn(src, None if src else self._x_src, *args, stream), "stem_fused")
            return launch

        self.custom(name, [xs] if xs is not None else [], [out, t3], make)
        fl = 2 * B * ((H // 2) * (W // 2) * 64 * 27 + (H // 4) * (W // 4) * 64 * 64 * (9 + (1 if tail is not None else 0)))
        self.flops += fl
        self.conv_log.append(dict(name=name, flops=float(fl)))
        return out if tail is None else (out, t3)

    def _transition1_dual(self: Plan, ys: List[VT], trans) -> bool:
        """transition1 as ONE launch (rsg_conv_dual): both new branches start with a 3x3 conv of the same 256-channel map."""
        if not sw.plan_on("trans_dual") or len(ys) != 1 or len(trans) != 2 or trans[0] is None or \
                trans[1] is None or len(trans[1]) != 1:
            return False
        ca, cb, x = trans[0][0], trans[1][0][0], ys[0]
        return ca.kernel_size == (3, 3) and ca.stride == (1, 1) and cb.kernel_size == (3, 3) and cb.stride == (2, 2) and \
            ca.in_channels == cb.in_channels == x.C and x.C % 64 == 0 and \
            ((ca.out_channels <= 32 and cb.out_channels == 64) or (32 < ca.out_channels <= 64 and cb.out_channels == 96)) and \
            ca.bias is None and cb.bias is None and x.coff == 0 and x.H % 2 == 0 and x.W % 2 == 0

    def conv_dual(self: Plan, x: VT, seq_a, seq_b, name: str) -> List[VT]:
        """Record transition1 (pose_hrnet.py:437-443): conv-BN-ReLU at stride 1 and at stride 2 from the same map."""
        ca, cb = seq_a[0], seq_b[0]
        sa, ha = self.bn_fold(seq_a[1])
        sb, hb = self.bn_fold(seq_b[1])
        wa, ba, _i, Oa = self.pack_conv(ca.weight, sa, ha, None)
        wb, bb, _i, Ob = self.pack_conv(cb.weight, sb, hb, None)
        oa = self.new(x.H, x.W, _round_up(Oa, 32), name=name + ".a")
        ob = self.new(x.H // 2, x.W // 2, Ob, name=name + ".b")
        lib, B, dt = self.lib, self.batch, self.dtype

        def make():
            d = hip.ConvDesc()
            d.inp, d.w_packed, d.bias, d.out16 = x.ptr(), wa.data_ptr(), ba.data_ptr(), oa.ptr()
            d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = B, x.H, x.W, x.C, (32 if x.planes else x.ld)
            d.layout = hip.RSG_LAYOUT_IN_PLANES if x.planes else 0
            d.out_h, d.out_w, d.out_c, d.out_ld = x.H, x.W, Oa, oa.ld
            d.kh = d.kw = 3
            d.stride, d.pad_y, d.pad_x = 1, 1, 1
            d.up, d.phys_h, d.phys_w, d.relu, d.dtype = 1, x.H, x.W, 1, dt
            self._keep.append(d)
            ref, pb, pbb, pob, ldb = C.byref(d), wb.data_ptr(), bb.data_ptr(), ob.ptr(), ob.ld

            def launch(stream):
                rc = lib.rsg_conv_dual(ref, pb, pbb, pob, ldb, Ob, 1, stream)
                if rc:
                    hip.check(rc, name)
            return launch
        self.custom(name, [x], [oa, ob], make)
        fl = 2 * B * x.C * 9 * (x.H * x.W * Oa + (x.H // 2) * (x.W // 2) * Ob)
        self.flops += fl
        self.conv_log.append(dict(name=name, flops=float(fl)))
        return [oa, ob]

    def _unit(self: Plan, x: VT, conv: nn.Conv2d, bn, **kw) -> VT:
        return self.conv(x, conv.weight, bn, conv.bias, stride=conv.stride[0], **kw)

    @staticmethod
    def _plain_bottleneck(blk) -> bool:
        return hasattr(blk, "conv3") and blk.downsample is None and blk.conv1.kernel_size == (1, 1) and \
            blk.conv1.stride == (1, 1) and blk.conv1.in_channels == 256 and blk.conv1.out_channels == 64 and \
            blk.conv1.bias is None and blk.conv2.kernel_size == (3, 3) and blk.conv2.stride == (1, 1) and \
            blk.conv2.in_channels == 64 and blk.conv2.out_channels == 64 and blk.conv3.kernel_size == (1, 1) and \
            blk.conv3.out_channels == 256

    def _residual_block(self: Plan, x: VT, blk, name: str) -> VT:
        """BasicBlock / Bottleneck (pose_hrnet.py:30-101)."""
        res = x
        if blk.downsample is not None:
            res = self._unit(x, blk.downsample[0], blk.downsample[1], name=name + ".downsample")
        # whole-Bottleneck fusion (rsg_bottleneck_fused) is opt-in: measured equal to conv1 + conv_pair in step time
        # (both are bound by exposed latency at 2 workgroups per CU), so the proven pair stays
        if hasattr(blk, "conv3") and blk.downsample is None and sw.plan_on("bottleneck") and \
                x.C == 256 and x.coff == 0 and blk.conv1.kernel_size == (1, 1) and blk.conv1.in_channels == 256 and \
                blk.conv1.out_channels == 64 and blk.conv2.kernel_size == (3, 3) and blk.conv2.stride == (1, 1) and \
                blk.conv2.out_channels == 64 and blk.conv3.kernel_size == (1, 1) and blk.conv3.out_channels == 256:
            return self.bottleneck(x, blk, name=name + ".conv1+conv2+conv3")
        y = self._unit(x, blk.conv1, blk.bn1, relu=True, name=name + ".conv1")
        if hasattr(blk, "conv3"):
            c2, c3 = blk.conv2, blk.conv3
            if sw.plan_on("pair") and c2.kernel_size == (3, 3) and c2.stride == (1, 1) and \
                    c2.out_channels == 64 and c2.in_channels % 64 == 0 and c3.kernel_size == (1, 1) and \
                    c3.out_channels % 32 == 0 and y.C == c2.in_channels:
                return self.conv_pair(y, c2, blk.bn2, c3, blk.bn3, res, name=name + ".conv2+conv3")
            y = self._unit(y, blk.conv2, blk.bn2, relu=True, name=name + ".conv2")
            return self._unit(y, blk.conv3, blk.bn3, relu=True, residual=res, name=name + ".conv3")
        return self._unit(y, blk.conv2, blk.bn2, relu=True, residual=res, name=name + ".conv2")

    def _hr_module(self: Plan, xs: List[VT], mod, name: str, rider=None) -> List[VT]:
        """HighResolutionModule.forward (pose_hrnet.py:250-268).  rider(y_last): records ops that read only the module's
        LAST output; when that output is finished one fuse level before the module's final launch (its sum rides on a conv of
        the previous level), they are recorded INTO that final launch and the result is kept in self._rider_out."""
        self._rider_out = None
        xs = list(xs)
        # branches: LDS-resident chains, grouped so that ONE launch runs every branch of the module
        # (longest chain first); a branch that needs row tiles contributes its blocks slice by slice
        slices: List[List[Tuple[int, list]]] = []      # slices[k] = [(branch, blocks)] of the k-th launch
        steps = {}
        chan = {}                                      # channels the branch's convs see (HRNet-W48: 48 of a 64-channel buffer)
        for b, branch in enumerate(mod.branches):
            blocks = list(branch)
            plain = all(not hasattr(blk, "conv3") and blk.downsample is None for blk in blocks)
            cc = chan[b] = blocks[0].conv1.in_channels
            plain = plain and (cc == xs[b].C or (cc == 48 and xs[b].C == 64 and xs[b].coff == 0))
            steps[b] = self.chain_mode(cc, xs[b].H, xs[b].W, len(blocks)) if plain else 0
        # RSGNET_HIP_PLAN chain_handoff (default on): ONE launch per module.  Whole-image branches run all their blocks in
        # one chain; a row-tiled branch (halo grows with the chain) runs as two chains that hand over inside the
        # launch, so its second half needs no launch of its own (where half the CUs would idle).
        info = (C.c_int32 * 4)()

        def whole(b):
            n = 2 * len(mod.branches[b])
            return n <= 8 and self.lib.rsg_basic_chain_supported(chan[b], xs[b].H, xs[b].W, n, self.batch, info) \
                and info[0] == xs[b].H

        handoff = self.use_group and len(steps) > 1 and all(steps.values()) and self.allow_handoff and \
            all(whole(b) or 2 * steps[b] == len(mod.branches[b]) for b in steps)
        if handoff:
            # workgroups that would WAIT inside the launch (the second halves of the row-tiled branches): the launch
            # is deadlock-free whatever the dispatch order only while they cannot fill the machine (csrc/chain.hip,
            # rsg_basic_chain_handoff_max_waiters); otherwise the second halves get a launch of their own
            waiters = 0
            for b in steps:
                if not whole(b):
                    self.lib.rsg_basic_chain_supported(chan[b], xs[b].H, xs[b].W, 2 * steps[b], self.batch, info)
                    waiters += info[3]
            handoff = waiters <= self.lib.rsg_basic_chain_handoff_max_waiters()
        if handoff:
            order = sorted(steps, key=lambda b: (not whole(b), -xs[b].C))       # whole-image chains first, widest first
            items, where = [], {}
            # the whole-image 64-channel chain as two row tiles (rsg_chain_item.split_rows): it is the launch's critical path
            # in the two-branch module (8 layers on 64 CUs).  RSGNET_HIP_PLAN chain_split64: 0 off (default), 1 two-branch modules,
            # 2 all.  Measured (round 5, profiles/r05_ab_chain_split64.txt): bit-identical; 1 changes nothing (the tiles' 1.5x
            # work gives back what the second 64 CUs buy), 2 costs a lone forward 14 %
            split64 = sw.plan_int("chain_split64")
            for b in order:
                blocks = list(mod.branches[b])
                if whole(b):
                    where[b] = len(items)
                    sp = chan[b] == 64 and xs[b].C == 64 and (split64 == 2 or (split64 == 1 and len(mod.branches) == 2)) \
                        and self.batch * 2 <= 256 and not self.throughput and \
                        bool(self.lib.rsg_basic_chain_supported_split(64, xs[b].H, xs[b].W, 2 * len(blocks), self.batch, info))
                    items.append((xs[b], blocks, sp))
            # A row-tiled branch as ONE chain whose tiles trade their halo rows half way, inside the launch
            # (rsg_chain_item.exchange): one staging and one store per tile instead of two -- ~9 % less CU-time for the 64x48x32
            # branch -- but twice as long workgroups (a longer launch when it runs alone).  RSGNET_HIP_PLAN chain_exchange =
            # auto (default): on for the plans of side streams (throughput, like chain_pack), off for the default stream's.
            xmode = sw.plan("chain_exchange")
            def xch(b):
                n = 2 * len(mod.branches[b])
                return (xmode == "1" or (xmode == "auto" and self.throughput)) and n % 4 == 0 and \
                    2 * steps[b] == len(mod.branches[b]) and \
                    bool(self.lib.rsg_basic_chain_exchange_bytes(chan[b], xs[b].H, xs[b].W, n, self.batch, None))
            for b in order:                      # first halves, then second halves: a producer precedes its consumer
                if not whole(b):
                    where[b] = len(items)
                    if xch(b):
                        items.append((xs[b], list(mod.branches[b]), False, True))
                    else:
                        items.append((xs[b], list(mod.branches[b])[:steps[b]]))
            for b in order:
                if not whole(b) and not xch(b):
                    first, where[b] = where[b], len(items)
                    items.append((first, list(mod.branches[b])[steps[b]:]))
            outs = self.chain_group(items, f"{name}.chains")
            for b in order:
                xs[b] = outs[where[b]]
            steps = {}
        # branches the chains do not cover (HRNet-W48's widths): conv k of EVERY such branch in one grouped launch
        # (they are independent; alone, a 12x9x384 conv fills a fraction of the 256 CUs and a launch costs ~3 us)
        loose = [b for b, s in steps.items() if not s]
        if self.use_group and len(loose) > 1 and sw.plan_on("zip_branches") and \
                len({len(mod.branches[b]) for b in loose}) == 1 and \
                all(not hasattr(blk, "conv3") and blk.downsample is None for b in loose for blk in mod.branches[b]):
            mid: Dict[int, VT] = {}
            for k in range(len(mod.branches[loose[0]])):
                self.begin_group()
                for b in loose:
                    blk = mod.branches[b][k]
                    mid[b] = self._unit(xs[b], blk.conv1, blk.bn1, relu=True, name=f"{name}.branches.{b}.{k}.conv1")
                self.end_group(f"{name}.branches.{k}.conv1")
                self.begin_group()
                for b in loose:
                    blk = mod.branches[b][k]
                    xs[b] = self._unit(mid[b], blk.conv2, blk.bn2, relu=True, residual=xs[b],
                                       name=f"{name}.branches.{b}.{k}.conv2")
                self.end_group(f"{name}.branches.{k}.conv2")
            for b in loose:
                del steps[b]
        for b, step in steps.items():
            blocks = list(mod.branches[b])
            if not step:
                with self.on_lane(b):
                    for k, blk in enumerate(blocks):
                        xs[b] = self._residual_block(xs[b], blk, f"{name}.branches.{b}.{k}")
                continue
            for k in range(0, len(blocks), step):
                while len(slices) <= k // step:
                    slices.append([])
                slices[k // step].append((b, blocks[k:k + step]))
        for k, sl in enumerate(slices):
            sl.sort(key=lambda e: (-len(e[1]), -xs[e[0]].C))     # whole-branch chains first, widest first
            if self.use_group and (len(sl) > 1 or any(chan[b] != xs[b].C for b, _ in sl)):
                outs = self.chain_group([(xs[b], blks) for b, blks in sl], f"{name}.chains{k}")
                for (b, _), o in zip(sl, outs):
                    xs[b] = o
            else:
                for b, blks in sl:
                    with self.on_lane(b):
                        xs[b] = self.chain(xs[b], blks, f"{name}.branches.{b}.chain{k}")
        if mod.fuse_layers is None:
            return xs
        # the LAST module of pose_hrnet (one output): its 1x1 convs, the sum and the classifier in one launch (HRNetPlan)
        if getattr(self, "_defer_sum0", None) == name and sw.plan_on("fuse_head4") \
                and len(mod.fuse_layers) == 1 and 2 <= len(xs) <= 4 and all(xs[j].C == 32 << j for j in range(len(xs))) \
                and all(xs[j].coff == 0 for j in range(len(xs))) and xs[0].H % 8 == 0 and xs[0].W % 8 == 0 \
                and all(mod.fuse_layers[0][j][0].kernel_size == (1, 1) and mod.fuse_layers[0][j][0].bias is None for j in range(1, len(xs))):
            self._deferred_head4 = (list(xs), mod.fuse_layers[0])
            return [xs[0]]
        # fuse layers, level by level: level k holds the k-th conv of every path (all 1x1 "up" convs and
        # the first strided conv of every "down" chain are level 0); the sum of output i can run once
        # its longest chain (i convs, from branch 0) is done.  One grouped launch per level.
        rows = mod.fuse_layers
        n_out = len(rows)
        cur = {}                                         # (i, j) -> tensor produced so far on that path
        depth = {}                                       # (i, j) -> number of convs on that path
        for i, row in enumerate(rows):
            for j, fl in enumerate(row):
                if j != i:
                    depth[(i, j)] = 1 if j > i else len(fl)
                    cur[(i, j)] = xs[j]
        outs: List[Optional[VT]] = [None] * n_out
        max_depth = max(depth.values()) if depth else 0
        ready = {i: max([d for (ii, _), d in depth.items() if ii == i], default=0) for i in range(n_out)}
        # output n_out-1 has no up-sampled addends and at most three others besides its longest chain: fold its sum
        fold_into = None
        last = n_out - 1
        if sw.plan_on("fold_sum") and last >= 1 and last == len(xs) - 1 \
                and last <= 3 and depth.get((last, 0), 0) == last and depth[(last, 0)] == max_depth:
            fold_into = (last, 0)
            ready[last] = -1                              # no separate sum for it
        # level 0 as ONE launch of row bands (rsg_fuse_module: every conv that reads a branch map, and the sums of outputs 0
        # and 1); a level-0 conv it does not take (the weight-heavy 2 -> 3 conv) runs one level later, in the grouped
        # launch of the paths' second convs
        shift = {ij: 0 for ij in depth}                  # levels a path's convs run later than their index
        done0 = self._fuse_module(xs, rows, name, cur, depth, outs, fold_into)
        if done0 is not None:
            for ij in depth:
                if ij not in done0:
                    shift[ij] = 1
            for i in range(n_out):
                ready[i] = -1 if i < 2 or ready[i] < 0 else max(d + shift[ij] for ij, d in depth.items() if ij[0] == i)
            if fold_into == (1, 0):
                fold_into = None                          # the two-branch module: the band launch folded y1's sum itself
            max_depth = max(d + shift[ij] for ij, d in depth.items())
        for level in range(max_depth + 1):
            if fold_into is not None and level == max_depth and not any(r == level for r in ready.values()):
                break                                     # nothing left for a last, sum-only level
            self.begin_group()
            # convs of this level (the level-th conv of every path that is that long), those that read the same tensor next
            # to each other in the launch (their workgroups share the staged lines in L2)
            buckets: Dict[tuple, list] = {}
            for (i, j), dep in depth.items():
                k = level - shift[(i, j)]                # which conv of the path runs at this level
                if 0 <= k < dep and not (k == 0 and done0 is not None and (i, j) in done0):
                    fl = rows[i][j]
                    conv, bnm = (fl[0], fl[1]) if j > i else (fl[k][0], fl[k][1])
                    relu = False if j > i else (k != dep - 1)
                    xin = cur[(i, j)]
                    key = (id(xin.base), xin.coff, xin.C, conv.kernel_size, conv.stride)
                    buckets.setdefault(key, []).append(((i, j), xin, conv, bnm, relu, k))
            for items in buckets.values():
                for ij, xin, conv, bnm, relu, k in items:
                    nm = f"{name}.fuse.{ij[0]}.{ij[1]}" + ("" if ij[1] > ij[0] else f".{k}")
                    if ij == fold_into and k == depth[ij] - 1:
                        # the sum of the lowest-resolution output rides on the last conv that feeds it: every other
                        # addend (the identity and the shorter strided chains) was finished at an earlier level
                        i = ij[0]
                        others = [xs[i]] + [cur[(i, j)] for j in range(1, i)]
                        outs[i] = self._unit(xin, conv, bnm, relu=True, residual=others[0], more_residuals=others[1:],
                                             name=nm + "+sum")
                        cur[ij] = outs[i]
                        continue
                    cur[ij] = self._unit(xin, conv, bnm, relu=relu, name=nm)
            for i in range(n_out):
                if ready[i] == level:                    # every addend was finished by the previous level
                    same = [cur[(i, j)] for j in range(i)] + [xs[i]]
                    low = [(cur[(i, j)], j - i) for j in range(i + 1, len(xs))]
                    if i == 0 and getattr(self, "_defer_sum0", None) == name and same[0].C == 32 and len(same) == 1:
                        self._deferred_sum0 = (same, low)     # HRNetPlan: summed inside the classifier's launch
                        outs[i] = xs[i]                       # (placeholder: the caller does not read it)
                        continue
                    outs[i] = self.fuse(same, low, name=f"{name}.fuse.{i}")
            if rider is not None and self._rider_out is None and outs[last] is not None and fold_into is not None \
                    and level > depth[fold_into] - 1 and sw.plan_on("trans_ride"):
                self._rider_out = rider(outs[last])       # y_last was finished by an EARLIER launch: independent of this one
            self.end_group(f"{name}.fuse.level{level}")
        return outs

    def _fuse_module(self: Plan, xs: List[VT], rows, name: str, cur: dict, depth: dict, outs: list, fold_into):
        """Record level 0 of a module's fuse layers as one rsg_fuse_module launch (csrc/fuse_module.hip;
        pose_hrnet.py:190-268).  Fills outs[0], outs[1] and cur[(i, j)] of the level-0 convs it computes and returns their
        set; None (nothing recorded) when the module is not one it covers -- the caller then records level 0 as before."""
        nbr = len(xs)
        if not sw.plan_on("fuse_module") or not self.use_group or not 2 <= nbr <= 4 or len(rows) != nbr:
            return None
        if getattr(self, "_defer_sum0", None) == name:
            return None                                  # the last module of pose_hrnet: its y0 sum rides on the classifier
        c0 = rows[1][0][0][0].in_channels                # channels of branch 0: 32 (HRNet-W32) or 48 (HRNet-W48, in 64-channel buffers)
        if c0 not in (32, 48) or any(x.C != _round_up(c0 << j, 32) or x.coff or x.planes or x.H != xs[0].H >> j or x.W != xs[0].W >> j
                                     for j, x in enumerate(xs)):
            return None
        if fold_into not in (None, (1, 0), (nbr - 1, 0)):
            return None
        # the convs of level 0: (i, j) -> (conv, bn, k, role)
        take = {}
        for (i, j), dep in depth.items():
            fl = rows[i][j]
            conv, bnm = (fl[0], fl[1]) if j > i else (fl[0][0], fl[0][1])
            if conv.bias is not None or conv.in_channels != c0 << j:
                return None
            if j > i:                                    # 1x1, BatchNorm, up-sampled by the reader
                if conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.out_channels != c0 << i:
                    return None
                role = hip.FM_UP_Y0 if i == 0 else hip.FM_UP_Y1 if i == 1 else hip.FM_TO_GLOBAL
                take[(i, j)] = (conv, bnm, 1, role, False)
            else:
                if conv.kernel_size != (3, 3) or conv.stride != (2, 2) or conv.padding != (1, 1):
                    return None
                if j > 1:
                    continue                             # 2 -> 3 (590 KB of weights for 12 pixels of a band): grouped launch
                if conv.out_channels != (c0 << j if dep > 1 else c0 << i):
                    return None
                role = hip.FM_DOWN_Y1 if (i, j) == (1, 0) else hip.FM_TO_GLOBAL
                take[(i, j)] = (conv, bnm, 3, role, dep > 1)
        # a conv left out runs one level later: its path must end there, and a folded sum must still find it finished
        for ij, dep in depth.items():
            if ij not in take and (dep != 1 or (fold_into is not None and fold_into[0] == ij[0] and depth[fold_into] < 3)):
                return None
        if (1, 0) not in take or len(take) > hip.FM_MAX_CONVS:
            return None
        y1_fold = fold_into == (1, 0)
        if y1_fold and nbr != 2:
            return None
        H, W, B, dt, lib = xs[0].H, xs[0].W, self.batch, self.dtype, self.lib
        packed, glob, ys = {}, {}, []

        def desc(fake: bool) -> "hip.FuseModuleDesc":
            d = hip.FuseModuleDesc()
            for j, x in enumerate(xs):
                d.x[j] = 0x10000 if fake else x.ptr()
                d.x_ld[j] = x.ld
            for i in range(2):
                d.y[i] = 0x10000 if fake else ys[i].ptr()
                d.y_ld[i] = _round_up(c0 << i, 32)
            d.batch, d.h, d.w, d.n_branches, d.dtype, d.y1_fold, d.c0 = B, H, W, nbr, dt, 1 if y1_fold else 0, c0
            d.n_convs = len(take)
            for c, (ij, (conv, bnm, k, role, relu)) in enumerate(take.items()):
                e = d.convs[c]
                e.w_packed = 0x10000 if fake else packed[ij][0].data_ptr()
                e.bias = 0x10000 if fake else packed[ij][1].data_ptr()
                e.src, e.k, e.cout, e.relu, e.role = ij[1], k, conv.out_channels, 1 if relu else 0, role
                if role == hip.FM_TO_GLOBAL:
                    e.out16 = 0x10000 if fake else glob[ij].ptr()
                    e.out_ld = _round_up(conv.out_channels, 32)
            return d

        nbytes = lib.rsg_fuse_module_table_bytes()
        launch2 = (C.c_int32 * 2)()
        probe = np.zeros(nbytes, np.uint8)
        if lib.rsg_fuse_module_plan(C.byref(desc(True)), probe.ctypes.data, launch2):
            return None                                  # a shape the band kernel does not cover (LDS): grouped launches
        for ij, (conv, bnm, k, role, relu) in take.items():
            sc, sh = self.bn_fold(bnm)
            wp, bo, _i, _o = self.pack_conv(conv.weight, sc, sh, None)
            packed[ij] = (wp, bo)
        y0 = self.new(H, W, _round_up(c0, 32), name=f"{name}.fuse.0")
        y1 = self.new(H >> 1, W >> 1, 2 * c0, name=f"{name}.fuse.1.0.0+sum" if y1_fold else f"{name}.fuse.1")
        ys += [y0, y1]
        for ij, (conv, bnm, k, role, relu) in take.items():
            if role == hip.FM_TO_GLOBAL:
                lvl = ij[1] + (1 if k == 3 else 0)
                glob[ij] = self.new(H >> lvl, W >> lvl, _round_up(conv.out_channels, 32),
                                    name=f"{name}.fuse.{ij[0]}.{ij[1]}" + ("" if ij[1] > ij[0] else ".0"))

        def make():
            table = np.zeros(nbytes, np.uint8)
            l2 = (C.c_int32 * 2)()
            hip.check(lib.rsg_fuse_module_plan(C.byref(desc(False)), table.ctypes.data, l2), name)
            self._keep.append(table)                     # the table is the kernel ARGUMENT: read from host memory at every launch
            tptr, blocks, lds = table.ctypes.data, l2[0], l2[1]

            def launch(stream):
                rc = lib.rsg_fuse_module_run(tptr, blocks, lds, dt, stream)
                if rc:
                    hip.check(rc, name)
            return launch

        self.custom(f"{name}.fuse.band", list(xs), [y0, y1, *glob.values()], make)
        fl = 0
        for ij, (conv, bnm, k, role, relu) in take.items():
            lvl = ij[1] + (1 if k == 3 else 0)
            fl += 2 * B * (H >> lvl) * (W >> lvl) * conv.out_channels * conv.in_channels * k * k
        self.flops += fl
        self.conv_log.append(dict(name=f"{name}.fuse.band", flops=float(fl), band_convs=[f"{i}.{j}" for i, j in take]))
        outs[0], outs[1] = y0, y1
        for ij, v in glob.items():
            cur[ij] = v
        return set(take)

    def _record_backbone(self: Plan, m, H: int, W: int) -> List[VT]:
        if sw.plan_on("s2d") and H % 2 == 0 and W % 2 == 0:
            x = self._record_stem_s2d(m, H, W)
        else:
            x = self._record_input(H, W)
            x = self._unit(x, m.conv1, m.bn1, relu=True, name="conv1")
        b0 = m.layer1[0]
        fold = (sw.plan_on("l1cat") and b0.downsample is not None
                and hasattr(b0, "conv3") and b0.downsample[0].kernel_size == (1, 1)
                and b0.downsample[0].stride == (1, 1) and b0.conv3.in_channels % 16 == 0
                and m.conv2.out_channels % 16 == 0)
        use_next = sw.plan_on("pair_next") and sw.plan_on("pair") \
            and not sw.plan_on("bottleneck")
        t_pre = None                # conv1 output of the NEXT block, already computed by the previous block's launch
        l1 = list(m.layer1)
        # layer1's 256-channel maps as channel-tile planes (rsgnet_hip.h, RSG_LAYOUT_*): when every one of them is written by
        # a conv_pair tail and read only as the next tail's residual or by transition1's conv_dual
        planes = sw.plan_on("l1_planes") and fold and use_next and len(l1) > 1 \
            and all(self._plain_bottleneck(b) for b in l1[1:]) and sw.plan_on("pair_cat") \
            and self._transition1_dual([VT(H // 4, W // 4, 256)], m.transition1) and b0.conv3.out_channels == 256
        if fold and use_next and len(l1) > 1 and self._plain_bottleneck(l1[1]) and b0.conv2.kernel_size == (3, 3) \
                and b0.conv2.in_channels == 64 and b0.conv2.out_channels == 64 and b0.conv3.out_channels == 256 \
                and m.conv2.out_channels == 64 and sw.plan_on("pair_cat"):
            # layer1.0 in TWO launches: conv1, then conv2 -> [conv3 | projected residual] as one K = 128 contraction over
            # [conv2 output | x] -> conv1 of layer1.1, all from LDS (rsg_conv_pair_next with cat_x)
            stem_tail = x is None and sw.plan_on("stem_tail") and b0.conv1.kernel_size == (1, 1) \
                and b0.conv1.in_channels == 64 and b0.conv1.out_channels == 64 and b0.conv1.bias is None
            if stem_tail:      # layer1.0.conv1 rides on the stem's launch
                xs, t = self.stem_to(m, None, tail=(b0.conv1, b0.bn1))
            else:
                xs = self.stem_to(m, None) if x is None else self._unit(x, m.conv2, m.bn2, relu=True, name="conv2")
                t = self._unit(xs, b0.conv1, b0.bn1, relu=True, name="layer1.0.conv1")
            s3, h3 = self.bn_fold(b0.bn3)
            sd, hd = self.bn_fold(b0.downsample[1])
            w3 = b0.conv3.weight.detach().float().cpu() * s3.view(-1, 1, 1, 1)
            wd = b0.downsample[0].weight.detach().float().cpu() * sd.view(-1, 1, 1, 1)
            x, t_pre = self.conv_pair(t, b0.conv2, b0.bn2, None, None, None, name="layer1.0.conv2+conv3+downsample+next.conv1",
                                      next_unit=(l1[1].conv1, l1[1].bn1), tail=(torch.cat([w3, wd], dim=1), h3 + hd), cat_x=xs,
                                      planes=planes)
            rest, k0 = l1[1:], 1
        elif fold:
            # layer1.0: relu(bn3(conv3(t)) + bn_d(conv_d(x))) is ONE 1x1 conv over the channel concat [t | x]
            # with weights [s3*W3 | sd*Wd]: the projected residual (100 MB at batch 64) is never stored.
            # The stem's conv2 writes x straight into its half of the concat buffer.
            cm, cx = b0.conv3.in_channels, m.conv2.out_channels
            hq, wq = (H // 4, W // 4) if x is None else (x.H // 2, x.W // 2)
            cat = self.new(hq, wq, cm + cx, name="layer1.0.cat")
            xv = cat.view(cm, cx)
            if x is None:
                self.stem_to(m, xv)
            else:
                self._unit(x, m.conv2, m.bn2, relu=True, out=xv, name="conv2")
            t = self._unit(xv, b0.conv1, b0.bn1, relu=True, name="layer1.0.conv1")
            self._unit(t, b0.conv2, b0.bn2, relu=True, out=cat.view(0, cm), name="layer1.0.conv2")
            s3, h3 = self.bn_fold(b0.bn3)
            sd, hd = self.bn_fold(b0.downsample[1])
            w3 = b0.conv3.weight.detach().float().cpu() * s3.view(-1, 1, 1, 1)
            wd = b0.downsample[0].weight.detach().float().cpu() * sd.view(-1, 1, 1, 1)
            x = self.conv(cat, torch.cat([w3, wd], dim=1), None, None, relu=True, name="layer1.0.conv3+downsample",
                          extra_shift=h3 + hd)
            rest = list(m.layer1)[1:]
            k0 = 1
        else:
            x = self.stem_to(m, None) if x is None else self._unit(x, m.conv2, m.bn2, relu=True, name="conv2")
            rest, k0 = list(m.layer1), 0
        # runs of plain Bottlenecks: each block's launch also computes the next block's conv1 (RSGNET_HIP_PLAN pair_next=0: off)
        k = 0
        while k < len(rest):
            run = []
            while use_next and k + len(run) < len(rest) and self._plain_bottleneck(rest[k + len(run)]) and x.C == 256 and x.coff == 0:
                rest[k + len(run)]._rsg_name = f"layer1.{k0 + k + len(run)}"
                run.append(rest[k + len(run)])
            if len(run) >= 2 or (run and t_pre is not None):
                t = t_pre if t_pre is not None else self._unit(x, run[0].conv1, run[0].bn1, relu=True, name=f"{run[0]._rsg_name}.conv1")
                t_pre = None
                for j, blk in enumerate(run):
                    if j + 1 < len(run):
                        x, t = self.conv_pair(t, blk.conv2, blk.bn2, blk.conv3, blk.bn3, x, planes=x.planes,
                                              name=f"{blk._rsg_name}.conv2+conv3+next.conv1", next_unit=(run[j + 1].conv1, run[j + 1].bn1))
                    else:
                        x = self.conv_pair(t, blk.conv2, blk.bn2, blk.conv3, blk.bn3, x, planes=x.planes, name=f"{blk._rsg_name}.conv2+conv3")
                k += len(run)
            else:
                x = self._residual_block(x, rest[k], f"layer1.{k0 + k}")
                k += 1
        ys = [x]
        ridden = None               # a transition conv already recorded inside the previous stage's last fuse launch
        for s in (2, 3, 4):
            trans = getattr(m, f"transition{s - 1}")
            xs = []
            if s == 2 and self._transition1_dual(ys, trans):
                xs = self.conv_dual(ys[0], trans[0], trans[1][0], "transition1.0+1.0")
                trans = []
            for i, tr in enumerate(trans):
                if i < len(ys):
                    xs.append(ys[i] if tr is None else self._unit(ys[i], tr[0], tr[1], relu=True,
                                                                  name=f"transition{s - 1}.{i}"))
                elif ridden is not None and i == len(ys) and len(tr) == 1:
                    xs.append(ridden)                  # already recorded inside the previous module's last fuse launch
                else:
                    t = ys[-1]
                    with self.on_lane(i):
                        for j, step in enumerate(tr):
                            t = self._unit(t, step[0], step[1], relu=True, name=f"transition{s - 1}.{i}.{j}")
                    xs.append(t)
            mods = list(getattr(m, f"stage{s}"))
            ridden = None
            for k, mod in enumerate(mods):
                rider = None
                nxt = getattr(m, f"transition{s}", None) if k == len(mods) - 1 and s < 4 else None
                if nxt is not None and len(nxt) == len(xs) + 1 and nxt[-1] is not None and len(nxt[-1]) == 1 \
                        and all(t is None for t in list(nxt)[:-1]):
                    step = nxt[-1][0]                  # the next stage's new branch: ONE strided conv of this stage's last output
                    rider = lambda y, step=step, s=s, nb=len(xs): self._unit(y, step[0], step[1], relu=True, name=f"transition{s}.{nb}.0")
                xs = self._hr_module(xs, mod, f"stage{s}.{k}", rider=rider)
                ridden = self._rider_out
            ys = xs
        return ys


class HRNetPlan(Plan, _BackboneMixin):
    """pose_hrnet.PoseHighResolutionNet.forward (pose_hrnet.py:428-463)."""

    def __init__(self, model, batch, device, dtype):
        super().__init__(model, batch, device, dtype)
        W, H = model.cfg.MODEL.IMAGE_SIZE
        self.H, self.W = H, W
        J = model.cfg.MODEL.NUM_JOINTS
        fl = model.final_layer
        if fl.kernel_size != (1, 1):
            raise NotImplementedError("FINAL_CONV_KERNEL=3")
        # the last module's only fuse sum (multi_scale_output = False) is evaluated inside the classifier's launch
        # (rsg_fuse_classifier) when the highest-resolution branch has 32 channels: y0 is never stored
        self._deferred_sum0 = None
        self._deferred_head4 = None
        n4 = len(model.stage4)
        if sw.plan_on("fuse_classifier") and J <= 32 and fl.in_channels == 32 \
                and model.stage4[n4 - 1].fuse_layers is not None and len(model.stage4[n4 - 1].fuse_layers) == 1:
            self._defer_sum0 = f"stage4.{n4 - 1}"
        ys = self._record_backbone(model, H, W)
        self.heat = torch.zeros(batch, J, H // 4, W // 4, dtype=torch.float32, device=device)
        self.heat_slot = OutSlot(self.heat)
        if self._deferred_head4 is not None:
            self._fuse_head4(self._deferred_head4, fl, J)
        elif self._deferred_sum0 is not None:
            self._fuse_classifier(self._deferred_sum0, fl, J, H // 4, W // 4)
        else:
            self.conv(ys[0], fl.weight, None, fl.bias, out_f32=self.heat_slot, want16=False, name="final_layer")
        self.n_tail = 1
        self.finalize()

    def _fuse_head4(self, pending, fl, J: int):
        """The last module's three 1x1 convs (+BatchNorm), its fuse sum and the classifier: rsg_fuse_head4."""
        xs, row = pending
        packs = []
        for j in range(1, len(xs)):
            sc, sh = self.bn_fold(row[j][1])
            packs.append(self.pack_conv(row[j][0].weight, sc, sh, None)[:2])
        wpc, boc = self.pack_conv(fl.weight, None, None, fl.bias)[:2]
        lib, B, dt, slot = self.lib, self.batch, self.dtype, self.heat_slot
        x0, h, w, nl = xs[0], xs[0].H, xs[0].W, len(xs) - 1

        def make():
            lp = (C.c_void_p * 3)(*[v.ptr() for v in xs[1:]])
            lld = (C.c_int32 * 3)(*[v.ld for v in xs[1:]])
            wl = (C.c_void_p * 3)(*[wp.data_ptr() for wp, _ in packs])
            bl = (C.c_void_p * 3)(*[bo.data_ptr() for _, bo in packs])
            self._keep += [lp, lld, wl, bl]

            def launch(stream):
                rc = lib.rsg_fuse_head4(x0.ptr(), x0.ld, lp, lld, nl, wl, bl, wpc.data_ptr(), boc.data_ptr(), slot.ptr, B, h, w, J, dt, stream)
                if rc:
                    hip.check(rc, "final_layer")
            return launch

        self.custom("stage4.fuse+final_layer", list(xs), [], make)
        fl_ops = 2 * B * (h * w * J * 32 + sum((h >> j) * (w >> j) * 32 * (32 << j) for j in range(1, len(xs))))
        self.flops += fl_ops
        self.conv_log.append(dict(name="stage4.fuse+final_layer", flops=float(fl_ops)))

    def _fuse_classifier(self, pending, fl, J: int, h: int, w: int):
        same, low = pending
        wp, bo = self.pack_conv(fl.weight, None, None, fl.bias)[:2]
        lib, B, dt, slot = self.lib, self.batch, self.dtype, self.heat_slot

        def make():
            ns, nl = len(same), len(low)
            sp = (C.c_void_p * 4)(*[v.ptr() for v in same])
            sld = (C.c_int32 * 4)(*[v.ld for v in same])
            lp = (C.c_void_p * 3)(*[v.ptr() for v, _ in low])
            lsh = (C.c_int32 * 3)(*[sh for _, sh in low])
            lld = (C.c_int32 * 3)(*[v.ld for v, _ in low])
            self._keep += [sp, sld, lp, lsh, lld]

            def launch(stream):
                rc = lib.rsg_fuse_classifier(sp, sld, ns, lp, lsh, lld, nl, wp.data_ptr(), bo.data_ptr(), slot.ptr, B, h, w, J, dt, stream)
                if rc:
                    hip.check(rc, "final_layer")
            return launch

        self.custom("stage4.fuse.0+final_layer", list(same) + [v for v, _ in low], [], make)
        fl_ops = 2 * B * h * w * J * 32
        self.flops += fl_ops
        self.conv_log.append(dict(name="stage4.fuse.0+final_layer", flops=float(fl_ops)))

    def run(self, x: torch.Tensor, clone: bool = True) -> torch.Tensor:
        if x.shape != self.x_static.shape:
            raise ValueError(f"input {tuple(x.shape)} != plan {tuple(self.x_static.shape)} "
                             "(H, W must equal cfg.MODEL.IMAGE_SIZE)")
        self._bind_input(x)
        if self.graph_mode == "dag":           # frozen kernel arguments: static output, cloned
            self.heat_slot.use_static()
            self._replay()
            return self.heat.clone() if clone else self.heat
        out = self.heat_slot.fresh()           # the classifier writes straight into the tensor handed out
        self._replay()
        return out


class RSGNetPlan(Plan, _BackboneMixin):
    """RSGNet.forward (pose_rsgnet.py:955-1021) — filled in by rsgnet_amd/plan_rsgnet.py."""

    def __init__(self, model, batch, device, dtype):
        super().__init__(model, batch, device, dtype)
        from .plan_rsgnet import record_rsgnet
        record_rsgnet(self, model)
        self.finalize()

    def run(self, x: torch.Tensor, clone: bool = True):
        if x.shape != self.x_static.shape:
            raise ValueError(f"input {tuple(x.shape)} != plan {tuple(self.x_static.shape)}")
        self._bind_input(x)
        slots = (self.multi_slot, self.kpt_slot, self.limbs_slot)
        if self.graph_mode == "dag":
            outs = tuple(sl.use_static() for sl in slots)
            self._replay()
            return tuple(o.clone() for o in outs) if clone else outs
        outs = tuple(sl.fresh() for sl in slots)
        self._replay()
        return outs
