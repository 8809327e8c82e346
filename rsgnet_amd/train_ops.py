"""HIP-backed autograd operators for the training-mode head of pose_rsgnet.

The backbone trains on a recorded launch list (rsgnet_amd/train.py).  The head (pose_rsgnet.py:959-1013) mixes
convolutions over maps with sub-graphs that depend on the weights alone (type_fc / KTMachine / the location clue), so it
runs as ordinary autograd: each operator below is ONE node whose forward and backward are the library's kernels, the same
ones the backbone's training plan launches:

  ConvBnRelu   nn.Conv2d(bias=False) -> nn.BatchNorm2d (training mode: batch statistics, running statistics updated with
               momentum 0.1) -> [ReLU]      rsg_conv_igemm, rsg_bn_stats/apply; backward rsg_bn_backward, rsg_conv_wgrad
               (MFMA weight gradient) and the data gradient as rsg_conv_igemm over the transposed, flipped pack
  GroupNormTail  W of the TRP (association.py:298-300): conv1x1 + bias -> GroupNorm(8, C) on the fp32 relation output E;
               rsg_trp_tail forward, rsg_trp_tail_backward (dy, dE, the per-workgroup sums of dgamma / dbeta)
  Bilinear2x   F.interpolate(scale_factor=2, 'bilinear', align_corners=True) [+ sigmoid] on fp32 NCHW maps
               (pose_rsgnet.py:1009-1013): rsg_bilinear2x_nchw_f32 / rsg_bilinear2x_backward_nchw_f32
  DeconvBnRelu nn.ConvTranspose2d(4, 2, 1, bias=False) -> BatchNorm2d -> ReLU (the keypoint branch's up-sampler) as four
               sub-pixel 2x2 convs; backward through the same conv / MFMA weight-gradient kernels
  Conv1x1Out   the 1x1 classifier convs (bias or not) writing fp32 NCHW maps (multi_final_layer, final_layer, the limb
               classifier whose weight is KTMachine's output)

Maps travel between them as dense NHWC bf16 tensors [B,H,W,C], C a multiple of 32 -- the training plan's own format, so
the backbone's output map and its gradient cross the boundary without a conversion.  fp32 master weights are re-packed
to the MFMA fragment order on the device in every forward (rsg_pack_conv_weights_device: forward and data-gradient forms
in one launch).  No CPU path: every operator raises off the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import hip

BF16 = hip.RSG_BF16


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


def _bn_chunk(c: int) -> int:
    """rsg_bn_* take a power-of-two channel count in [8, 256]: wider or odd maps are normalised in channel chunks
    (BatchNorm is per channel, so this is exact)."""
    k = 256
    while k > 8 and c % k:
        k //= 2
    if c % k:
        raise NotImplementedError(f"BatchNorm over {c} channels")
    return k


class _ConvState:
    """Per-conv device buffers that outlive a step: the two weight packs, the packer's item table, workspaces."""

    def __init__(self, cout: int, cin: int, k: int, device: torch.device):
        lib = hip.lib()
        self.cout, self.cin, self.k, self.device = cout, cin, k, device
        self.cin_pad = int(lib.rsg_conv_padded_cin(cin))
        self.cout_pad = _round_up(cout, 32)
        u8 = dict(dtype=torch.uint8, device=device)
        self.pack_fwd = torch.zeros(lib.rsg_conv_packed_bytes(cin, cout, k, k), **u8)
        self.pack_bwd = torch.zeros(lib.rsg_conv_packed_bytes(cout, cin, k, k), **u8)     # transposed: packed cin = cout
        blocks_fwd = int(lib.rsg_pack_item_blocks(cin, cout, k, k))
        blocks_bwd = int(lib.rsg_pack_item_blocks(cout, cin, k, k))                         # as packed: cin and cout swapped
        self.blocks = blocks_fwd + blocks_bwd
        self.w_stage = torch.zeros(cout, cin, k, k, dtype=torch.float32, device=device)   # static address for the packer
        items = (hip.PackItem * 2)()
        for it, dst, tr, first in ((items[0], self.pack_fwd, 0, 0), (items[1], self.pack_bwd, 1, blocks_fwd)):
            it.w, it.dst, it.cin, it.cout, it.kh, it.kw = self.w_stage.data_ptr(), dst.data_ptr(), cin, cout, k, k
            it.transpose, it.first_block = tr, first
        self.table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(device)
        self.zero_bias = torch.zeros(max(self.cout_pad, _round_up(self.cin_pad, 32)), dtype=torch.float32, device=device)
        self._wgrad_ws: Dict[tuple, torch.Tensor] = {}
        self._bn_ws: Optional[torch.Tensor] = None

    def pack(self, w: torch.Tensor):
        """fp32 [O,I,k,k] (a parameter, or a tensor computed from parameters) -> both 16-bit packs."""
        lib = hip.lib()
        self.w_stage.copy_(w.detach().reshape(self.w_stage.shape))
        hip.check(lib.rsg_pack_conv_weights_device(self.table.data_ptr(), 2, self.blocks, BF16, _stream(self.w_stage)),
                  "pack_weights")

    def wgrad_ws(self, b: int, h: int, w: int, cout_pad: Optional[int] = None, xc: Optional[int] = None) -> torch.Tensor:
        cout_pad = cout_pad or self.cout_pad
        xc = xc or self.cin_pad
        key = (b, h, w, cout_pad, xc)
        ws = self._wgrad_ws.get(key)
        if ws is None:
            pad = (self.k - 1) // 2
            need = int(hip.lib().rsg_conv_wgrad_workspace_bytes(b, h, w, xc, h, w, cout_pad, self.k, self.k, 1, pad))
            if not need:
                raise NotImplementedError(f"weight gradient of a {self.k}x{self.k} conv {self.cin}->{self.cout}")
            ws = self._wgrad_ws[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def wgrad_ws3(self, b: int, h: int, w: int) -> torch.Tensor:
        """workspace of a 3x3 pad-1 weight gradient over this conv's channels (the deconv phases' route)."""
        key = (b, h, w, -3, 0)
        ws = self._wgrad_ws.get(key)
        if ws is None:
            need = int(hip.lib().rsg_conv_wgrad_workspace_bytes(b, h, w, self.cin_pad, h, w, self.cout_pad, 3, 3, 1, 1))
            if not need:
                raise NotImplementedError(f"3x3 weight gradient {self.cin}->{self.cout}")
            ws = self._wgrad_ws[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def bn_ws(self) -> torch.Tensor:
        if self._bn_ws is None:
            self._bn_ws = torch.empty(hip.lib().rsg_bn_workspace_bytes(256), dtype=torch.uint8, device=self.device)
        return self._bn_ws


def conv_state(owner: nn.Module, key: str, cout: int, cin: int, k: int, device: torch.device) -> _ConvState:
    cache = owner.__dict__.setdefault("_hip_conv_states", {})
    st = cache.get(key)
    if st is None or st.device != device or (st.cout, st.cin, st.k) != (cout, cin, k):
        st = cache[key] = _ConvState(cout, cin, k, device)
    return st


def _conv(st: _ConvState, x16: torch.Tensor, pack: torch.Tensor, in_c: int, out_c: int, bias: torch.Tensor,
          out16: Optional[torch.Tensor], out_f32: Optional[torch.Tensor], what: str, pads: Optional[Tuple[int, int]] = None,
          up: Tuple[int, int, int] = (1, 0, 0), residual: Optional[torch.Tensor] = None):
    """stride-1 conv of a dense NHWC bf16 map through rsg_conv_igemm: 'same' padding unless `pads` = (pad_y, pad_x) is
    given (the 2x2 sub-pixel convs of a ConvTranspose2d); up = (factor, y, x) places output pixel (y,x) at
    (factor*y + up_y, factor*x + up_x) of the physical output map; residual: a 16-bit map added in the epilogue."""
    b, h, w, c = x16.shape
    if c < in_c or c % 8 or x16.dtype != torch.bfloat16 or not x16.is_contiguous():
        raise ValueError(f"{what}: input {tuple(x16.shape)} {x16.dtype}, the conv wants dense bf16 with at least {in_c} channels")
    d = hip.ConvDesc()
    pad_y, pad_x = pads if pads is not None else ((st.k - 1) // 2, (st.k - 1) // 2)
    d.inp, d.w_packed, d.bias = x16.data_ptr(), pack.data_ptr(), bias.data_ptr()
    d.out16 = out16.data_ptr() if out16 is not None else None
    d.out_nchw_f32 = out_f32.data_ptr() if out_f32 is not None else None
    d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = b, h, w, in_c, c          # channels [in_c, c) of the buffer are not read
    d.out_h, d.out_w, d.out_c, d.out_ld = h, w, out_c, _round_up(out_c, 32)
    d.kh, d.kw, d.stride, d.pad_y, d.pad_x = st.k, st.k, 1, pad_y, pad_x
    d.up, d.up_y, d.up_x, d.phys_h, d.phys_w = up[0], up[1], up[2], up[0] * h, up[0] * w
    if residual is not None:
        d.residual, d.res_ld = residual.data_ptr(), residual.shape[-1]
    d.relu, d.dtype = 0, BF16
    hip.check(hip.lib().rsg_conv_igemm(C.byref(d), _stream(x16)), what)


def _need_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what} runs on the HIP kernels of librsgnet_hip.so only; got a {t.device.type} tensor")


class ConvBnRelu(torch.autograd.Function):
    """z = [ReLU](BatchNorm_train(conv(x))) on dense NHWC bf16 maps.  `w` is the fp32 [O,I,k,k] weight (gradient: fp32,
    same shape), gamma/beta the BatchNorm's affine terms; `bn` supplies the running statistics (updated in place)."""

    @staticmethod
    def forward(ctx, x16, w, gamma, beta, st: _ConvState, bn: nn.BatchNorm2d, relu: bool):
        _need_gpu(x16, "ConvBnRelu")
        b, h, wd, _ = x16.shape
        O = st.cout
        if O != st.cout_pad:
            raise NotImplementedError("ConvBnRelu: output channels must be a multiple of 32")
        st.pack(w)
        y = torch.empty(b, h, wd, O, dtype=torch.bfloat16, device=x16.device)
        _conv(st, x16, st.pack_fwd, st.cin_pad, O, st.zero_bias, y, None, "conv forward")
        z, mean, invstd, g32, ys, ck = _bn_forward(st, y, gamma, beta, bn, relu)
        ctx.st, ctx.relu, ctx.ck = st, relu, ck
        ctx.save_for_backward(x16, z, mean, invstd, g32, *ys)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = hip.lib()
        st, relu, ck = ctx.st, ctx.relu, ctx.ck
        x16, z, mean, invstd, g32, *ys = ctx.saved_tensors
        b, h, wd, _ = x16.shape
        O, s = st.cout, _stream(x16)
        dy, dgamma, dbeta = _bn_backward(st, dz.to(torch.bfloat16).contiguous(), z, ys, mean, invstd, g32, relu, ck)
        dw = torch.empty(O, st.cin, st.k, st.k, dtype=torch.float32, device=x16.device)
        pad = (st.k - 1) // 2
        xc = x16.shape[3]                                  # the map's channel stride (>= the conv's padded input channels)
        hip.check(lib.rsg_conv_wgrad(x16.data_ptr(), dy.data_ptr(), dw.data_ptr(), st.wgrad_ws(b, h, wd, xc=xc).data_ptr(), b, h, wd,
                                     st.cin, xc, h, wd, O, st.cout_pad, st.k, st.k, 1, pad, BF16, s), "conv_wgrad")
        dx = None
        if ctx.needs_input_grad[0]:
            cin_out = _round_up(st.cin, 32)
            if cin_out != x16.shape[3]:
                raise NotImplementedError(f"data gradient into a {x16.shape[3]}-channel map of a conv with {st.cin} inputs")
            dx = torch.empty_like(x16)
            _conv(st, dy, st.pack_bwd, st.cout_pad, st.cin, st.zero_bias, dx, None, "conv data gradient")
        return dx, dw, dgamma, dbeta, None, None, None


def conv_bn_relu(owner: nn.Module, key: str, x16: torch.Tensor, conv_w: torch.Tensor, bn: nn.BatchNorm2d, relu: bool = True):
    """`owner`/`key` name the persistent device state (packs, workspaces) of this conv."""
    O, I, k, _ = conv_w.shape
    st = conv_state(owner, key, O, I, k, x16.device)
    return ConvBnRelu.apply(x16, conv_w, bn.weight, bn.bias, st, bn, relu)


def _bn_forward(st: _ConvState, y: torch.Tensor, gamma, beta, bn: nn.BatchNorm2d, relu: bool):
    """Training-mode BatchNorm [+ReLU] of a dense NHWC bf16 map in channel chunks; returns (z, mean, invstd, chunks of y)."""
    lib = hip.lib()
    b, h, wd, O = y.shape
    px, s = b * h * wd, _stream(y)
    if bn.momentum is None or not bn.track_running_stats or not bn.affine:
        raise NotImplementedError("BatchNorm2d without momentum / running statistics / affine terms")
    ck = _bn_chunk(O)
    n = O // ck
    ys = [y] if n == 1 else [y[..., i * ck:(i + 1) * ck].contiguous() for i in range(n)]
    mean = torch.empty(O, dtype=torch.float32, device=y.device)
    invstd = torch.empty_like(mean)
    g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
    zs = []
    for i, yc in enumerate(ys):
        lo = i * ck
        rm, rv = bn.running_mean[lo:lo + ck], bn.running_var[lo:lo + ck]
        hip.check(lib.rsg_bn_stats(yc.data_ptr(), px, ck, BF16, st.bn_ws().data_ptr(), mean[lo:].data_ptr(), invstd[lo:].data_ptr(),
                                   rm.data_ptr(), rv.data_ptr(), float(bn.momentum), float(bn.eps), s), "bn_stats")
        z = torch.empty_like(yc)
        hip.check(lib.rsg_bn_apply(yc.data_ptr(), mean[lo:].data_ptr(), invstd[lo:].data_ptr(), g32[lo:].data_ptr(),
                                   b32[lo:].data_ptr(), None, int(relu), z.data_ptr(), px, ck, BF16, s), "bn_apply")
        zs.append(z)
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return (zs[0] if n == 1 else torch.cat(zs, dim=3)), mean, invstd, g32, ys, ck


def _bn_backward(st: _ConvState, dz: torch.Tensor, z: torch.Tensor, ys, mean, invstd, g32, relu: bool, ck: int):
    """-> (dy [B,H,W,O] bf16, dgamma, dbeta)."""
    lib = hip.lib()
    n, O = len(ys), mean.numel()
    b, h, wd, _ = dz.shape
    px, s = b * h * wd, _stream(dz)
    dgamma = torch.empty(O, dtype=torch.float32, device=dz.device)
    dbeta = torch.empty_like(dgamma)
    dys = []
    for i, yc in enumerate(ys):
        lo = i * ck
        dzc = dz if n == 1 else dz[..., lo:lo + ck].contiguous()
        zc = z if n == 1 else z[..., lo:lo + ck].contiguous()
        dy = torch.empty_like(yc)
        hip.check(lib.rsg_bn_backward(dzc.data_ptr(), zc.data_ptr() if relu else None, yc.data_ptr(), mean[lo:].data_ptr(),
                                      invstd[lo:].data_ptr(), g32[lo:].data_ptr(), dy.data_ptr(), None,
                                      dgamma[lo:].data_ptr(), dbeta[lo:].data_ptr(), st.bn_ws().data_ptr(), px, ck, BF16, s),
                  "bn_backward")
        dys.append(dy)
    return (dys[0] if n == 1 else torch.cat(dys, dim=3)), dgamma, dbeta


_DECONV_TAPS = {0: (3, 1), 1: (2, 0)}       # output phase p, tap a of the 2x2 conv -> index into the 4-wide deconv kernel
_DECONV_PHASE = (1, 0, 1, 0)                # deconv kernel index k -> the output phase that uses it ...
_DECONV_ROW3 = (2, 1, 1, 0)                 # ... and its row a + p in the 3x3 pad-1 window of that phase's weight gradient


def _deconv_index(device):
    """Index arrays: phase weights [4,O,I,2,2] out of W^T [O,I,4,4]; deconv weight gradient [4,4,...] out of the four 3x3
    weight gradients [4,O,I,3,3]."""
    ky = torch.tensor([[_DECONV_TAPS[py][a] for a in (0, 1)] for py in (0, 1) for _ in (0, 1)], device=device)     # [4,2]
    kx = torch.tensor([[_DECONV_TAPS[px][b] for b in (0, 1)] for _ in (0, 1) for px in (0, 1)], device=device)
    ph = torch.tensor([[2 * _DECONV_PHASE[y] + _DECONV_PHASE[x] for x in range(4)] for y in range(4)], device=device)
    ry = torch.tensor([[_DECONV_ROW3[y] for _ in range(4)] for y in range(4)], device=device)
    rx = torch.tensor([[_DECONV_ROW3[x] for x in range(4)] for _ in range(4)], device=device)
    return ky[:, :, None].expand(4, 2, 2), kx[:, None, :].expand(4, 2, 2), ph, ry, rx


class DeconvBnRelu(torch.autograd.Function):
    """z = ReLU(BatchNorm_train(ConvTranspose2d(k=4, s=2, p=1, bias=False)(x))) on dense NHWC bf16 maps [B,H,W,C] ->
    [B,2H,2W,O] (pose_rsgnet.py:741-750, the keypoint branch's up-sampler).  The transposed conv runs as four sub-pixel 2x2
    convs, out[2y+py, 2x+px] = sum_ab W[i,o,ky(py,a),kx(px,b)] in[y+a-(1-py), x+b-(1-px)], each writing its phase of the
    output (rsg_conv_igemm with up = 2).  Backward: per phase the data gradient is the 2x2 conv of that phase of dy with the
    transposed, flipped taps (accumulated through the conv's residual operand), the weight gradient the 3x3 MFMA weight
    gradient of (x, phase of dy), of which the phase's four taps are kept.  `wt`: the module's weight [I,O,4,4]."""

    @staticmethod
    def forward(ctx, x16, wt, gamma, beta, states, bn: nn.BatchNorm2d):
        _need_gpu(x16, "DeconvBnRelu")
        b, h, wd, cin = x16.shape
        I, O = int(wt.shape[0]), int(wt.shape[1])
        if tuple(wt.shape[2:]) != (4, 4) or I != cin or O % 32 or I % 32:
            raise NotImplementedError(f"DeconvBnRelu: weight {tuple(wt.shape)} on a {cin}-channel map")
        y = torch.empty(b, 2 * h, 2 * wd, O, dtype=torch.bfloat16, device=x16.device)
        idx = getattr(states[0], "deconv_index", None)
        if idx is None:
            idx = states[0].deconv_index = _deconv_index(x16.device)
        wph = wt.detach().float().permute(1, 0, 2, 3)[:, :, idx[0], idx[1]].permute(2, 0, 1, 3, 4)      # [4,O,I,2,2]
        for py in (0, 1):
            for px in (0, 1):
                st = states[2 * py + px]
                st.pack(wph[2 * py + px])
                _conv(st, x16, st.pack_fwd, I, O, st.zero_bias, y, None, "deconv phase", pads=(1 - py, 1 - px), up=(2, py, px))
        z, mean, invstd, g32, ys, ck = _bn_forward(states[0], y, gamma, beta, bn, True)
        ctx.states, ctx.ck, ctx.io = states, ck, (I, O)
        ctx.save_for_backward(x16, z, mean, invstd, g32, *ys)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = hip.lib()
        states, ck, (I, O) = ctx.states, ctx.ck, ctx.io
        x16, z, mean, invstd, g32, *ys = ctx.saved_tensors
        b, h, wd, _ = x16.shape
        s = _stream(x16)
        dy, dgamma, dbeta = _bn_backward(states[0], dz.to(torch.bfloat16).contiguous(), z, ys, mean, invstd, g32, True, ck)
        dw3 = torch.empty(4, O, I, 3, 3, dtype=torch.float32, device=x16.device)
        dx = torch.empty_like(x16) if ctx.needs_input_grad[0] else None
        first = True
        for py in (0, 1):
            for px in (0, 1):
                st = states[2 * py + px]
                dph = dy[:, py::2, px::2, :].contiguous()
                # weight gradient: the 2x2 taps sit at rows a - pad_y + 1 of a 3x3 pad-1 window
                hip.check(lib.rsg_conv_wgrad(x16.data_ptr(), dph.data_ptr(), dw3[2 * py + px].data_ptr(), st.wgrad_ws3(b, h, wd).data_ptr(),
                                             b, h, wd, I, I, h, wd, O, O, 3, 3, 1, 1, BF16, s), "deconv wgrad")
                if dx is not None:
                    _conv(st, dph, st.pack_bwd, O, I, st.zero_bias, dx, None, "deconv data gradient", pads=(py, px),
                          residual=None if first else dx)
                    first = False
        _, _, ph, ry, rx = states[0].deconv_index
        dwt = dw3[ph, :, :, ry, rx].permute(3, 2, 0, 1).contiguous()      # [4,4,O,I] -> [I,O,4,4]
        return dx, dwt, dgamma, dbeta, None, None


def deconv_bn_relu(owner: nn.Module, key: str, x16: torch.Tensor, deconv: nn.ConvTranspose2d, bn: nn.BatchNorm2d):
    if deconv.kernel_size != (4, 4) or deconv.stride != (2, 2) or deconv.padding != (1, 1) or deconv.output_padding != (0, 0) \
            or deconv.bias is not None:
        raise NotImplementedError("only ConvTranspose2d(k=4, s=2, p=1, bias=False) is built (every shipped config)")
    I, O = int(deconv.weight.shape[0]), int(deconv.weight.shape[1])
    states = [conv_state(owner, f"{key}.phase{q}", O, I, 2, x16.device) for q in range(4)]
    return DeconvBnRelu.apply(x16, deconv.weight, bn.weight, bn.bias, states, bn)


class Conv1x1Out(torch.autograd.Function):
    """out = conv1x1(x16, w) [+ bias] as an fp32 NCHW map [B,J,H,W] (the classifier heads, pose_rsgnet.py:961,1000,1005)."""

    @staticmethod
    def forward(ctx, x16, w, bias, st: _ConvState):
        _need_gpu(x16, "Conv1x1Out")
        b, h, wd, _ = x16.shape
        J = st.cout
        st.pack(w)
        bias32 = torch.zeros(st.cout_pad, dtype=torch.float32, device=x16.device)
        if bias is not None:
            bias32[:J].copy_(bias.detach())
        out = torch.empty(b, J, h, wd, dtype=torch.float32, device=x16.device)
        _conv(st, x16, st.pack_fwd, st.cin_pad, J, bias32, None, out, "classifier conv")
        ctx.st, ctx.has_bias = st, bias is not None
        ctx.save_for_backward(x16)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = hip.lib()
        st = ctx.st
        x16, = ctx.saved_tensors
        b, h, wd, _ = x16.shape
        J, s = st.cout, _stream(x16)
        dout = dout.float().contiguous()
        dpad = int(lib.rsg_conv_padded_cin(J))           # the gradient map is the data-gradient conv's INPUT
        d16 = torch.empty(b, h, wd, dpad, dtype=torch.bfloat16, device=x16.device)
        hip.check(lib.rsg_nchw_f32_to_nhwc16(dout.data_ptr(), d16.data_ptr(), b, J, h, wd, dpad, BF16, s), "dheat")
        dbias = None
        if ctx.has_bias:
            dbias = torch.empty(J, dtype=torch.float32, device=x16.device)
            csw = torch.empty(lib.rsg_channel_sum_workspace_bytes(J), dtype=torch.uint8, device=x16.device)
            hip.check(lib.rsg_channel_sum_nchw_f32(dout.data_ptr(), dbias.data_ptr(), csw.data_ptr(), b, J, h * wd, s), "dbias")
        dw = torch.empty(J, st.cin, 1, 1, dtype=torch.float32, device=x16.device)
        xc = x16.shape[3]
        hip.check(lib.rsg_conv_wgrad(x16.data_ptr(), d16.data_ptr(), dw.data_ptr(), st.wgrad_ws(b, h, wd, dpad, xc).data_ptr(), b, h, wd,
                                     st.cin, xc, h, wd, J, dpad, 1, 1, 1, 0, BF16, s), "conv_wgrad")
        dx = None
        if ctx.needs_input_grad[0]:
            if _round_up(st.cin, 32) != x16.shape[3]:
                raise NotImplementedError(f"data gradient into a {x16.shape[3]}-channel map of a conv with {st.cin} inputs")
            dx = torch.empty_like(x16)
            _conv(st, d16, st.pack_bwd, dpad, st.cin, st.zero_bias, dx, None, "classifier data gradient")
        return dx, dw, dbias, None


def conv1x1_out(owner: nn.Module, key: str, x16: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]):
    J, I = int(w.shape[0]), int(w.shape[1])
    st = conv_state(owner, key, J, I, 1, x16.device)
    return Conv1x1Out.apply(x16, w.reshape(J, I, 1, 1), bias, st)


def rows_gram(a: torch.Tensor, b: torch.Tensor, with_colsum: bool = False):
    """a^T b for two tall fp32 row matrices [N,32] (N ~ 1e5) on rsg_rows_gram_f32: per-block partial sums over contiguous
    row ranges, summed in block order by a second launch (fixed order, no atomics).  with_colsum: also sum_n a[n,:]."""
    lib = hip.lib()
    n, c = a.shape
    a, b = a.contiguous(), b.contiguous()
    gram = torch.empty(c, c, dtype=torch.float32, device=a.device)
    cs = torch.empty(c, dtype=torch.float32, device=a.device) if with_colsum else None
    ws = torch.empty(lib.rsg_rows_gram_workspace_bytes(c), dtype=torch.uint8, device=a.device)
    hip.check(lib.rsg_rows_gram_f32(a.data_ptr(), b.data_ptr(), gram.data_ptr(), cs.data_ptr() if with_colsum else None,
                                    n, c, ws.data_ptr(), _stream(a)), "rows_gram")
    return (gram, cs) if with_colsum else gram


def rows_project(a: torch.Tensor, m: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a [N,C] @ m [C,C] in fp32 on rsg_rows_project_f32, added to `out` when given."""
    lib = hip.lib()
    n, c = a.shape
    a, m = a.contiguous(), m.contiguous()
    acc = out is not None
    if out is None:
        out = torch.empty_like(a)
    hip.check(lib.rsg_rows_project_f32(a.data_ptr(), m.data_ptr(), out.data_ptr(), n, c, int(acc), _stream(a)), "rows_project")
    return out


class SgmGraphConv(torch.autograd.Function):
    """The SGM's graph convolution (KTMachine.forward, /root/reference/lib/models/pose_rsgnet.py:576-600) on rsg_sgm_graph_conv:
    Wl [L,D] = Linear2(LeakyReLU_0.02(Linear1((matrix_limb * prior) @ Wk))) with Wk [J,D] the joint classifier's weight; the
    backward (rsg_sgm_graph_conv_backward) returns the gradients of matrix_limb, Wk and the two Linears.  fp32, one launch each."""

    @staticmethod
    def forward(ctx, limb, prior, wk, w1, b1, w2, b2, slope: float):
        _need_gpu(wk, "SgmGraphConv")
        lib = hip.lib()
        L, J = limb.shape
        D = wk.shape[1]
        args = [t.detach().float().contiguous() for t in (limb, prior, wk, w1, b1, w2, b2)]
        wl = torch.empty(L, D, dtype=torch.float32, device=wk.device)
        t, h = torch.empty_like(wl), torch.empty_like(wl)
        hip.check(lib.rsg_sgm_graph_conv(*[a.data_ptr() for a in args], wl.data_ptr(), t.data_ptr(), h.data_ptr(), L, J, D,
                                         float(slope), _stream(wk)), "sgm_graph_conv")
        ctx.save_for_backward(*args[:4], args[5], t, h)
        ctx.slope = float(slope)
        return wl

    @staticmethod
    def backward(ctx, d_wl):
        lib = hip.lib()
        limb, prior, wk, w1, w2, t, h = ctx.saved_tensors
        L, J = limb.shape
        D = wk.shape[1]
        d_wl = d_wl.float().contiguous()
        d_limb, d_wk, d_w1, d_w2 = (torch.empty_like(x) for x in (limb, wk, w1, w2))
        d_b1 = torch.empty(D, dtype=torch.float32, device=wk.device)
        d_b2 = torch.empty_like(d_b1)
        hip.check(lib.rsg_sgm_graph_conv_backward(limb.data_ptr(), prior.data_ptr(), wk.data_ptr(), w1.data_ptr(), w2.data_ptr(),
                                                  t.data_ptr(), h.data_ptr(), d_wl.data_ptr(), d_limb.data_ptr(), d_wk.data_ptr(),
                                                  d_w1.data_ptr(), d_b1.data_ptr(), d_w2.data_ptr(), d_b2.data_ptr(), L, J, D, ctx.slope,
                                                  _stream(wk)), "sgm_graph_conv_backward")
        return d_limb, None, d_wk, d_w1, d_b1, d_w2, d_b2, None


def sgm_graph_conv(kt: nn.Module, weight: torch.Tensor) -> torch.Tensor:
    """kt_machine(final_layer.weight) -> [L,C0,1,1] on the HIP kernel (1x1 classifiers: the reshape of pose_rsgnet.py:591-593
    is the identity on [J,C0])."""
    n_out, n_in, kh, kw = weight.shape
    l0, act, l2 = kt.kpt_transformer[0], kt.kpt_transformer[1], kt.kpt_transformer[2]
    wl = SgmGraphConv.apply(kt.matrix_limb, kt.real_matrix_limb, weight.reshape(n_out, n_in * kh * kw), l0.weight, l0.bias,
                            l2.weight, l2.bias, float(act.negative_slope))
    return wl.reshape(wl.shape[0], n_in, kh, kw)


class GroupNormTail(torch.autograd.Function):
    """out16 [B,S,C] bf16 = GroupNorm(8, C)(e @ w^T + bias) from the TRP's fp32 rows e [B,S,C]."""

    @staticmethod
    def forward(ctx, e, w, bias, gamma, beta):
        _need_gpu(e, "GroupNormTail")
        lib = hip.lib()
        b, s, c = e.shape
        e32 = e.detach().float().contiguous()
        w32 = w.detach().reshape(c, c).float().contiguous()
        b32, g32, be32 = (t.detach().float().contiguous() for t in (bias, gamma, beta))
        ws = torch.empty(lib.rsg_trp_tail_workspace_bytes(b, s, c) // 4, dtype=torch.float32, device=e.device)
        out = torch.empty(b, s, c, dtype=torch.bfloat16, device=e.device)
        hip.check(lib.rsg_trp_tail(e32.data_ptr(), w32.data_ptr(), b32.data_ptr(), g32.data_ptr(), be32.data_ptr(), out.data_ptr(), c,
                                   b, s, c, BF16, ws.data_ptr(), _stream(e)), "trp_tail")
        ctx.save_for_backward(e32, w32, g32, ws)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = hip.lib()
        e32, w32, g32, ws = ctx.saved_tensors
        b, s, c = e32.shape
        dout = dout.to(torch.bfloat16).contiguous()
        dy = torch.empty_like(e32)
        de = torch.empty_like(e32)
        part = torch.empty(lib.rsg_trp_tail_backward_partial_floats(b, s, c), dtype=torch.float32, device=e32.device)
        hip.check(lib.rsg_trp_tail_backward(dout.data_ptr(), c, ws.data_ptr(), w32.data_ptr(), g32.data_ptr(), dy.data_ptr(),
                                            de.data_ptr(), part.data_ptr(), b, s, c, BF16, _stream(e32)), "trp_tail_backward")
        pc = part[:b * (s // 256) * 2 * c].reshape(-1, 2, c).sum(0)        # [2,C]: dgamma, dbeta
        dw, dbias = rows_gram(dy.reshape(-1, c), e32.reshape(-1, c), with_colsum=True)
        return de, dw.reshape(c, c, 1, 1), dbias, pc[0], pc[1]


class Bilinear2x(torch.autograd.Function):
    """fp32 NCHW [B,C,H,W] -> [B,C,2H,2W], bilinear with align_corners=True, optionally followed by a sigmoid."""

    @staticmethod
    def forward(ctx, x, sigmoid: bool):
        _need_gpu(x, "Bilinear2x")
        b, c, h, w = x.shape
        x32 = x.detach().float().contiguous()
        out = torch.empty(b, c, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
        hip.check(hip.lib().rsg_bilinear2x_nchw_f32(x32.data_ptr(), out.data_ptr(), b * c, h, w, int(sigmoid), _stream(x)), "bilinear2x")
        ctx.sigmoid = sigmoid
        ctx.save_for_backward(out if sigmoid else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        prob, = ctx.saved_tensors
        b, c, oh, ow = dout.shape
        dout = dout.float().contiguous()
        din = torch.empty(b, c, oh // 2, ow // 2, dtype=torch.float32, device=dout.device)
        hip.check(hip.lib().rsg_bilinear2x_backward_nchw_f32(dout.data_ptr(), prob.data_ptr() if prob is not None else None,
                                                             din.data_ptr(), b * c, oh // 2, ow // 2, _stream(dout)), "bilinear2x_backward")
        return din, None


def nchw_to_nhwc16(t: torch.Tensor, c_pad: Optional[int] = None) -> torch.Tensor:
    """fp32 NCHW -> dense NHWC bf16, channels zero-padded to c_pad (differentiable: torch ops)."""
    b, c, h, w = t.shape
    c_pad = c_pad or _round_up(c, 32)
    x = t.permute(0, 2, 3, 1)
    if c_pad != c:
        x = torch.nn.functional.pad(x, (0, c_pad - c))
    return x.to(torch.bfloat16).contiguous()


def nhwc16_to_nchw(t: torch.Tensor, c: Optional[int] = None) -> torch.Tensor:
    x = t if c is None else t[..., :c]
    return x.permute(0, 3, 1, 2).float().contiguous()
