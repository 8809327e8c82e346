"""Training-mode forward of `pose_rsgnet` (/root/reference/lib/models/pose_rsgnet.py:955-1021 under
`model.train()`, driven by `rsgnet_train`, /root/reference/lib/core/function.py:239-323).

  * the HRNet backbone (stem .. stage4: 99 % of the step's arithmetic) runs on the HIP training plan of
    rsgnet_amd/train.py -- the same launch lists as `pose_hrnet`, recorded up to the backbone's output map, which comes out
    as ONE autograd node (NHWC bf16, the plan's own format);
  * every conv -> BatchNorm -> ReLU unit of the head (vis_conv, type_conv, contact_conv, predict_contact_net, kpt_net,
    predict_net, limbs_net) and the three 1x1 classifiers (multi_final_layer, final_layer, the limb classifier) are autograd
    nodes on the library's kernels (rsgnet_amd/train_ops.py: rsg_conv_igemm, rsg_bn_*, rsg_conv_wgrad, the data gradient as a
    conv over the transposed pack);
  * the TRP's relation (association.py:288-297) and the relation MSE against the factored target (pose_rsgnet.py:1014-1018)
    run on csrc/trp.hip in both directions (`TrpRelation` below): neither pass stores the [B,S,S] map (1.2 GB at batch 32,
    S = 3072, which the reference's autograd keeps alive between forward and backward);
  * what depends on the weights alone stays an ordinary autograd expression over a few KB, evaluated on the device by
    PyTorch-ROCm: type_fc (Linear -> BatchNorm1d over the J rows -> ReLU) folded with type_conv's weight into a 3x3 conv
    J -> C0, W'[o,j] = sum_c type_conv.W[o,c] * Tm[j,c] (the [B,600,H,W] map of pose_rsgnet.py:974 is never built);
    the location clue loc_conv(loc_features); KTMachine -- the SGM's graph convolution, the limb classifier's weight from
    final_layer's -- is the library's rsg_sgm_graph_conv in both directions (train_ops.SgmGraphConv);
  * W behind the TRP (1x1 conv + bias -> GroupNorm, association.py:298-300: `GroupNormTail`) and the two x2 bilinear
    up-samplings with the limb maps' sigmoid (`Bilinear2x`) are autograd nodes on the library's kernels too;
  * the keypoint branch's up-sampler (ConvTranspose2d + BatchNorm + ReLU) runs as four sub-pixel 2x2 convs with their
    backward on the same kernels (`DeconvBnRelu`); only the sub-sampled TRP variant's up-sampler
    (RELATION_SUB_SAMPLE, one shipped CrowdPose config) is still a PyTorch-ROCm operator in training.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import switches as sw


def _loc_clue(m) -> torch.Tensor:
    """loc_conv(loc_features) (pose_rsgnet.py:693-697,979-980) with its 1x1 conv over the four constant planes written as a
    broadcast product and a fixed-order sum: the framework's weight-gradient kernel for this 4-channel convolution
    accumulates atomically, so its low bits differed between identical steps (one parameter of 1 830, 7e-12 of 3e-5: found by
    test_rsgnet_training_call_of_the_reference_reaches_the_hip_trp) -- the training step is bit-reproducible, this included."""
    conv, rest = m.loc_conv[0], m.loc_conv[1:]
    y = (conv.weight * m.loc_features).sum(1).unsqueeze(0)          # [C0,4,1,1] * [1,4,h,w] -> [1,C0,h,w]
    if conv.bias is not None:
        y = y + conv.bias.reshape(1, -1, 1, 1)
    return rest(y)


def kt_machine_forward(kt, weight: torch.Tensor) -> torch.Tensor:
    """KTMachine.forward (pose_rsgnet.py:576-600): the limb classifier from the joint classifier's weights.  On the GPU: the
    SGM graph-conv kernel in both directions (train_ops.SgmGraphConv); the expression below is its CPU statement (host-side
    tests of the head's algebra against the fp64 oracle)."""
    n_out, n_in, h, w = weight.shape
    if weight.is_cuda and n_out <= 32 and kt.matrix_limb.shape[0] <= 32 and n_in * h * w <= 64:
        from .train_ops import sgm_graph_conv
        return sgm_graph_conv(kt, weight)
    k = kt.matrix_limb * kt.real_matrix_limb                              # learned L x J masked by the 0/1 skeleton prior
    t = k @ weight.reshape(n_out, n_in * h * w)
    t = kt.kpt_transformer(t)                                            # Linear -> LeakyReLU(0.02) -> Linear
    return t.reshape(t.shape[0], n_in, h, w)


_KEY_ORDER = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)   # rsg_trp_project_g's order of the keys in a 16-block


_KEY_INDEX = {}


def _keys_transposed(rows: torch.Tensor) -> torch.Tensor:
    """[B,S,C] -> [B,C,S] with every 16-block of keys in the kernel's order."""
    b, s, c = rows.shape
    idx = _KEY_INDEX.get(rows.device)
    if idx is None:                       # built once per device (a host-to-device copy has no place in a captured graph)
        idx = _KEY_INDEX[rows.device] = torch.tensor(_KEY_ORDER, device=rows.device)
    return rows.transpose(1, 2).reshape(b, c, s // 16, 16).index_select(3, idx).reshape(b, c, s).contiguous()


class TrpRelation(torch.autograd.Function):
    """E = sigmoid(X X^T) (X Wg^T + bg) and mse_b = mean_ij (T_ij - sigmoid(X X^T)_ij)^2 on the HIP kernels, bf16
    operands and fp32 accumulation as the backbone's training plan.  x: rows [B,S,C] (bf16, or fp32 rounded here); the
    relation target either as its factor p [B,S] (T = p p^T, rsgnet_amd.loss.relation_target_vector) or as the [B,S,S] tensor
    the reference's loop builds (function.py:261-272), which the kernels read tile by tile.  Returns E [B,S,C] fp32 and
    mse [B].  forward: rsg_trp_project_g (values transposed for the attention and as rows for the backward) +
    rsg_trp_attention + rsg_trp_relation_mse[_dense]; backward: rsg_trp_backward[_dense] (sigmoid recomputed per tile) and
    the g projection's own gradients on rsg_rows_project_f32 / rsg_rows_gram_f32."""

    @staticmethod
    def forward(ctx, x, wg, bg, p):
        from . import hip
        if not x.is_cuda:
            raise RuntimeError("TrpRelation runs on the HIP kernels: the input must be on the GPU")
        lib = hip.lib()
        b, s, c = x.shape
        st = torch.cuda.current_stream(x.device).cuda_stream
        x16 = x.detach().to(torch.bfloat16).contiguous()
        w32 = wg.detach().reshape(c, c).float().contiguous()
        b32 = bg.detach().float().contiguous()
        g_hi = torch.empty(b, c, s, dtype=torch.bfloat16, device=x.device)
        g16 = torch.empty(b, s, c, dtype=torch.bfloat16, device=x.device)
        hip.check(lib.rsg_trp_project_g(x16.data_ptr(), c, w32.data_ptr(), b32.data_ptr(), g_hi.data_ptr(), g16.data_ptr(),
                                        b, s, c, hip.RSG_BF16, st), "trp_project_g")
        e = torch.empty(b, s, c, dtype=torch.float32, device=x.device)
        hip.check(lib.rsg_trp_attention(x16.data_ptr(), c, g_hi.data_ptr(), None, e.data_ptr(), b, s, c,
                                        hip.RSG_BF16, st), "trp_attention")
        mse = torch.zeros(b, dtype=torch.float32, device=x.device)
        p32 = None
        if p is not None:
            p32 = p.detach().to(x.device, torch.float32).contiguous()
            ws = torch.empty(int(lib.rsg_trp_mse_workspace_bytes(b, s)), dtype=torch.uint8, device=x.device)
            if p32.dim() == 3:
                if p32.shape != (b, s, s):
                    raise ValueError(f"relation_target of shape {tuple(p32.shape)} for {s} positions")
                hip.check(lib.rsg_trp_relation_mse_dense(x16.data_ptr(), c, p32.data_ptr(), mse.data_ptr(), ws.data_ptr(), b, s, c,
                                                         hip.RSG_BF16, st), "trp_relation_mse_dense")
            else:
                hip.check(lib.rsg_trp_relation_mse(x16.data_ptr(), c, p32.data_ptr(), mse.data_ptr(), ws.data_ptr(), b, s, c,
                                                   hip.RSG_BF16, st), "trp_relation_mse")
        ctx.save_for_backward(x16, g16, w32, p32)
        ctx.in_dtype = x.dtype
        return e, mse

    @staticmethod
    def backward(ctx, d_e, d_mse):
        from . import hip
        from .train_ops import rows_gram, rows_project
        lib = hip.lib()
        x16, g16, w32, p32 = ctx.saved_tensors
        b, s, c = x16.shape
        st = torch.cuda.current_stream(x16.device).cuda_stream
        de16 = d_e.to(torch.bfloat16).contiguous() if d_e is not None else torch.zeros_like(x16)
        x_t, de_t = _keys_transposed(x16), _keys_transposed(de16)
        coef = None
        if p32 is not None:
            coef = (torch.zeros(b, device=x16.device) if d_mse is None else d_mse.float() * (2.0 / (s * s))).contiguous()
        dx = torch.empty(b, s, c, dtype=torch.float32, device=x16.device)
        dg = torch.empty_like(dx)
        fn = lib.rsg_trp_backward_dense if (p32 is not None and p32.dim() == 3) else lib.rsg_trp_backward
        hip.check(fn(x16.data_ptr(), c, g16.data_ptr(), c, de16.data_ptr(), c, x_t.data_ptr(), de_t.data_ptr(),
                     None if p32 is None else p32.data_ptr(), None if coef is None else coef.data_ptr(),
                     dx.data_ptr(), dg.data_ptr(), b, s, c, hip.RSG_BF16, st), "trp_backward")
        dg2 = dg.reshape(-1, c)
        rows_project(dg2, w32, out=dx.reshape(-1, c))                        # g = x Wg^T + bg:  dx += dg Wg
        d_w, d_b = rows_gram(dg2, x16.float().reshape(-1, c), with_colsum=True)
        return dx.to(ctx.in_dtype), d_w.reshape(c, c, 1, 1), d_b, None


def trp_on_hip(rh, channels: int, positions: int, relation_target) -> bool:
    """The HIP pair covers what training uses: 32 channels, S a multiple of 128, the target as its factor [B,S] or as the
    [B,S,S] tensor of the reference's loop (function.py:261-272)."""
    if relation_target is None or relation_target.dim() not in (2, 3):
        return False
    s = positions // (4 if getattr(rh, "sub_sample", False) else 1)
    return channels == 32 and s % 128 == 0


def type_conv_weight(m) -> torch.Tensor:
    """type_conv[0] applied to scores^T @ type_fc(type_features) (pose_rsgnet.py:971-977) == a 3x3 conv J -> C0 over the
    scores with this weight, exactly (both maps are linear in the scores)."""
    tm = m.type_fc(m.type_features)                                       # [J,T], training-mode BatchNorm1d over the J rows
    return torch.einsum("ocyx,jc->ojyx", m.type_conv[0].weight, tm)


def relation_head_forward(rh, x: torch.Tensor):
    """SpatialRelationHead.forward (association.py:280-301) with the score map materialised ([B,S,S] fp32: what the
    reference does; 1.2 GB at batch 32, S = 3072): the path for a target given as [B,S,S] or not at all."""
    if getattr(rh, "sub_sample", False):
        x = F.max_pool2d(x, 2)
    b, c, h, w = x.shape
    g = rh.g(x).view(b, c, -1).permute(0, 2, 1)
    xf = x.view(b, c, -1)
    score = torch.sigmoid(torch.matmul(xf.permute(0, 2, 1), xf))
    e = torch.matmul(score, g).permute(0, 2, 1).contiguous().view(b, c, h, w)
    return rh.W(e), score


def head_forward(model, feat16: torch.Tensor, relation_target: Optional[torch.Tensor]):
    """pose_rsgnet.py:959-1021 from the backbone's output map `feat16` [B,H/4,W/4,C0] (NHWC bf16, requires grad)."""
    from .train_ops import Bilinear2x, GroupNormTail, conv1x1_out, conv_bn_relu, deconv_bn_relu, nchw_to_nhwc16, nhwc16_to_nchw
    m = model
    B, H, W, c0 = feat16.shape
    taps = getattr(m, "_train_taps", None)              # tests: the units' outputs by BatchNorm name (ReLU decisions)
    if taps is not None:
        def conv_bn_relu(owner, key, x16, w, bn, _f=conv_bn_relu):
            z = _f(owner, key, x16, w, bn)
            taps[key + ".1"] = z.detach()
            return z

        def deconv_bn_relu(owner, key, x16, dc, bn, _f=deconv_bn_relu):
            z = _f(owner, key, x16, dc, bn)
            taps[key + ".1"] = z.detach()
            return z
    multi = conv1x1_out(m, "multi_final_layer", feat16, m.multi_final_layer.weight, m.multi_final_layer.bias)   # :961
    vis = conv_bn_relu(m, "vis_conv", feat16, m.vis_conv[0].weight, m.vis_conv[1])                            # :965
    typ = conv_bn_relu(m, "type_conv", nchw_to_nhwc16(multi), type_conv_weight(m), m.type_conv[1])           # :971-977
    loc = nchw_to_nhwc16(_loc_clue(m)).expand(B, -1, -1, -1)                                                  # :979-980
    f = conv_bn_relu(m, "contact_conv", torch.cat((vis, typ, loc), dim=3), m.contact_conv[0].weight, m.contact_conv[1])
    f = conv_bn_relu(m, "predict_contact_net", f, m.predict_contact_net[0].weight, m.predict_contact_net[1])   # :982-984
    rh = m.relation_head
    hip_trp = trp_on_hip(rh, c0, H * W, relation_target)
    if hip_trp:                                                           # :987 and :1014-1018
        x = f
        if getattr(rh, "sub_sample", False):
            x = nchw_to_nhwc16(F.max_pool2d(nhwc16_to_nchw(f), 2))
        b, hr, wr, _ = x.shape
        e, rel = TrpRelation.apply(x.reshape(b, hr * wr, c0), rh.g.weight, rh.g.bias, relation_target)
        if not getattr(rh, "sub_sample", False) and (hr * wr) % 256 == 0:       # W = conv1x1 + bias -> GroupNorm(8, C)
            rel16 = GroupNormTail.apply(e, rh.W[0].weight, rh.W[0].bias, rh.W[1].weight, rh.W[1].bias).reshape(b, hr, wr, c0)
        else:                                                             # the sub-sampled variant's up-sampler: torch
            rel16 = nchw_to_nhwc16(rh.W(e.reshape(b, hr, wr, c0).permute(0, 3, 1, 2)))
    else:
        rel_feats, score = relation_head_forward(rh, nhwc16_to_nchw(f))
        rel16 = nchw_to_nhwc16(rel_feats)
    k = conv_bn_relu(m, "kpt_net", torch.cat((rel16, f), dim=3), m.kpt_net[0].weight, m.kpt_net[1])          # :991-992
    if m.scale_factor > 1:
        k = deconv_bn_relu(m, "predict_convtranspose", k, m.predict_convtranspose[0], m.predict_convtranspose[1])
    k = conv_bn_relu(m, "predict_net", k, m.predict_net[0].weight, m.predict_net[1])
    kpt = conv1x1_out(m, "final_layer", k, m.final_layer.weight, m.final_layer.bias)                           # :999-1000
    lm = conv_bn_relu(m, "limbs_net", f, m.limbs_net[0].weight, m.limbs_net[1])
    limbs = conv1x1_out(m, "limbs", lm, kt_machine_forward(m.kt_machine, m.final_layer.weight), None)         # :1003-1005
    if m.scale_factor > 1:                                                # :1008-1013
        multi = Bilinear2x.apply(multi, False)
        limbs = Bilinear2x.apply(limbs, True)
    else:
        limbs = torch.sigmoid(limbs)
    if hip_trp:
        return multi, kpt, limbs, rel
    rel = score
    if relation_target is not None:                                      # :1014-1018
        with torch.no_grad():
            t = relation_target.to(score.device, score.dtype)
            if t.dim() == 2:                                              # the factor p of T = p p^T (rsgnet_amd.loss.relation_target_vector)
                t = t[:, :, None] * t[:, None, :]
        rel = ((t - score) ** 2).mean(dim=(1, 2))
    return multi, kpt, limbs, rel


class _Head(torch.nn.Module):
    """head_forward as a module of tensors in, tensors out (what torch.cuda.make_graphed_callables captures)."""

    def __init__(self, model, with_target: bool):
        super().__init__()
        self.model = model
        self.with_target = with_target

    def forward(self, feat16, *target):
        return head_forward(self.model, feat16, target[0] if self.with_target else None)


def _head_buffers(model):
    return [b for b in model.buffers() if b.is_cuda]


def graphed_head(model, feat16: torch.Tensor, relation_target: Optional[torch.Tensor]):
    """The head's forward and backward as two HIP graphs (torch.cuda.make_graphed_callables), built once per input
    geometry and replayed afterwards: ~110 launches forward and ~190 backward become two graph launches.  The capture's
    three warm-up passes would apply the BatchNorm momentum update three extra times, so the buffers are put back as they
    were before the first real step.  (Round 3 recorded a segmentation fault in hipStreamEndCapture here; it does not
    occur with this head -- tools/head_capture_probe.py, profiles/r04_head_capture.txt.  Slower than the eager head at
    batch 8 and 32 on this runtime, hence opt-in.)"""
    # the captured graphs freeze more than the input geometry: the mode (train-mode BatchNorm updates its buffers inside the
    # graph) and WHICH tensors are the parameters / which of them take gradients (a replaced or frozen parameter would be
    # read, or have its gradient written, through the pointer captured for the old one) -- advisor, round 4
    params = tuple((id(q), q.data_ptr(), q.requires_grad) for q in model.parameters())
    key = (tuple(feat16.shape), feat16.dtype, feat16.device, bool(model.training), hash(params),
           None if relation_target is None else (tuple(relation_target.shape), relation_target.dtype))
    cache = model.__dict__.setdefault("_head_graphs", {})
    for k in [k for k in cache if k[:3] == key[:3] and k != key]:
        del cache[k]                                  # a stale capture of the same geometry holds its pools for nothing
    fn = cache.get(key)
    if fn is None:
        bufs = _head_buffers(model)
        keep = [b.clone() for b in bufs]
        sample = [feat16.detach().clone().requires_grad_(True)]
        if relation_target is not None:
            sample.append(relation_target.detach().clone())
        fn = torch.cuda.make_graphed_callables(_Head(model, relation_target is not None), tuple(sample), allow_unused_input=True)
        with torch.no_grad():
            for b, k in zip(bufs, keep):
                b.copy_(k)
        cache[key] = fn
    return fn(feat16) if relation_target is None else fn(feat16, relation_target)


def train_forward(model, x: torch.Tensor, relation_target: Optional[torch.Tensor] = None):
    """`model(x, relation_target)` under `model.train()` for pose_rsgnet.  RSGNET_HIP_TRAIN="head_graph=1" replays the head
    (forward and backward) as HIP graphs (`graphed_head`); by default its ~300 launches per step are enqueued eagerly, which
    the host hides behind the backbone's graphs at the shipped batch sizes (profiles/r04_head_capture.txt)."""
    from .train import train_forward as backbone_forward
    feat16 = backbone_forward(model, x)                                   # HIP: one autograd node for the whole backbone
    mode = sw.train("head_graph")
    if mode == "1" and getattr(model, "_train_taps", None) is None and torch.is_grad_enabled():
        return graphed_head(model, feat16, relation_target)
    return head_forward(model, feat16, relation_target)
