"""ORACLE — test infrastructure, NOT product code.

CPU restatement (torch CPU fp32, optionally fp64) of the reference's per-crop forward hot path:

  * plain HRNet           /root/reference/lib/models/pose_hrnet.py:428-463
  * RSGNet                /root/reference/lib/models/pose_rsgnet.py:921-1021
  * TRP (SpatialRelationHead) /root/reference/lib/models/association.py:280-301
  * SGM (KTMachine)       /root/reference/lib/models/pose_rsgnet.py:576-600

It is a *functional* restatement: it takes a reference-named ``state_dict`` (a mapping name -> tensor)
plus the cfg and walks the graph with ``torch.nn.functional`` primitives.  The arithmetic itself lives
in PyTorch's CPU operators (third-party; version used here: torch 2.10.0) exactly as it does for the
reference.

Parity pin: the reference holds no golden vectors or tests for this path.  This oracle is
pinned by fixtures under ``tests/golden/`` that ``oracle/make_goldens.py`` produced by importing the
reference's own ``lib/models`` in the build container and running it on CPU; ``tests/test_oracle.py``
re-checks this file against those fixtures (≤1e-5 max-abs).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product path (``rsgnet_amd``) never does.

Two extra modes exist purely for testing:
  * ``calibrate=True``  — every BatchNorm takes its statistics from the current batch and writes them
    back into the state dict as running stats (what ``model.train()`` with ``momentum=1`` does in the
    reference: normalise with the biased batch variance, store the unbiased one).
    ``make_goldens.py`` uses it to give the synthetic weights realistic, non-identity BN statistics.
  * ``emulate='fp16'|'bf16'`` — rounds activations and BN-folded weights where the HIP path stores
    them in 16 bits (accumulation stays fp32), to predict the HIP path's error on CPU.
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Optional

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d / nn.GroupNorm default, pose_rsgnet.py:31, association.py:244

# pose_rsgnet.py:523-555 — KTMachine always uses the 14 CrowdPose names / 14 limb rules
_KPT_NAMES = ("left_shoulder", "right_shoulder", "left_elbow", "right_elbow", "left_wrist",
              "right_wrist", "left_hip", "right_hip", "left_knee", "right_knee", "left_ankle",
              "right_ankle", "head", "neck")
_LIMB_RULES = (("head", "neck"), ("right_shoulder", "neck"), ("left_shoulder", "neck"),
               ("left_shoulder", "left_elbow"), ("right_shoulder", "right_elbow"),
               ("left_elbow", "left_wrist"), ("right_elbow", "right_wrist"),
               ("left_hip", "right_hip"), ("left_hip", "left_knee"), ("right_hip", "right_knee"),
               ("left_knee", "left_ankle"), ("right_knee", "right_ankle"),
               ("left_shoulder", "left_hip"), ("right_shoulder", "right_hip"))


def limb_prior(num_limbs: int, num_joints: int) -> np.ndarray:
    """pose_rsgnet.py:558-573 — binary limb x joint prior (rows beyond 14 stay zero)."""
    idx = {n: i for i, n in enumerate(_KPT_NAMES)}
    m = np.zeros((num_limbs, num_joints), dtype=np.float32)
    for i, (a, b) in enumerate(_LIMB_RULES):
        m[i, idx[a]] = 1
        m[i, idx[b]] = 1
    return m


def geometry_embeddings(image_wh) -> np.ndarray:
    """pose_rsgnet.py:800-815 — the constant 4-channel location map [1,4,H/4,W/4]."""
    w, h = int(image_wh[0] / 4), int(image_wh[1] / 4)
    xs = np.arange(w, dtype=np.float32)
    ys = np.arange(h, dtype=np.float32)
    m = np.zeros((4, h, w))
    m[0] += (xs / w)[None, :]
    m[1] += (ys / h)[:, None]
    m[2] += (xs - w / 2.0)[None, :]
    m[3] += (ys - h / 2.0)[:, None]
    return m[None].astype(np.float32)


class Ctx:
    """Walk state: the weights, the mode switches and a few taps for block-level tests."""

    def __init__(self, sd: Mapping[str, torch.Tensor], dtype=torch.float32, calibrate=False,
                 emulate: Optional[str] = None, relu_masks: Optional[Mapping[str, torch.Tensor]] = None):
        self.sd: Dict[str, torch.Tensor] = sd if calibrate else dict(sd)
        self.dtype = dtype
        self.calibrate = calibrate
        self.emu = {None: None, "fp16": torch.float16, "bf16": torch.bfloat16}[emulate]
        self.taps: Dict[str, torch.Tensor] = {}
        # teacher forcing (tests of the training path): {BatchNorm name or "<module>.fuse.<i>": 0/1 map [B,C,H,W]} taken
        # from ANOTHER implementation's activations; a ReLU whose key is present multiplies by that map instead of
        # deciding for itself, so both implementations differentiate the same piecewise-linear function
        self.relu_masks = relu_masks
        # fault injection for the SENSITIVITY tests of the training parity bars (tests/test_train_step.py): forward values
        # unchanged, one backward term missing.  "<module>.fuse.<i>.<j>": that addend of a fuse sum passes no gradient;
        # "<BatchNorm name>": the batch variance passes no gradient (one of the three terms of BatchNorm's backward).
        self.drop_grad: set = set()
        # precision study (tests/test_precision_floor.py): which of the two 16-bit roundings of the emulation to apply -- the
        # folded weights' (MFMA A operand) and the stored activations' (the next conv's B operand)
        self.emu_weights = True
        self.emu_acts = True

    def relu(self, y: torch.Tensor, key: Optional[str]) -> torch.Tensor:
        if self.relu_masks is not None and key is not None and key in self.relu_masks:
            return y * self.relu_masks[key].to(y.dtype)
        return F.relu(y)

    def p(self, name: str) -> torch.Tensor:
        return self.sd[name].to(self.dtype)

    def has(self, name: str) -> bool:
        return name in self.sd

    def store(self, x: torch.Tensor) -> torch.Tensor:
        """Where the HIP path writes an activation in 16 bits."""
        if self.emu is None or not self.emu_acts:
            return x
        return x.to(self.emu).to(self.dtype)


def _bn_coeffs(ctx: Ctx, name: str, x: Optional[torch.Tensor]):
    """(scale, shift) of an eval-mode BatchNorm; in calibrate mode first set the running stats."""
    if ctx.calibrate:
        dims = [d for d in range(x.dim()) if d != 1]
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=dims)
        var_b = x.var(dim=dims, unbiased=False)
        if name in ctx.drop_grad:
            var_b = var_b.detach()
        ctx.sd[name + ".running_mean"] = mean.to(torch.float32)
        ctx.sd[name + ".running_var"] = (var_b * (n / max(n - 1, 1))).to(torch.float32)
        inv = torch.rsqrt(var_b + BN_EPS)  # training normalises with the biased variance
    else:
        mean = ctx.p(name + ".running_mean")
        inv = torch.rsqrt(ctx.p(name + ".running_var") + BN_EPS)
    scale = ctx.p(name + ".weight") * inv
    shift = ctx.p(name + ".bias") - mean * scale
    return scale, shift


def conv_unit(ctx: Ctx, x, conv: str, bn: Optional[str] = None, *, stride=1, relu=False,
              residual=None, transposed=False):
    """conv (+bias) -> [BatchNorm] -> [+residual] -> [ReLU]: the unit the HIP path fuses.

    Exact mode does the four steps as the reference does (separate ops, fp32).  Emulation mode folds
    the BN scale into the weight, rounds weight and output to 16 bits and keeps the bias in fp32,
    which is what the HIP kernels compute.
    """
    w = ctx.p(conv + ".weight")
    b = ctx.p(conv + ".bias") if ctx.has(conv + ".bias") else None
    k = w.shape[-1]
    pad = (k - 1) // 2 if not transposed else 1

    def run(xx, ww, bb):
        if transposed:  # ConvTranspose2d(k=4, s=2, p=1, op=0), pose_rsgnet.py:727-734
            return F.conv_transpose2d(xx, ww, bb, stride=2, padding=pad)
        return F.conv2d(xx, ww, bb, stride=stride, padding=pad)

    if ctx.emu is None or ctx.calibrate:
        y = run(x, w, b)
        if bn is not None:
            s, t = _bn_coeffs(ctx, bn, y)
            y = y * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)
        if residual is not None:
            y = y + residual
        return ctx.relu(y, bn) if relu else y

    if bn is not None:
        s, t = _bn_coeffs(ctx, bn, None)
    else:
        s, t = torch.ones(w.shape[1 if transposed else 0], dtype=ctx.dtype), 0.0
    shape = (1, -1, 1, 1) if transposed else (-1, 1, 1, 1)
    wq = w * s.view(shape)
    if ctx.emu_weights:
        wq = wq.to(ctx.emu).to(ctx.dtype)
    bias = t if b is None else (b * s + t)
    y = run(x, wq, None) + (bias.view(1, -1, 1, 1) if torch.is_tensor(bias) else bias)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return ctx.store(y)


# ------------------------------------------------------------------------------------------------
# backbone (shared by pose_hrnet.py and pose_rsgnet.py — the reference duplicates the code)
# ------------------------------------------------------------------------------------------------

def bottleneck(ctx: Ctx, x, pre: str):
    """pose_rsgnet.py:57-95."""
    res = x
    if ctx.has(pre + ".downsample.0.weight"):
        res = conv_unit(ctx, x, pre + ".downsample.0", pre + ".downsample.1")
    y = conv_unit(ctx, x, pre + ".conv1", pre + ".bn1", relu=True)
    y = conv_unit(ctx, y, pre + ".conv2", pre + ".bn2", relu=True)
    return conv_unit(ctx, y, pre + ".conv3", pre + ".bn3", relu=True, residual=res)


def basic_block(ctx: Ctx, x, pre: str):
    """pose_rsgnet.py:25-54 (HRNet branches never carry a downsample)."""
    y = conv_unit(ctx, x, pre + ".conv1", pre + ".bn1", relu=True)
    return conv_unit(ctx, y, pre + ".conv2", pre + ".bn2", relu=True, residual=x)


def hr_module(ctx: Ctx, xs: List[torch.Tensor], pre: str, n_out: int):
    """HighResolutionModule.forward, pose_rsgnet.py:254-272; fuse layers :194-249."""
    nb = len(xs)
    xs = list(xs)
    for b in range(nb):
        k = 0
        while ctx.has(f"{pre}.branches.{b}.{k}.conv1.weight"):
            xs[b] = basic_block(ctx, xs[b], f"{pre}.branches.{b}.{k}")
            k += 1
    if nb == 1:
        return xs
    outs = []
    for i in range(n_out):
        y = None
        for j in range(nb):
            fl = f"{pre}.fuse_layers.{i}.{j}"
            if j == i:
                t = xs[j]
            elif j > i:  # 1x1 + BN then nearest upsample by 2^(j-i)
                t = conv_unit(ctx, xs[j], fl + ".0", fl + ".1")
                t = F.interpolate(t, scale_factor=2 ** (j - i), mode="nearest")
            else:  # (i-j) strided 3x3; ReLU on all but the last
                t = xs[j]
                for k in range(i - j):
                    t = conv_unit(ctx, t, f"{fl}.{k}.0", f"{fl}.{k}.1", stride=2,
                                  relu=(k != i - j - 1))
            if f"{pre}.fuse.{i}.{j}" in ctx.drop_grad:
                t = t.detach()
            y = t if y is None else y + t
        outs.append(ctx.store(ctx.relu(y, f"{pre}.fuse.{i}")))
    return outs


def _count(ctx: Ctx, fmt: str) -> int:
    n = 0
    while ctx.has(fmt.format(n)):
        n += 1
    return n


def backbone(ctx: Ctx, x, cfg) -> List[torch.Tensor]:
    """_forward_visual_encoder, pose_rsgnet.py:921-953 == pose_hrnet.py:429-459."""
    extra = cfg["MODEL"]["EXTRA"]
    x = ctx.store(x.to(ctx.dtype))
    x = conv_unit(ctx, x, "conv1", "bn1", stride=2, relu=True)
    x = conv_unit(ctx, x, "conv2", "bn2", stride=2, relu=True)
    ctx.taps["stem"] = x
    for k in range(4):
        x = bottleneck(ctx, x, f"layer1.{k}")
    ctx.taps["layer1"] = x

    ys = [x]
    for sidx in (2, 3, 4):
        scfg = extra[f"STAGE{sidx}"]
        nb = scfg["NUM_BRANCHES"]
        tr = f"transition{sidx - 1}"
        xs = []
        for i in range(nb):
            if i < len(ys):
                if ctx.has(f"{tr}.{i}.0.weight"):
                    xs.append(conv_unit(ctx, ys[i], f"{tr}.{i}.0", f"{tr}.{i}.1", relu=True))
                else:
                    xs.append(ys[i])
            else:  # new branch: strided conv(s) from the lowest-resolution map, :938-951
                t = ys[-1]
                for j in range(_count(ctx, tr + f".{i}." + "{}.0.weight")):
                    t = conv_unit(ctx, t, f"{tr}.{i}.{j}.0", f"{tr}.{i}.{j}.1", stride=2, relu=True)
                xs.append(t)
        nmod = scfg["NUM_MODULES"]
        for m in range(nmod):
            last = sidx == 4 and m == nmod - 1  # multi_scale_output=False, :899-904
            xs = hr_module(ctx, xs, f"stage{sidx}.{m}", 1 if last else nb)
        ys = xs
        ctx.taps[f"stage{sidx}"] = ys[0]
    return ys


def hrnet_forward(sd, cfg, x, **kw) -> torch.Tensor:
    """pose_hrnet.PoseHighResolutionNet.forward, pose_hrnet.py:428-463 -> [B,J,H/4,W/4]."""
    ctx = kw.pop("ctx", None) or Ctx(sd, **kw)
    with torch.no_grad():
        ys = backbone(ctx, x, cfg)
        w, b = ctx.p("final_layer.weight"), ctx.p("final_layer.bias")
        if ctx.emu is not None and ctx.emu_weights:
            w = w.to(ctx.emu).to(ctx.dtype)
        return F.conv2d(ys[0], w, b, padding=(w.shape[-1] - 1) // 2)


# ------------------------------------------------------------------------------------------------
# RSGNet head
# ------------------------------------------------------------------------------------------------

def type_templates(ctx: Ctx) -> torch.Tensor:
    """type_fc(type_features): Linear(no bias) -> BatchNorm1d -> ReLU on [J,T], pose_rsgnet.py:679-683,972."""
    t = ctx.p("type_features") @ ctx.p("type_fc.0.weight").t()
    s, sh = _bn_coeffs(ctx, "type_fc.1", t)
    return F.relu(t * s + sh)


def kt_machine(ctx: Ctx, wk: torch.Tensor) -> torch.Tensor:
    """KTMachine.forward, pose_rsgnet.py:592-600: joint classifier weights -> limb classifier weights."""
    n_out, n_in, h, w = wk.shape
    k = ctx.p("kt_machine.matrix_limb") * ctx.p("kt_machine.real_matrix_limb")
    t = k @ wk.reshape(n_out, n_in * h * w)
    t = F.linear(t, ctx.p("kt_machine.kpt_transformer.0.weight"), ctx.p("kt_machine.kpt_transformer.0.bias"))
    t = F.leaky_relu(t, 0.02)
    t = F.linear(t, ctx.p("kt_machine.kpt_transformer.2.weight"), ctx.p("kt_machine.kpt_transformer.2.bias"))
    return t.reshape(t.shape[0], n_in, h, w)


def relation_head(ctx: Ctx, x, want_scores=True):
    """SpatialRelationHead.forward, association.py:280-301.

    Returns (relation_embs [B,C,H,W], relation_score [B,S,S] or None).  theta = phi = x (no
    projection), attention is an un-normalised sigmoid of the Gram matrix.  With sub_sample
    (association.py:254-268,286-287; recognised by the W.0.0 deconv in the state dict) x is 2x2 max-pooled first
    (S -> S/4) and W = deconv4x4s2 -> BatchNorm -> ReLU -> 1x1 -> GroupNorm brings the map back to H x W.
    """
    sub = ctx.has("relation_head.W.0.0.weight")
    if sub:
        x = F.max_pool2d(x, 2)                                  # nn.MaxPool2d(kernel_size=(2, 2)), :286-287
    b, c, h, w = x.shape
    g = F.conv2d(x, ctx.p("relation_head.g.weight"), ctx.p("relation_head.g.bias"))
    g = g.view(b, c, -1).permute(0, 2, 1)                      # [B,S,C]
    xf = x.view(b, c, -1)                                       # [B,C,S]
    embs = torch.empty(b, h * w, c, dtype=x.dtype)
    scores = torch.empty(b, h * w, h * w, dtype=x.dtype) if want_scores else None
    for i in range(b):                                          # per crop: keeps the S*S map small
        r = torch.sigmoid(xf[i].t() @ xf[i])
        embs[i] = r @ g[i]
        if want_scores:
            scores[i] = r
    e = embs.permute(0, 2, 1).contiguous().view(b, c, h, w)
    ctx.taps["relation_embs_raw"] = e
    if sub:
        e = conv_unit(ctx, ctx.store(e), "relation_head.W.0.0", "relation_head.W.0.1", relu=True, transposed=True)
        e = F.conv2d(e, ctx.p("relation_head.W.1.0.weight"), ctx.p("relation_head.W.1.0.bias"))
        e = F.group_norm(e, 8, ctx.p("relation_head.W.1.1.weight"), ctx.p("relation_head.W.1.1.bias"), BN_EPS)
    else:
        e = F.conv2d(e, ctx.p("relation_head.W.0.weight"), ctx.p("relation_head.W.0.bias"))
        e = F.group_norm(e, 8, ctx.p("relation_head.W.1.weight"), ctx.p("relation_head.W.1.bias"), BN_EPS)
    return ctx.store(e), scores


def rsgnet_forward(sd, cfg, x, relation_target=None, want_scores=True, **kw):
    """RSGNet.forward, pose_rsgnet.py:955-1021.

    Returns (multi_kpt_scores, kpt_scores, limbs_scores, relation_scores) like the reference;
    relation_scores is [B,S,S], or [B] when ``relation_target`` is given, or None when
    ``want_scores=False`` (a test convenience: the 37.7 MB/crop map is the one thing the HIP path
    never materialises unless asked).
    """
    grad = kw.pop("grad", False)             # True: build the autograd graph (oracle/train_step_oracle.rsgnet_train_step)
    ctx = kw.pop("ctx", None) or Ctx(sd, **kw)
    model = cfg["MODEL"]
    with (torch.enable_grad() if grad else torch.no_grad()):
        vis = backbone(ctx, x, cfg)[0]
        ctx.taps["vis"] = vis
        wm, bm = ctx.p("multi_final_layer.weight"), ctx.p("multi_final_layer.bias")
        if ctx.emu is not None:
            wm = wm.to(ctx.emu).to(ctx.dtype)
        multi = F.conv2d(vis, wm, bm)                                            # :961
        vis = conv_unit(ctx, vis, "vis_conv.0", "vis_conv.1", relu=True)          # :965
        b, j, h, w = multi.shape
        tm = type_templates(ctx)                                                  # [J,T] :971-972
        if ctx.emu is None or ctx.calibrate:
            tmap = (multi.reshape(b, j, h * w).permute(0, 2, 1) @ tm)             # [B,S,T] :974
            tmap = tmap.permute(0, 2, 1).reshape(b, tm.shape[1], h, w)
            typ = conv_unit(ctx, tmap, "type_conv.0", "type_conv.1", relu=True)   # :977
        else:
            # the HIP path's exact fold: a 3x3 conv J->C0 with W'[o,j] = sum_c W[o,c] Tm[j,c]
            s, t = _bn_coeffs(ctx, "type_conv.1", None)
            wf = torch.einsum("ocyx,jc->ojyx", ctx.p("type_conv.0.weight"), tm) * s.view(-1, 1, 1, 1)
            typ = F.conv2d(ctx.store(multi), wf.to(ctx.emu).to(ctx.dtype), None, padding=1)
            typ = ctx.store(F.relu(typ + t.view(1, -1, 1, 1)))
        loc = conv_unit(ctx, ctx.p("loc_features"), "loc_conv.0", "loc_conv.1", relu=True)  # :979
        f = torch.cat((vis, typ, loc.repeat(b, 1, 1, 1)), dim=1)                  # :982
        f = conv_unit(ctx, f, "contact_conv.0", "contact_conv.1", relu=True)
        f = conv_unit(ctx, f, "predict_contact_net.0", "predict_contact_net.1", relu=True)
        ctx.taps["final_vis"] = f
        rel, scores = relation_head(ctx, f, want_scores or relation_target is not None)   # :987
        ctx.taps["relation_feats"] = rel
        k = conv_unit(ctx, torch.cat((rel, f), dim=1), "kpt_net.0", "kpt_net.1", relu=True)  # :991-992
        if model["UP_SCALE"] > 1:
            k = conv_unit(ctx, k, "predict_convtranspose.0", "predict_convtranspose.1", relu=True,
                          transposed=True)                                        # :994
        k = conv_unit(ctx, k, "predict_net.0", "predict_net.1", relu=True)        # :999
        wf_, bf_ = ctx.p("final_layer.weight"), ctx.p("final_layer.bias")
        kpt = F.conv2d(k, wf_.to(ctx.emu).to(ctx.dtype) if ctx.emu is not None else wf_, bf_)  # :1000
        lf = conv_unit(ctx, f, "limbs_net.0", "limbs_net.1", relu=True)           # :1003
        wl = kt_machine(ctx, ctx.p("final_layer.weight"))                         # :1004
        ctx.taps["limb_weights"] = wl
        limbs = F.conv2d(lf, wl.to(ctx.emu).to(ctx.dtype) if ctx.emu is not None else wl)  # :1005
        ctx.taps["limbs_logits"] = limbs                                          # before x2 upsampling and sigmoid
        if model["UP_SCALE"] > 1:                                                 # :1008-1012
            multi = F.interpolate(multi, scale_factor=2, mode="bilinear", align_corners=True)
            limbs = F.interpolate(limbs, scale_factor=2, mode="bilinear", align_corners=True)
        limbs = torch.sigmoid(limbs)                                              # :1013
        if relation_target is not None:                                           # :1014-1018
            scores = ((relation_target.to(ctx.dtype) - scores) ** 2).mean(dim=(1, 2))
        return multi, kpt, limbs, scores


def flip_test_merge(output, output_flipped, flip_pairs, shift_heatmap=True):
    """lib/core/function.py:414-427 with lib/utils/transforms.py:23-37 (flip_back), restated with torch ops:
    reverse W, swap the paired joint channels, shift one pixel to the right, average."""
    f = output_flipped.flip(3).clone()
    for a, b in flip_pairs:
        tmp = f[:, a].clone()
        f[:, a] = f[:, b]
        f[:, b] = tmp
    if shift_heatmap:
        f[:, :, :, 1:] = f.clone()[:, :, :, 0:-1]
    return (output + f) * 0.5


def _affine_from_points(src, dst):
    """What cv2.getAffineTransform(src, dst) returns (third-party, not installed here): the 2x3 matrix
    mapping the three src points onto the three dst points, solved in float64."""
    a = np.hstack([np.asarray(src, np.float64), np.ones((3, 1))])
    return np.linalg.solve(a, np.asarray(dst, np.float64)).T


def inverse_crop_affine(center, scale, output_size):
    """get_affine_transform(center, scale, rot=0, output_size, inv=1), lib/utils/transforms.py:65-101."""
    scale_tmp = np.asarray(scale, np.float32) * 200.0
    src_w, dst_w, dst_h = scale_tmp[0], output_size[0], output_size[1]
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = center
    src[1] = np.asarray(center, np.float32) + np.array([0, src_w * -0.5], np.float32)
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5]) + np.array([0, dst_w * -0.5], np.float32)
    for p in (src, dst):                     # get_3rd_point: a + (-(a-b).y, (a-b).x) rotated by 90 degrees
        d = p[0] - p[1]
        p[2] = p[1] + np.array([-d[1], d[0]], np.float32)
    return _affine_from_points(dst, src)


def final_preds(batch_heatmaps, center, scale, post_process=True):
    """get_max_preds + get_final_preds, lib/core/inference.py:21-82 (numpy, loops and all)."""
    hm = np.asarray(batch_heatmaps, np.float32)
    b, j, h, w = hm.shape
    flat = hm.reshape(b, j, -1)
    idx = np.argmax(flat, 2)
    maxvals = np.amax(flat, 2).reshape(b, j, 1)
    coords = np.zeros((b, j, 2), np.float32)
    coords[:, :, 0] = idx % w
    coords[:, :, 1] = np.floor(idx / w)
    coords *= np.tile(np.greater(maxvals, 0.0), (1, 1, 2)).astype(np.float32)
    if post_process:
        for n in range(b):
            for p in range(j):
                m = hm[n][p]
                px = int(math.floor(coords[n][p][0] + 0.5))
                py = int(math.floor(coords[n][p][1] + 0.5))
                if 1 < px < w - 1 and 1 < py < h - 1:
                    diff = np.array([m[py][px + 1] - m[py][px - 1], m[py + 1][px] - m[py - 1][px]])
                    coords[n][p] += np.sign(diff) * .25
    preds = coords.copy()
    for i in range(b):
        t = inverse_crop_affine(center[i], scale[i], [w, h])
        for p in range(j):
            preds[i, p] = t @ np.array([coords[i, p, 0], coords[i, p, 1], 1.0])
    return preds, maxvals
