"""Training-mode forward of `pose_rsgnet` (/root/reference/lib/models/pose_rsgnet.py:955-1021 under
`model.train()`, driven by `rsgnet_train`, /root/reference/lib/core/function.py:239-323).

Two parts:
  * the HRNet backbone (stem .. stage4: 99 % of the step's arithmetic) runs on the HIP training plan of
    rsgnet_amd/train.py -- the same launch lists as `pose_hrnet`, recorded up to the backbone's output map, which comes out
    as ONE autograd node;
  * the head (clue encoders, TRP, joints-of-target branch, SGM) is written here with PyTorch-ROCm operators on the model's
    own modules, so autograd supplies its backward.  INTERIM: the HIP inference kernels of the head (csrc/trp.hip, the
    grouped convs) have no backward yet; until they do, the head's training arithmetic is torch's (GPU, fp32).  The two
    weight-only sub-graphs keep their exact algebraic folds, which autograd differentiates: the type path is a 3x3 conv
    J -> C0 with W'[o,j] = sum_c type_conv.W[o,c] * Tm[j,c] (the [B,600,H,W] map of pose_rsgnet.py:974 is never built), and
    the relation target stays factored (T = p p^T) when it is given as [B,S].
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F


def kt_machine_forward(kt, weight: torch.Tensor) -> torch.Tensor:
    """KTMachine.forward (pose_rsgnet.py:576-600): the limb classifier from the joint classifier's weights."""
    n_out, n_in, h, w = weight.shape
    k = kt.matrix_limb * kt.real_matrix_limb                              # learned L x J masked by the 0/1 skeleton prior
    t = k @ weight.reshape(n_out, n_in * h * w)
    t = kt.kpt_transformer(t)                                            # Linear -> LeakyReLU(0.02) -> Linear
    return t.reshape(t.shape[0], n_in, h, w)


def relation_head_forward(rh, x: torch.Tensor):
    """SpatialRelationHead.forward (association.py:280-301) with the score map materialised ([B,S,S] fp32: what the
    reference does; 1.2 GB at batch 32, S = 3072)."""
    if getattr(rh, "sub_sample", False):
        x = F.max_pool2d(x, 2)
    b, c, h, w = x.shape
    g = rh.g(x).view(b, c, -1).permute(0, 2, 1)
    xf = x.view(b, c, -1)
    score = torch.sigmoid(torch.matmul(xf.permute(0, 2, 1), xf))
    e = torch.matmul(score, g).permute(0, 2, 1).contiguous().view(b, c, h, w)
    return rh.W(e), score


def head_forward(model, feat: torch.Tensor, relation_target: Optional[torch.Tensor]):
    """pose_rsgnet.py:959-1021 from the backbone's output map `feat` [B,C0,H/4,W/4] (fp32 NCHW, requires grad)."""
    m = model
    multi = m.multi_final_layer(feat)                                     # :961
    vis = m.vis_conv(feat)                                                # :965
    tm = m.type_fc(m.type_features)                                       # [J,T], training-mode BatchNorm1d over the J rows (:971-972)
    w_fold = torch.einsum("ocyx,jc->ojyx", m.type_conv[0].weight, tm)     # == type_conv(scores^T @ Tm), :974-977, exactly
    typ = m.type_conv[2](m.type_conv[1](F.conv2d(multi, w_fold, None, padding=1)))
    loc = m.loc_conv(m.loc_features).repeat(feat.shape[0], 1, 1, 1)       # :979-980
    f = m.predict_contact_net(m.contact_conv(torch.cat((vis, typ, loc), dim=1)))   # :982-984
    rel_feats, score = relation_head_forward(m.relation_head, f)          # :987
    k = m.kpt_net(torch.cat((rel_feats, f), dim=1))                       # :991-992
    if m.scale_factor > 1:
        k = m.predict_convtranspose(k)
    kpt = m.final_layer(m.predict_net(k))                                 # :999-1000
    limbs = F.conv2d(m.limbs_net(f), kt_machine_forward(m.kt_machine, m.final_layer.weight))   # :1003-1005
    if m.scale_factor > 1:
        multi = F.interpolate(multi, scale_factor=2, mode="bilinear", align_corners=True)
        limbs = F.interpolate(limbs, scale_factor=2, mode="bilinear", align_corners=True)
    limbs = torch.sigmoid(limbs)                                          # :1013
    rel = score
    if relation_target is not None:                                      # :1014-1018
        with torch.no_grad():
            t = relation_target.to(score.device, score.dtype)
            if t.dim() == 2:                                              # the factor p of T = p p^T (rsgnet_amd.loss.relation_target_vector)
                t = t[:, :, None] * t[:, None, :]
        rel = ((t - score) ** 2).mean(dim=(1, 2))
    return multi, kpt, limbs, rel


def train_forward(model, x: torch.Tensor, relation_target: Optional[torch.Tensor] = None):
    """`model(x, relation_target)` under `model.train()` for pose_rsgnet."""
    from .train import train_forward as backbone_forward
    feat = backbone_forward(model, x)                                     # HIP: one autograd node for the whole backbone
    return head_forward(model, feat, relation_target)
