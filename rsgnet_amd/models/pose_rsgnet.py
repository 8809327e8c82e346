"""Drop-in for the reference's ``models.pose_rsgnet`` (HRNet + TRP + SGM).

Boundary: ``get_pose_net(cfg, is_train, **kw)`` (/root/reference/lib/models/pose_rsgnet.py:1066-1075),
``forward(x, relation_target=None) -> (multi_kpt_scores, kpt_scores, limbs_scores, relation_scores)``
(pose_rsgnet.py:955-1021), reference state_dict names.

Differences from the reference that a caller can observe (all deliberate):
  * the keypoint word embeddings come from ``rsgnet_amd/data/{kpt,cp_kpt}_word_embs.npy`` chosen by
    NUM_JOINTS instead of ``open('kpt_word_embs.pkl')`` relative to the CWD (pose_rsgnet.py:674-675,
    SURVEY §0 defect 3);
  * ``relation_scores`` (the [B,S,S] fp32 sigmoid map, 37.7 MB per crop) is returned as a lazy
    ``RelationScores`` handle unless ``model.materialize_relation_scores = True``; eval callers drop
    it (lib/core/function.py:389), so by default it is never written to HBM.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from ..config import as_cfg
from ._base import HipPoseModule
from ._tree import bn, build_backbone, conv, conv_bn

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")

# The 14 CrowdPose keypoint names and limb rules KTMachine uses for every dataset
# (pose_rsgnet.py:523-555; SURVEY §0 defect 5: the COCO prior is 18x17 with rows 14-17 zero).
_NAMES = ("left_shoulder right_shoulder left_elbow right_elbow left_wrist right_wrist left_hip "
          "right_hip left_knee right_knee left_ankle right_ankle head neck").split()
_LIMBS = ("head-neck right_shoulder-neck left_shoulder-neck left_shoulder-left_elbow "
          "right_shoulder-right_elbow left_elbow-left_wrist right_elbow-right_wrist left_hip-right_hip "
          "left_hip-left_knee right_hip-right_knee left_knee-left_ankle right_knee-right_ankle "
          "left_shoulder-left_hip right_shoulder-right_hip").split()


def skeleton_prior(num_limbs: int, num_joints: int) -> torch.Tensor:
    m = torch.zeros(num_limbs, num_joints)
    for i, pair in enumerate(_LIMBS):
        for name in pair.split("-"):
            m[i, _NAMES.index(name)] = 1.0
    return m


def location_map(image_wh) -> torch.Tensor:
    """[1,4,H/4,W/4]: x/W4, y/H4, x-W4/2, y-H4/2 (pose_rsgnet.py:800-815)."""
    w4, h4 = int(image_wh[0] / 4), int(image_wh[1] / 4)
    xs = torch.arange(w4, dtype=torch.float64)
    ys = torch.arange(h4, dtype=torch.float64)
    m = torch.stack([(xs / w4).expand(h4, w4), (ys / h4)[:, None].expand(h4, w4),
                     (xs - w4 / 2.0).expand(h4, w4), (ys - h4 / 2.0)[:, None].expand(h4, w4)])
    return m[None].float()


class KTMachine(nn.Module):
    """SGM parameters: learned limb x joint relation masked by the skeleton prior, and the two-layer
    transformer from joint-classifier to limb-classifier weights (pose_rsgnet.py:499-521)."""

    def __init__(self, cfg, weight_size: int):
        super().__init__()
        j, l = cfg.MODEL.NUM_JOINTS, cfg.MODEL.NUM_LIMBS
        self.matrix_limb = nn.Parameter(torch.randn(l, j))
        self.real_matrix_limb = nn.Parameter(skeleton_prior(l, j), requires_grad=False)
        self.kpt_transformer = nn.Sequential(nn.Linear(weight_size, weight_size), nn.LeakyReLU(0.02),
                                             nn.Linear(weight_size, weight_size))
        for name, p in self.named_parameters():      # pose_rsgnet.py:581-589
            if "bias" in name:
                nn.init.constant_(p, 0)
            elif "weight" in name:
                nn.init.kaiming_normal_(p, mode="fan_out", nonlinearity="relu")


class SpatialRelationHead(nn.Module):
    """TRP parameters (association.py:199-278): g 1x1 (+bias), W = 1x1 (+bias) -> GroupNorm(8).

    sub_sample (MODEL.RELATION_SUB_SAMPLE; experiments/crowdpose/hrnet/"rsgnet_w48_256x192_adam_lr1e-3 .yaml":42):
    the relation is computed on the 2x2 max-pooled map and W becomes Sequential(up_sample_layer, W) with
    up_sample_layer = ConvTranspose2d(4, stride 2, pad 1, no bias) -> BatchNorm2d -> ReLU (association.py:254-268), i.e.
    parameters W.0.0.weight, W.0.1.*, W.1.0.{weight,bias}, W.1.1.{weight,bias}."""

    def __init__(self, channels: int, sub_sample: bool = False):
        super().__init__()
        self.sub_sample = bool(sub_sample)
        self.g = nn.Conv2d(channels, channels, 1)
        self.W = nn.Sequential(nn.Conv2d(channels, channels, 1), nn.GroupNorm(8, channels))
        if self.sub_sample:
            up = nn.Sequential(nn.ConvTranspose2d(channels, channels, 4, 2, 1, 0, bias=False),
                               nn.BatchNorm2d(channels, momentum=0.1), nn.ReLU(inplace=True))
            self.W = nn.Sequential(up, self.W)
        for name, p in self.named_parameters():      # association.py:271-278
            if "bias" in name:
                nn.init.constant_(p, 0)
            elif "weight" in name:
                nn.init.normal_(p, std=0.001)


class RelationScores:
    """Lazy handle for TRP's [B,S,S] sigmoid map; ``materialize()`` computes it on the GPU."""

    def __init__(self, plan):
        self._plan = plan
        S = plan.S_rel                         # S, or S/4 with RELATION_SUB_SAMPLE (the relation of the pooled map)
        self.shape = torch.Size((plan.batch, S, S))

    def materialize(self) -> torch.Tensor:
        return self._plan.relation_scores()


class RSGNet(HipPoseModule):
    _init_linear = True

    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.cfg = cfg = as_cfg(cfg)
        model, extra = cfg.MODEL, cfg.MODEL.EXTRA
        if model.UDP_POSE_ON:
            raise NotImplementedError("UDP_POSE_ON: dead branch in the reference (SURVEY §0 defect 7)")
        self.UDP_ON = False
        chans = build_backbone(self, extra)
        k = extra.FINAL_CONV_KERNEL
        if k != 1:
            raise NotImplementedError("FINAL_CONV_KERNEL != 1 is not used by any shipped config")
        J, c0, T = model.NUM_JOINTS, extra["OUTPUT_CONVS"][0], model.NUM_TYPE_VECTOR
        self.num_joints = J
        self.multi_final_layer = nn.Conv2d(chans[0], J, k, 1, 0)
        self.vis_conv = conv_bn(c0, c0, 3)
        emb = np.load(os.path.join(_DATA, "kpt_word_embs.npy" if J == 17 else "cp_kpt_word_embs.npy"))
        if emb.shape != (J, T):
            raise ValueError(f"word embeddings are {emb.shape}, cfg wants ({J}, {T})")
        self.type_features = nn.Parameter(torch.from_numpy(emb).float())
        self.type_fc = nn.Sequential(nn.Linear(T, T, bias=False), nn.BatchNorm1d(T, momentum=0.1),
                                     nn.ReLU(inplace=True))
        self.type_conv = conv_bn(T, c0, 3)
        self.loc_features = nn.Parameter(location_map(model.IMAGE_SIZE), requires_grad=False)
        self.loc_conv = conv_bn(4, c0, 1)
        self.contact_conv = conv_bn(3 * c0, 3 * c0, 3)
        self.predict_contact_net = conv_bn(3 * c0, c0, 3)
        self.relation_head = SpatialRelationHead(c0, model.RELATION_SUB_SAMPLE)
        self.kpt_net = conv_bn(2 * c0, c0, 3)
        self.scale_factor = model.UP_SCALE
        if self.scale_factor not in (1, 2):
            raise NotImplementedError("UP_SCALE must be 1 or 2")
        if self.scale_factor > 1:
            if model.FINAL_DECONV_KERNEL_SIZE != 4:
                raise NotImplementedError("only FINAL_DECONV_KERNEL_SIZE=4 is used by shipped configs")
            self.predict_convtranspose = nn.Sequential(
                nn.ConvTranspose2d(c0, c0, 4, 2, 1, 0, bias=False), bn(c0), nn.ReLU(inplace=True))
        self.predict_net = conv_bn(c0, c0, 3)
        self.final_layer = nn.Conv2d(chans[0], J, k, 1, 0)
        self.kt_machine = KTMachine(cfg, chans[0] * k * k)
        self.limbs_net = conv_bn(c0, c0, 3)
        self.pretrained_layers = extra.PRETRAINED_LAYERS
        self.materialize_relation_scores = False

    def _build_plan(self, batch, device):
        from ..plan import RSGNetPlan
        return RSGNetPlan(self, batch, device, self.compute_dtype)

    #: training mode: backbone on the HIP training plan, head as autograd nodes whose forward and backward are
    #: the library's kernels (rsgnet_amd/train_rsgnet.py, rsgnet_amd/train_ops.py)
    supports_training = True
    train_head_on_torch = True
    train_feat_nhwc16 = True          # the backbone's output map and its gradient cross the boundary as NHWC bf16

    def forward(self, x: torch.Tensor, relation_target=None):
        if self.training:
            if x.device.type != "cuda":
                raise RuntimeError(f"{type(self).__name__} runs only on an AMD GPU through librsgnet_hip.so; got a "
                                   f"{x.device.type} tensor. There is no CPU fallback.")
            from ..train_rsgnet import train_forward
            return train_forward(self, x, relation_target)
        plan, (multi, kpt, limbs) = self.run_plan(x)
        with torch.cuda.device(x.device):
            if relation_target is not None:
                rel = plan.relation_mse(relation_target)          # [B], pose_rsgnet.py:1014-1018
            elif self.materialize_relation_scores:
                rel = plan.relation_scores()
            else:
                rel = RelationScores(plan)
        return multi, kpt, limbs, rel


def get_pose_net(cfg, is_train, **kwargs):
    model = RSGNet(cfg, **kwargs)
    cfg = model.cfg
    if is_train and cfg.MODEL.INIT_WEIGHTS:
        model.init_weights(cfg.MODEL.PRETRAINED)
    return model
