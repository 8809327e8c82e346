"""Drop-in for the reference's ``models.pose_hrnet`` (plain HRNet pose network).

Boundary: ``get_pose_net(cfg, is_train, **kw)`` (/root/reference/lib/models/pose_hrnet.py:778-786),
``forward(x[B,3,H,W] fp32 NCHW) -> heatmaps [B,J,H/4,W/4] fp32`` (pose_hrnet.py:428-463), reference
state_dict names.  Compute: hand-written HIP kernels via ``rsgnet_amd.plan``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..config import as_cfg
from ._base import HipPoseModule
from ._tree import build_backbone


class PoseHighResolutionNet(HipPoseModule):
    def __init__(self, cfg, **kwargs):
        super().__init__()
        self.cfg = cfg = as_cfg(cfg)
        extra = cfg.MODEL.EXTRA
        chans = build_backbone(self, extra)
        k = extra.FINAL_CONV_KERNEL
        self.final_layer = nn.Conv2d(chans[0], cfg.MODEL.NUM_JOINTS, k, 1, 1 if k == 3 else 0)
        self.pretrained_layers = extra.PRETRAINED_LAYERS

    def _build_plan(self, batch, device):
        from ..plan import HRNetPlan
        return HRNetPlan(self, batch, device, self.compute_dtype)

    supports_training = True

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            return self.train_forward(x)
        return self.run_plan(x)[1]


def get_pose_net(cfg, is_train, **kwargs):
    model = PoseHighResolutionNet(cfg, **kwargs)
    cfg = model.cfg
    if is_train and cfg.MODEL.INIT_WEIGHTS:
        model.init_weights(cfg.MODEL.PRETRAINED)
    return model
