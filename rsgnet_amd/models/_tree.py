"""Parameter trees with the reference's state_dict names.

The torch layers created here are *containers only*: they give the model the reference's parameter
and buffer names/shapes (so reference checkpoints load and ``optim.Adam(model.parameters())`` works)
but their ``forward`` is never called — the compute goes through the HIP execution plan
(``rsgnet_amd/plan.py``).  Names follow /root/reference/lib/models/pose_hrnet.py:279-334 and
pose_rsgnet.py:605-777.
"""
from __future__ import annotations

from typing import List, Sequence

import torch.nn as nn

BN_MOMENTUM = 0.1  # pose_hrnet.py:19


def conv(cin: int, cout: int, k: int, stride: int = 1, bias: bool = False) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=bias)


def bn(c: int) -> nn.BatchNorm2d:
    return nn.BatchNorm2d(c, momentum=BN_MOMENTUM)


def conv_bn(cin: int, cout: int, k: int, stride: int = 1, relu: bool = True) -> nn.Sequential:
    """Sequential(conv, bn[, relu]) — children '0' and '1' carry the parameters."""
    mods: List[nn.Module] = [conv(cin, cout, k, stride), bn(cout)]
    if relu:
        mods.append(nn.ReLU(inplace=True))
    return nn.Sequential(*mods)


class Residual(nn.Module):
    """BasicBlock (kind 'BASIC', convs 3x3,3x3) or Bottleneck ('BOTTLENECK', 1x1,3x3,1x1 with x4)."""

    def __init__(self, kind: str, cin: int, planes: int, project: bool = False):
        super().__init__()
        if kind == "BASIC":
            shapes = [(cin, planes, 3), (planes, planes, 3)]
            cout = planes
        elif kind == "BOTTLENECK":
            shapes = [(cin, planes, 1), (planes, planes, 3), (planes, planes * 4, 1)]
            cout = planes * 4
        else:
            raise ValueError(f"unknown block type {kind!r}")
        for n, (a, b, k) in enumerate(shapes, 1):
            setattr(self, f"conv{n}", conv(a, b, k))
            setattr(self, f"bn{n}", bn(b))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = nn.Sequential(conv(cin, cout, 1), bn(cout)) if project else None
        self.out_channels = cout


EXPANSION = {"BASIC": 1, "BOTTLENECK": 4}


class HRModule(nn.Module):
    """Containers of one HighResolutionModule: ``branches`` and ``fuse_layers``."""

    def __init__(self, kind: str, num_blocks: Sequence[int], in_ch: List[int], ch: Sequence[int],
                 multi_scale_output: bool = True):
        super().__init__()
        nb = len(ch)
        if not (nb == len(num_blocks) == len(in_ch)):
            # same check and message family as pose_hrnet.py:122-140
            raise ValueError(f"NUM_BRANCHES({nb}) <> NUM_BLOCKS({len(num_blocks)}) / "
                             f"NUM_INCHANNELS({len(in_ch)})")
        exp = EXPANSION[kind]
        branches = []
        for b in range(nb):
            blocks = []
            for k in range(num_blocks[b]):
                blocks.append(Residual(kind, in_ch[b], ch[b], project=in_ch[b] != ch[b] * exp))
                in_ch[b] = ch[b] * exp
            branches.append(nn.Sequential(*blocks))
        self.branches = nn.ModuleList(branches)
        self.num_branches = nb
        self.num_out = nb if multi_scale_output else 1
        self.channels = list(in_ch)
        if nb == 1:
            self.fuse_layers = None
            return
        rows = []
        for i in range(self.num_out):
            row = []
            for j in range(nb):
                if j > i:    # 1x1 + BN, then nearest x2^(j-i)
                    row.append(nn.Sequential(conv(in_ch[j], in_ch[i], 1), nn.BatchNorm2d(in_ch[i]),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode="nearest")))
                elif j == i:
                    row.append(None)
                else:        # (i-j) strided 3x3 convs, the last one maps to in_ch[i] without ReLU
                    chain = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        co = in_ch[i] if last else in_ch[j]
                        mods = [conv(in_ch[j], co, 3, 2), nn.BatchNorm2d(co)]
                        if not last:
                            mods.append(nn.ReLU(True))
                        chain.append(nn.Sequential(*mods))
                    row.append(nn.Sequential(*chain))
            rows.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(rows)
        self.relu = nn.ReLU(True)


def make_transition(pre: Sequence[int], cur: Sequence[int]) -> nn.ModuleList:
    """pose_hrnet.py:336-375."""
    out = []
    for i, c in enumerate(cur):
        if i < len(pre):
            out.append(conv_bn(pre[i], c, 3) if c != pre[i] else None)
        else:
            steps = []
            for j in range(i + 1 - len(pre)):
                steps.append(conv_bn(pre[-1], c if j == i - len(pre) else pre[-1], 3, 2))
            out.append(nn.Sequential(*steps))
    return nn.ModuleList(out)


def make_stage(scfg, in_ch: List[int], multi_scale_output: bool = True):
    """pose_hrnet.py:396-426."""
    mods = []
    n = scfg["NUM_MODULES"]
    for m in range(n):
        mso = multi_scale_output or m != n - 1
        mod = HRModule(scfg["BLOCK"], scfg["NUM_BLOCKS"], in_ch, scfg["NUM_CHANNELS"], mso)
        in_ch = mod.channels
        mods.append(mod)
    return nn.Sequential(*mods), in_ch


def build_backbone(net: nn.Module, extra) -> List[int]:
    """Attach stem, layer1, transitions and stages 2-4 to ``net``; returns stage-4 channels."""
    net.conv1 = conv(3, 64, 3, 2)
    net.bn1 = bn(64)
    net.conv2 = conv(64, 64, 3, 2)
    net.bn2 = bn(64)
    net.relu = nn.ReLU(inplace=True)
    blocks, cin = [], 64
    for k in range(4):
        blocks.append(Residual("BOTTLENECK", cin, 64, project=(k == 0)))
        cin = 256
    net.layer1 = nn.Sequential(*blocks)
    pre = [256]
    for s in (2, 3, 4):
        scfg = extra[f"STAGE{s}"]
        exp = EXPANSION[scfg["BLOCK"]]
        ch = [c * exp for c in scfg["NUM_CHANNELS"]]
        setattr(net, f"stage{s}_cfg", scfg)
        setattr(net, f"transition{s - 1}", make_transition(pre, ch))
        stage, pre = make_stage(scfg, list(ch), multi_scale_output=(s != 4))
        setattr(net, f"stage{s}", stage)
    return pre
