"""On-device flip test: what lib/core/function.py:125-146 / :400-427 do per batch.

The reference runs a second forward on ``input.flip(3)``, copies the heatmaps to the host, reverses W and
swaps left/right joints in numpy (lib/utils/transforms.py:23-37), copies back, shifts by one pixel and
averages.  Here the original and the mirrored crops go through two forwards of batch B that OVERLAP on two
side streams (one plan per stream, models/_base.py; round 3: 4.0 -> 3.6 ms for 64 CrowdPose crops against
one forward of batch 2B) and the merge is a single kernel (``rsg_flip_merge_nchw_f32``).
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import hip


def flip_permutation(num_joints: int, flip_pairs: Sequence[Sequence[int]]):
    perm = list(range(num_joints))
    for a, b in flip_pairs:
        perm[a], perm[b] = perm[b], perm[a]
    return perm


def flip_merge(h: torch.Tensor, h_flipped: torch.Tensor, flip_pairs, shift_heatmap: bool = True) -> torch.Tensor:
    """0.5 * (h + shift(flip_back(h_flipped))) for fp32 NCHW CUDA tensors."""
    if h.device.type != "cuda" or h.dtype != torch.float32 or h.shape != h_flipped.shape or h.dim() != 4:
        raise ValueError("flip_merge expects two fp32 NCHW CUDA tensors of the same shape")
    h, h_flipped = h.contiguous(), h_flipped.contiguous()
    B, J, H, W = h.shape
    perm = (C.c_int32 * J)(*flip_permutation(J, flip_pairs))
    out = torch.empty_like(h)
    st = torch.cuda.current_stream(h.device).cuda_stream
    hip.check(hip.lib().rsg_flip_merge_nchw_f32(h.data_ptr(), h_flipped.data_ptr(), perm, out.data_ptr(), B, J, H, W,
                                                int(bool(shift_heatmap)), st), "flip_merge")
    return out


_SIDE = {}          # device index -> the two side streams of forward_flip_test


def forward_flip_test(model, x: torch.Tensor, flip_pairs, shift_heatmap: bool = True, prepared: bool = False,
                      overlap: bool = True) -> torch.Tensor:
    """Flip-tested joint heatmaps [B,J,h,w]: the forward of x and the forward of x mirrored, then the fused merge.
    prepared: ``x`` already is the 2B batch [x ; x mirrored] (``CropProducer(..., with_flip=True)`` writes both halves
    in its one pass).  overlap=False: ONE forward of batch 2B on the caller's stream instead of two of batch B beside
    each other (same bits either way: a crop's result does not depend on its batch).

    For ``pose_rsgnet`` the merged map is ``kpt_scores`` (item 1 of the 4-tuple), as in
    ``rsgnet_validate`` (lib/core/function.py:389-427)."""
    B = x.shape[0] // 2 if prepared else x.shape[0]
    if not overlap or x.device.type != "cuda":
        out = model(x if prepared else torch.cat((x, x.flip(3)), dim=0))
        heat = out if torch.is_tensor(out) else out[1]
        return flip_merge(heat[:B], heat[B:], flip_pairs, shift_heatmap)
    cur = torch.cuda.current_stream(x.device)
    if x.device.index not in _SIDE:
        _SIDE[x.device.index] = (torch.cuda.Stream(x.device), torch.cuda.Stream(x.device))
    heats = []
    for k, st in enumerate(_SIDE[x.device.index]):
        st.wait_stream(cur)                              # x is final where the caller's stream stands
        with torch.cuda.stream(st):
            xi = (x[:B], x[B:])[k] if prepared else (x if k == 0 else x.flip(3))
            out = model(xi.contiguous())
            heat = out if torch.is_tensor(out) else out[1]
        heat.record_stream(cur)                          # allocated on the side stream, consumed on the caller's
        heats.append(heat)
    for st in _SIDE[x.device.index]:
        cur.wait_stream(st)
    return flip_merge(heats[0], heats[1], flip_pairs, shift_heatmap)
