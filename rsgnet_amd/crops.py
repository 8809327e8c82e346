"""Crop producer on the device: what ``JointsDataset.__getitem__`` does per person box on the
CPU — ``get_affine_transform`` + ``cv2.warpAffine(..., INTER_LINEAR)`` + ToTensor + Normalize
(lib/dataset/CPJointsDataset.py:1281-1290, lib/utils/transforms.py:64-101, tools/test.py:105-112) — as one launch
per batch over pictures already resident in HBM, writing the fp32 NCHW tensor ``RSGNet.forward`` takes.

Host side: the 2x3 maps (a few flops per box, float64); device side: ``rsg_crop_warp_affine``.  No CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import hip

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _solve_triangle_map(p, q):
    """2x3 map with map(p_i) = q_i for three points (cv2.getAffineTransform), by Cramer's rule in float64."""
    p = np.asarray(p, np.float64)
    q = np.asarray(q, np.float64)
    (x0, y0), (x1, y1), (x2, y2) = p
    det = x0 * (y1 - y2) - y0 * (x1 - x2) + (x1 * y2 - x2 * y1)
    if det == 0:
        raise ValueError("get_affine_transform: degenerate box (zero scale)")
    out = np.empty((2, 3))
    for r in range(2):
        u0, u1, u2 = q[:, r]
        out[r, 0] = (u0 * (y1 - y2) - y0 * (u1 - u2) + (u1 * y2 - u2 * y1)) / det
        out[r, 1] = (x0 * (u1 - u2) - u0 * (x1 - x2) + (x1 * u2 - x2 * u1)) / det
        out[r, 2] = (x0 * (y1 * u2 - y2 * u1) - y0 * (x1 * u2 - x2 * u1) + u0 * (x1 * y2 - x2 * y1)) / det
    return out


def get_affine_transform(center, scale, rot, output_size, shift=(0.0, 0.0), inv=0):
    """lib/utils/transforms.py:64-101: same arguments, same float32 roundings of the two point triples."""
    scale = np.asarray(scale, dtype=np.float32)
    if scale.ndim == 0:
        scale = np.array([scale, scale], dtype=np.float32)
    f32 = np.float32
    span = scale * f32(200.0)
    off = span * np.asarray(shift, dtype=np.float32)
    ow, oh = output_size[0], output_size[1]
    a = np.pi * rot / 180
    half = span[0] * -0.5
    direction = [0 * np.cos(a) - half * np.sin(a), 0 * np.sin(a) + half * np.cos(a)]
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0] = np.asarray(center) + off
    src[1] = np.asarray(center) + direction + off
    dst[0] = [ow * 0.5, oh * 0.5]
    dst[1] = np.array([ow * 0.5, oh * 0.5]) + np.array([0, ow * -0.5], np.float32)
    for t in (src, dst):
        d = t[0] - t[1]
        t[2] = t[1] + np.array([-d[1], d[0]], dtype=np.float32)
    return _solve_triangle_map(dst, src) if inv else _solve_triangle_map(src, dst)


def affine_transform(pt, t):
    """lib/utils/transforms.py:104-107."""
    t = np.asarray(t, np.float64)
    return np.array([t[0, 0] * pt[0] + t[0, 1] * pt[1] + t[0, 2], t[1, 0] * pt[0] + t[1, 1] * pt[1] + t[1, 2]])


def box_to_center_scale(box, aspect_ratio, pixel_std=200):
    """lib/dataset/coco.py:211-230 (``_box2cs``): [x, y, w, h] -> (center, scale) with the 1.25 margin."""
    x, y, w, h = [float(v) for v in box[:4]]
    center = np.array([x + w * 0.5, y + h * 0.5], dtype=np.float32)
    if w > aspect_ratio * h:
        h = w * 1.0 / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    scale = np.array([w * 1.0 / pixel_std, h * 1.0 / pixel_std], dtype=np.float32)
    if center[0] != -1:
        scale = scale * 1.25
    return center, scale


_DESC_DTYPE = np.dtype([("image", np.uint64), ("height", np.int32), ("width", np.int32), ("stride", np.int32),
                        ("reserved", np.int32), ("m", np.float64, (6,))])      # struct rsg_crop_desc
assert _DESC_DTYPE.itemsize == C.sizeof(hip.CropDesc)


def affine_maps(centers, scales, rots, output_size) -> np.ndarray:
    """``get_affine_transform`` for a batch of boxes at once: [n,2,3] float64.  Same float32 roundings of the point
    triples as the scalar function (which the tests compare it with)."""
    c = np.asarray(centers, np.float32).reshape(-1, 2)
    sc = np.asarray(scales, np.float32).reshape(-1, 2)
    n = len(c)
    a = np.pi * (np.zeros(n) if rots is None else np.asarray(rots, np.float64)) / 180
    ow, oh = output_size[0], output_size[1]
    half = (sc[:, 0] * np.float32(200.0)) * -0.5                      # float32 * python float, as the reference
    direction = np.stack([0 * np.cos(a) - half * np.sin(a), 0 * np.sin(a) + half * np.cos(a)], axis=1)     # float64
    src = np.zeros((n, 3, 2), np.float32)
    dst = np.zeros((n, 3, 2), np.float32)
    src[:, 0] = c
    src[:, 1] = c + direction
    dst[:, 0] = [ow * 0.5, oh * 0.5]
    dst[:, 1] = np.array([ow * 0.5, oh * 0.5]) + np.array([0, ow * -0.5], np.float32)
    for t in (src, dst):
        d = t[:, 0] - t[:, 1]
        t[:, 2] = t[:, 1] + np.stack([-d[:, 1], d[:, 0]], axis=1)
    p, q = src.astype(np.float64), dst.astype(np.float64)
    x0, y0, x1, y1, x2, y2 = p[:, 0, 0], p[:, 0, 1], p[:, 1, 0], p[:, 1, 1], p[:, 2, 0], p[:, 2, 1]
    det = x0 * (y1 - y2) - y0 * (x1 - x2) + (x1 * y2 - x2 * y1)
    if np.any(det == 0):
        raise ValueError("get_affine_transform: degenerate box (zero scale)")
    out = np.empty((n, 2, 3))
    for r in range(2):
        u0, u1, u2 = q[:, 0, r], q[:, 1, r], q[:, 2, r]
        out[:, r, 0] = (u0 * (y1 - y2) - y0 * (u1 - u2) + (u1 * y2 - u2 * y1)) / det
        out[:, r, 1] = (x0 * (u1 - u2) - u0 * (x1 - x2) + (x1 * u2 - x2 * u1)) / det
        out[:, r, 2] = (x0 * (y1 * u2 - y2 * u1) - y0 * (x1 * u2 - x2 * u1) + u0 * (x1 * y2 - x2 * y1)) / det
    return out


class CropProducer:
    """Batches of network inputs from uint8 pictures in device memory.

    pictures: uint8 CUDA tensors [H,W,3] in the channel order ``cv2.imread`` gives (BGR); ``color_rgb``
    (cfg.DATASET.COLOR_RGB) reverses it on the way, as the datasets do with ``cv2.cvtColor``."""

    def __init__(self, image_size: Sequence[int], mean=IMAGENET_MEAN, std=IMAGENET_STD, color_rgb: bool = True):
        self.out_w, self.out_h = int(image_size[0]), int(image_size[1])      # cfg.MODEL.IMAGE_SIZE = [w, h]
        self._mean = (C.c_float * 3)(*mean)
        self._std = (C.c_float * 3)(*std)
        self.color_rgb = bool(color_rgb)

    def __call__(self, pictures: Sequence[torch.Tensor], picture_of_box: Sequence[int], centers, scales,
                 rots: Optional[Sequence[float]] = None, out: Optional[torch.Tensor] = None,
                 with_flip: bool = False) -> torch.Tensor:
        """Returns the input batch [n,3,H,W]; with_flip: [2n,3,H,W] = [crops ; crops mirrored left-right], the batch the
        flip test forwards in one go (rsgnet_amd.flip.forward_flip_test(..., prepared=True))."""
        n = len(picture_of_box)
        if n == 0:
            raise ValueError("CropProducer: empty batch")
        centers = np.asarray(centers, np.float32).reshape(n, 2)
        scales = np.asarray(scales, np.float32).reshape(n, 2)
        dev = pictures[0].device
        maps = affine_maps(centers, scales, rots, (self.out_w, self.out_h))
        rec = np.zeros(n, dtype=_DESC_DTYPE)
        for i, k in enumerate(picture_of_box):
            p = pictures[k]
            if p.device != dev or p.device.type != "cuda" or p.dtype != torch.uint8 or p.dim() != 3 \
                    or p.shape[2] != 3 or p.stride(2) != 1 or p.stride(1) != 3:
                raise ValueError("CropProducer expects uint8 CUDA pictures [H,W,3] with interleaved channels")
            rec[i] = (p.data_ptr(), p.shape[0], p.shape[1], p.stride(0), 0, 0)
        rec["m"] = maps.reshape(n, 6)
        desc = rec
        # pinned staging + asynchronous copy: the host never waits for the stream (torch's host allocator keeps the
        # pinned block alive until the copy has run)
        staged = torch.empty(desc.nbytes, dtype=torch.uint8, pin_memory=True)
        staged.numpy()[:] = desc.view(np.uint8)
        table = staged.to(dev, non_blocking=True)
        rows = 2 * n if with_flip else n
        if out is None:
            out = torch.empty(rows, 3, self.out_h, self.out_w, dtype=torch.float32, device=dev)
        elif out.shape != (rows, 3, self.out_h, self.out_w) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("CropProducer: out must be a contiguous fp32 [n,3,H,W] tensor ([2n,3,H,W] with_flip)")
        st = torch.cuda.current_stream(dev).cuda_stream
        hip.check(hip.lib().rsg_crop_warp_affine(table.data_ptr(), out.data_ptr(), out[n:].data_ptr() if with_flip else None,
                                                 n, self.out_h, self.out_w,
                                                 self._mean, self._std, int(self.color_rgb), st), "crop_warp_affine")
        table.record_stream(torch.cuda.current_stream(dev))
        return out
