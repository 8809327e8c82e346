"""CPU restatement of the crop producer and of OKS-NMS.  TEST INFRASTRUCTURE ONLY: imported by
tests/, ``__graft_entry__.smoke()`` and bench's cpu_baseline leg, never by the product path.

Crop producer — lib/dataset/CPJointsDataset.py:1281-1290 (JointsDataset.py does the same), tools/test.py:105-112:
``get_affine_transform`` (lib/utils/transforms.py:64-101) builds a 2x3 map from (center, scale, rot);
``cv2.warpAffine(img, trans, size, flags=cv2.INTER_LINEAR)`` resamples the uint8 picture; ToTensor + Normalize
give the fp32 NCHW input.

PARITY UNPINNED for the warp: the arithmetic lives in OpenCV (``cv2``), a third-party dependency the reference does
not vendor or pin (requirements.txt does not even list it) and which is absent from this image, and the reference
holds no fixtures for this path.  What follows restates OpenCV's published fixed-point algorithm (modules/imgproc/
src/imgwarp.cpp: ``WarpAffineInvoker`` + ``remapBilinear`` with ``FixedPtCast<int, uchar, INTER_REMAP_COEF_BITS>``,
INTER_BITS = 5, AB_BITS = 10, BORDER_CONSTANT value 0, the path OpenCV 3.x - 4.10 take for 8-bit pictures) and is
pinned by properties only (identity, integer shifts, flips, the float bilinear formula within one grey level).

OKS-NMS — lib/nms/nms.py:75-124,138-180 and the rescoring of lib/dataset/coco.py:322-357; pinned against the
reference's own functions by tests/golden/oks_nms.npz (oracle/make_crop_goldens.py).
"""
from __future__ import annotations

import numpy as np

IMAGENET_MEAN = (0.485, 0.456, 0.406)     # tools/test.py:105-107
IMAGENET_STD = (0.229, 0.224, 0.225)
COCO_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0


# ---- lib/utils/transforms.py:64-119 ----------------------------------------------------------------------
def _third_point(a, b):
    """transforms.py:110-112: b + perp(a - b), in float32."""
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def affine_from_triangles(src, dst):
    """cv2.getAffineTransform: the 2x3 map taking three points onto three points, solved in float64."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    A = np.zeros((6, 6))
    rhs = np.zeros(6)
    for i in range(3):
        A[i, 0:2], A[i, 2] = src[i], 1.0
        A[i + 3, 3:5], A[i + 3, 5] = src[i], 1.0
        rhs[i], rhs[i + 3] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(A, rhs).reshape(2, 3)


def get_affine_transform(center, scale, rot, output_size, shift=(0.0, 0.0), inv=False):
    """transforms.py:64-101.  The three point pairs are float32 arrays there; so are they here."""
    scale = np.asarray(scale, dtype=np.float32)
    if scale.ndim == 0:
        scale = np.array([scale, scale], dtype=np.float32)
    shift = np.asarray(shift, dtype=np.float32)
    center = np.asarray(center)
    scale_px = scale * 200.0
    src_w, dst_w, dst_h = scale_px[0], output_size[0], output_size[1]
    rad = np.pi * rot / 180
    sn, cs = np.sin(rad), np.cos(rad)
    p = [0, src_w * -0.5]
    src_dir = [p[0] * cs - p[1] * sn, p[0] * sn + p[1] * cs]                  # transforms.py:115-119
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_px * shift
    src[1, :] = center + src_dir + scale_px * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    src[2, :] = _third_point(src[0, :], src[1, :])
    dst[2, :] = _third_point(dst[0, :], dst[1, :])
    return affine_from_triangles(dst, src) if inv else affine_from_triangles(src, dst)


def affine_transform(pt, t):
    """transforms.py:104-107."""
    return (np.asarray(t, np.float64) @ np.array([pt[0], pt[1], 1.0]))[:2]


def xywh_to_center_scale(x, y, w, h, aspect_ratio, pixel_std=200):
    """lib/dataset/coco.py:215-230 (``_xywh2cs``)."""
    center = np.zeros(2, dtype=np.float32)
    center[0] = x + w * 0.5
    center[1] = y + h * 0.5
    if w > aspect_ratio * h:
        h = w * 1.0 / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    scale = np.array([w * 1.0 / pixel_std, h * 1.0 / pixel_std], dtype=np.float32)
    if center[0] != -1:
        scale = scale * 1.25
    return center, scale


# ---- cv2.warpAffine(..., flags=INTER_LINEAR), 8-bit, BORDER_CONSTANT 0 --------------------------------------
def _sat_rint(v):
    return np.clip(np.rint(v), -2147483648.0, 2147483647.0).astype(np.int64)


def invert_affine(m):
    """The inversion cv::warpAffine performs on the forward map (double, this operation order)."""
    m = np.asarray(m, np.float64).reshape(6).copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def warp_affine_u8(img, trans, out_w, out_h):
    """img: uint8 [H,W,3]; trans: forward 2x3 map; returns uint8 [out_h,out_w,3]."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    m = invert_affine(trans)
    x = np.arange(out_w, dtype=np.float64)
    y = np.arange(out_h, dtype=np.float64)
    adelta = _sat_rint(m[0] * x * 1024.0)
    bdelta = _sat_rint(m[3] * x * 1024.0)
    X0 = _sat_rint((m[1] * y + m[2]) * 1024.0) + 16
    Y0 = _sat_rint((m[4] * y + m[5]) * 1024.0) + 16

    def wrap32(v):                      # int arithmetic of the C code
        return ((v + 2 ** 31) % 2 ** 32) - 2 ** 31
    X = wrap32(X0[:, None] + adelta[None, :]) >> 5
    Y = wrap32(Y0[:, None] + bdelta[None, :]) >> 5
    sx = np.clip(X >> 5, -32768, 32767)
    sy = np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w = [np.minimum((32 - fx) * (32 - fy) * 32, 32767), fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32]
    acc = np.zeros((out_h, out_w, 3), dtype=np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        tap = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64) * ok[..., None]
        acc += tap * w[k][..., None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def to_tensor_normalize(crop_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """transforms.ToTensor + transforms.Normalize (tools/test.py:105-112): uint8 HWC -> fp32 CHW."""
    t = crop_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    mean = np.asarray(mean, np.float32)[:, None, None]
    std = np.asarray(std, np.float32)[:, None, None]
    return ((t - mean) / std).astype(np.float32)


def make_crop(img_bgr, center, scale, rot, image_size, color_rgb=True, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """One dataset item's input tensor (eval path: JointsDataset.__getitem__ without augmentation)."""
    img = img_bgr[:, :, ::-1] if color_rgb else img_bgr          # cv2.COLOR_BGR2RGB
    trans = get_affine_transform(center, scale, rot, image_size)
    return to_tensor_normalize(warp_affine_u8(np.ascontiguousarray(img), trans, int(image_size[0]),
                                              int(image_size[1])), mean, std)


# ---- lib/nms/nms.py:75-124, 138-180 and coco.py:322-357 ------------------------------------------------------
def oks_iou(g, d, a_g, a_d, sigmas=None, in_vis_thre=None):
    """nms.py:75-96.  g: [3J]; d: [n,3J].  With in_vis_thre the reference's ``list(vg > t) and list(vd > t)``
    evaluates to the second list: only the candidate's visibilities select joints."""
    sig = COCO_SIGMAS if sigmas is None else np.asarray(sigmas)
    var = (sig * 2) ** 2
    xg, yg = g[0::3], g[1::3]
    out = np.zeros(d.shape[0])
    for n in range(d.shape[0]):
        dx, dy = d[n, 0::3] - xg, d[n, 1::3] - yg
        e = (dx ** 2 + dy ** 2) / var / ((a_g + a_d[n]) / 2 + np.spacing(1)) / 2
        if in_vis_thre is not None:
            e = e[d[n, 2::3] > in_vis_thre]
        out[n] = np.sum(np.exp(-e)) / e.shape[0] if e.shape[0] != 0 else 0.0
    return out


def oks_nms(kpts, scores, areas, thresh, sigmas=None, in_vis_thre=None):
    """nms.py:99-124 on arrays: kpts [n,J,3], scores [n], areas [n] -> kept indices, best first."""
    n = len(scores)
    if n == 0:
        return []
    flat = np.asarray(kpts, np.float64).reshape(n, -1)
    order = np.asarray(scores).argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        ovr = oks_iou(flat[i], flat[order[1:]], areas[i], np.asarray(areas)[order[1:]], sigmas, in_vis_thre)
        order = order[np.where(ovr <= thresh)[0] + 1]
    return keep


def soft_oks_nms(kpts, scores, areas, thresh, sigmas=None, in_vis_thre=None, max_dets=20):
    """nms.py:138-180 (gaussian rescoring, nms.py:127-135)."""
    n = len(scores)
    if n == 0:
        return []
    flat = np.asarray(kpts, np.float64).reshape(n, -1)
    areas = np.asarray(areas)
    order = np.asarray(scores).argsort()[::-1]
    sc = np.asarray(scores, np.float64)[order]
    keep = []
    while order.size > 0 and len(keep) < max_dets:
        i = order[0]
        ovr = oks_iou(flat[i], flat[order[1:]], areas[i], areas[order[1:]], sigmas, in_vis_thre)
        order = order[1:]
        sc = sc[1:] * np.exp(-ovr ** 2 / thresh)
        by = sc.argsort()[::-1]
        order, sc = order[by], sc[by]
        keep.append(int(i))
    return keep


def rescore(kpts, box_scores, in_vis_thre):
    """coco.py:329-342: box score x mean confidence of the joints above in_vis_thre."""
    conf = np.asarray(kpts, np.float64)[:, :, 2]
    out = np.zeros(len(box_scores))
    for p in range(len(box_scores)):
        total, cnt = 0, 0
        for j in range(conf.shape[1]):
            if conf[p, j] > in_vis_thre:
                total, cnt = total + conf[p, j], cnt + 1
        if cnt != 0:
            total = total / cnt
        out[p] = total * box_scores[p]
    return out
