"""Crop producer and OKS-NMS: oracle pins, host mirrors, and the HIP warp against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import crop_oracle as co
from rsgnet_amd import crops, nms
from tests.util import GOLD


def _nms_cases():
    g = np.load(os.path.join(GOLD, "oks_nms.npz"))
    for c in range(int(g["n_cases"])):
        sig = g[f"c{c}_sigmas"]
        vis = float(g[f"c{c}_vis"])
        yield (g[f"c{c}_kpts"], g[f"c{c}_scores"], g[f"c{c}_areas"], float(g[f"c{c}_thresh"]),
               None if sig.size == 0 else sig, None if np.isnan(vis) else vis,
               list(g[f"c{c}_keep"]), list(g[f"c{c}_soft"]), g[f"c{c}_iou0"])


def test_nms_oracle_matches_reference_goldens():
    """oracle/crop_oracle.py against outputs of the reference's lib/nms/nms.py (oracle/make_crop_goldens.py)."""
    for k, s, a, thr, sig, vis, keep, soft, iou0 in _nms_cases():
        assert co.oks_nms(k, s, a, thr, sig, vis) == keep
        assert co.soft_oks_nms(k, s, a, thr, sig, vis) == soft
        assert np.array_equal(co.oks_iou(k[0].flatten(), k.reshape(len(s), -1), a[0], a, sig, vis), iou0)


def test_nms_host_mirror_matches_reference_goldens():
    for k, s, a, thr, sig, vis, keep, soft, iou0 in _nms_cases():
        db = [{"score": s[i], "keypoints": k[i], "area": a[i]} for i in range(len(s))]
        assert list(nms.oks_nms(db, thr, sig, vis)) == keep
        assert list(nms.soft_oks_nms(db, thr, sig, vis)) == soft
        np.testing.assert_allclose(nms.oks_iou(k[0].flatten(), k.reshape(len(s), -1), a[0], a, sig, vis), iou0,
                                   rtol=1e-13)
    assert nms.oks_nms([], 0.9) == [] and nms.soft_oks_nms([], 0.9) == []          # nms.py:106-107


def test_rescoring_matches_oracle():
    rs = np.random.RandomState(5)
    k = np.concatenate([rs.uniform(0, 300, (9, 17, 2)), rs.uniform(0, 1, (9, 17, 1))], axis=2)
    box = rs.uniform(0.2, 1, 9)
    want = co.rescore(k, box, 0.2)
    db = [{"score": box[i], "keypoints": k[i], "area": 5000.0 + i} for i in range(9)]
    kept = nms.rescore_and_nms(db, 17, 0.2, 0.9)
    np.testing.assert_allclose([p["score"] for p in db], want, rtol=1e-14)
    assert [id(p) for p in kept] == [id(db[i]) for i in co.oks_nms(k, want, [5000.0 + i for i in range(9)], 0.9)]


def _smooth_picture(h, w, seed):
    rs = np.random.RandomState(seed)
    ys, xs = np.mgrid[0:h, 0:w]
    ch = [127 + 120 * np.sin(xs / rs.uniform(20, 50) + rs.uniform(0, 6)) * np.cos(ys / rs.uniform(20, 50)) for _ in range(3)]
    return np.clip(np.stack(ch, axis=2), 0, 255).astype(np.uint8)


def test_warp_oracle_properties():
    """The restated cv2.warpAffine (parity unpinned: cv2 is absent) on cases with known answers."""
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, size=(120, 160, 3)).astype(np.uint8)
    ident = np.array([[1, 0, 0], [0, 1, 0]], float)
    assert np.array_equal(co.warp_affine_u8(img, ident, 160, 120), img)
    shift = np.array([[1, 0, -7], [0, 1, 5]], float)                     # crop(x, y) = img(x + 7, y - 5)
    want = np.zeros((100, 90, 3), np.uint8)
    want[5:, :] = img[0:95, 7:97]
    assert np.array_equal(co.warp_affine_u8(img, shift, 90, 100), want)
    mirror = np.array([[-1, 0, 159], [0, 1, 0]], float)
    assert np.array_equal(co.warp_affine_u8(img, mirror, 160, 120), img[:, ::-1])
    half = np.array([[1, 0, -0.5], [0, 1, 0]], float)                     # midpoints: (a + b + 1) >> 1 rounding of the fixed point
    got = co.warp_affine_u8(img, half, 100, 120).astype(int)
    a, b = img[:, 0:100].astype(int), img[:, 1:101].astype(int)
    assert np.array_equal(got, (a * 16384 + b * 16384 + 16384) >> 15)
    # a smooth picture under rotation + scale: within 2 grey levels of real-valued bilinear sampling
    pic = _smooth_picture(300, 400, 1)
    m = co.get_affine_transform(np.array([200, 150], np.float32), np.array([1.0, 1.33], np.float32), 20, (192, 256))
    got = co.warp_affine_u8(pic, m, 192, 256).astype(float)
    mi = np.linalg.inv(np.vstack([m, [0, 0, 1]]))[:2]
    ys, xs = np.mgrid[0:256, 0:192]
    sx, sy = mi[0, 0] * xs + mi[0, 1] * ys + mi[0, 2], mi[1, 0] * xs + mi[1, 1] * ys + mi[1, 2]
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]

    def tap(y, x):
        ok = (y >= 0) & (y < 300) & (x >= 0) & (x < 400)
        return pic[np.clip(y, 0, 299), np.clip(x, 0, 399)].astype(float) * ok[..., None]
    want = tap(y0, x0) * (1 - fx) * (1 - fy) + tap(y0, x0 + 1) * fx * (1 - fy) + tap(y0 + 1, x0) * (1 - fx) * fy \
        + tap(y0 + 1, x0 + 1) * fx * fy
    inside = (x0 >= 1) & (x0 < 398) & (y0 >= 1) & (y0 < 298)
    assert np.abs(got - want)[inside].max() <= 2.0


def test_affine_transform_mirror_matches_oracle():
    for c, s, r in [((100.5, 80.2), (1.2, 1.6), 0), ((320, 240), (2.0, 2.67), 30), ((10, 400), (0.5, 0.66), -47.5)]:
        for inv in (0, 1):
            a = co.get_affine_transform(np.array(c, np.float32), np.array(s, np.float32), r, (192, 256), inv=bool(inv))
            b = crops.get_affine_transform(np.array(c, np.float32), np.array(s, np.float32), r, (192, 256), inv=inv)
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
    # rot = 0: the map is the scaling that takes the 200*scale box onto the crop (transforms.py:64-101)
    m = crops.get_affine_transform(np.array([50, 60], np.float32), np.array([0.96, 1.28], np.float32), 0, (192, 256))
    np.testing.assert_allclose(m, [[1.0, 0, 96 - 50], [0, 1.0, 128 - 60]], atol=1e-5)
    np.testing.assert_allclose(crops.affine_transform((50, 60), m), (96, 128), atol=1e-5)
    # the batched form the producer uses is the scalar function, bit for bit
    rs = np.random.RandomState(1)
    c, s = rs.uniform(0, 500, (40, 2)).astype(np.float32), rs.uniform(0.2, 3, (40, 2)).astype(np.float32)
    r = rs.uniform(-80, 80, 40)
    r[::3] = 0
    batched = crops.affine_maps(c, s, r, (192, 256))
    for i in range(40):
        assert np.array_equal(batched[i], crops.get_affine_transform(c[i], s[i], float(r[i]), (192, 256)))
    for box in ([10, 20, 50, 120], [0, 0, 300, 100]):
        c0, s0 = co.xywh_to_center_scale(*box, aspect_ratio=0.75)
        c1, s1 = crops.box_to_center_scale(box, 0.75)
        assert np.array_equal(c0, c1) and np.array_equal(s0, s1)


@pytest.mark.gpu
@pytest.mark.parametrize("color_rgb", [True, False])
def test_crop_kernel_bit_exact(color_rgb):
    """rsg_crop_warp_affine against the oracle: ragged pictures, rotations, boxes hanging over every edge."""
    rs = np.random.RandomState(3)
    pics = [rs.randint(0, 256, size=s + (3,)).astype(np.uint8) for s in ((480, 640), (333, 501), (64, 48), (720, 1280))]
    dev = [torch.from_numpy(p).cuda() for p in pics]
    which, centers, scales, rots = [], [], [], []
    for i in range(24):
        k = i % len(pics)
        h, w = pics[k].shape[:2]
        which.append(k)
        centers.append((rs.uniform(-0.1 * w, 1.1 * w), rs.uniform(-0.1 * h, 1.1 * h)))
        sw = rs.uniform(0.1, 1.5) * w / 200
        scales.append((sw, sw / 0.75))
        rots.append(0.0 if i % 3 == 0 else rs.uniform(-80, 80))
    prod = crops.CropProducer((192, 256), color_rgb=color_rgb)
    got = prod(dev, which, centers, scales, rots).cpu().numpy()
    for i in range(24):
        # the same 2x3 map on both sides (the two solvers agree to 1e-12, which could still flip an exact .5 tie)
        m = crops.get_affine_transform(np.array(centers[i], np.float32), np.array(scales[i], np.float32), rots[i],
                                       (192, 256))
        src = pics[which[i]][:, :, ::-1] if color_rgb else pics[which[i]]
        want = co.to_tensor_normalize(co.warp_affine_u8(np.ascontiguousarray(src), m, 192, 256))
        assert np.array_equal(got[i], want), i


@pytest.mark.gpu
def test_crop_kernel_feeds_the_network_input_shape():
    pic = torch.from_numpy(_smooth_picture(480, 640, 2)).cuda()
    c, s = crops.box_to_center_scale([200, 100, 120, 300], 192 / 256)
    out = crops.CropProducer((288, 384))([pic], [0, 0], [c, c], [s, s])
    assert out.shape == (2, 3, 384, 288) and out.dtype == torch.float32
    assert torch.equal(out[0], out[1])
    both = crops.CropProducer((288, 384))([pic], [0, 0], [c, c], [s, s], with_flip=True)     # [crops ; crops.flip(3)]
    assert both.shape == (4, 3, 384, 288)
    assert torch.equal(both[:2], out) and torch.equal(both[2:], out.flip(3))
    with pytest.raises(ValueError):
        crops.CropProducer((192, 256))([pic.cpu()], [0], [c], [s])
