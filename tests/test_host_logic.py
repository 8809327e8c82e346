"""Host-side logic that runs without a GPU: the weight packer's fragment order against the MFMA
operand/accumulator maps the kernel assumes, the auto-tiler, and the plan-time weight folds."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import pose_oracle as po
from rsgnet_amd import hip, plan_rsgnet
from tests.util import load_case, maxabs


def _pack(w, scale=None, shift=None, bias=None, dtype=0):
    lib = hip.lib()
    O, I, kh, kw = w.shape
    wp = np.zeros(lib.rsg_conv_packed_bytes(I, O, kh, kw), np.uint8)
    bo = np.zeros((O + 31) // 32 * 32, np.float32)
    ptr = lambda a: None if a is None else a.ctypes.data
    rc = lib.rsg_pack_conv_weights_host(w.ctypes.data, ptr(scale), ptr(shift), ptr(bias), I, O, kh, kw, dtype,
                                        wp.ctypes.data, bo.ctypes.data)
    assert rc == 0, lib.rsg_last_error()
    return wp, bo


@pytest.mark.parametrize("cin,cout,k", [(32, 32, 3), (3, 64, 3), (17, 32, 3), (64, 128, 3), (96, 96, 3),
                                        (256, 64, 1), (32, 17, 1), (48, 48, 3), (32, 32, 2)])
def test_packer_matches_kernel_mfma_convention(cin, cout, k):
    """Undo the packing with the lane maps of v_mfma_f32_32x32x16 (A: row=lane&31, k=8*(lane>>5)+j;
    D: col=lane&31, row=(r&3)+8*(r>>2)+4*(lane>>5)) and the kernel's epilogue rule (lane half h,
    register r -> channel 32*ct + 16*h + r): the folded weights must come back."""
    lib = hip.lib()
    rs = np.random.RandomState(cin * 1000 + cout)
    w = rs.standard_normal((cout, cin, k, k)).astype(np.float32)
    scale = rs.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = rs.standard_normal(cout).astype(np.float32)
    cb = rs.standard_normal(cout).astype(np.float32)
    wp, bo = _pack(w, scale, shift, cb)
    cin_pad = lib.rsg_conv_padded_cin(cin)
    cinb = 64 if cin_pad % 64 == 0 else (32 if cin_pad % 32 == 0 else (48 if cin_pad == 48 else 16))   # csrc/conv_igemm.hip, pick_cinb
    ntaps, kpt, nch = k * k, cinb // 16, cin_pad // cinb
    KS = nch * ntaps * kpt
    ntile = (cout + 31) // 32
    arr = wp.view(np.float16).reshape(ntile, KS, 64, 8).astype(np.float32)
    want = np.zeros((ntile * 32, cin_pad, ntaps), np.float32)
    want[:cout, :cin] = (w * scale[:, None, None, None]).astype(np.float16).astype(np.float32).reshape(cout, cin, ntaps)
    got = np.zeros_like(want)
    row_to_ch = {}
    for h in range(2):
        for r in range(16):
            row_to_ch[(r & 3) + 8 * (r >> 2) + 4 * h] = 16 * h + r
    for ct in range(ntile):
        for ch in range(nch):
            for t in range(ntaps):
                for c in range(kpt):
                    ks = (ch * ntaps + t) * kpt + c
                    for lane in range(64):
                        i, h = lane & 31, lane >> 5
                        for j in range(8):
                            got[ct * 32 + row_to_ch[i], ch * cinb + 16 * c + 8 * h + j, t] = arr[ct, ks, lane, j]
    assert np.array_equal(got, want)
    assert np.allclose(bo[:cout], cb * scale + shift, atol=1e-6) and not bo[cout:].any()


def test_auto_tiler_picks_valid_tiles():
    lib = hip.lib()
    shapes = [  # cin, cout, k, stride, out_h, out_w  (HRNet-W32 256x192 and W48 384x288 maps)
        (16, 64, 3, 2, 128, 96), (64, 64, 3, 2, 64, 48), (64, 64, 1, 1, 64, 48), (64, 256, 1, 1, 64, 48),
        (256, 64, 1, 1, 64, 48), (256, 32, 3, 1, 64, 48), (256, 64, 3, 2, 32, 24), (32, 32, 3, 1, 64, 48),
        (64, 64, 3, 1, 32, 24), (128, 128, 3, 1, 16, 12), (256, 256, 3, 1, 8, 6), (128, 32, 1, 1, 16, 12),
        (32, 256, 3, 2, 8, 6), (96, 96, 3, 1, 64, 48), (32, 32, 3, 1, 128, 96), (32, 17, 1, 1, 128, 96),
        (48, 48, 3, 1, 96, 72), (384, 384, 3, 1, 12, 9), (32, 32, 2, 1, 64, 48),
    ]
    for cin, cout, k, s, oh, ow in shapes:
        d = hip.ConvDesc()
        d.batch, d.in_c, d.in_ld, d.out_c, d.out_ld = 4, cin, cin, cout, (cout + 31) // 32 * 32
        d.kh = d.kw = k
        d.stride, d.out_h, d.out_w = s, oh, ow
        d.in_h, d.in_w = oh * s, ow * s
        info = (C.c_int32 * 8)()
        assert lib.rsg_conv_plan(C.byref(d), info) == 0, lib.rsg_last_error()
        cinb, wm, wn, nt, th, tw, lds, grid = list(info)
        assert cin % cinb == 0 and wm * wn == 4 and th * tw <= 32 * nt * wn and lds <= 64 * 1024
        assert ((cout + 31) // 32 * 32) % (32 * wm) == 0 and grid >= 4


def test_plan_time_folds_match_oracle():
    cfg, model, sd, x, gold = load_case("rsgnet_w32_256x192_coco")
    ctx = po.Ctx(sd)
    # SGM limb classifier (KTMachine) and the constant location clue
    assert maxabs(plan_rsgnet.limb_classifier(model), gold["limb_weights"]) < 1e-6
    loc = po.conv_unit(ctx, ctx.p("loc_features"), "loc_conv.0", "loc_conv.1", relu=True)
    assert maxabs(plan_rsgnet.location_features(model), loc) < 1e-5
    # type path: folded 3x3 J->C0 conv == matmul with the [J,600] templates followed by the 600->C0 conv
    scores = torch.randn(1, 17, 16, 12)
    tm = po.type_templates(ctx)
    tmap = (scores.reshape(1, 17, -1).permute(0, 2, 1) @ tm).permute(0, 2, 1).reshape(1, 600, 16, 12)
    want = F.conv2d(tmap, sd["type_conv.0.weight"], padding=1)
    got = F.conv2d(scores, plan_rsgnet.folded_type_conv_weight(model), padding=1)
    assert maxabs(got, want) < 2e-4 * float(want.abs().max())


def test_deconv_phase_decomposition():
    wt = torch.randn(8, 6, 4, 4)
    x = torch.randn(2, 8, 5, 7)
    want = F.conv_transpose2d(x, wt, stride=2, padding=1)
    got = torch.zeros_like(want)
    for py, px, pad_y, pad_x, w in plan_rsgnet.deconv_phase_weights(wt):
        xp = F.pad(x, (pad_x, 1 - pad_x, pad_y, 1 - pad_y))
        got[:, :, py::2, px::2] = F.conv2d(xp, w)
    assert maxabs(got, want) < 1e-5
