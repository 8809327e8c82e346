"""GPU parity of every HIP kernel family against torch-CPU fp32 restatements, through the C ABI."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from rsgnet_amd import hip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def run_conv(x_nchw, w, scale=None, shift=None, bias=None, *, stride=1, relu=False, residual=None,
             want_f32=False, tile_hint=0, in_pad_extra=0, dtype="fp16"):
    """x_nchw fp32 [B,Cin,H,W] (CPU) -> (out16 as fp32 NCHW, out32) computed by rsg_conv_igemm."""
    lib = hip.lib()
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    B, Cin, H, W = x_nchw.shape
    O, _, kh, kw = w.shape
    cin_pad = lib.rsg_conv_padded_cin(Cin)
    ld = cin_pad + in_pad_extra
    xin = torch.zeros(B, H, W, ld, dtype=tdt)
    xin[..., :Cin] = x_nchw.permute(0, 2, 3, 1).to(tdt)
    xin = xin.to(DEV)
    wp = np.zeros(lib.rsg_conv_packed_bytes(Cin, O, kh, kw), np.uint8)
    cpad = (O + 31) // 32 * 32
    bo = np.zeros(cpad, np.float32)
    ptr = lambda a: None if a is None else a.ctypes.data
    wn = np.ascontiguousarray(w.numpy())
    sn = None if scale is None else np.ascontiguousarray(scale.numpy())
    hn = None if shift is None else np.ascontiguousarray(shift.numpy())
    bn = None if bias is None else np.ascontiguousarray(bias.numpy())
    hip.check(lib.rsg_pack_conv_weights_host(wn.ctypes.data, ptr(sn), ptr(hn), ptr(bn), Cin, O, kh, kw,
                                             hip.DTYPES[dtype], wp.ctypes.data, bo.ctypes.data))
    wpd, bod = torch.from_numpy(wp).to(DEV), torch.from_numpy(bo).to(DEV)
    pad = (kh - 1) // 2
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    out = torch.full((B, Ho, Wo, cpad), float("nan"), dtype=tdt, device=DEV)
    out32 = torch.full((B, O, Ho, Wo), float("nan"), dtype=torch.float32, device=DEV) if want_f32 else None
    res = None
    if residual is not None:
        res = torch.zeros(B, Ho, Wo, cpad, dtype=tdt)
        res[..., :O] = residual.permute(0, 2, 3, 1).to(tdt)
        res = res.to(DEV)
    d = hip.ConvDesc()
    d.inp, d.w_packed, d.bias = xin.data_ptr(), wpd.data_ptr(), bod.data_ptr()
    d.residual = res.data_ptr() if res is not None else None
    d.out16 = out.data_ptr()
    d.out_nchw_f32 = out32.data_ptr() if want_f32 else None
    d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = B, H, W, cin_pad, ld
    d.out_h, d.out_w, d.out_c, d.out_ld, d.res_ld = Ho, Wo, O, cpad, cpad
    d.kh, d.kw, d.stride, d.pad_y, d.pad_x = kh, kw, stride, pad, pad
    d.up, d.up_y, d.up_x, d.phys_h, d.phys_w = 1, 0, 0, Ho, Wo
    d.relu, d.dtype, d.tile_hint = int(relu), hip.DTYPES[dtype], tile_hint
    hip.check(lib.rsg_conv_igemm(C.byref(d), _stream()), "conv")
    torch.cuda.synchronize()
    o16 = out.float().cpu().permute(0, 3, 1, 2)
    return o16, (out32.cpu() if want_f32 else None)


def ref_conv(x, w, scale, shift, bias, stride, relu, residual, dtype=torch.float16):
    """What the kernel computes, in fp64 on the CPU: 16-bit-rounded inputs and folded weights."""
    xq = x.to(dtype).double()
    wf = w.double() * (scale.double().view(-1, 1, 1, 1) if scale is not None else 1.0)
    wq = wf.float().to(dtype).double()
    b = torch.zeros(w.shape[0], dtype=torch.float64)
    if bias is not None:
        b = b + bias.double() * (scale.double() if scale is not None else 1.0)
    if shift is not None:
        b = b + shift.double()
    y = F.conv2d(xq, wq, b, stride=stride, padding=(w.shape[-1] - 1) // 2)
    if residual is not None:
        y = y + residual.to(dtype).double()
    return F.relu(y) if relu else y


CONV_CASES = [
    # B, Cin, Cout, H, W, k, stride, relu, residual        -- the HRNet-W32 256x192 shapes ...
    (2, 3, 64, 32, 24, 3, 2, True, False),       # stem conv1 (3 -> padded 16)
    (2, 64, 64, 32, 24, 3, 2, True, False),      # stem conv2
    (2, 64, 64, 64, 48, 1, 1, True, False),      # layer1 1x1
    (2, 64, 256, 16, 12, 1, 1, True, True),      # layer1 expand + residual
    (2, 256, 64, 16, 12, 1, 1, True, False),     # layer1 reduce (4 channel blocks)
    (1, 256, 32, 64, 48, 3, 1, True, False),     # transition1.0
    (2, 32, 32, 64, 48, 3, 1, True, True),       # branch 0 BasicBlock conv2
    (2, 64, 64, 32, 24, 3, 1, True, True),       # branch 1
    (2, 128, 128, 16, 12, 3, 1, True, True),     # branch 2
    (3, 256, 256, 8, 6, 3, 1, True, True),       # branch 3
    (2, 32, 64, 64, 48, 3, 2, False, False),     # fuse down, last step (no ReLU)
    (2, 128, 32, 16, 12, 1, 1, False, False),    # fuse up 1x1
    (2, 96, 96, 16, 12, 3, 1, True, False),      # contact_conv
    (2, 17, 32, 16, 12, 3, 1, True, False),      # folded type conv (17 -> padded 32)
    (2, 32, 32, 128, 96, 3, 1, True, False),     # predict_net
    (2, 64, 32, 16, 24, 3, 1, True, False),      # kpt_net
    (2, 32, 64, 16, 24, 3, 1, False, False),     # data gradient of kpt_net (32 -> 64 channels, stride 1)
    (2, 32, 96, 16, 24, 3, 1, False, False),     # data gradient of predict_contact_net
    # ... and ragged / odd tails
    (1, 32, 32, 7, 5, 3, 1, True, True),
    (2, 64, 64, 13, 9, 3, 2, True, False),
    (1, 48, 48, 12, 9, 3, 1, True, True),        # W48 width (CINB=16 path)
    (2, 32, 32, 1, 1, 1, 1, False, False),
]


def _to_planes(t_nhwc: torch.Tensor) -> torch.Tensor:
    """dense [B,H,W,C] -> channel-tile planes [C/32][B,H,W][32] (include/rsgnet_hip.h, RSG_LAYOUT_*), flat."""
    B, H, W, Cc = t_nhwc.shape
    return t_nhwc.reshape(B, H, W, Cc // 32, 32).permute(3, 0, 1, 2, 4).contiguous()


def _from_planes(t: torch.Tensor, B, H, W, Cc) -> torch.Tensor:
    return t.reshape(Cc // 32, B, H, W, 32).permute(1, 2, 3, 0, 4).reshape(B, H, W, Cc)


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_igemm_matches_cpu(case):
    B, Cin, Cout, H, W, k, stride, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Cout, Ho, Wo, generator=g) if use_res else None
    got16, got32 = run_conv(x, w, scale, shift, None, stride=stride, relu=relu, residual=res, want_f32=True)
    want = ref_conv(x, w, scale, shift, None, stride, relu, res)
    tol32 = 1e-4 * float(want.abs().max()) + 1e-5          # fp32 accumulation-order differences only
    assert float((got32.double() - want).abs().max()) <= tol32, "fp32 NCHW output"
    err16 = (got16[:, :Cout].double() - want).abs()
    assert float((err16 - want.abs() * 2.0 ** -10).max()) <= tol32, "16-bit NHWC output"
    if got16.shape[1] > Cout:
        assert float(got16[:, Cout:].abs().max()) == 0.0, "padding channels must be written as zeros"


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[1] >= 32 and c[3] * c[4] >= 48][:12], ids=lambda c: "x".join(map(str, c)))
def test_conv_spread_changes_the_tiling_not_the_bits(case):
    """rsg_tune("conv_spread", n) re-tiles a conv whose grid has fewer than n workgroups (smaller output-channel and pixel
    tiles: more, shorter workgroups).  The contraction order per output element is the same in every tiling, so the
    results are equal bit for bit; rsg_conv_plan shows that the tiling did change for the small grids of this test."""
    B, Cin, Cout, H, W, k, stride, relu, use_res = case
    g = torch.Generator().manual_seed(hash(case) & 0xffff)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(B, Cout, Ho, Wo, generator=g) if use_res else None
    outs = []
    lib = hip.lib()
    try:
        for n in (0, 4096):
            hip.check(lib.rsg_tune(b"conv_spread", n))
            outs.append(run_conv(x, w, scale, shift, None, stride=stride, relu=relu, residual=res, want_f32=True))
    finally:
        hip.check(lib.rsg_tune(b"conv_spread", 128))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_conv_stride2_shape_sweep():
    """Stride-2 3x3 convs keep even and odd input columns in separate LDS halves (conv_body `deint`): seeded random
    shapes with odd/even sizes, tiles narrower than the map, every channel-block width."""
    rs = np.random.RandomState(2)
    for it in range(14):
        H, W = int(rs.randint(2, 41)), int(rs.randint(2, 41))
        Cin = int(rs.choice([3, 32, 64, 128, 256]))
        Cout = int(rs.choice([32, 64, 128, 256]))
        B = int(rs.randint(1, 4))
        g = torch.Generator().manual_seed(100 + it)
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
        scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
        got16, got32 = run_conv(x, w, scale, shift, None, stride=2, relu=bool(it & 1), want_f32=True)
        want = ref_conv(x, w, scale, shift, None, 2, bool(it & 1), None)
        tol32 = 1e-4 * float(want.abs().max()) + 1e-5
        assert float((got32.double() - want).abs().max()) <= tol32, (it, B, Cin, Cout, H, W)
        err16 = (got16[:, :Cout].double() - want).abs()
        assert float((err16 - want.abs() * 2.0 ** -10).max()) <= tol32, (it, B, Cin, Cout, H, W)


def test_conv_bias_only_classifier_and_bf16():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 16, 12, generator=g)
    w = torch.randn(17, 32, 1, 1, generator=g) * 0.2
    b = torch.randn(17, generator=g)
    _o16, o32 = run_conv(x, w, None, None, b, want_f32=True)
    want = ref_conv(x, w, None, None, b, 1, False, None)
    assert float((o32.double() - want).abs().max()) < 1e-4
    _o16, o32 = run_conv(x, w, None, None, b, want_f32=True, dtype="bf16")
    want = ref_conv(x, w, None, None, b, 1, False, None, dtype=torch.bfloat16)
    assert float((o32.double() - want).abs().max()) < 1e-4


def test_conv_reads_channel_slice_of_wider_buffer():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 32, 16, 12, generator=g)
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.1
    o16, _ = run_conv(x, w, in_pad_extra=64)
    want = ref_conv(x, w, None, None, None, 1, False, None)
    assert float((o16.double() - want).abs().max()) < 2e-3 * float(want.abs().max())


def test_conv_rejects_bad_descriptors():
    lib = hip.lib()
    d = hip.ConvDesc()
    assert lib.rsg_conv_igemm(C.byref(d), None) != 0
    assert b"null" in lib.rsg_last_error()


def test_nchw_to_nhwc16():
    lib = hip.lib()
    x = torch.randn(3, 3, 10, 6, generator=torch.Generator().manual_seed(8))
    out = torch.full((3, 10, 6, 16), float("nan"), dtype=torch.float16, device=DEV)
    xd = x.to(DEV)
    hip.check(lib.rsg_nchw_f32_to_nhwc16(xd.data_ptr(), out.data_ptr(), 3, 3, 10, 6, 16, 0, _stream()))
    torch.cuda.synchronize()
    o = out.cpu()
    assert torch.equal(o[..., :3], x.permute(0, 2, 3, 1).half()) and not o[..., 3:].any()


def test_hr_fuse_sum():
    lib = hip.lib()
    B, H, W, Cc = 2, 16, 8, 32
    g = torch.Generator().manual_seed(5)
    same = [torch.randn(B, H, W, Cc, generator=g).half() for _ in range(2)]
    low = [torch.randn(B, H >> s, W >> s, Cc, generator=g).half() for s in (1, 2, 3)]
    want = sum(t.float() for t in same)
    for s, t in zip((1, 2, 3), low):
        want = want + F.interpolate(t.float().permute(0, 3, 1, 2), scale_factor=2 ** s, mode="nearest").permute(0, 2, 3, 1)
    want = F.relu(want).half()
    sd, ld_ = [t.to(DEV) for t in same], [t.to(DEV) for t in low]
    out = torch.empty(B, H, W, Cc, dtype=torch.float16, device=DEV)
    sp = (C.c_void_p * 4)(*[t.data_ptr() for t in sd])
    lp = (C.c_void_p * 3)(*[t.data_ptr() for t in ld_])
    sh = (C.c_int32 * 3)(1, 2, 3)
    hip.check(lib.rsg_hr_fuse_sum(sp, 2, lp, sh, 3, out.data_ptr(), B, H, W, Cc, 0, _stream()))
    torch.cuda.synchronize()
    assert float((out.cpu().float() - want.float()).abs().max()) <= 2 ** -9 * float(want.abs().max())


@pytest.mark.parametrize("hw", [(16, 12), (64, 48), (7, 6)], ids=str)
@pytest.mark.parametrize("sig", [0, 1])
def test_bilinear2x(sig, hw):
    lib = hip.lib()
    h, w = hw
    x = torch.randn(2, 5, h, w, generator=torch.Generator().manual_seed(7))
    want = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    if sig:
        want = torch.sigmoid(want)
    xd = x.to(DEV)
    out = torch.empty(2, 5, 2 * h, 2 * w, device=DEV)
    hip.check(lib.rsg_bilinear2x_nchw_f32(xd.data_ptr(), out.data_ptr(), 10, h, w, sig, _stream()))
    torch.cuda.synchronize()
    assert float((out.cpu() - want).abs().max()) < 1e-5      # a few fp32 ulps at |x| ~ 4


def _trp_inputs(B=2, S=256, C_=32, ld=64, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = F.relu(torch.randn(B, S, C_, generator=g)) * 0.35          # post-ReLU features, affinity O(1)
    buf = torch.zeros(B, S, ld, dtype=torch.float16)
    buf[..., ld - C_:] = x.half()                                   # slice of a wider (concat) buffer
    return buf, buf[..., ld - C_:].float()


def test_trp_attention_project_tail_scores_mse():
    lib = hip.lib()
    B, S, Cc, ld = 2, 256, 32, 64
    buf, xq = _trp_inputs(B, S, Cc, ld)
    g = torch.Generator().manual_seed(1)
    wg, bg = torch.randn(Cc, Cc, generator=g) * 0.2, torch.randn(Cc, generator=g) * 0.1
    ww, bw = torch.randn(Cc, Cc, generator=g) * 0.2, torch.randn(Cc, generator=g) * 0.1
    gam, bet = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    bufd = buf.to(DEV)
    xptr = bufd.data_ptr() + (ld - Cc) * 2
    dev = lambda t: t.contiguous().to(DEV)
    wgd, bgd, wwd, bwd, gd, bd = map(dev, (wg, bg, ww, bw, gam, bet))
    gT = torch.zeros(B, Cc, S, dtype=torch.float16, device=DEV)
    e = torch.zeros(B, S, Cc, device=DEV)
    out = torch.zeros(B, S, ld, dtype=torch.float16, device=DEV)
    scores = torch.zeros(B, S, S, device=DEV)
    st = _stream()
    hip.check(lib.rsg_trp_project_g(xptr, ld, wgd.data_ptr(), bgd.data_ptr(), gT.data_ptr(), None, B, S, Cc, 0, st))
    hip.check(lib.rsg_trp_attention(xptr, ld, gT.data_ptr(), None, e.data_ptr(), B, S, Cc, 0, st))
    ws = torch.zeros(lib.rsg_trp_tail_workspace_bytes(B, S, Cc) // 4, device=DEV)
    hip.check(lib.rsg_trp_tail(e.data_ptr(), wwd.data_ptr(), bwd.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                               out.data_ptr(), ld, B, S, Cc, 0, ws.data_ptr(), st))
    hip.check(lib.rsg_trp_scores(xptr, ld, scores.data_ptr(), B, S, Cc, 0, st))
    p = torch.rand(B, S, generator=g)
    pd = p.to(DEV)
    mse = torch.full((B,), float("nan"), device=DEV)          # overwritten, not accumulated into
    mse2 = torch.full((B,), float("nan"), device=DEV)
    mws = torch.empty(lib.rsg_trp_mse_workspace_bytes(B, S), dtype=torch.uint8, device=DEV)
    hip.check(lib.rsg_trp_relation_mse(xptr, ld, pd.data_ptr(), mse.data_ptr(), mws.data_ptr(), B, S, Cc, 0, st))
    hip.check(lib.rsg_trp_relation_mse(xptr, ld, pd.data_ptr(), mse2.data_ptr(), mws.data_ptr(), B, S, Cc, 0, st))
    torch.cuda.synchronize()
    assert torch.equal(mse, mse2)                             # fixed summation order: the same bits every time
    # CPU restatement (association.py:288-300) in fp64 on the same 16-bit-rounded x
    xd = xq.double()
    gref = xd @ wg.double().t() + bg.double()
    perm = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
    g_unperm = gT.cpu().float().view(B, Cc, S // 16, 16)
    g_back = torch.empty_like(g_unperm)
    g_back[..., perm] = g_unperm                      # storage position j holds key perm[j]
    g_back = g_back.view(B, Cc, S).permute(0, 2, 1)
    assert float((g_back.double() - gref).abs().max()) < 2 ** -10 * float(gref.abs().max()) + 1e-6
    R = torch.sigmoid(xd @ xd.transpose(1, 2))
    assert float((scores.cpu().double() - R).abs().max()) < 2e-6
    eref = R @ g_back.double()                        # with the 16-bit g the kernel actually used
    assert float((e.cpu().double() - eref).abs().max()) < 3e-4 * float(eref.abs().max())
    y = e.cpu().double() @ ww.double().t() + bw.double()
    y = F.group_norm(y.permute(0, 2, 1).reshape(B, Cc, S, 1), 8, gam.double(), bet.double(), 1e-5)
    y = y.reshape(B, Cc, S).permute(0, 2, 1)
    got = out.cpu()[..., :Cc].double()
    assert float((got - y).abs().max()) < 2 ** -9 * float(y.abs().max())
    assert not out.cpu()[..., Cc:].any()              # the neighbouring slice is untouched
    mref = ((p.double()[:, :, None] * p.double()[:, None, :] - R) ** 2).mean(dim=(1, 2))
    assert float((mse.cpu().double() - mref).abs().max()) < 1e-5


def _pack_chain_layer(lib, w, scale, shift, Cc):
    """One 3x3 Cc->Cc layer in the order its chain kernel reads (rsg_chain_pack_kind: three packs by width)."""
    return hip.pack_chain_layer(w, scale, shift, 0)


@pytest.mark.parametrize("Cc,H,W,L", [(128, 16, 12, 8), (256, 8, 6, 8), (64, 32, 24, 8), (32, 64, 48, 2),
                                      (256, 8, 6, 2), (32, 30, 48, 2), (32, 64, 48, 4), (32, 64, 48, 8),
                                      (32, 61, 48, 4), (64, 32, 24, 4),
                                      # HRNet-W48 widths (12 waves per workgroup; 48 = 1.5 output-channel tiles)
                                      (48, 96, 72, 2), (96, 48, 36, 2), (192, 24, 18, 2), (384, 12, 9, 2),
                                      (384, 12, 9, 8), (96, 21, 36, 4), (48, 64, 48, 2), (192, 16, 12, 4)])
def test_basic_chain_matches_cpu(Cc, H, W, L):
    """rsg_basic_chain vs a CPU chain that rounds to fp16 where the kernel does (after every conv)."""
    lib = hip.lib()
    B = 3
    g = torch.Generator().manual_seed(Cc + H + L)
    x = torch.randn(B, Cc, H, W, generator=g)
    ws, scs, shs = [], [], []
    for l in range(L):
        ws.append(torch.randn(Cc, Cc, 3, 3, generator=g) * (2.0 / (Cc * 9)) ** 0.5)
        scs.append((torch.rand(Cc, generator=g) * 0.4 + 0.1) if l % 2 else (torch.rand(Cc, generator=g) + 0.5))
        shs.append(torch.randn(Cc, generator=g) * 0.1)
    info = (C.c_int32 * 4)()
    assert lib.rsg_basic_chain_supported(Cc, H, W, L, B, info) == 1
    keep, wps, bps = [], [], []
    for l in range(L):
        wn, sn, hn = ws[l].numpy(), scs[l].numpy(), shs[l].numpy()
        wp, bo = _pack_chain_layer(lib, wn, sn, hn, Cc)
        wd, bd = torch.from_numpy(wp).to(DEV), torch.from_numpy(bo).to(DEV)
        keep += [wd, bd]
        wps.append(wd.data_ptr())
        bps.append(bd.data_ptr())
    xin = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    out = torch.full_like(xin, float("nan"))
    need = lib.rsg_basic_chain_workspace_bytes(Cc, H, W, L, B)
    assert need > 0
    wsp = torch.empty(need, dtype=torch.uint8, device=DEV)
    hip.check(lib.rsg_basic_chain(xin.data_ptr(), out.data_ptr(), (C.c_void_p * L)(*wps), (C.c_void_p * L)(*bps),
                                  L, B, H, W, Cc, 0, wsp.data_ptr(), _stream()), "chain")
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    t = x.half().double()
    for l in range(L):
        if l % 2 == 0:
            blk_in = t
        y = ref_conv(t.float(), ws[l], scs[l], shs[l], None, 1, False, None)
        if l % 2:
            y = y + blk_in
        t = F.relu(y).float().half().double()
    assert torch.isfinite(got).all()
    err = float((got.double() - t).abs().max())
    # identical roundings except where fp32 summation order flips a 16-bit tie: a few ulps, amplified
    # mildly along the chain
    assert err <= 2.0 ** -7 * float(t.abs().max()), (err, float(t.abs().max()), list(info))


def _chain_layers(lib, Cc, L, seed):
    """L packed 3x3 Cc->Cc layers (device weights, biases) with BasicBlock-like scales."""
    g = torch.Generator().manual_seed(seed)
    keep, wps, bps = [], [], []
    for l in range(L):
        w = (torch.randn(Cc, Cc, 3, 3, generator=g) * (2.0 / (Cc * 9)) ** 0.5).numpy()
        sc = ((torch.rand(Cc, generator=g) * 0.4 + 0.1) if l % 2 else (torch.rand(Cc, generator=g) + 0.5)).numpy()
        sh = (torch.randn(Cc, generator=g) * 0.1).numpy()
        wp, bo = _pack_chain_layer(lib, w, sc, sh, Cc)
        wd, bd = torch.from_numpy(wp).to(DEV), torch.from_numpy(bo).to(DEV)
        keep += [wd, bd]
        wps.append(wd.data_ptr())
        bps.append(bd.data_ptr())
    return keep, wps, bps


def test_chain_group_handoff_inside_one_launch():
    """A row-tiled branch cut in two chains that hand over INSIDE one grouped launch (done/wait flags) gives
    bit for bit what two separate launches give, next to a long whole-image chain that keeps the grid uneven."""
    lib = hip.lib()
    B, L = 64, 4
    g = torch.Generator().manual_seed(5)
    x32 = torch.randn(B, 64, 48, 32, generator=g).half().to(DEV)
    x64 = torch.randn(B, 32, 24, 64, generator=g).half().to(DEV)
    k32a, w32a, b32a = _chain_layers(lib, 32, L, 1)
    k32b, w32b, b32b = _chain_layers(lib, 32, L, 2)
    k64, w64, b64 = _chain_layers(lib, 64, 2 * L, 3)

    def ws(Cc, H, W, n):
        return torch.empty(lib.rsg_basic_chain_workspace_bytes(Cc, H, W, n, B), dtype=torch.uint8, device=DEV)

    def arr(v):
        return (C.c_void_p * len(v))(*v)

    # reference: three separate launches
    mid, ref32, ref64 = torch.empty_like(x32), torch.empty_like(x32), torch.empty_like(x64)
    wsa, wsb, wsc = ws(32, 64, 48, L), ws(32, 64, 48, L), ws(64, 32, 24, 2 * L)
    hip.check(lib.rsg_basic_chain(x32.data_ptr(), mid.data_ptr(), arr(w32a), arr(b32a), L, B, 64, 48, 32, 0, wsa.data_ptr(), _stream()))
    hip.check(lib.rsg_basic_chain(mid.data_ptr(), ref32.data_ptr(), arr(w32b), arr(b32b), L, B, 64, 48, 32, 0, wsb.data_ptr(), _stream()))
    hip.check(lib.rsg_basic_chain(x64.data_ptr(), ref64.data_ptr(), arr(w64), arr(b64), 2 * L, B, 32, 24, 64, 0, wsc.data_ptr(), _stream()))
    torch.cuda.synchronize()

    info = (C.c_int32 * 4)()
    assert lib.rsg_basic_chain_supported(32, 64, 48, L, B, info) == 1
    flags = torch.zeros(info[3] * 3, dtype=torch.int32, device=DEV)      # zeroed once: launches leave them clean
    items = (hip.ChainItem * 3)()
    keep = []
    spec = [(x64, None, 64, 32, 24, 2 * L, w64, b64, wsc, None, None),
            (x32, None, 32, 64, 48, L, w32a, b32a, wsa, flags, None),
            (None, None, 32, 64, 48, L, w32b, b32b, wsb, None, flags)]
    outs = [torch.empty_like(x64), torch.empty_like(x32), torch.empty_like(x32)]
    for it, (xin, _o, Cc, H, W, n, wp, bp, wsp, done, wait), out in zip(items, spec, outs):
        wa, ba = arr(wp), arr(bp)
        keep += [wa, ba]
        it.inp = (outs[1] if xin is None else xin).data_ptr()
        it.out = out.data_ptr()
        it.w_packed, it.bias = C.cast(wa, C.POINTER(C.c_void_p)), C.cast(ba, C.POINTER(C.c_void_p))
        it.nlayers, it.batch, it.h, it.w, it.c = n, B, H, W, Cc
        it.workspace = wsp.data_ptr()
        it.done_flags = done.data_ptr() if done is not None else None
        it.wait_flags = wait.data_ptr() if wait is not None else None
    for rep in range(8):
        for o in outs:
            o.fill_(float("nan"))
        hip.check(lib.rsg_basic_chain_group(items, 3, 0, _stream()), "chain_group")
        torch.cuda.synchronize()
        f = flags.view(-1, info[3] // B, 3).cpu()           # [image, producer tile, consumer tile - producer tile + 1]
        assert not f[:, 0, 1:].any() and not f[:, -1, :2].any()      # every word that has a reader is clear again
        assert torch.equal(outs[0], ref64), rep
        assert torch.equal(outs[1], mid), rep
        assert torch.equal(outs[2], ref32), rep
    # the consumer must come after its producer
    items[1], items[2] = items[2], items[1]
    assert lib.rsg_basic_chain_group(items, 3, 0, _stream()) != 0


def test_chain_group_halo_exchange_inside_one_launch():
    """A row-tiled 8-layer chain whose tiles trade their halo rows after layer 4 INSIDE the launch (rsg_chain_item.exchange:
    one staging and one store per tile, a 4-row halo instead of two 4-layer chains' staging + store each) gives bit for bit
    what the two 4-layer launches give, next to a long whole-image chain; every flag word is clear again afterwards; a
    producer that never raises its flag (debug bit 4) ends in a counted time-out, not a hang."""
    lib = hip.lib()
    B, L = 64, 4
    g = torch.Generator().manual_seed(6)
    x32 = torch.randn(B, 64, 48, 32, generator=g).half().to(DEV)
    x64 = torch.randn(B, 32, 24, 64, generator=g).half().to(DEV)
    k32, w32, b32 = _chain_layers(lib, 32, 2 * L, 1)
    k64, w64, b64 = _chain_layers(lib, 64, 2 * L, 3)

    def ws(Cc, H, W, n):
        return torch.empty(lib.rsg_basic_chain_workspace_bytes(Cc, H, W, n, B), dtype=torch.uint8, device=DEV)

    def arr(v):
        return (C.c_void_p * len(v))(*v)

    mid, ref32, ref64 = torch.empty_like(x32), torch.empty_like(x32), torch.empty_like(x64)
    wsa, wsc = ws(32, 64, 48, L), ws(64, 32, 24, 2 * L)
    hip.check(lib.rsg_basic_chain(x32.data_ptr(), mid.data_ptr(), arr(w32[:L]), arr(b32[:L]), L, B, 64, 48, 32, 0, wsa.data_ptr(), _stream()))
    hip.check(lib.rsg_basic_chain(mid.data_ptr(), ref32.data_ptr(), arr(w32[L:]), arr(b32[L:]), L, B, 64, 48, 32, 0, wsa.data_ptr(), _stream()))
    hip.check(lib.rsg_basic_chain(x64.data_ptr(), ref64.data_ptr(), arr(w64), arr(b64), 2 * L, B, 32, 24, 64, 0, wsc.data_ptr(), _stream()))
    torch.cuda.synchronize()

    info = (C.c_int32 * 4)()
    need = lib.rsg_basic_chain_exchange_bytes(32, 64, 48, 2 * L, B, info)
    assert need > 0 and info[0] == 32 and info[3] == 2 * B                  # two 32-row tiles per image, 4-row halo
    assert lib.rsg_basic_chain_exchange_bytes(32, 64, 48, 2 * L + 2, B, None) == 0      # layers in fours (the trade is half way)
    xbuf = torch.empty(need, dtype=torch.uint8, device=DEV)
    flags = torch.zeros(info[3] * 3, dtype=torch.int32, device=DEV)
    items = (hip.ChainItem * 2)()
    outs = [torch.empty_like(x64), torch.empty_like(x32)]
    keep = []
    for it, (xin, Cc, H, W, wp, bp), out in zip(items, [(x64, 64, 32, 24, w64, b64), (x32, 32, 64, 48, w32, b32)], outs):
        wa, ba = arr(wp), arr(bp)
        keep += [wa, ba]
        it.inp, it.out = xin.data_ptr(), out.data_ptr()
        it.w_packed, it.bias = C.cast(wa, C.POINTER(C.c_void_p)), C.cast(ba, C.POINTER(C.c_void_p))
        it.nlayers, it.batch, it.h, it.w, it.c = 2 * L, B, H, W, Cc
    items[0].workspace = wsc.data_ptr()
    items[1].workspace, items[1].done_flags, items[1].exchange = xbuf.data_ptr(), flags.data_ptr(), 1
    for rep in range(8):
        for o in outs:
            o.fill_(float("nan"))
        xbuf.fill_(0xff)
        hip.check(lib.rsg_basic_chain_group(items, 2, 0, _stream()), "chain_group")
        torch.cuda.synchronize()
        assert not flags.any()                               # self-clearing
        assert torch.equal(outs[0], ref64), rep
        assert torch.equal(outs[1], ref32), rep
    n0, n1 = C.c_uint32(0), C.c_uint32(0)
    hip.check(lib.rsg_chain_handoff_timeouts(C.byref(n0)))
    # fault injection: nobody raises a flag -> every poll gives up, the launch still drains, the error is counted
    hip.check(lib.rsg_tune(b"chain_debug", 4))
    try:
        hip.check(lib.rsg_basic_chain_group(items, 2, 0, _stream()), "chain_group")
        torch.cuda.synchronize()
    finally:
        hip.check(lib.rsg_tune(b"chain_debug", 0))
    hip.check(lib.rsg_chain_handoff_timeouts(C.byref(n1)))
    assert n1.value > n0.value
    flags.zero_()
    hip.check(lib.rsg_basic_chain_group(items, 2, 0, _stream()), "chain_group")        # ... and the next launch is right again
    torch.cuda.synchronize()
    assert torch.equal(outs[1], ref32) and not flags.any()
    # an exchange item needs its flags, and no hand-off of the other kind
    items[1].done_flags = None
    assert lib.rsg_basic_chain_group(items, 2, 0, _stream()) != 0


@pytest.mark.parametrize("tag", ["hrnet_w32_256x192"])
def test_chain_exchange_plan_equals_the_handoff_plan(tag):
    """RSGNET_HIP_PLAN chain_exchange=0/1 on the same weights and input: the plan with the in-launch halo trade (default for the
    plans of side streams) gives the heat maps of the plan with the two-chain hand-off, bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from util import load_case
    outs, names = {}, {}
    for flag in ("0", "1"):
        os.environ["RSGNET_HIP_PLAN"] = f"chain_exchange={flag}"
        try:
            cfg, model, sd, x, gold = load_case(tag)
            model = model.cuda()
            with torch.no_grad():
                outs[flag] = model(torch.cat([x] * 8).cuda()).cpu()
                outs[flag + "b"] = model(torch.cat([x] * 8).cuda()).cpu()          # flags clean for the second replay
        finally:
            os.environ.pop("RSGNET_HIP_PLAN", None)
    assert torch.equal(outs["0"], outs["1"]) and torch.equal(outs["1"], outs["1b"])


def test_chain_group_w48_widths_one_launch():
    """One BasicBlock of every HRNet-W48 stage-4 branch (48 / 96 / 192 / 384 channels at 96x72 ... 12x9) in ONE grouped
    launch of the 12-wave kernel gives bit for bit what the separate launches give; the 48-channel map lives in a
    64-channel buffer (pixel stride 64) whose upper 16 channels the chain neither reads nor writes; the 12-wave and the
    8-wave widths cannot share a launch."""
    lib = hip.lib()
    B, L = 4, 2
    g = torch.Generator().manual_seed(11)
    dims = [(48, 96, 72), (96, 48, 36), (192, 24, 18), (384, 12, 9)]
    xs = [torch.randn(B, H, W, Cc, generator=g).half().to(DEV) for Cc, H, W in dims]
    lay = [_chain_layers(lib, Cc, L, 20 + k) for k, (Cc, _, _) in enumerate(dims)]

    def ws(Cc, H, W):
        return torch.empty(lib.rsg_basic_chain_workspace_bytes(Cc, H, W, L, B), dtype=torch.uint8, device=DEV)

    def arr(v):
        return (C.c_void_p * len(v))(*v)

    refs = []
    for (Cc, H, W), x, (_k, wp, bp) in zip(dims, xs, lay):
        r = torch.empty_like(x)
        hip.check(lib.rsg_basic_chain(x.data_ptr(), r.data_ptr(), arr(wp), arr(bp), L, B, H, W, Cc, 0,
                                      ws(Cc, H, W).data_ptr(), _stream()), "chain")
        refs.append(r)
    torch.cuda.synchronize()
    # the 48-channel operand in padded buffers: channels 48..63 hold a marker that must survive in the output buffer
    xpad = torch.full((B, 96, 72, 64), 7.0, dtype=torch.float16, device=DEV)
    xpad[..., :48] = xs[0]
    opad = torch.full((B, 96, 72, 64), -3.0, dtype=torch.float16, device=DEV)
    outs = [opad] + [torch.full_like(x, float("nan")) for x in xs[1:]]
    ins = [xpad] + xs[1:]
    items = (hip.ChainItem * 4)()
    keep = []
    for it, (Cc, H, W), xin, out, (_k, wp, bp) in zip(items, dims, ins, outs, lay):
        wa, ba, wsp = arr(wp), arr(bp), ws(Cc, H, W)
        keep += [wa, ba, wsp]
        it.inp, it.out = xin.data_ptr(), out.data_ptr()
        it.w_packed, it.bias = C.cast(wa, C.POINTER(C.c_void_p)), C.cast(ba, C.POINTER(C.c_void_p))
        it.nlayers, it.batch, it.h, it.w, it.c = L, B, H, W, Cc
        it.workspace = wsp.data_ptr()
        if Cc == 48:
            it.in_ld = it.out_ld = 64
    hip.check(lib.rsg_basic_chain_group(items, 4, 0, _stream()), "chain_group wide")
    torch.cuda.synchronize()
    assert torch.equal(opad[..., :48], refs[0]) and bool((opad[..., 48:] == -3.0).all())
    for o, r in zip(outs[1:], refs[1:]):
        assert torch.equal(o, r)
    # padded strides are for the 48-channel chain only; 8-wave widths do not mix with 12-wave ones
    items[1].in_ld = 128
    assert lib.rsg_basic_chain_group(items, 4, 0, _stream()) != 0
    items[1].in_ld = 0
    items[3].c, items[3].h, items[3].w = 256, 8, 6
    assert lib.rsg_basic_chain_group(items, 4, 0, _stream()) != 0


@pytest.mark.parametrize("shape", [(3, 64, 48, 256, 32, 64), (2, 40, 24, 256, 32, 64), (1, 18, 14, 64, 24, 64),
                                   (2, 96, 72, 256, 48, 96), (1, 22, 18, 128, 64, 96)], ids=str)
def test_conv_dual_equals_two_convs(shape):
    """rsg_conv_dual (transition1: the stride-1 and the stride-2 3x3 conv of the same map from ONE staged tile) gives bit for
    bit what two rsg_conv_igemm launches give -- whole tiles, tiles that hang over the map's edge, a narrow first conv."""
    B, H, W, Cin, Oa, Ob = shape
    lib = hip.lib()
    g = torch.Generator().manual_seed(H * W + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    wa = torch.randn(Oa, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    wb = torch.randn(Ob, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5
    sa, ha = torch.rand(Oa, generator=g) + 0.5, torch.randn(Oa, generator=g) * 0.1
    sb, hb = torch.rand(Ob, generator=g) + 0.5, torch.randn(Ob, generator=g) * 0.1
    PA = (Oa + 31) // 32 * 32
    ref_a, _ = run_conv(x, wa, sa, ha, relu=True)
    ref_b, _ = run_conv(x, wb, sb, hb, stride=2, relu=True)

    def pack(w, s, h):
        O = w.shape[0]
        wp = np.zeros(lib.rsg_conv_packed_bytes(Cin, O, 3, 3), np.uint8)
        bo = np.zeros((O + 31) // 32 * 32, np.float32)
        hip.check(lib.rsg_pack_conv_weights_host(np.ascontiguousarray(w.numpy()).ctypes.data, s.numpy().ctypes.data, h.numpy().ctypes.data,
                                                 None, Cin, O, 3, 3, 0, wp.ctypes.data, bo.ctypes.data))
        return torch.from_numpy(wp).to(DEV), torch.from_numpy(bo).to(DEV)

    wpa, boa = pack(wa, sa, ha)
    wpb, bob = pack(wb, sb, hb)
    xin = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    oa = torch.full((B, H, W, PA), float("nan"), dtype=torch.float16, device=DEV)
    ob = torch.full((B, H // 2, W // 2, Ob), float("nan"), dtype=torch.float16, device=DEV)
    d = hip.ConvDesc()
    d.inp, d.w_packed, d.bias, d.out16 = xin.data_ptr(), wpa.data_ptr(), boa.data_ptr(), oa.data_ptr()
    d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = B, H, W, Cin, Cin
    d.out_h, d.out_w, d.out_c, d.out_ld = H, W, Oa, PA
    d.kh = d.kw = 3
    d.stride, d.pad_y, d.pad_x = 1, 1, 1
    d.up, d.phys_h, d.phys_w, d.relu, d.dtype = 1, H, W, 1, 0
    hip.check(lib.rsg_conv_dual(C.byref(d), wpb.data_ptr(), bob.data_ptr(), ob.data_ptr(), Ob, Ob, 1, _stream()), "conv_dual")
    torch.cuda.synchronize()
    assert torch.equal(oa.float().cpu().permute(0, 3, 1, 2), ref_a)
    assert torch.equal(ob.float().cpu().permute(0, 3, 1, 2), ref_b)
    # the input as channel-tile planes (RSG_LAYOUT_IN_PLANES): the same bits
    xin_p = _to_planes(xin)
    oa2, ob2 = torch.full_like(oa, float("nan")), torch.full_like(ob, float("nan"))
    d.inp, d.in_ld, d.out16, d.layout = xin_p.data_ptr(), 32, oa2.data_ptr(), hip.RSG_LAYOUT_IN_PLANES
    hip.check(lib.rsg_conv_dual(C.byref(d), wpb.data_ptr(), bob.data_ptr(), ob2.data_ptr(), Ob, Ob, 1, _stream()), "conv_dual(planes)")
    torch.cuda.synchronize()
    assert torch.equal(oa2, oa) and torch.equal(ob2, ob)
    d.inp, d.in_ld, d.out16, d.layout = xin.data_ptr(), Cin, oa.data_ptr(), 0
    d.stride = 2
    assert lib.rsg_conv_dual(C.byref(d), wpb.data_ptr(), bob.data_ptr(), ob.data_ptr(), Ob, Ob, 1, _stream()) != 0


@pytest.mark.parametrize("shape", [(2, 64, 48, 256), (3, 13, 9, 128), (1, 64, 48, 64)], ids=str)
def test_conv_pair_equals_two_convs(shape):
    """rsg_conv_pair (Bottleneck conv2 3x3 -> conv3 1x1 + residual + ReLU in one launch, the 64-channel map in LDS)
    gives bit for bit what the two rsg_conv_igemm launches give."""
    B, H, W, C2 = shape
    lib = hip.lib()
    g = torch.Generator().manual_seed(H * W + C2)
    x = torch.randn(B, 64, H, W, generator=g)
    w1 = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5
    s1, h1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    w2 = torch.randn(C2, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    s2, h2 = torch.rand(C2, generator=g) * 0.4 + 0.1, torch.randn(C2, generator=g) * 0.1
    res = torch.randn(B, C2, H, W, generator=g)
    mid, _ = run_conv(x, w1, s1, h1, relu=True)
    want, _ = run_conv(mid, w2, s2, h2, relu=True, residual=res)

    def pack(w, s, h):
        O, I, kh, kw = w.shape
        wp = np.zeros(lib.rsg_conv_packed_bytes(I, O, kh, kw), np.uint8)
        bo = np.zeros(O, np.float32)
        hip.check(lib.rsg_pack_conv_weights_host(np.ascontiguousarray(w.numpy()).ctypes.data, s.numpy().ctypes.data,
                                                 h.numpy().ctypes.data, None, I, O, kh, kw, 0, wp.ctypes.data, bo.ctypes.data))
        return torch.from_numpy(wp).to(DEV), torch.from_numpy(bo).to(DEV)

    wp1, b1 = pack(w1, s1, h1)
    wp2, b2 = pack(w2, s2, h2)
    xin = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    rin = res.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    out = torch.full((B, H, W, C2), float("nan"), dtype=torch.float16, device=DEV)
    d = hip.ConvDesc()
    d.inp, d.w_packed, d.bias = xin.data_ptr(), wp1.data_ptr(), b1.data_ptr()
    d.out16 = out.data_ptr()
    d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = B, H, W, 64, 64
    d.out_h, d.out_w, d.out_c, d.out_ld = H, W, 64, 64
    d.kh, d.kw, d.stride, d.pad_y, d.pad_x = 3, 3, 1, 1, 1
    d.up, d.up_y, d.up_x, d.phys_h, d.phys_w = 1, 0, 0, H, W
    d.relu, d.dtype = 1, 0
    hip.check(lib.rsg_conv_pair(C.byref(d), wp2.data_ptr(), b2.data_ptr(), C2, rin.data_ptr(), C2, out.data_ptr(), C2, 1,
                                _stream()), "conv_pair")
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    assert torch.equal(got, want[:, :C2])
    # output and residual as channel-tile planes (RSG_LAYOUT_*): the same bits
    out_p = torch.full((C2 // 32, B, H, W, 32), float("nan"), dtype=torch.float16, device=DEV)
    rin_p = _to_planes(rin[..., :C2].contiguous())
    d.layout = hip.RSG_LAYOUT_OUT_PLANES | hip.RSG_LAYOUT_RES_PLANES
    hip.check(lib.rsg_conv_pair(C.byref(d), wp2.data_ptr(), b2.data_ptr(), C2, rin_p.data_ptr(), 32, out_p.data_ptr(), 32, 1,
                                _stream()), "conv_pair(planes)")
    torch.cuda.synchronize()
    assert torch.equal(_from_planes(out_p, B, H, W, C2), out)


@pytest.mark.parametrize("shape", [(2, 64, 48), (3, 13, 9), (1, 96, 72), (1, 5, 70)], ids=str)
def test_conv_pair_next_equals_pair_plus_conv1(shape):
    """rsg_conv_pair_next (conv2 -> conv3 + x -> the NEXT Bottleneck's conv1 in one launch, the finished accumulator tiles
    as conv1's B operand) against rsg_conv_pair followed by rsg_conv_igemm: the 256-channel block output is bit-identical;
    conv1's output sums its 256 input channels in another order (tile by tile, the lane-half and k-step bits swapped), so
    it agrees to fp32 summation noise rounded to 16 bits."""
    B, H, W = shape
    lib = hip.lib()
    g = torch.Generator().manual_seed(7 * H + W)
    x = torch.randn(B, 64, H, W, generator=g)
    w1 = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5
    s1, h1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    w2 = torch.randn(256, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    s2, h2 = torch.rand(256, generator=g) * 0.4 + 0.1, torch.randn(256, generator=g) * 0.1
    w3 = torch.randn(64, 256, 1, 1, generator=g) * (2.0 / 256) ** 0.5
    s3, h3 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    res = torch.randn(B, 256, H, W, generator=g)
    mid, _ = run_conv(x, w1, s1, h1, relu=True)
    want, _ = run_conv(mid, w2, s2, h2, relu=True, residual=res)
    want_t, _ = run_conv(want, w3, s3, h3, relu=True)

    def pack(w, s, h):
        O, I, kh, kw = w.shape
        wp = np.zeros(lib.rsg_conv_packed_bytes(I, O, kh, kw), np.uint8)
        bo = np.zeros(O, np.float32)
        hip.check(lib.rsg_pack_conv_weights_host(np.ascontiguousarray(w.numpy()).ctypes.data, s.numpy().ctypes.data,
                                                 h.numpy().ctypes.data, None, I, O, kh, kw, 0, wp.ctypes.data, bo.ctypes.data))
        return torch.from_numpy(wp).to(DEV), torch.from_numpy(bo).to(DEV)

    perm = [lib.rsg_conv_pair_next_perm(l) for l in range(256)]
    assert sorted(perm) == list(range(256))
    wp1, b1 = pack(w1, s1, h1)
    wp2, b2 = pack(w2, s2, h2)
    wp3, b3 = pack(w3[:, perm].contiguous(), s3, h3)
    xin = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    rin = res.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    out = torch.full((B, H, W, 256), float("nan"), dtype=torch.float16, device=DEV)
    tn = torch.full((B, H, W, 64), float("nan"), dtype=torch.float16, device=DEV)
    d = hip.ConvDesc()
    d.inp, d.w_packed, d.bias = xin.data_ptr(), wp1.data_ptr(), b1.data_ptr()
    d.out16 = out.data_ptr()
    d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = B, H, W, 64, 64
    d.out_h, d.out_w, d.out_c, d.out_ld = H, W, 64, 64
    d.kh, d.kw, d.stride, d.pad_y, d.pad_x = 3, 3, 1, 1, 1
    d.up, d.up_y, d.up_x, d.phys_h, d.phys_w = 1, 0, 0, H, W
    d.relu, d.dtype = 1, 0
    hip.check(lib.rsg_conv_pair_next(C.byref(d), wp2.data_ptr(), b2.data_ptr(), rin.data_ptr(), 256, out.data_ptr(), 256,
                                     wp3.data_ptr(), b3.data_ptr(), tn.data_ptr(), 64, None, 0, _stream()), "conv_pair_next")
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    got_t = tn.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all() and torch.isfinite(got_t).all()
    assert torch.equal(got, want)
    err = (got_t - want_t).abs().max().item()
    assert err <= 2 ** -9 * want_t.abs().max().item(), (err, want_t.abs().max().item())
    assert (got_t != want_t).float().mean().item() < 0.05           # and only the odd last bit at that
    # the same launch with the block output and the residual stored as channel-tile planes (RSG_LAYOUT_*): same bits
    out_p = torch.full((256 // 32, B, H, W, 32), float("nan"), dtype=torch.float16, device=DEV)
    tn_p = torch.full((B, H, W, 64), float("nan"), dtype=torch.float16, device=DEV)
    rin_p = _to_planes(rin)
    d.layout = hip.RSG_LAYOUT_OUT_PLANES | hip.RSG_LAYOUT_RES_PLANES
    hip.check(lib.rsg_conv_pair_next(C.byref(d), wp2.data_ptr(), b2.data_ptr(), rin_p.data_ptr(), 32, out_p.data_ptr(), 32,
                                     wp3.data_ptr(), b3.data_ptr(), tn_p.data_ptr(), 64, None, 0, _stream()), "conv_pair_next(planes)")
    torch.cuda.synchronize()
    assert torch.equal(_from_planes(out_p, B, H, W, 256), out) and torch.equal(tn_p, tn)
    # mixed: dense residual, output in planes
    out_p.fill_(float("nan"))
    d.layout = hip.RSG_LAYOUT_OUT_PLANES
    hip.check(lib.rsg_conv_pair_next(C.byref(d), wp2.data_ptr(), b2.data_ptr(), rin.data_ptr(), 256, out_p.data_ptr(), 32,
                                     wp3.data_ptr(), b3.data_ptr(), tn_p.data_ptr(), 64, None, 0, _stream()), "conv_pair_next(planes out)")
    torch.cuda.synchronize()
    assert torch.equal(_from_planes(out_p, B, H, W, 256), out)
    # a planes map has a pixel stride of 32: anything else is refused
    assert lib.rsg_conv_pair_next(C.byref(d), wp2.data_ptr(), b2.data_ptr(), rin.data_ptr(), 256, out_p.data_ptr(), 256,
                                  wp3.data_ptr(), b3.data_ptr(), tn_p.data_ptr(), 64, None, 0, _stream()) != 0


@pytest.mark.parametrize("shape", [(2, 64, 48), (3, 13, 9), (1, 5, 70)], ids=str)
def test_conv_pair_next_with_projected_residual(shape):
    """The first Bottleneck (pose_hrnet.py:93-101 with `downsample`): conv2 -> ONE 1x1 conv over [conv2 output | x] with
    weights [s3*W3 | sd*Wd] (conv3 + projected residual) -> ReLU -> conv1 of the next block, one launch
    (rsg_conv_pair_next with cat_x) against the three rsg_conv_igemm launches it replaces."""
    B, H, W = shape
    lib = hip.lib()
    g = torch.Generator().manual_seed(11 * H + W)
    xblk = torch.randn(B, 64, H, W, generator=g)                    # block input (stem output)
    t = torch.relu(torch.randn(B, 64, H, W, generator=g))           # conv1 output
    w2 = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5
    s2, h2 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    wcat = torch.randn(256, 128, 1, 1, generator=g) * (2.0 / 128) ** 0.5
    hcat = torch.randn(256, generator=g) * 0.1
    w3 = torch.randn(64, 256, 1, 1, generator=g) * (2.0 / 256) ** 0.5
    s3, h3 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    mid, _ = run_conv(t, w2, s2, h2, relu=True)
    want, _ = run_conv(torch.cat([mid, xblk.half().float()], 1), wcat, None, hcat, relu=True)
    want_t, _ = run_conv(want, w3, s3, h3, relu=True)

    def pack(w, s, h):
        O, I, kh, kw = w.shape
        wp = np.zeros(lib.rsg_conv_packed_bytes(I, O, kh, kw), np.uint8)
        bo = np.zeros(O, np.float32)
        hip.check(lib.rsg_pack_conv_weights_host(np.ascontiguousarray(w.numpy()).ctypes.data, None if s is None else s.numpy().ctypes.data,
                                                 h.numpy().ctypes.data, None, I, O, kh, kw, 0, wp.ctypes.data, bo.ctypes.data))
        return torch.from_numpy(wp).to(DEV), torch.from_numpy(bo).to(DEV)

    perm = [lib.rsg_conv_pair_next_perm(l) for l in range(256)]
    wp2, b2 = pack(w2, s2, h2)
    wpc, bc = pack(wcat, None, hcat)
    wp3, b3 = pack(w3[:, perm].contiguous(), s3, h3)
    tin = t.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    xin = xblk.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    out = torch.full((B, H, W, 256), float("nan"), dtype=torch.float16, device=DEV)
    tn = torch.full((B, H, W, 64), float("nan"), dtype=torch.float16, device=DEV)
    d = hip.ConvDesc()
    d.inp, d.w_packed, d.bias = tin.data_ptr(), wp2.data_ptr(), b2.data_ptr()
    d.out16 = out.data_ptr()
    d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = B, H, W, 64, 64
    d.out_h, d.out_w, d.out_c, d.out_ld = H, W, 64, 64
    d.kh, d.kw, d.stride, d.pad_y, d.pad_x = 3, 3, 1, 1, 1
    d.up, d.up_y, d.up_x, d.phys_h, d.phys_w = 1, 0, 0, H, W
    d.relu, d.dtype = 1, 0
    hip.check(lib.rsg_conv_pair_next(C.byref(d), wpc.data_ptr(), bc.data_ptr(), None, 0, out.data_ptr(), 256,
                                     wp3.data_ptr(), b3.data_ptr(), tn.data_ptr(), 64, xin.data_ptr(), 64, _stream()), "conv_pair_next cat")
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    got_t = tn.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all() and torch.isfinite(got_t).all()
    assert torch.equal(got, want)                                   # same k order as the K-concat conv_igemm
    err = (got_t - want_t).abs().max().item()
    assert err <= 2 ** -9 * want_t.abs().max().item(), (err, want_t.abs().max().item())


@pytest.mark.parametrize("shape", [(2, 64, 48), (3, 13, 9), (1, 96, 72), (1, 5, 70)], ids=str)
def test_bottleneck_fused_equals_three_convs(shape):
    """rsg_bottleneck_fused (conv1 1x1 evaluated per tile into LDS -> conv2 3x3 -> conv3 1x1 + x, one launch) against
    the three rsg_conv_igemm launches.  The head sums its 256 input channels in a different order, so the 16-bit
    64-channel map may differ by one rounding step here and there: a few fp16 ulps on the output, not bit for bit."""
    B, H, W = shape
    lib = hip.lib()
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(B, 256, H, W, generator=g)
    w0 = torch.randn(64, 256, 1, 1, generator=g) * (2.0 / 256) ** 0.5
    s0, h0 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    w1 = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5
    s1, h1 = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    w2 = torch.randn(256, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    s2, h2 = torch.rand(256, generator=g) * 0.4 + 0.1, torch.randn(256, generator=g) * 0.1
    m0, _ = run_conv(x, w0, s0, h0, relu=True)
    m1, _ = run_conv(m0, w1, s1, h1, relu=True)
    want, _ = run_conv(m1, w2, s2, h2, relu=True, residual=x)

    def pack(w, s, h):
        O, I, kh, kw = w.shape
        wp = np.zeros(lib.rsg_conv_packed_bytes(I, O, kh, kw), np.uint8)
        bo = np.zeros(O, np.float32)
        hip.check(lib.rsg_pack_conv_weights_host(np.ascontiguousarray(w.numpy()).ctypes.data, s.numpy().ctypes.data,
                                                 h.numpy().ctypes.data, None, I, O, kh, kw, 0, wp.ctypes.data, bo.ctypes.data))
        return torch.from_numpy(wp).to(DEV), torch.from_numpy(bo).to(DEV)

    perm = [lib.rsg_bottleneck_head_perm(l) for l in range(256)]
    assert sorted(perm) == list(range(256))
    wp0, b0 = pack(w0[:, perm].contiguous(), s0, h0)
    wp1, b1 = pack(w1, s1, h1)
    wp2, b2 = pack(w2, s2, h2)
    xin = x.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    out = torch.full((B, H, W, 256), float("nan"), dtype=torch.float16, device=DEV)
    d = hip.ConvDesc()
    d.inp, d.w_packed, d.bias = xin.data_ptr(), wp1.data_ptr(), b1.data_ptr()
    d.out16 = out.data_ptr()
    d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = B, H, W, 64, 64
    d.out_h, d.out_w, d.out_c, d.out_ld = H, W, 64, 64
    d.kh, d.kw, d.stride, d.pad_y, d.pad_x = 3, 3, 1, 1, 1
    d.up, d.up_y, d.up_x, d.phys_h, d.phys_w = 1, 0, 0, H, W
    d.relu, d.dtype = 1, 0
    hip.check(lib.rsg_bottleneck_fused(C.byref(d), xin.data_ptr(), 256, wp0.data_ptr(), b0.data_ptr(), wp2.data_ptr(),
                                       b2.data_ptr(), out.data_ptr(), 256, _stream()), "bottleneck_fused")
    torch.cuda.synchronize()
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    diff = (got - want).abs()
    print(f"bottleneck {shape}: max diff {float(diff.max()):.3e}, differing {float((diff > 0).float().mean()):.4f}")
    assert torch.allclose(got, want, rtol=4e-3, atol=4e-3)
    assert float(diff.mean()) < 2e-4 * float(want.std())
    assert lib.rsg_bottleneck_fused(C.byref(d), xin.data_ptr(), 256, wp0.data_ptr(), b0.data_ptr(), wp2.data_ptr(),
                                    b2.data_ptr(), xin.data_ptr(), 256, _stream()) != 0          # in place: refused


def test_stem_fused_equals_two_convs():
    """rsg_stem_fused (conv1 as 2x2 s2d conv + conv2 3x3 s2 in one launch) against the two-launch path, bit for bit,
    through the model plan: RSGNET_HIP_PLAN stem=0/1 on the same weights and input."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from util import load_case
    outs = {}
    for flag in ("0", "1"):
        os.environ["RSGNET_HIP_PLAN"] = f"stem={flag}"
        try:
            cfg, model, sd, x, gold = load_case("hrnet_w32_256x192")
            model = model.cuda()
            with torch.no_grad():
                outs[flag] = model(x.cuda()).cpu()
            names = [n for n, _ in list(model._plans.values())[0].ops]
            assert any(n.startswith("stem(conv1+conv2)") for n in names) == (flag == "1")
        finally:
            os.environ.pop("RSGNET_HIP_PLAN", None)
    assert torch.equal(outs["0"], outs["1"])


def test_stem_tail_equals_stem_plus_conv1():
    """rsg_stem_fused_conv1 (layer1.0's 1x1 conv1 on the stem's output in the stem's launch) against the stem launch followed
    by the rsg_conv_igemm launch, bit for bit, through the model plan: RSGNET_HIP_PLAN stem_tail=0/1 on the same weights and
    input; the default plan has one launch fewer."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from util import load_case
    outs, nops = {}, {}
    for flag in ("0", "1"):
        os.environ["RSGNET_HIP_PLAN"] = f"stem_tail={flag}"
        try:
            cfg, model, sd, x, gold = load_case("hrnet_w32_256x192")
            model = model.cuda()
            with torch.no_grad():
                outs[flag] = model(x.cuda()).cpu()
            names = [n for n, _ in list(model._plans.values())[0].ops]
            nops[flag] = len(names)
            assert ("stem(conv1+conv2)+layer1.0.conv1" in names) == (flag == "1") and ("layer1.0.conv1" in names) == (flag == "0")
        finally:
            os.environ.pop("RSGNET_HIP_PLAN", None)
    assert torch.equal(outs["0"], outs["1"]) and nops["1"] == nops["0"] - 1


def test_bottleneck_fused_through_the_plan():
    """RSGNET_HIP_PLAN="bottleneck=1" (opt-in) replaces conv1 + conv_pair of layer1's plain Bottlenecks by one launch each;
    heatmaps stay within the golden tolerance of the default path."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from util import load_case
    outs = {}
    for flag in ("0", "1"):
        os.environ["RSGNET_HIP_PLAN"] = f"bottleneck={flag}"
        try:
            cfg, model, sd, x, gold = load_case("hrnet_w32_256x192")
            model = model.cuda()
            with torch.no_grad():
                outs[flag] = model(x.cuda()).cpu()
            names = [n for n, _ in list(model._plans.values())[0].ops]
            assert sum("conv1+conv2+conv3" in n for n in names) == (3 if flag == "1" else 0)
        finally:
            os.environ.pop("RSGNET_HIP_PLAN", None)
    scale = float(outs["0"].std())
    # two fp16 evaluation orders of layer1 seen through 30 more layers: the fp16 noise floor of the whole network
    assert float((outs["0"] - outs["1"]).abs().max()) < 3e-2 * scale


def _pack(w, scale, shift, dtype="fp16"):
    lib = hip.lib()
    O, I, kh, kw = w.shape
    wp = np.zeros(lib.rsg_conv_packed_bytes(I, O, kh, kw), np.uint8)
    bo = np.zeros((O + 31) // 32 * 32, np.float32)
    wn, sn, hn = (np.ascontiguousarray(t.numpy()) for t in (w, scale, shift))
    hip.check(lib.rsg_pack_conv_weights_host(wn.ctypes.data, sn.ctypes.data, hn.ctypes.data, None, I, O, kh, kw,
                                             hip.DTYPES[dtype], wp.ctypes.data, bo.ctypes.data))
    return torch.from_numpy(wp).to(DEV), torch.from_numpy(bo).to(DEV)


@pytest.mark.parametrize("batch", [3, 8])
def test_conv_fuse_group_matches_cpu(batch):
    """The PRODUCTION fuse launches of a HighResolutionModule (rsg_conv_fuse_group_plan / _run, csrc/group.hip; what
    plan.py records for pose_hrnet.py:190-268) on a three-branch module, against fp64 on the CPU with 16-bit rounding
    where the launches store 16 bits: level 0 = every 1x1 "up" conv, the first strided conv of every "down" path;
    level 1 = the second strided conv of path 2<-0 with the module's last sum folded into its epilogue (three residual
    operands), plus the fuse-sums of outputs 0 and 1 (same-resolution and nearest-up-sampled operands)."""
    lib = hip.lib()
    g = torch.Generator().manual_seed(40 + batch)
    B, H, W = batch, 16, 12
    chans = [32, 64, 128]
    xs = [torch.randn(B, c, H >> i, W >> i, generator=g).relu() for i, c in enumerate(chans)]
    h16 = lambda t: t.to(torch.float16)

    def dev(t):     # NCHW fp32 -> NHWC fp16 on the device
        return h16(t.permute(0, 2, 3, 1)).contiguous().to(DEV)

    xd = [dev(t) for t in xs]
    convs = {}      # name -> (src index or name, cout, k, stride, relu)
    spec = [("u01", 1, 32, 1, 1, False), ("u02", 2, 32, 1, 1, False), ("u12", 2, 64, 1, 1, False),
            ("d10", 0, 64, 3, 2, False), ("d20a", 0, 32, 3, 2, True), ("d21", 1, 128, 3, 2, False)]
    params = {}
    for name, src, cout, k, s, relu in spec + [("d20b", "d20a", 128, 3, 2, True)]:
        cin = chans[src] if isinstance(src, int) else 32
        w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
        params[name] = (w, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, src, k, s, relu)
    keep = []

    def desc(name, inp, in_hw, out, res=()):
        w, sc, sh, src, k, s, relu = params[name]
        wp, bo = _pack(w, sc, sh)
        keep.extend([wp, bo])
        d = hip.ConvDesc()
        d.inp, d.w_packed, d.bias, d.out16 = inp.data_ptr(), wp.data_ptr(), bo.data_ptr(), out.data_ptr()
        cin, cout = w.shape[1], w.shape[0]
        pad = (k - 1) // 2
        ho, wo = (in_hw[0] + 2 * pad - k) // s + 1, (in_hw[1] + 2 * pad - k) // s + 1
        d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = B, in_hw[0], in_hw[1], cin, cin
        d.out_h, d.out_w, d.out_c, d.out_ld = ho, wo, cout, cout
        d.kh = d.kw = k
        d.stride, d.pad_y, d.pad_x = s, pad, pad
        d.up, d.phys_h, d.phys_w, d.relu, d.dtype = 1, ho, wo, int(relu), hip.RSG_F16
        if res:
            d.residual, d.res_ld = res[0].data_ptr(), cout
            if len(res) > 1:
                d.residual2, d.res2_ld = res[1].data_ptr(), cout
            if len(res) > 2:
                d.residual3, d.res3_ld = res[2].data_ptr(), cout
        return d

    nan16 = lambda c, h, w: torch.full((B, h, w, c), float("nan"), dtype=torch.float16, device=DEV)
    o = {"u01": nan16(32, 8, 6), "u02": nan16(32, 4, 3), "u12": nan16(64, 4, 3), "d10": nan16(64, 8, 6),
         "d20a": nan16(32, 8, 6), "d21": nan16(128, 4, 3), "out2": nan16(128, 4, 3), "out0": nan16(32, 16, 12),
         "out1": nan16(64, 8, 6)}

    def run(descs, fuses):
        cd = (hip.ConvDesc * max(len(descs), 1))(*descs)
        fi = (hip.FuseItem * max(len(fuses), 1))(*fuses)
        table = np.zeros(lib.rsg_group_table_bytes(), np.uint8)
        l2 = (C.c_int32 * 2)()
        hip.check(lib.rsg_conv_fuse_group_plan(cd, len(descs), fi, len(fuses), hip.RSG_F16, table.ctypes.data, l2), "plan")
        hip.check(lib.rsg_conv_fuse_group_run(table.ctypes.data, l2[0], l2[1], hip.RSG_F16, _stream()), "run")
        torch.cuda.synchronize()

    run([desc(n, xd[params[n][3]], (H >> params[n][3], W >> params[n][3]), o[n]) for n, *_ in spec], [])

    def fuse_item(same, low, out, h, w, c):
        it = hip.FuseItem()
        for k, t in enumerate(same):
            it.same[k], it.same_ld[k] = t.data_ptr(), c
        for k, (t, sh) in enumerate(low):
            it.low[k], it.low_shift[k], it.low_ld[k] = t.data_ptr(), sh, c
        it.n_same, it.n_low, it.out, it.batch, it.h, it.w, it.c = len(same), len(low), out.data_ptr(), B, h, w, c
        return it

    run([desc("d20b", o["d20a"], (8, 6), o["out2"], res=(xd[2], o["d21"]))],
        [fuse_item([xd[0]], [(o["u01"], 1), (o["u02"], 2)], o["out0"], 16, 12, 32),
         fuse_item([o["d10"], xd[1]], [(o["u12"], 1)], o["out1"], 8, 6, 64)])

    # ---- fp64 on the CPU, rounding to 16 bits wherever a launch stores 16 bits
    def cref(name, x):
        w, sc, sh, _src, k, s, relu = params[name]
        return ref_conv(x, w, sc, sh, None, s, relu, None)

    r = {n: h16(cref(n, xs[params[n][3]].float()).float()).double() for n, *_ in spec}
    x16 = [h16(t).double() for t in xs]
    up = lambda t, f: F.interpolate(t, scale_factor=f, mode="nearest")
    want0 = F.relu(x16[0] + up(r["u01"], 2) + up(r["u02"], 4))
    want1 = F.relu(r["d10"] + x16[1] + up(r["u12"], 2))
    want2 = F.relu(ref_conv(r["d20a"].float(), params["d20b"][0], params["d20b"][1], params["d20b"][2], None, 2, False, None)
                   + x16[2] + r["d21"])
    for name, want in (("out0", want0), ("out1", want1), ("out2", want2)):
        got = o[name].float().cpu().permute(0, 3, 1, 2).double()
        err = (got - want).abs()
        assert torch.isfinite(got).all(), name
        # every addend went through a 16-bit store of its own: a different fp32 summation order inside a conv may flip one
        # of those roundings (one ulp of the ADDEND, 2^-10 of a value up to the map's maximum), then the sum is rounded
        assert float(err.max()) <= 2.0 ** -9 * float(want.abs().max()), (name, float(err.max()))
    for n, *_ in spec:       # the level-0 maps themselves
        got = o[n].float().cpu().permute(0, 3, 1, 2).double()
        want = cref(n, xs[params[n][3]].float())
        assert float(((got - want).abs() - want.abs() * 2.0 ** -10).max()) <= 1e-4 * float(want.abs().max()) + 1e-5, n


@pytest.mark.parametrize("Cc,dtype", [(32, "fp16"), (48, "fp16"), (48, "bf16")])
def test_trp_sub_sample_kernels(Cc, dtype):
    """The three kernels RELATION_SUB_SAMPLE adds (association.py:254-268,286-287): 2x2 max pool of a 16-bit NHWC
    channel slice (exact), fp32 rows -> 16-bit rows (exact rounding), and the W tail (1x1 + bias -> GroupNorm(8)) reading
    the up-sampler's 16-bit output instead of fp32 E, against fp64 on the CPU."""
    lib = hip.lib()
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    dt = hip.DTYPES[dtype]
    B, H, W, ld = 3, 16, 32, 2 * Cc
    g = torch.Generator().manual_seed(7 + Cc)
    buf = torch.randn(B, H, W, ld, generator=g).to(tdt)
    bufd = buf.to(DEV)
    pooled = torch.full((B, H // 2, W // 2, Cc), float("nan"), dtype=tdt, device=DEV)
    st = _stream()
    hip.check(lib.rsg_maxpool2x2_nhwc16(bufd.data_ptr() + Cc * 2, ld, pooled.data_ptr(), Cc, B, H, W, Cc, dt, st), "pool")
    want = F.max_pool2d(buf[..., Cc:].float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    torch.cuda.synchronize()
    assert torch.equal(pooled.float().cpu(), want)
    # fp32 rows -> 16-bit rows of a wider buffer
    e = torch.randn(B * 128, Cc, generator=g) * 5
    dst = torch.zeros(B * 128, ld, dtype=tdt, device=DEV)
    hip.check(lib.rsg_f32_rows_to_nhwc16(e.to(DEV).data_ptr(), dst.data_ptr() + Cc * 2, ld, B * 128, Cc, dt, st), "cast")
    torch.cuda.synchronize()
    assert torch.equal(dst.cpu()[:, Cc:], e.to(tdt)) and not dst.cpu()[:, :Cc].any()
    # W tail on a 16-bit input
    S = 256
    x16 = (torch.randn(B, S, ld, generator=g)).to(tdt)
    ww, bw = torch.randn(Cc, Cc, generator=g) * 0.2, torch.randn(Cc, generator=g) * 0.1
    gam, bet = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    dev = lambda t: t.contiguous().to(DEV)
    xd, wwd, bwd, gd, bd = map(dev, (x16, ww, bw, gam, bet))
    out = torch.zeros(B, S, ld, dtype=tdt, device=DEV)
    ws = torch.zeros(lib.rsg_trp_tail_workspace_bytes(B, S, Cc) // 4, device=DEV)
    hip.check(lib.rsg_trp_tail_in16(xd.data_ptr() + Cc * 2, ld, wwd.data_ptr(), bwd.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                                    out.data_ptr(), ld, B, S, Cc, dt, ws.data_ptr(), st), "tail_in16")
    torch.cuda.synchronize()
    y = x16[..., Cc:].double() @ ww.double().t() + bw.double()
    y = F.group_norm(y.permute(0, 2, 1).reshape(B, Cc, S, 1), 8, gam.double(), bet.double(), 1e-5)
    y = y.reshape(B, Cc, S).permute(0, 2, 1)
    got = out.cpu()[..., :Cc].double()
    ulp = 2.0 ** -9 if dtype == "fp16" else 2.0 ** -7
    assert float((got - y).abs().max()) < ulp * float(y.abs().max())
    assert not out.cpu()[..., Cc:].any()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("with_target", [True, False])
def test_trp_backward_kernel_matches_autograd(dtype, with_target):
    """rsg_trp_backward against torch autograd (fp64, CPU) of the materialised formula on the same 16-bit operands:
    L = sum(dE * (sigmoid(X X^T) G)) + sum_b dmse_b * mean_ij (p_i p_j - sigmoid(X X^T)_ij)^2, gradients wrt X and G
    (association.py:288-297, pose_rsgnet.py:1014-1018).  The kernel rounds sigmoid and M to 16 bits before the second
    product, so the bar is a few 16-bit ulps of the largest gradient."""
    lib = hip.lib()
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    dt = hip.DTYPES[dtype]
    B, S, Cc = 2, 384, 32
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(B, S, Cc, generator=g).abs() * 0.35).to(tdt)          # post-ReLU features: scores spread over (0.5, 1)
    x[:, ::3] *= -1                                                        # and some below 0.5
    gg = (torch.randn(B, S, Cc, generator=g)).to(tdt)
    de = (torch.randn(B, S, Cc, generator=g) * 0.01).to(tdt)
    p = torch.rand(B, S, generator=g)
    dmse = torch.tensor([0.7, 1.3]) * 50
    perm = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
    tr = lambda t: t.transpose(1, 2).reshape(B, Cc, S // 16, 16)[..., perm].reshape(B, Cc, S).contiguous()
    dev = lambda t: t.contiguous().to(DEV)
    xd, gd, ded, xtd, detd, pd = map(dev, (x, gg, de, tr(x), tr(de), p))
    coef = dev(dmse * 2.0 / (S * S))
    dx = torch.full((B, S, Cc), float("nan"), device=DEV)
    dg = torch.full((B, S, Cc), float("nan"), device=DEV)
    hip.check(lib.rsg_trp_backward(xd.data_ptr(), Cc, gd.data_ptr(), Cc, ded.data_ptr(), Cc, xtd.data_ptr(), detd.data_ptr(),
                                   pd.data_ptr() if with_target else None, coef.data_ptr() if with_target else None,
                                   dx.data_ptr(), dg.data_ptr(), B, S, Cc, dt, _stream()), "trp_backward")
    torch.cuda.synchronize()
    X = x.double().requires_grad_(True)
    G = gg.double().requires_grad_(True)
    R = torch.sigmoid(X @ X.transpose(1, 2))
    L = (de.double() * (R @ G)).sum()
    if with_target:
        T = p.double()[:, :, None] * p.double()[:, None, :]
        L = L + (dmse.double() * ((T - R) ** 2).mean(dim=(1, 2))).sum()
    wx, wg = torch.autograd.grad(L, [X, G])
    ulp = 2.0 ** -10 if dtype == "fp16" else 2.0 ** -7
    ex = float((dx.cpu().double() - wx).abs().max()) / float(wx.abs().max())
    eg = float((dg.cpu().double() - wg).abs().max()) / float(wg.abs().max())
    print(f"  trp_backward {dtype} target={with_target}: dX err {ex:.2e}  dG err {eg:.2e} of the largest gradient")
    assert ex < 2 * ulp and eg < 2 * ulp


@pytest.mark.gpu
def test_trp_relation_function_matches_materialised_autograd():
    """rsgnet_amd.train_rsgnet.TrpRelation (HIP forward + backward, nothing S x S stored) against torch autograd of the
    reference's materialised formulation (association.py:288-297 + the factored relation MSE) in fp64 on the CPU, from
    the same bf16-rounded input: E, mse, and the gradients wrt the input map, g.weight and g.bias."""
    from rsgnet_amd.train_rsgnet import TrpRelation
    B, Cc, H, W = 2, 32, 16, 24
    S = H * W
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B, Cc, H, W, generator=g).abs() * 0.35).to(torch.bfloat16).float()
    x[:, :, ::2, 1::3] *= -1
    wg, bg = torch.randn(Cc, Cc, 1, 1, generator=g) * 0.2, torch.randn(Cc, generator=g) * 0.1
    p = torch.rand(B, S, generator=g)
    de = torch.randn(B, Cc, H, W, generator=g) * 0.01
    dm = torch.tensor([30.0, 60.0])
    xd = x.permute(0, 2, 3, 1).reshape(B, S, Cc).contiguous().to(DEV).requires_grad_(True)     # rows [B,S,C], as the head hands them over
    wd, bd = wg.to(DEV).requires_grad_(True), bg.to(DEV).requires_grad_(True)
    e, mse = TrpRelation.apply(xd, wd, bd, p.to(DEV))
    de_rows = de.permute(0, 2, 3, 1).reshape(B, S, Cc).to(DEV)
    loss = (e * de_rows).sum() + (mse * dm.to(DEV)).sum()
    gx, gw, gb = torch.autograd.grad(loss, [xd, wd, bd])
    e = e.reshape(B, H, W, Cc).permute(0, 3, 1, 2)
    gx = gx.reshape(B, H, W, Cc).permute(0, 3, 1, 2)
    X = x.double().requires_grad_(True)
    Wg, Bg = wg.double().requires_grad_(True), bg.double().requires_grad_(True)
    xf = X.reshape(B, Cc, S)
    R = torch.sigmoid(xf.transpose(1, 2) @ xf)
    G = F.conv2d(X, Wg, Bg).reshape(B, Cc, S).transpose(1, 2)
    E = (R @ G).transpose(1, 2).reshape(B, Cc, H, W)
    T = p.double()[:, :, None] * p.double()[:, None, :]
    M = ((T - R) ** 2).mean(dim=(1, 2))
    wx, ww, wb = torch.autograd.grad((E * de.double()).sum() + (M * dm.double()).sum(), [X, Wg, Bg])
    rel = lambda a, b: float((a.detach().cpu().double() - b.detach()).abs().max()) / float(b.detach().abs().max())
    errs = dict(E=rel(e, E), mse=rel(mse, M), dX=rel(gx, wx), dW=rel(gw, ww), db=rel(gb, wb))
    print("  TrpRelation vs fp64:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["E"] < 3e-3 and errs["mse"] < 1e-4
    assert errs["dX"] < 2 ** -6 and errs["dW"] < 2 ** -6 and errs["db"] < 2 ** -6


@pytest.mark.gpu
def test_trp_dense_relation_target_equals_the_factored_one_and_autograd():
    """The reference's training loop hands the model the relation target as a [B,S,S] tensor (function.py:261-272: p p^T).
    rsg_trp_relation_mse_dense / rsg_trp_backward_dense read it tile by tile: (a) given p p^T they reproduce the factored
    kernels bit for bit; (b) given a symmetric target that is NOT rank one they match fp64 autograd of the materialised
    formula; (c) through TrpRelation, model-level: the [B,S,S] form gives the factored form's outputs and gradients."""
    from rsgnet_amd.train_rsgnet import TrpRelation
    lib = hip.lib()
    B, S, Cc = 2, 384, 32
    g = torch.Generator().manual_seed(23)
    x = (torch.randn(B, S, Cc, generator=g).abs() * 0.35).to(torch.bfloat16)
    x[:, ::3] *= -1
    gg = torch.randn(B, S, Cc, generator=g).to(torch.bfloat16)
    de = (torch.randn(B, S, Cc, generator=g) * 0.01).to(torch.bfloat16)
    p = torch.rand(B, S, generator=g)
    dmse = torch.tensor([0.7, 1.3]) * 50
    perm = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
    tr = lambda t: t.transpose(1, 2).reshape(B, Cc, S // 16, 16)[..., perm].reshape(B, Cc, S).contiguous()
    dev = lambda t: t.contiguous().to(DEV)
    xd, gd, ded, xtd, detd, pd = map(dev, (x, gg, de, tr(x), tr(de), p))
    coef = dev(dmse * 2.0 / (S * S))
    dt = hip.DTYPES["bf16"]

    def run(target, dense):
        dx = torch.full((B, S, Cc), float("nan"), device=DEV)
        dg = torch.full((B, S, Cc), float("nan"), device=DEV)
        mse = torch.full((B,), float("nan"), device=DEV)
        fb = lib.rsg_trp_backward_dense if dense else lib.rsg_trp_backward
        fm = lib.rsg_trp_relation_mse_dense if dense else lib.rsg_trp_relation_mse
        hip.check(fb(xd.data_ptr(), Cc, gd.data_ptr(), Cc, ded.data_ptr(), Cc, xtd.data_ptr(), detd.data_ptr(), target.data_ptr(),
                     coef.data_ptr(), dx.data_ptr(), dg.data_ptr(), B, S, Cc, dt, _stream()), "trp_backward")
        mws = torch.empty(lib.rsg_trp_mse_workspace_bytes(B, S), dtype=torch.uint8, device=DEV)
        hip.check(fm(xd.data_ptr(), Cc, target.data_ptr(), mse.data_ptr(), mws.data_ptr(), B, S, Cc, dt, _stream()), "trp_relation_mse")
        torch.cuda.synchronize()
        return dx.cpu(), dg.cpu(), mse.cpu()

    # (a) p p^T in fp32 is what the factored kernels multiply out per element
    t1 = dev(p[:, :, None] * p[:, None, :])
    dx_f, dg_f, mse_f = run(pd, False)
    dx_d, dg_d, mse_d = run(t1, True)
    assert torch.equal(dx_f, dx_d) and torch.equal(dg_f, dg_d)
    assert torch.equal(mse_f, mse_d)                  # per-wave partials summed in index order: no atomics
    # (b) a symmetric target of full rank against fp64 autograd
    a = torch.rand(B, S, S, generator=g)
    t2 = 0.5 * (a + a.transpose(1, 2))
    dx2, dg2, mse2 = run(dev(t2), True)
    X, G = x.double().requires_grad_(True), gg.double().requires_grad_(True)
    R = torch.sigmoid(X @ X.transpose(1, 2))
    M = ((t2.double() - R) ** 2).mean(dim=(1, 2))
    wx, wg = torch.autograd.grad((de.double() * (R @ G)).sum() + (dmse.double() * M).sum(), [X, G])
    ex = float((dx2.double() - wx).abs().max()) / float(wx.abs().max())
    eg = float((dg2.double() - wg).abs().max()) / float(wg.abs().max())
    em = float((mse2.double() - M.detach()).abs().max()) / float(M.abs().max())
    print(f"  dense relation target: dX err {ex:.2e}  dG err {eg:.2e}  mse err {em:.2e}")
    assert ex < 2 * 2.0 ** -7 and eg < 2 * 2.0 ** -7 and em < 1e-4
    # (b') a target that is NOT symmetric: the gradient takes (T_ij + T_ji) / 2, the loss value T itself (advisor, round 4:
    # the kernel read T_ji alone, exact only for the reference's symmetric p p^T)
    t3 = torch.rand(B, S, S, generator=g)
    dx3, dg3, mse3 = run(dev(t3), True)
    X, G = x.double().requires_grad_(True), gg.double().requires_grad_(True)
    R = torch.sigmoid(X @ X.transpose(1, 2))
    M = ((t3.double() - R) ** 2).mean(dim=(1, 2))
    wx, wg = torch.autograd.grad((de.double() * (R @ G)).sum() + (dmse.double() * M).sum(), [X, G])
    ex = float((dx3.double() - wx).abs().max()) / float(wx.abs().max())
    eg = float((dg3.double() - wg).abs().max()) / float(wg.abs().max())
    em = float((mse3.double() - M.detach()).abs().max()) / float(M.abs().max())
    print(f"  asymmetric dense relation target: dX err {ex:.2e}  dG err {eg:.2e}  mse err {em:.2e}")
    assert ex < 2 * 2.0 ** -7 and eg < 2 * 2.0 ** -7 and em < 1e-4
    # (c) the autograd node with either form of the target
    wgt, bgt = (torch.randn(Cc, Cc, 1, 1, generator=g) * 0.2).to(DEV), (torch.randn(Cc, generator=g) * 0.1).to(DEV)
    outs = []
    for tgt in (pd, t1):
        xr = xd.float().requires_grad_(True)
        w_, b_ = wgt.clone().requires_grad_(True), bgt.clone().requires_grad_(True)
        e, mse = TrpRelation.apply(xr, w_, b_, tgt)
        grads = torch.autograd.grad((e * ded.float()).sum() + (mse * dmse.to(DEV)).sum(), [xr, w_, b_])
        outs.append((e.detach(), *[t.detach() for t in grads]))
    for u, v in zip(*outs):
        assert torch.equal(u, v)


@pytest.mark.gpu
def test_rows_gram_and_rows_project_match_fp64():
    """rsg_rows_gram_f32 / rsg_rows_project_f32 (the TRP projections' own gradients in training) against fp64, and bit for
    bit the same from run to run (fixed-order sums)."""
    from rsgnet_amd.train_ops import rows_gram, rows_project
    g = torch.Generator().manual_seed(3)
    n, c = 98304 + 37, 32                       # not a multiple of the row tile
    a, b = torch.randn(n, c, generator=g), torch.randn(n, c, generator=g)
    m = torch.randn(c, c, generator=g)
    ad, bd, md = a.to(DEV), b.to(DEV), m.to(DEV)
    gram, cs = rows_gram(ad, bd, with_colsum=True)
    gram2, cs2 = rows_gram(ad, bd, with_colsum=True)
    assert torch.equal(gram, gram2) and torch.equal(cs, cs2)
    ref = a.double().t() @ b.double()
    assert float((gram.cpu().double() - ref).abs().max()) < 1e-5 * float(ref.abs().max()) + 1e-3
    assert float((cs.cpu().double() - a.double().sum(0)).abs().max()) < 1e-3
    out = torch.ones(n, c, device=DEV)
    rows_project(ad, md, out=out)
    ref = 1.0 + a.double() @ m.double()
    assert float((out.cpu().double() - ref).abs().max()) < 1e-4
    assert float((rows_project(ad, md).cpu().double() - (ref - 1.0)).abs().max()) < 1e-4


def test_fuse_classifier_and_head4_equal_the_separate_launches():
    """rsg_fuse_classifier (the last module's fuse sum inside the classifier's launch) and rsg_fuse_head4 (that module's 1x1
    convs as well) against the separate launches (grouped 1x1 convs, fuse sum, rsg_conv_igemm classifier), bit for bit,
    through the model plan on the same weights and input (batch 3: the last workgroup of the first form runs past the map)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from util import load_case
    outs, nops = {}, {}
    forms = {"separate": ("0", "0", "final_layer"), "sum+classifier": ("1", "0", "stage4.fuse.0+final_layer"),
             "module+classifier": ("1", "1", "stage4.fuse+final_layer")}
    for form, (fc, h4, last) in forms.items():
        os.environ["RSGNET_HIP_PLAN"] = f"fuse_classifier={fc},fuse_head4={h4}"
        try:
            cfg, model, sd, x, gold = load_case("hrnet_w32_256x192")
            model = model.cuda()
            x3 = torch.cat((x, x[:1] * 0.5)).cuda()
            with torch.no_grad():
                outs[form] = model(x3).cpu()
            names = [n for n, _ in list(model._plans.values())[0].ops]
            nops[form] = len(names)
            assert names[-1] == last, (form, names[-3:])
        finally:
            os.environ.pop("RSGNET_HIP_PLAN", None)
    assert torch.isfinite(outs["separate"]).all()
    assert torch.equal(outs["separate"], outs["sum+classifier"]) and torch.equal(outs["separate"], outs["module+classifier"])
    assert nops["sum+classifier"] == nops["separate"] - 1 and nops["module+classifier"] == nops["separate"] - 2


@pytest.mark.parametrize("shape", [(3, 64, 48, 0), (5, 100, 52, 0), (140, 64, 48, 0), (9, 256, 192, 0), (2, 384, 288, 1), (40, 96, 80, 1)])
def test_persistent_stem_equals_the_per_tile_stem(shape):
    """rsg_stem_fused_conv1 as persistent workgroups with conv2's weights in registers (stem_persist_kernel, the default) against
    the one-tile-per-workgroup kernel (rsg_tune stem_persist 0), bit for bit, through the C ABI: both outputs (stem, layer1.0's
    conv1), tiles that hang over the right / bottom edge (W/4 = 12, 13, 20), fewer tiles than workgroups and more (140 pictures:
    560 tiles on 512 workgroups, so the walk over tiles and the prefetch of the next window run), the picture read directly
    (fp32 NCHW) and through the space-to-depth map, fp16 and bf16."""
    B, H, W, dt = shape
    lib = hip.lib()
    g = torch.Generator().manual_seed(H * 7 + W + B)
    x = torch.randn(B, 3, H, W, generator=g).to(DEV)
    w1 = torch.randn(64, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5
    w2 = torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5
    w3 = torch.randn(64, 64, 1, 1, generator=g) * (2.0 / 64) ** 0.5
    w1s = torch.zeros(64, 16, 2, 2)                      # conv1 as a 2x2 conv over the space-to-depth picture (plan.py, _record_stem_s2d)
    for a in range(2):
        for py in range(2):
            ky = 2 * (a - 1) + py + 1
            if not 0 <= ky <= 2:
                continue
            for b in range(2):
                for px in range(2):
                    kx = 2 * (b - 1) + px + 1
                    if 0 <= kx <= 2:
                        w1s[:, (py * 2 + px) * 3:(py * 2 + px) * 3 + 3, a, b] = w1[:, :, ky, kx]

    def pack(w):
        O, I, kh, kw = w.shape
        s, h = torch.rand(O, generator=g) + 0.5, torch.randn(O, generator=g) * 0.1
        wp = np.zeros(lib.rsg_conv_packed_bytes(I, O, kh, kw), np.uint8)
        bo = np.zeros(O, np.float32)
        hip.check(lib.rsg_pack_conv_weights_host(np.ascontiguousarray(w.numpy()).ctypes.data, s.numpy().ctypes.data, h.numpy().ctypes.data,
                                                 None, I, O, kh, kw, dt, wp.ctypes.data, bo.ctypes.data))
        return torch.from_numpy(wp).to(DEV), torch.from_numpy(bo).to(DEV)

    (p1, b1), (p2, b2), (p3, b3) = pack(w1s), pack(w2), pack(w3)
    t16 = torch.bfloat16 if dt else torch.float16
    xs = torch.empty(B, H // 2, W // 2, 16, dtype=t16, device=DEV)
    hip.check(lib.rsg_nchw_f32_to_s2d_nhwc16(x.data_ptr(), xs.data_ptr(), B, 3, H, W, dt, _stream()), "s2d")
    res = {}
    try:
        for persist in (0, 1):
            assert lib.rsg_tune(b"stem_persist", persist) == 0
            for direct in (0, 1):
                out = torch.full((B, H // 4, W // 4, 64), float("nan"), dtype=t16, device=DEV)
                out3 = torch.full_like(out, float("nan"))
                hip.check(lib.rsg_stem_fused_conv1(None if direct else xs.data_ptr(), x.data_ptr() if direct else None, p1.data_ptr(),
                                                   b1.data_ptr(), p2.data_ptr(), b2.data_ptr(), out.data_ptr(), 64, p3.data_ptr(),
                                                   b3.data_ptr(), out3.data_ptr(), 64, B, H, W, dt, _stream()), "stem_fused_conv1")
                torch.cuda.synchronize()
                assert torch.isfinite(out.float()).all() and torch.isfinite(out3.float()).all()
                res[persist, direct] = (out, out3)
    finally:
        lib.rsg_tune(b"stem_persist", 1)
    for direct in (0, 1):
        assert torch.equal(res[0, direct][0], res[1, direct][0]) and torch.equal(res[0, direct][1], res[1, direct][1]), direct
    assert torch.equal(res[1, 0][0], res[1, 1][0])                       # both input paths round the picture alike
    assert float(res[1, 1][0].float().abs().max()) > 0.1


@pytest.mark.parametrize("hw", [(64, 48), (96, 72), (7, 10)])
def test_bilinear2x_pair_equals_two_launches(hw):
    """rsg_bilinear2x_pair_nchw_f32 (RSGNet's two x2 up-samplings, pose_rsgnet.py:1009-1013, as one launch: the second map
    through the sigmoid) gives bit for bit what two rsg_bilinear2x_nchw_f32 launches give; row bands that end inside the map
    (2h not a multiple of 16) included."""
    lib = hip.lib()
    h, w = hw
    g = torch.Generator().manual_seed(h * w)
    a, b = torch.randn(3, 17, h, w, generator=g).to(DEV), torch.randn(3, 18, h, w, generator=g).to(DEV)
    oa, ob = torch.empty(3, 17, 2 * h, 2 * w, device=DEV), torch.empty(3, 18, 2 * h, 2 * w, device=DEV)
    pa, pb = torch.full_like(oa, float("nan")), torch.full_like(ob, float("nan"))
    hip.check(lib.rsg_bilinear2x_nchw_f32(a.data_ptr(), oa.data_ptr(), 51, h, w, 0, _stream()))
    hip.check(lib.rsg_bilinear2x_nchw_f32(b.data_ptr(), ob.data_ptr(), 54, h, w, 1, _stream()))
    hip.check(lib.rsg_bilinear2x_pair_nchw_f32(a.data_ptr(), pa.data_ptr(), 51, 0, b.data_ptr(), pb.data_ptr(), 54, 1, h, w, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(pa, oa) and torch.equal(pb, ob)
    want = torch.sigmoid(F.interpolate(b.cpu(), scale_factor=2, mode="bilinear", align_corners=True))
    assert float((pb.cpu() - want).abs().max()) < 1e-5


@pytest.mark.parametrize("cfg", [(32, 0, 0), (48, 0, 0), (32, 64, 0), (48, 64, 1)])
def test_trp_tail_without_the_projection_round_trip_equals_the_tail(cfg):
    """rsg_trp_tail_ex with keep_y = 0 (inference: the fp32 projection W e + b is recomputed by the GroupNorm pass instead of
    written to the workspace and read back) gives bit for bit the output of rsg_trp_tail / rsg_trp_tail_in16 (keep_y = 1),
    for fp32 rows and for the 16-bit map of the sub-sampled variant, C = 32 and 48."""
    lib = hip.lib()
    Cc, e_ld, dt = cfg
    B, S = 3, 768
    g = torch.Generator().manual_seed(Cc + e_ld)
    t16 = torch.bfloat16 if dt else torch.float16
    if e_ld:
        e = (torch.randn(B, S, e_ld, generator=g) * 2).to(t16).to(DEV)
    else:
        e = (torch.randn(B, S, Cc, generator=g) * 30 + 100).to(DEV)      # E is ~S times larger than its variation (trp.hip header)
    w = (torch.randn(Cc, Cc, generator=g) * 0.2).to(DEV)
    bias, gam, bet = ((torch.randn(Cc, generator=g) * 0.3 + k).to(DEV) for k in (0.0, 1.0, 0.0))
    ws = torch.empty(lib.rsg_trp_tail_workspace_bytes(B, S, Cc), dtype=torch.uint8, device=DEV)
    outs = []
    for keep in (1, 0):
        ws.fill_(0xFF)                                                    # NaN patterns: a pass that read y without writing it would show
        out = torch.full((B, S, 64), float("nan"), dtype=t16, device=DEV)
        hip.check(lib.rsg_trp_tail_ex(e.data_ptr(), e_ld, w.data_ptr(), bias.data_ptr(), gam.data_ptr(), bet.data_ptr(), out.data_ptr(),
                                      64, B, S, Cc, dt, keep, ws.data_ptr(), _stream()), "trp_tail_ex")
        torch.cuda.synchronize()
        outs.append(out[..., :Cc].clone())
    assert torch.isfinite(outs[0].float()).all() and torch.equal(outs[0], outs[1])
    x = (e[..., :Cc] if e_ld else e).float().cpu()
    y = x @ w.cpu().t() + bias.cpu()
    want = F.group_norm(y.permute(0, 2, 1), 8, gam.cpu(), bet.cpu(), 1e-5).permute(0, 2, 1)
    assert float((outs[1].float().cpu() - want).abs().max()) < (0.06 if dt else 0.01)


@pytest.mark.parametrize("tag", ["rsgnet_w32_256x192_coco", "rsgnet_w32_256x192_crowdpose"])
def test_head_tail_equals_the_two_launches(tag):
    """rsg_conv_tail1x1 (predict_net's accumulators as final_layer's MFMA operand, pose_rsgnet.py:999-1000: one launch, the
    32-channel map between them never stored) against the two rsg_conv_igemm launches, bit for bit, through the model plan:
    RSGNET_HIP_PLAN head_tail=0/1 on the same weights and input (17 and 14 joints); the default plan has one launch fewer."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from util import load_case
    outs, names = {}, {}
    for flag in ("0", "1"):
        os.environ["RSGNET_HIP_PLAN"] = f"head_tail={flag}"
        try:
            cfg, model, sd, x, gold = load_case(tag)
            model = model.cuda()
            with torch.no_grad():
                outs[flag] = [t.cpu() for t in model(torch.cat([x] * 3).cuda())[:3]]
            names[flag] = [n for n, _ in list(model._plans.values())[0].ops]
        finally:
            os.environ.pop("RSGNET_HIP_PLAN", None)
    assert "predict_net+final_layer" in names["1"] and "final_layer" in names["0"] and len(names["1"]) == len(names["0"]) - 1
    for a, b in zip(outs["0"], outs["1"]):
        assert torch.equal(a, b)
