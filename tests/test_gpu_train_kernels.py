"""GPU parity of the training-step kernels (csrc/train.hip, csrc/wgrad.hip, the device weight packer) against torch
CPU fp32/fp64 autograd on the same 16-bit-rounded operands, through the C ABI."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from rsgnet_amd import hip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = hip.DTYPES["bf16"]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _nhwc16(x_nchw, cpad=None):
    B, Cc, H, W = x_nchw.shape
    cpad = cpad or Cc
    t = torch.zeros(B, H, W, cpad, dtype=torch.bfloat16)
    t[..., :Cc] = x_nchw.permute(0, 2, 3, 1).to(torch.bfloat16)
    return t.to(DEV)


def _q(x):
    return x.to(torch.bfloat16).float()


def pack_device(ws, transpose):
    """Pack a list of fp32 [O,I,kh,kw] weights in one launch; returns the packed byte tensors."""
    lib = hip.lib()
    items = (hip.PackItem * len(ws))()
    keep, outs, blocks = [], [], 0
    for it, w in zip(items, ws):
        O, I, kh, kw = w.shape
        pc_in, pc_out = (O, I) if transpose else (I, O)
        wd = w.contiguous().to(DEV)
        dst = torch.zeros(lib.rsg_conv_packed_bytes(pc_in, pc_out, kh, kw), dtype=torch.uint8, device=DEV)
        keep.append(wd)
        outs.append(dst)
        it.w, it.dst, it.cin, it.cout, it.kh, it.kw = wd.data_ptr(), dst.data_ptr(), I, O, kh, kw
        it.transpose, it.first_block = int(transpose), blocks
        blocks += lib.rsg_pack_item_blocks(pc_in, pc_out, kh, kw)
    dev_items = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(DEV)
    hip.check(lib.rsg_pack_conv_weights_device(dev_items.data_ptr(), len(ws), blocks, BF, _stream()), "pack_device")
    torch.cuda.synchronize()
    return outs


def pack_host(w):
    lib = hip.lib()
    O, I, kh, kw = w.shape
    wp = np.zeros(lib.rsg_conv_packed_bytes(I, O, kh, kw), np.uint8)
    bo = np.zeros((O + 31) // 32 * 32, np.float32)
    wn = np.ascontiguousarray(w.numpy())
    hip.check(lib.rsg_pack_conv_weights_host(wn.ctypes.data, None, None, None, I, O, kh, kw, BF, wp.ctypes.data, bo.ctypes.data))
    return wp


def test_device_packer_equals_host_packer():
    g = torch.Generator().manual_seed(1)
    ws = [torch.randn(s, generator=g) for s in ((32, 32, 3, 3), (64, 3, 3, 3), (17, 32, 1, 1), (256, 128, 3, 3), (128, 64, 1, 1))]
    for dev, w in zip(pack_device(ws, False), ws):
        assert np.array_equal(dev.cpu().numpy(), pack_host(w))
    for dev, w in zip(pack_device(ws, True), ws):       # data-gradient form: in/out swapped, taps flipped
        assert np.array_equal(dev.cpu().numpy(), pack_host(w.transpose(0, 1).flip(2, 3).contiguous()))


def conv_packed(x16, wp, cin_pad, O, kh, stride, B, H, W):
    """rsg_conv_igemm with device-packed weights, zero bias, no ReLU -> NHWC bf16 [B,Ho,Wo,round_up(O,32)]."""
    lib = hip.lib()
    pad = (kh - 1) // 2
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kh) // stride + 1
    cpad = (O + 31) // 32 * 32
    out = torch.full((B, Ho, Wo, cpad), float("nan"), dtype=torch.bfloat16, device=DEV)
    bias = torch.zeros(cpad, dtype=torch.float32, device=DEV)
    d = hip.ConvDesc()
    d.inp, d.w_packed, d.bias, d.out16 = x16.data_ptr(), wp.data_ptr(), bias.data_ptr(), out.data_ptr()
    d.batch, d.in_h, d.in_w, d.in_c, d.in_ld = B, H, W, cin_pad, cin_pad
    d.out_h, d.out_w, d.out_c, d.out_ld = Ho, Wo, O, cpad
    d.kh, d.kw, d.stride, d.pad_y, d.pad_x = kh, kh, stride, pad, pad
    d.up, d.up_y, d.up_x, d.phys_h, d.phys_w = 1, 0, 0, Ho, Wo
    d.relu, d.dtype = 0, BF
    hip.check(lib.rsg_conv_igemm(C.byref(d), _stream()), "conv")
    torch.cuda.synchronize()
    return out


GRAD_CASES = [
    # B, Cin, Cout, H, W, k, stride
    (3, 32, 32, 64, 48, 3, 1),        # branch 0
    (2, 64, 128, 32, 24, 3, 2),       # fuse down / transition
    (3, 256, 256, 8, 6, 3, 1),        # branch 3
    (2, 128, 32, 16, 12, 1, 1),       # fuse up 1x1
    (2, 3, 64, 32, 24, 3, 2),         # stem conv1 (cin 3 in 16 channels)
    (2, 32, 17, 64, 48, 1, 1),        # classifier (cout 17 in 32 channels)
    (2, 64, 64, 13, 10, 3, 2),        # ragged: partial k-steps, odd bands
    (1, 256, 64, 64, 48, 1, 1),       # layer1 reduce
]


@pytest.mark.parametrize("case", GRAD_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_weight_and_data_gradients_match_autograd(case):
    """dW from rsg_conv_wgrad and dX from rsg_conv_igemm over the transposed/flipped pack (stride 2: behind
    rsg_zero_insert2x) against torch autograd in fp64 on the same bf16-rounded x, dy, w."""
    B, Cin, Cout, H, W, k, stride = case
    lib = hip.lib()
    g = torch.Generator().manual_seed(sum(case))
    x = _q(torch.randn(B, Cin, H, W, generator=g))
    w = _q(torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5)
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = _q(torch.randn(B, Cout, Ho, Wo, generator=g))
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv2d(xd, wd, None, stride=stride, padding=pad).backward(dy.double())
    cin_pad, cout_pad = lib.rsg_conv_padded_cin(Cin), (Cout + 31) // 32 * 32
    x16, dy16 = _nhwc16(x, cin_pad), _nhwc16(dy, cout_pad)
    # ---- weight gradient
    ws_bytes = lib.rsg_conv_wgrad_workspace_bytes(B, H, W, cin_pad, Ho, Wo, cout_pad, k, k, stride, pad)
    assert ws_bytes > 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    dw = torch.full((Cout, Cin, k, k), float("nan"), dtype=torch.float32, device=DEV)
    for rep in range(2):                                   # twice: bit-reproducible
        hip.check(lib.rsg_conv_wgrad(x16.data_ptr(), dy16.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, H, W, Cin, cin_pad, Ho, Wo,
                                     Cout, cout_pad, k, k, stride, pad, BF, _stream()), "wgrad")
        torch.cuda.synchronize()
        if rep == 0:
            first = dw.clone()
    assert torch.equal(first, dw)
    ref = wd.grad
    err = (dw.cpu().double() - ref).abs().max().item()
    assert err <= 1e-4 * ref.abs().max().item(), (err, ref.abs().max().item())
    # ---- data gradient (not needed for the picture itself)
    if Cin >= 16:
        wT = pack_device([w], True)[0]
        if stride == 2:
            dz = torch.full((B, 2 * Ho, 2 * Wo, cout_pad), float("nan"), dtype=torch.bfloat16, device=DEV)
            hip.check(lib.rsg_zero_insert2x(dy16.data_ptr(), dz.data_ptr(), B, Ho, Wo, cout_pad, _stream()), "zero_insert")
            assert (2 * Ho, 2 * Wo) == (H + H % 2, W + W % 2)
            src, sh, sw = dz, 2 * Ho, 2 * Wo
        else:
            src, sh, sw = dy16, Ho, Wo
        dcin = lib.rsg_conv_padded_cin(Cout)
        assert dcin == cout_pad or Cout % 16 == 0
        if dcin != cout_pad:                               # 16-aligned couts that are not 32-aligned do not occur in HRNet-W32
            pytest.skip("cout not a multiple of 32")
        dx = conv_packed(src, wT, dcin, Cin, k, 1, B, sh, sw)[:, :H, :W, :Cin].float().cpu().permute(0, 3, 1, 2)
        ref = xd.grad
        err = (dx.double() - ref).abs().max().item()
        assert err <= 1.2e-2 * ref.abs().max().item(), (err, ref.abs().max().item())     # bf16 output rounding


@pytest.mark.parametrize("shape", [(2, 64, 48, 32), (3, 16, 12, 128), (2, 8, 6, 256), (1, 7, 5, 64)], ids=str)
@pytest.mark.parametrize("relu,res", [(True, True), (True, False), (False, False)])
def test_batchnorm_training_forward_and_backward(shape, relu, res):
    """rsg_bn_stats + rsg_bn_apply + rsg_bn_backward against F.batch_norm(training=True) (+ residual, ReLU) and its
    autograd, fp64 on the CPU over the same bf16 inputs (nn.BatchNorm2d in train mode, pose_hrnet.py:37)."""
    B, H, W, Cc = shape
    lib = hip.lib()
    g = torch.Generator().manual_seed(B * 1000 + Cc)
    y = _q(torch.randn(B, Cc, H, W, generator=g) * 1.7 + 0.3)
    r = _q(torch.randn(B, Cc, H, W, generator=g)) if res else None
    gamma = torch.rand(Cc, generator=g) + 0.5
    beta = torch.randn(Cc, generator=g) * 0.1
    rm0, rv0 = torch.randn(Cc, generator=g) * 0.1, torch.rand(Cc, generator=g) + 0.5
    dz = _q(torch.randn(B, Cc, H, W, generator=g))
    # reference
    yd, rd = y.double().requires_grad_(True), (r.double().requires_grad_(True) if res else None)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = rm0.double().clone(), rv0.double().clone()
    z = F.batch_norm(yd, rm, rv, gd, bd, True, 0.1, 1e-5)
    if res:
        z = z + rd
    if relu:
        z = F.relu(z)
    # device forward
    pixels = B * H * W
    y16 = _nhwc16(y)
    r16 = _nhwc16(r) if res else None
    ws = torch.empty(lib.rsg_bn_workspace_bytes(Cc), dtype=torch.uint8, device=DEV)
    mean, invstd = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
    rmd, rvd = rm0.to(DEV), rv0.to(DEV)
    gam, bet = gamma.to(DEV), beta.to(DEV)
    hip.check(lib.rsg_bn_stats(y16.data_ptr(), pixels, Cc, BF, ws.data_ptr(), mean.data_ptr(), invstd.data_ptr(), rmd.data_ptr(),
                               rvd.data_ptr(), 0.1, 1e-5, _stream()), "bn_stats")
    z16 = torch.full((B, H, W, Cc), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.check(lib.rsg_bn_apply(y16.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gam.data_ptr(), bet.data_ptr(),
                               r16.data_ptr() if res else None, int(relu), z16.data_ptr(), pixels, Cc, BF, _stream()), "bn_apply")
    torch.cuda.synchronize()
    assert (rmd.cpu().double() - rm).abs().max() < 1e-5 and (rvd.cpu().double() - rv).abs().max() < 1e-5
    zref = z.detach().permute(0, 2, 3, 1)
    assert (z16.float().cpu().double() - zref).abs().max() <= 2 ** -8 * zref.abs().max() + 1e-6
    # backward: the device masks with ITS z (bf16); give the reference the same mask by using the device output
    z.backward(dz.double())
    dz16 = _nhwc16(dz)
    dy16 = torch.full((B, H, W, Cc), float("nan"), dtype=torch.bfloat16, device=DEV)
    dres = torch.full((B, H, W, Cc), float("nan"), dtype=torch.bfloat16, device=DEV) if res else None
    dgam, dbet = torch.empty(Cc, device=DEV), torch.empty(Cc, device=DEV)
    hip.check(lib.rsg_bn_backward(dz16.data_ptr(), z16.data_ptr() if relu else None, y16.data_ptr(), mean.data_ptr(),
                                  invstd.data_ptr(), gam.data_ptr(), dy16.data_ptr(), dres.data_ptr() if res else None,
                                  dgam.data_ptr(), dbet.data_ptr(), ws.data_ptr(), pixels, Cc, BF, _stream()), "bn_backward")
    torch.cuda.synchronize()
    scale = max(gd.grad.abs().max().item(), 1.0)
    assert (dgam.cpu().double() - gd.grad).abs().max().item() < 2e-3 * scale
    assert (dbet.cpu().double() - bd.grad).abs().max().item() < 2e-3 * max(bd.grad.abs().max().item(), 1.0)
    dyref = yd.grad.permute(0, 2, 3, 1)
    assert (dy16.float().cpu().double() - dyref).abs().max().item() <= 1.5e-2 * dyref.abs().max().item()
    if res:
        drref = rd.grad.permute(0, 2, 3, 1)
        assert (dres.float().cpu().double() - drref).abs().max().item() <= 2 ** -8 * drref.abs().max().item() + 1e-6


def test_gradient_plumbing_kernels():
    """ReLU mask, two-consumer add, nearest-upsample backward, zero insertion, per-channel sum."""
    lib = hip.lib()
    g = torch.Generator().manual_seed(9)
    a, b = _q(torch.randn(2, 32, 16, 12, generator=g)), _q(torch.randn(2, 32, 16, 12, generator=g))
    a16, b16 = _nhwc16(a), _nhwc16(b)
    o16 = torch.empty_like(a16)
    n = a16.numel()
    hip.check(lib.rsg_relu_backward(a16.data_ptr(), b16.data_ptr(), o16.data_ptr(), n, BF, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(o16.float().cpu(), (a * (b > 0)).permute(0, 2, 3, 1))
    hip.check(lib.rsg_add16(a16.data_ptr(), b16.data_ptr(), o16.data_ptr(), n, BF, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(o16.cpu(), (a + b).to(torch.bfloat16).permute(0, 2, 3, 1))
    for shift in (1, 2, 3):
        hi = _q(torch.randn(2, 32, 8 << shift, 6 << shift, generator=g))
        hid = hi.double().requires_grad_(False)
        lo = torch.zeros(2, 32, 8, 6, dtype=torch.float64, requires_grad=True)
        F.interpolate(lo, scale_factor=2 ** shift, mode="nearest").backward(hid)
        hi16 = _nhwc16(hi)
        lo16 = torch.empty(2, 8, 6, 32, dtype=torch.bfloat16, device=DEV)
        hip.check(lib.rsg_upsample_nearest_backward(hi16.data_ptr(), lo16.data_ptr(), 2, 8, 6, 32, shift, BF, _stream()))
        torch.cuda.synchronize()
        ref = lo.grad.permute(0, 2, 3, 1)
        assert (lo16.float().cpu().double() - ref).abs().max() <= 2 ** -8 * ref.abs().max()
    z16 = torch.full((2, 32, 24, 32), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.check(lib.rsg_zero_insert2x(a16.data_ptr(), z16.data_ptr(), 2, 16, 12, 32, _stream()))
    torch.cuda.synchronize()
    want = torch.zeros(2, 32, 24, 32)
    want[:, ::2, ::2] = a.permute(0, 2, 3, 1)
    assert torch.equal(z16.float().cpu(), want)
    x = torch.randn(3, 17, 64 * 48, generator=g)
    out = torch.empty(17, device=DEV)
    xd = x.to(DEV)
    csw = torch.empty(lib.rsg_channel_sum_workspace_bytes(17), dtype=torch.uint8, device=DEV)
    hip.check(lib.rsg_channel_sum_nchw_f32(xd.data_ptr(), out.data_ptr(), csw.data_ptr(), 3, 17, 64 * 48, _stream()))
    torch.cuda.synchronize()
    assert (out.cpu().double() - x.double().sum((0, 2))).abs().max() < 1e-3


def test_grouped_batchnorm_and_wgrad_equal_the_single_launches():
    """rsg_bn_forward_group / rsg_bn_backward_group over three differently shaped items give bit for bit what the per-map
    entry points give (same arithmetic, same reduction order); rsg_conv_wgrad_group agrees with rsg_conv_wgrad to fp32
    summation-order noise and with itself bit for bit."""
    lib = hip.lib()
    g = torch.Generator().manual_seed(77)
    shapes = [(2, 64, 48, 32), (2, 16, 12, 128), (3, 8, 6, 256)]
    n = len(shapes)
    T = lambda *s: torch.randn(*s, generator=g)
    ys = [T(*s).to(torch.bfloat16).to(DEV) for s in shapes]
    rs = [T(*s).to(torch.bfloat16).to(DEV) for s in shapes]
    dzs = [T(*s).to(torch.bfloat16).to(DEV) for s in shapes]
    gam = [(torch.rand(s[3], generator=g) + 0.5).to(DEV) for s in shapes]
    bet = [(T(s[3]) * 0.1).to(DEV) for s in shapes]

    def run(grouped):
        outs = []
        mean = [torch.zeros(s[3], device=DEV) for s in shapes]
        inv = [torch.zeros(s[3], device=DEV) for s in shapes]
        rm = [torch.zeros(s[3], device=DEV) for s in shapes]
        rv = [torch.ones(s[3], device=DEV) for s in shapes]
        z = [torch.empty_like(y) for y in ys]
        dy = [torch.empty_like(y) for y in ys]
        dres = [torch.empty_like(y) for y in ys]
        dg = [torch.zeros(s[3], device=DEV) for s in shapes]
        db = [torch.zeros(s[3], device=DEV) for s in shapes]
        ws = [torch.empty(lib.rsg_bn_workspace_bytes(s[3]), dtype=torch.uint8, device=DEV) for s in shapes]
        px = [s[0] * s[1] * s[2] for s in shapes]
        if grouped:
            it = (hip.BnItem * n)()
            for k in range(n):
                it[k].y, it[k].out, it[k].residual = ys[k].data_ptr(), z[k].data_ptr(), rs[k].data_ptr()
                it[k].gamma, it[k].beta, it[k].mean, it[k].invstd = gam[k].data_ptr(), bet[k].data_ptr(), mean[k].data_ptr(), inv[k].data_ptr()
                it[k].running_mean, it[k].running_var, it[k].workspace = rm[k].data_ptr(), rv[k].data_ptr(), ws[k].data_ptr()
                it[k].pixels, it[k].c, it[k].relu = px[k], shapes[k][3], 1
            hip.check(lib.rsg_bn_forward_group(it, n, 0.1, 1e-5, BF, _stream()), "bn_forward_group")
            for k in range(n):
                it[k].dz, it[k].z, it[k].out, it[k].dres = dzs[k].data_ptr(), z[k].data_ptr(), dy[k].data_ptr(), dres[k].data_ptr()
                it[k].dgamma, it[k].dbeta = dg[k].data_ptr(), db[k].data_ptr()
            hip.check(lib.rsg_bn_backward_group(it, n, BF, _stream()), "bn_backward_group")
        else:
            for k in range(n):
                Cc = shapes[k][3]
                hip.check(lib.rsg_bn_stats(ys[k].data_ptr(), px[k], Cc, BF, ws[k].data_ptr(), mean[k].data_ptr(), inv[k].data_ptr(),
                                           rm[k].data_ptr(), rv[k].data_ptr(), 0.1, 1e-5, _stream()))
                hip.check(lib.rsg_bn_apply(ys[k].data_ptr(), mean[k].data_ptr(), inv[k].data_ptr(), gam[k].data_ptr(), bet[k].data_ptr(),
                                           rs[k].data_ptr(), 1, z[k].data_ptr(), px[k], Cc, BF, _stream()))
                hip.check(lib.rsg_bn_backward(dzs[k].data_ptr(), z[k].data_ptr(), ys[k].data_ptr(), mean[k].data_ptr(), inv[k].data_ptr(),
                                              gam[k].data_ptr(), dy[k].data_ptr(), dres[k].data_ptr(), dg[k].data_ptr(), db[k].data_ptr(),
                                              ws[k].data_ptr(), px[k], Cc, BF, _stream()))
        torch.cuda.synchronize()
        return mean + inv + rm + rv + z + dy + dres + dg + db

    for a, b in zip(run(True), run(False)):
        assert torch.equal(a, b)
    # weight gradients: 3x3 stride 1 at three widths
    xs = [T(*s).to(torch.bfloat16).to(DEV) for s in shapes]
    dws = [[torch.full((s[3], s[3], 3, 3), float("nan"), device=DEV) for s in shapes] for _ in range(2)]
    wss = [torch.empty(lib.rsg_conv_wgrad_workspace_bytes(s[0], s[1], s[2], s[3], s[1], s[2], s[3], 3, 3, 1, 1), dtype=torch.uint8, device=DEV)
           for s in shapes]
    wi = (hip.WgradItem * n)()
    for k, s in enumerate(shapes):
        wi[k].x, wi[k].dy, wi[k].dw, wi[k].workspace = xs[k].data_ptr(), dzs[k].data_ptr(), dws[0][k].data_ptr(), wss[k].data_ptr()
        wi[k].batch, wi[k].in_h, wi[k].in_w, wi[k].cin, wi[k].cin_pad = s[0], s[1], s[2], s[3], s[3]
        wi[k].out_h, wi[k].out_w, wi[k].cout, wi[k].cout_pad, wi[k].kh, wi[k].kw, wi[k].stride, wi[k].pad = s[1], s[2], s[3], s[3], 3, 3, 1, 1
    hip.check(lib.rsg_conv_wgrad_group(wi, n, BF, _stream()), "wgrad_group")
    for k, s in enumerate(shapes):
        hip.check(lib.rsg_conv_wgrad(xs[k].data_ptr(), dzs[k].data_ptr(), dws[1][k].data_ptr(), wss[k].data_ptr(), s[0], s[1], s[2], s[3], s[3],
                                     s[1], s[2], s[3], s[3], 3, 3, 1, 1, BF, _stream()))
        torch.cuda.synchronize()
        # the grouped launch cuts the pixels into fewer slices per item (the group fills the chip together): the same
        # products, another fixed summation tree
        ref = dws[1][k]
        assert (dws[0][k] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), k
    again = [d.clone() for d in dws[0]]
    hip.check(lib.rsg_conv_wgrad_group(wi, n, BF, _stream()), "wgrad_group")
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(again, dws[0]))          # ... and reproducible bit for bit
    # pipelined form: group A defers its reduction, the next launch (group B, other workspaces) carries it in its first
    # workgroups; a launch without items flushes B.  Same gradients (another slice count), the same bits every time.
    wss2 = [torch.empty_like(w) for w in wss]
    dws2 = [torch.full_like(d, float("nan")) for d in dws[0]]
    wj = (hip.WgradItem * n)()
    for k in range(n):
        for f, _t in hip.WgradItem._fields_:
            setattr(wj[k], f, getattr(wi[k], f))
        wj[k].dw, wj[k].workspace = dws2[k].data_ptr(), wss2[k].data_ptr()
    outs = []
    for rep in range(2):
        for d in dws[0] + dws2:
            d.fill_(float("nan"))
        hip.check(lib.rsg_conv_wgrad_group_ex(wi, n, None, 0, 1, BF, _stream()), "wgrad_group_ex A")
        torch.cuda.synchronize()
        assert all(torch.isnan(d).all() for d in dws[0])                 # nothing reduced yet
        hip.check(lib.rsg_conv_wgrad_group_ex(wj, n, wi, n, 1, BF, _stream()), "wgrad_group_ex B + reduce A")
        hip.check(lib.rsg_conv_wgrad_group_ex(None, 0, wj, n, 0, BF, _stream()), "wgrad_group_ex flush B")
        torch.cuda.synchronize()
        for k in range(n):
            ref = dws[1][k]
            assert (dws[0][k] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), k
            assert torch.equal(dws[0][k], dws2[k]), k                     # A and B: the same inputs through the same plan
        outs.append([d.clone() for d in dws[0]])
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    # the group being reduced must not share workspaces with the one being computed
    assert lib.rsg_conv_wgrad_group_ex(wi, n, wi, n, 1, BF, _stream()) != 0
