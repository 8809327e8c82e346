"""HIP-backed autograd operators of the RSGNet training head (rsgnet_amd/train_ops.py) against torch autograd in fp64 on
the CPU, from the same bf16-rounded inputs (pose_rsgnet.py:961-1005: the conv -> BatchNorm -> ReLU units and the 1x1
classifiers)."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    b = b.detach().double()
    return float((a.detach().cpu().double() - b).norm() / (b.norm() + 1e-300))


@pytest.mark.parametrize("cin,cout,k", [(32, 32, 3), (96, 96, 3), (96, 32, 3), (64, 32, 3), (17, 32, 3), (14, 32, 3)])
def test_conv_bn_relu_matches_autograd(cin, cout, k):
    from rsgnet_amd import hip
    from rsgnet_amd.train_ops import conv_bn_relu
    B, H, W = 2, 16, 24
    g = torch.Generator().manual_seed(cin * 100 + cout + k)
    cpad = int(hip.lib().rsg_conv_padded_cin(cin))
    cpad = (cpad + 31) // 32 * 32
    x = torch.zeros(B, H, W, cpad)
    x[..., :cin] = torch.randn(B, H, W, cin, generator=g)
    x = x.to(torch.bfloat16)
    w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16).float()
    bn = nn.BatchNorm2d(cout, momentum=0.1)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(cout, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(cout, generator=g) * 0.2)
    dz = torch.randn(B, H, W, cout, generator=g).to(torch.bfloat16)
    # HIP
    owner = nn.Module()
    gam, bet = bn.weight.detach().clone(), bn.bias.detach().clone()
    bnd = bn.cuda().train()
    xd = x.cuda().requires_grad_(True)
    wd = w.cuda().requires_grad_(True)
    z = conv_bn_relu(owner, "unit", xd, wd, bnd)
    gx, gw, gg, gb = torch.autograd.grad((z.float() * dz.cuda().float()).sum(), [xd, wd, bnd.weight, bnd.bias])
    # fp64 reference (NCHW), teacher-forced: the ReLU takes its decisions from the HIP output, so both sides differentiate
    # the same piecewise-linear function (a flipped decision at a near-zero pixel is rounding, not an error of the backward)
    X = x[..., :cin].double().permute(0, 3, 1, 2).requires_grad_(True)
    Wt = w.double().requires_grad_(True)
    G, Bt = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    rm, rv = torch.zeros(cout, dtype=torch.float64), torch.ones(cout, dtype=torch.float64)
    Y = F.conv2d(X, Wt, None, padding=(k - 1) // 2)
    Z = F.batch_norm(Y, rm, rv, G, Bt, True, 0.1, 1e-5) * (z.detach() > 0).permute(0, 3, 1, 2).cpu().double()
    wx, ww, wg, wb = torch.autograd.grad((Z * dz.double().permute(0, 3, 1, 2)).sum(), [X, Wt, G, Bt])
    torch.cuda.synchronize()
    ez = float((z.detach().cpu().double().permute(0, 3, 1, 2) - Z.detach()).abs().max()) / float(Z.detach().abs().max())
    errs = dict(z=ez, dx=_rel(gx[..., :cin].permute(0, 3, 1, 2), wx), dw=_rel(gw, ww), dgamma=_rel(gg, wg), dbeta=_rel(gb, wb))
    print(f"  conv_bn_relu {cin}->{cout} k{k}:", {n: f"{v:.2e}" for n, v in errs.items()})
    assert ez < 2 ** -6                                            # two bf16 roundings (y, z) of values up to |Z|max
    assert errs["dx"] < 1e-2 and errs["dw"] < 1e-2 and errs["dgamma"] < 1e-2 and errs["dbeta"] < 1e-2, errs
    assert not gx[..., cin:].any()                                 # padding channels carry no gradient
    assert float((bnd.running_mean.cpu().double() - rm).abs().max()) < 2e-3 * max(1.0, float(rm.abs().max()))
    assert float((bnd.running_var.cpu().double() - rv).abs().max()) < 1e-2 * float(rv.abs().max())
    assert int(bnd.num_batches_tracked) == 1


@pytest.mark.parametrize("J,bias", [(17, True), (18, False), (14, True)])
def test_conv1x1_out_matches_autograd(J, bias):
    from rsgnet_amd.train_ops import conv1x1_out
    B, H, W, Cc = 2, 16, 24, 32
    g = torch.Generator().manual_seed(J)
    x = torch.randn(B, H, W, Cc, generator=g).to(torch.bfloat16)
    w = (torch.randn(J, Cc, 1, 1, generator=g) * 0.2).to(torch.bfloat16).float()
    b = torch.randn(J, generator=g) * 0.1 if bias else None
    dout = torch.randn(B, J, H, W, generator=g).to(torch.bfloat16).float()
    X = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    Wt = w.double().requires_grad_(True)
    Bt = b.double().requires_grad_(True) if bias else None
    O = F.conv2d(X, Wt, Bt)
    want = torch.autograd.grad((O * dout.double()).sum(), [X, Wt] + ([Bt] if bias else []))
    owner = nn.Module()
    xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True) if bias else None
    out = conv1x1_out(owner, "cls", xd, wd, bd)
    got = torch.autograd.grad((out * dout.cuda()).sum(), [xd, wd] + ([bd] if bias else []))
    torch.cuda.synchronize()
    assert out.shape == (B, J, H, W) and out.dtype == torch.float32
    eo = float((out.detach().cpu().double() - O.detach()).abs().max()) / float(O.abs().max())
    errs = dict(out=eo, dx=_rel(got[0].permute(0, 3, 1, 2), want[0]), dw=_rel(got[1], want[1]))
    if bias:
        errs["db"] = _rel(got[2], want[2])
    print(f"  conv1x1_out 32->{J}:", {n: f"{v:.2e}" for n, v in errs.items()})
    assert eo < 1e-5 and errs["dx"] < 1e-2 and errs["dw"] < 1e-4 and errs.get("db", 0) < 1e-5, errs


@pytest.mark.parametrize("sigmoid", [False, True])
@pytest.mark.parametrize("shape", [(2, 17, 16, 24), (1, 3, 5, 6), (2, 18, 64, 48)])
def test_bilinear2x_backward_matches_autograd(shape, sigmoid):
    """rsg_bilinear2x_backward_nchw_f32 (gather form) against torch autograd of F.interpolate(x2, bilinear,
    align_corners=True) [+ sigmoid] in fp64 (pose_rsgnet.py:1009-1013)."""
    from rsgnet_amd.train_ops import Bilinear2x
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    dout = torch.randn(shape[0], shape[1], 2 * shape[2], 2 * shape[3], generator=g)
    X = x.double().requires_grad_(True)
    O = F.interpolate(X, scale_factor=2, mode="bilinear", align_corners=True)
    if sigmoid:
        O = torch.sigmoid(O)
    want, = torch.autograd.grad((O * dout.double()).sum(), [X])
    xd = x.cuda().requires_grad_(True)
    out = Bilinear2x.apply(xd, sigmoid)
    got, = torch.autograd.grad((out * dout.cuda()).sum(), [xd])
    torch.cuda.synchronize()
    assert float((out.detach().cpu().double() - O.detach()).abs().max()) < 3e-5          # fp32 source coordinates against fp64 ones
    e = float((got.cpu().double() - want).abs().max()) / float(want.abs().max())
    print(f"  bilinear2x backward {shape} sigmoid={sigmoid}: {e:.2e} of the largest gradient")
    assert e < 5e-6


@pytest.mark.parametrize("B,S", [(2, 256), (3, 768)])
def test_groupnorm_tail_matches_autograd(B, S):
    """GroupNormTail (rsg_trp_tail forward, rsg_trp_tail_backward) against torch autograd in fp64 of
    GroupNorm(8, 32)(conv1x1(e) + bias) (association.py:298-300): output, dE, dW, dbias, dgamma, dbeta."""
    from rsgnet_amd.train_ops import GroupNormTail
    Cc = 32
    g = torch.Generator().manual_seed(B * S)
    e = torch.randn(B, S, Cc, generator=g) * 3
    w, bias = torch.randn(Cc, Cc, 1, 1, generator=g) * 0.2, torch.randn(Cc, generator=g) * 0.1
    gam, bet = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.1
    dout = torch.randn(B, S, Cc, generator=g).to(torch.bfloat16)
    E, Wt, Bi, Ga, Be = (t.double().requires_grad_(True) for t in (e, w, bias, gam, bet))
    Y = E @ Wt.reshape(Cc, Cc).t() + Bi
    O = F.group_norm(Y.permute(0, 2, 1).reshape(B, Cc, S, 1), 8, Ga, Be, 1e-5).reshape(B, Cc, S).permute(0, 2, 1)
    want = torch.autograd.grad((O * dout.double()).sum(), [E, Wt, Bi, Ga, Be])
    dev = [t.cuda().requires_grad_(True) for t in (e, w, bias, gam, bet)]
    out = GroupNormTail.apply(*dev)
    got = torch.autograd.grad((out.float() * dout.cuda().float()).sum(), dev)
    torch.cuda.synchronize()
    eo = float((out.detach().cpu().double() - O.detach()).abs().max()) / float(O.detach().abs().max())
    errs = {n: _rel(a, b) for n, a, b in zip(("dE", "dW", "dbias", "dgamma", "dbeta"), got, want)}
    print(f"  groupnorm tail B={B} S={S}: out {eo:.2e}", {n: f"{v:.2e}" for n, v in errs.items()})
    assert eo < 2 ** -7                                            # one bf16 rounding of the output
    assert all(v < 1e-4 for v in errs.values()), errs


def test_deconv_bn_relu_matches_autograd():
    """DeconvBnRelu (ConvTranspose2d(4,2,1) as four sub-pixel 2x2 convs -> training-mode BatchNorm -> ReLU; backward through
    the conv and MFMA weight-gradient kernels) against torch autograd in fp64, the ReLU decisions taken from the HIP
    output (pose_rsgnet.py:741-750)."""
    from rsgnet_amd.train_ops import deconv_bn_relu
    B, H, W, Cc = 2, 16, 24, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, Cc, generator=g).to(torch.bfloat16)
    dc = nn.ConvTranspose2d(Cc, Cc, 4, 2, 1, 0, bias=False)
    bn = nn.BatchNorm2d(Cc, momentum=0.1)
    with torch.no_grad():
        dc.weight.copy_((torch.randn(Cc, Cc, 4, 4, generator=g) * 0.1).to(torch.bfloat16).float())
        bn.weight.copy_(torch.rand(Cc, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(Cc, generator=g) * 0.2)
    dz = torch.randn(B, 2 * H, 2 * W, Cc, generator=g).to(torch.bfloat16)
    w0, gam, bet = dc.weight.detach().clone(), bn.weight.detach().clone(), bn.bias.detach().clone()
    owner = nn.Module()
    dcd, bnd = dc.cuda(), bn.cuda().train()
    xd = x.cuda().requires_grad_(True)
    z = deconv_bn_relu(owner, "up", xd, dcd, bnd)
    gx, gw, gg, gb = torch.autograd.grad((z.float() * dz.cuda().float()).sum(), [xd, dcd.weight, bnd.weight, bnd.bias])
    torch.cuda.synchronize()
    X = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    Wt, G, Bt = w0.double().requires_grad_(True), gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    rm, rv = torch.zeros(Cc, dtype=torch.float64), torch.ones(Cc, dtype=torch.float64)
    Y = F.conv_transpose2d(X, Wt, None, stride=2, padding=1)
    Z = F.batch_norm(Y, rm, rv, G, Bt, True, 0.1, 1e-5) * (z.detach() > 0).permute(0, 3, 1, 2).cpu().double()
    wx, ww, wg, wb = torch.autograd.grad((Z * dz.double().permute(0, 3, 1, 2)).sum(), [X, Wt, G, Bt])
    assert z.shape == (B, 2 * H, 2 * W, Cc)
    ez = float((z.detach().cpu().double().permute(0, 3, 1, 2) - Z.detach()).abs().max()) / float(Z.detach().abs().max())
    errs = dict(z=ez, dx=_rel(gx.permute(0, 3, 1, 2), wx), dw=_rel(gw, ww), dgamma=_rel(gg, wg), dbeta=_rel(gb, wb))
    print("  deconv_bn_relu:", {n: f"{v:.2e}" for n, v in errs.items()})
    assert ez < 2 ** -6
    assert errs["dx"] < 1e-2 and errs["dw"] < 1e-2 and errs["dgamma"] < 1e-2 and errs["dbeta"] < 1e-2, errs


@pytest.mark.parametrize("limbs,joints,channels", [(18, 17, 32), (14, 14, 32), (14, 14, 48), (32, 32, 64), (1, 1, 8)])
def test_sgm_graph_conv_matches_the_reference_module_and_its_autograd(limbs, joints, channels):
    """rsg_sgm_graph_conv[_backward] against KTMachine's four statements (/root/reference/lib/models/pose_rsgnet.py:576-600) in
    fp64 autograd on the CPU: the limb classifier and the gradients of matrix_limb, the joint classifier's weight and both
    Linears; the prior's zeros stay zeros in d matrix_limb; two runs give the same bits."""
    from rsgnet_amd.train_ops import SgmGraphConv
    g = torch.Generator().manual_seed(limbs * 1000 + joints * 10 + channels)
    limb = torch.randn(limbs, joints, generator=g)
    prior = (torch.rand(limbs, joints, generator=g) < 0.2).float()
    prior[0, 0] = 1.0
    wk = torch.randn(joints, channels, generator=g) * 0.3
    w1, w2 = (torch.randn(channels, channels, generator=g) * (2.0 / channels) ** 0.5 for _ in range(2))
    b1, b2 = (torch.randn(channels, generator=g) * 0.1 for _ in range(2))
    dwl = torch.randn(limbs, channels, generator=g)
    dev = [t.cuda().requires_grad_(True) for t in (limb, wk, w1, b1, w2, b2)]
    wl = SgmGraphConv.apply(dev[0], prior.cuda(), *dev[1:], 0.02)
    grads = torch.autograd.grad((wl * dwl.cuda()).sum(), dev)
    wl2 = SgmGraphConv.apply(dev[0], prior.cuda(), *dev[1:], 0.02)
    grads2 = torch.autograd.grad((wl2 * dwl.cuda()).sum(), dev)
    assert torch.equal(wl, wl2) and all(torch.equal(a, b) for a, b in zip(grads, grads2))
    ref = [t.double().requires_grad_(True) for t in (limb, wk, w1, b1, w2, b2)]
    t = (ref[0] * prior.double()) @ ref[1]
    out = F.linear(F.leaky_relu(F.linear(t, ref[2], ref[3]), 0.02), ref[4], ref[5])
    want = torch.autograd.grad((out * dwl.double()).sum(), ref)
    assert _rel(wl, out) < 2e-6
    for name, a, b in zip(("matrix_limb", "wk", "w1", "b1", "w2", "b2"), grads, want):
        assert _rel(a, b) < 5e-6, name
    assert not (grads[0].cpu()[prior == 0]).any()
