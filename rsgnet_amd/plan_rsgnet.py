"""Recorder for RSGNet's head (pose_rsgnet.py:955-1021) on top of the shared HRNet backbone recorder.

Weight-only sub-graphs are folded at plan-build time (they do not depend on the input in eval mode,
SURVEY §0): ``type_fc(type_features)`` and its product with ``type_conv`` (an exact 3x3 conv J->C0),
``loc_conv(loc_features)`` (a constant map written once into the concat buffer) and
``kt_machine(final_layer.weight)`` (the SGM limb classifier).  These folds are fp32 tensor algebra on
the *parameters*; every per-crop operation runs in librsgnet_hip.so.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import hip
from . import switches as sw
from .plan import BN_EPS, VT, Plan


def _bn_affine(bn):
    s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    return s, bn.bias.detach().double() - bn.running_mean.detach().double() * s


def type_templates(model) -> torch.Tensor:
    """ReLU(BN1d(Linear(type_features)))  [J,T]  (pose_rsgnet.py:679-683,971-972), fp64."""
    t = model.type_features.detach().double().cpu() @ model.type_fc[0].weight.detach().double().cpu().t()
    s, sh = _bn_affine(model.type_fc[1])
    return torch.relu(t * s.cpu() + sh.cpu())


def folded_type_conv_weight(model) -> torch.Tensor:
    """W'[o,j,ky,kx] = sum_c type_conv.W[o,c,ky,kx] * Tm[j,c]  (the [B,600,H,W] map never exists)."""
    tm = type_templates(model)
    w = model.type_conv[0].weight.detach().double().cpu()
    return torch.einsum("ocyx,jc->ojyx", w, tm).float()


def limb_classifier(model) -> torch.Tensor:
    """KTMachine.forward(final_layer.weight) -> [L,C0,1,1]  (pose_rsgnet.py:576-600,1004).  With the parameters on the GPU
    (every plan build) the SGM's graph convolution runs on the library's kernel (rsg_sgm_graph_conv, fp32); on the CPU
    (host-side tests against the goldens) the same four statements in fp64."""
    kt = model.kt_machine
    if model.final_layer.weight.is_cuda:
        from .train_ops import sgm_graph_conv
        with torch.no_grad():
            return sgm_graph_conv(kt, model.final_layer.weight.detach()).float().cpu()
    wk = model.final_layer.weight.detach().double().cpu()
    n_out, n_in, h, w = wk.shape
    k = kt.matrix_limb.detach().double().cpu() * kt.real_matrix_limb.detach().double().cpu()
    t = k @ wk.reshape(n_out, n_in * h * w)
    l0, l2 = kt.kpt_transformer[0], kt.kpt_transformer[2]
    t = F.linear(t, l0.weight.detach().double().cpu(), l0.bias.detach().double().cpu())
    t = F.leaky_relu(t, 0.02)
    t = F.linear(t, l2.weight.detach().double().cpu(), l2.bias.detach().double().cpu())
    return t.reshape(t.shape[0], n_in, h, w).float()


def location_features(model) -> torch.Tensor:
    """ReLU(BN(conv1x1(loc_features)))  [1,C0,H4,W4] (pose_rsgnet.py:693-697,979), fp64 -> fp32."""
    x = model.loc_features.detach().double().cpu()
    y = F.conv2d(x, model.loc_conv[0].weight.detach().double().cpu())
    s, sh = _bn_affine(model.loc_conv[1])
    return torch.relu(y * s.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1)).float()


def deconv_phase_weights(wt: torch.Tensor):
    """ConvTranspose2d(k=4,s=2,p=1) weight [I,O,4,4] -> four 2x2 convs (sub-pixel form).

    out[2y+py, 2x+px] = sum_{a,b in {0,1}} W[i,o,ky(py,a),kx(px,b)] * in[y+a-pad_y, x+b-pad_x]
    with ky(0,a) = 3-2a, pad 1 (rows y-1, y) and ky(1,a) = 2-2a, pad 0 (rows y, y+1)
    (nn.ConvTranspose2d semantics).  Yields (py, px, pad_y, pad_x, W_phase[O,I,2,2]).
    """
    wt = wt.detach().float().cpu()
    for py in (0, 1):
        kys = [3, 1] if py == 0 else [2, 0]
        for px in (0, 1):
            kxs = [3, 1] if px == 0 else [2, 0]
            w = wt[:, :, kys, :][:, :, :, kxs]            # [I,O,2,2]
            yield py, px, 1 - py, 1 - px, w.permute(1, 0, 2, 3).contiguous()


def record_rsgnet(p: Plan, model) -> None:
    cfg = model.cfg.MODEL
    W, H = cfg.IMAGE_SIZE
    J, L = cfg.NUM_JOINTS, cfg.NUM_LIMBS
    c0 = cfg.EXTRA["OUTPUT_CONVS"][0]
    if c0 not in (32, 48):
        raise NotImplementedError(f"RSGNet head with C0={c0}: the TRP kernels are built for 32 and 48 channels")
    if model.scale_factor != 2:
        raise NotImplementedError("UP_SCALE=1")
    B, dev, lib, dt = p.batch, p.device, p.lib, p.dtype
    H4, W4 = H // 4, W // 4
    S = H4 * W4
    p.H, p.W, p.S, p.C0 = H, W, S, c0
    f32 = dict(dtype=torch.float32, device=dev)

    vis = p._record_backbone(model, H, W)[0]

    # ---- multi-joint scores: fp32 NCHW for the output path, 16-bit NHWC (J padded to 32) for the type conv
    p.multi_lo = torch.zeros(B, J, H4, W4, **f32)
    mfl = model.multi_final_layer
    # ---- the three clues land in one [.,96] buffer (torch.cat at :982 becomes a channel offset)
    cat = p.new(H4, W4, 3 * c0, pinned=True, name="clues")
    p.begin_group()                                      # both read the backbone output: one launch
    scores16 = p.conv(vis, mfl.weight, None, mfl.bias, out_f32=p.multi_lo, name="multi_final_layer")
    p._unit(vis, model.vis_conv[0], model.vis_conv[1], relu=True, out=cat.view(0, c0), name="vis_conv")
    p.end_group("multi_final_layer+vis_conv")
    # the conv reads the first padded_cin(J) channels of the 32-wide score map (14 -> 16, 17 -> 32)
    p.conv(scores16.view(0, lib.rsg_conv_padded_cin(J)), folded_type_conv_weight(model), model.type_conv[1], None, relu=True,
           out=cat.view(c0, c0), name="type_conv(folded)")
    loc = location_features(model)                               # constant: written once, below

    t = p._unit(cat, model.contact_conv[0], model.contact_conv[1], relu=True, name="contact_conv")
    cat2 = p.new(H4, W4, 2 * c0, pinned=True, name="rel|vis")   # read on demand after the run
    fvis = cat2.view(c0, c0)
    p._unit(t, model.predict_contact_net[0], model.predict_contact_net[1], relu=True, out=fvis,
            name="predict_contact_net")
    p.fvis = fvis

    # ---- TRP (association.py:280-301)
    rh = model.relation_head
    sub = bool(getattr(rh, "sub_sample", False))
    Wseq = rh.W[1] if sub else rh.W                  # conv1x1(+bias) -> GroupNorm(8); sub_sample puts the up-sampler in front
    wg = p._dev(rh.g.weight.detach().float().cpu().reshape(c0, c0).numpy())
    bg = p._dev(rh.g.bias.detach().float().cpu().numpy())
    ww = p._dev(Wseq[0].weight.detach().float().cpu().reshape(c0, c0).numpy())
    bw = p._dev(Wseq[0].bias.detach().float().cpu().numpy())
    gam = p._dev(Wseq[1].weight.detach().float().cpu().numpy())
    bet = p._dev(Wseq[1].bias.detach().float().cpu().numpy())
    if abs(Wseq[1].eps - BN_EPS) > 0:
        raise NotImplementedError("GroupNorm eps != 1e-5")
    if sub:
        # RELATION_SUB_SAMPLE (association.py:254-268,286-287): theta = phi = g's input = the 2x2 max-pooled map
        if H4 % 2 or W4 % 2:
            raise NotImplementedError("RELATION_SUB_SAMPLE on an odd map")
        Hr, Wr = H4 // 2, W4 // 2
        xr = p.new(Hr, Wr, c0, pinned=True, name="trp.pooled")      # pinned: relation scores are computed on demand

        def make_pool():
            a = (fvis.ptr(), fvis.ld, xr.ptr(), xr.ld, B, H4, W4, c0, dt)
            return lambda s: hip.check(lib.rsg_maxpool2x2_nhwc16(*a, s), "maxpool2x2")

        p.custom("trp_maxpool", [fvis], [xr], make_pool)
    else:
        Hr, Wr, xr = H4, W4, fvis
    Sr = Hr * Wr
    p.S_rel, p.trp_x = Sr, xr
    gT = p.new(c0, Sr, 1, name="gT")
    p.e32 = torch.zeros(B, Sr, c0, **f32)

    def make_g():
        a = (xr.ptr(), xr.ld, wg.data_ptr(), bg.data_ptr(), gT.ptr(), None, B, Sr, c0, dt)
        return lambda s: hip.check(lib.rsg_trp_project_g(*a, s), "trp_project_g")

    def make_attn():
        a = (xr.ptr(), xr.ld, gT.ptr(), None, p.e32.data_ptr(), B, Sr, c0, dt)
        return lambda s: hip.check(lib.rsg_trp_attention(*a, s), "trp_attention")

    rel = cat2.view(0, c0)

    p.tail_ws = torch.zeros(lib.rsg_trp_tail_workspace_bytes(B, S, c0) // 4, **f32)
    p.custom("trp_project_g", [xr], [gT], make_g)
    p.custom("trp_attention", [xr, gT], [p.e32], make_attn)
    p.flops += B * (2 * 2 * Sr * Sr * c0 + 2 * 2 * Sr * c0 * c0)
    if sub:
        # E (fp32) -> 16 bits -> ConvTranspose2d(4, 2, 1) as four sub-pixel 2x2 convs (+BatchNorm, ReLU folded) -> W
        e16 = p.new(Hr, Wr, c0, name="trp.e16")

        def make_cast():
            a = (p.e32.data_ptr(), e16.ptr(), e16.ld, B * Sr, c0, dt)
            return lambda s: hip.check(lib.rsg_f32_rows_to_nhwc16(*a, s), "f32_rows_to_nhwc16")

        p.custom("trp_e_to_16", [p.e32], [e16], make_cast)
        up = p.new(H4, W4, c0, name="trp.up")
        dcv, dbn_ = rh.W[0][0], rh.W[0][1]
        p.begin_group()
        for py, px, pad_y, pad_x, wph in deconv_phase_weights(dcv.weight):
            p.conv(e16, wph, dbn_, None, relu=True, out=up, pad=(pad_y, pad_x), up=(2, py, px),
                   out_hw=(Hr, Wr), phys_hw=(H4, W4), name=f"trp.up.phase{py}{px}")
        p.end_group("trp.up(deconv)")

        def make_tail():
            a = (up.ptr(), up.ld, ww.data_ptr(), bw.data_ptr(), gam.data_ptr(), bet.data_ptr(), rel.ptr(),
                 rel.ld, B, S, c0, dt, 0, p.tail_ws.data_ptr())           # keep_y = 0: nothing reads the fp32 projection in inference
            return lambda s: hip.check(lib.rsg_trp_tail_ex(*a, s), "trp_tail_in16")

        p.custom("trp_tail", [up], [rel], make_tail)
        p.flops += B * 2 * S * c0 * c0
    else:
        def make_tail():
            a = (p.e32.data_ptr(), 0, ww.data_ptr(), bw.data_ptr(), gam.data_ptr(), bet.data_ptr(), rel.ptr(),
                 rel.ld, B, S, c0, dt, 0, p.tail_ws.data_ptr())           # keep_y = 0
            return lambda s: hip.check(lib.rsg_trp_tail_ex(*a, s), "trp_tail")

        p.custom("trp_tail", [p.e32], [rel], make_tail)

    # ---- joints-of-target branch (:991-1000); the limbs branch = SGM (:1003-1005) is independent of it and of the TRP, so
    # its two convs ride in this branch's launches (limbs_net beside kpt_net, the limb classifier beside the deconv phases)
    ride = sw.plan_on("limbs_ride")
    p.limbs_lo = torch.zeros(B, L, H4, W4, **f32)
    p.begin_group()
    k = p._unit(cat2, model.kpt_net[0], model.kpt_net[1], relu=True, name="kpt_net")
    lf = p._unit(fvis, model.limbs_net[0], model.limbs_net[1], relu=True, name="limbs_net") if ride else None
    p.end_group("kpt_net+limbs_net")
    dc, dbn = model.predict_convtranspose[0], model.predict_convtranspose[1]
    k2 = p.new(2 * H4, 2 * W4, c0, name="deconv")       # the four phases store exactly c0 channels
    p.begin_group()                                      # four independent sub-pixel convs: one launch
    for py, px, pad_y, pad_x, wph in deconv_phase_weights(dc.weight):
        p.conv(k, wph, dbn, None, relu=True, out=k2, pad=(pad_y, pad_x), up=(2, py, px),
               out_hw=(H4, W4), phys_hw=(2 * H4, 2 * W4), name=f"deconv.phase{py}{px}")
    if ride:
        p.conv(lf, limb_classifier(model), None, None, out_f32=p.limbs_lo, want16=False, name="limbs(sgm)")
    p.end_group("deconv+limbs(sgm)" if ride else "deconv")
    # predict_net -> final_layer (:999-1000) as ONE launch when the head is 32 channels wide: the 3x3 conv's accumulators feed the
    # classifier's MFMAs in registers (rsg_conv_tail1x1); the 50 MB map between them (64 crops at 128x96) is never stored
    head_tail = sw.plan_on("head_tail") and c0 == 32 and J <= 32 and model.predict_net[0].kernel_size == (3, 3) and \
        model.final_layer.kernel_size == (1, 1)
    k3 = None if head_tail else p._unit(k2, model.predict_net[0], model.predict_net[1], relu=True, name="predict_net")

    # ---- limbs branch = SGM (:1003-1013)
    p.kpt = torch.zeros(B, J, 2 * H4, 2 * W4, **f32)
    p.multi_up = torch.zeros(B, J, 2 * H4, 2 * W4, **f32)
    p.limbs = torch.zeros(B, L, 2 * H4, 2 * W4, **f32)
    from .plan import OutSlot
    p.kpt_slot, p.multi_slot, p.limbs_slot = OutSlot(p.kpt), OutSlot(p.multi_up), OutSlot(p.limbs)

    def make_up():              # both x2 up-samplings (:1009-1013) in one launch: the multi-joint scores, then the limb maps through the sigmoid
        ma, la = p.multi_lo.data_ptr(), p.limbs_lo.data_ptr()
        return lambda s: hip.check(lib.rsg_bilinear2x_pair_nchw_f32(ma, p.multi_slot.ptr, B * J, 0, la, p.limbs_slot.ptr, B * L, 1,
                                                                    H4, W4, s), "bilinear2x_pair")

    if not ride:
        with p.on_lane(1):      # independent of TRP and the joints-of-target branch: its own stream
            lf = p._unit(fvis, model.limbs_net[0], model.limbs_net[1], relu=True, name="limbs_net")
            p.conv(lf, limb_classifier(model), None, None, out_f32=p.limbs_lo, want16=False, name="limbs(sgm)")
    # the two ops that write the model's outputs come last: they run eagerly behind the captured graph and write
    # straight into the tensors a forward hands out (Plan.n_tail, OutSlot)
    fl = model.final_layer
    if head_tail:
        p.conv_tail1x1(k2, model.predict_net[0], model.predict_net[1], fl.weight, fl.bias, p.kpt_slot, name="predict_net+final_layer")
    else:
        p.conv(k3, fl.weight, None, fl.bias, out_f32=p.kpt_slot, want16=False, name="final_layer")
    p.custom("bilinear2x(multi)+bilinear2x+sigmoid(limbs)", [p.multi_lo, p.limbs_lo], [p.multi_up, p.limbs], make_up)
    p.n_tail = 2

    # the constant location clue: filled once, right after allocation, for every crop slot
    orig_finalize = p.finalize

    def finalize_with_loc():
        orig_finalize()
        dst = cat.tensor(B)[..., 2 * c0: 3 * c0]
        dst.copy_(loc[0].permute(1, 2, 0).to(dev).to(p.tdtype).unsqueeze(0).expand(B, -1, -1, -1))

    p.finalize = finalize_with_loc

    # ---- on-demand relation outputs -----------------------------------------------------------------
    def relation_scores() -> torch.Tensor:
        out = torch.empty(B, Sr, Sr, **f32)
        st = torch.cuda.current_stream(dev).cuda_stream
        hip.check(lib.rsg_trp_scores(xr.ptr(), xr.ld, out.data_ptr(), B, Sr, c0, dt, st), "trp_scores")
        return out

    def relation_mse(target: torch.Tensor) -> torch.Tensor:
        """pose_rsgnet.py:1014-1018 for the factored target p [B,S] (T = p p^T) or the reference's [B,S,S] tensor
        (lib/core/function.py:269), read tile by tile: nothing S x S is materialised either way."""
        st = torch.cuda.current_stream(dev).cuda_stream
        tv = target.to(dev, torch.float32).contiguous()
        if tv.dim() not in (2, 3) or tuple(tv.shape) != (B,) + (Sr,) * (tv.dim() - 1):
            raise ValueError(f"relation_target of shape {tuple(tv.shape)} for batch {B} and {Sr} positions")
        out = torch.empty(B, **f32)
        ws = torch.empty(int(lib.rsg_trp_mse_workspace_bytes(B, Sr)), dtype=torch.uint8, device=dev)
        fn = lib.rsg_trp_relation_mse if tv.dim() == 2 else lib.rsg_trp_relation_mse_dense
        hip.check(fn(xr.ptr(), xr.ld, tv.data_ptr(), out.data_ptr(), ws.data_ptr(), B, Sr, c0, dt, st), "trp_relation_mse")
        return out

    p.relation_scores = relation_scores
    p.relation_mse = relation_mse
