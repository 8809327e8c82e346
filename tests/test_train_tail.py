"""Training tail: host logic on CPU (gloo), HIP kernels against the oracle."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import train_oracle as to  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "train_tail.npz")


def _reducer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rsgnet_amd.parallel import FlatGradReducer
    n = 1000003
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = FlatGradReducer(g, bucket_bytes=1 << 20)
    red.reduce()
    want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = bool(torch.equal(g, want))
    # overlap protocol: the tail of the buffer is announced final first (its buckets go out at once), the rest at the
    # end -- every bucket must be exchanged exactly once (a second all-reduce would double it)
    g2 = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red2 = FlatGradReducer(g2, bucket_bytes=1 << 20)
    red2.begin()
    red2.ready(n // 2)
    early = sum(red2._done)
    red2.ready(n // 4)
    red2.reduce()
    ok = ok and bool(torch.equal(g2, want)) and 0 < early < len(red2.buckets) and not any(red2._done)
    q.put((rank, ok, len(red.buckets), red.grad_scale))
    dist.destroy_process_group()


def test_flat_grad_reducer_two_ranks_gloo():
    """Bucketed all-reduce of a flat gradient buffer, world size 2 (gloo on CPU): every element is the sum over
    ranks, buckets cover the buffer exactly once (from its end), grad_scale = 1/world."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(30)
    assert all(ok for _, ok, _, _ in res), res
    assert res[0][2] == 4 and res[0][3] == 0.5


def test_losses_refuse_cpu_tensors():
    from rsgnet_amd.loss import JointsMSELoss
    with pytest.raises(RuntimeError):
        JointsMSELoss(True)(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4), torch.ones(1, 2, 1))


@pytest.mark.gpu
def test_joints_mse_and_bce_match_goldens():
    """Fused value+gradient kernels against the reference's JointsMSELoss / torch BCELoss fixtures."""
    from rsgnet_amd.loss import JointsMSELoss, limbs_bce_loss, relation_target_vector
    g = np.load(GOLD)
    pred = torch.from_numpy(g["mse_pred"]).cuda().requires_grad_(True)
    gt, w = torch.from_numpy(g["mse_gt"]).cuda(), torch.from_numpy(g["mse_w"]).cuda()
    loss = JointsMSELoss(True)(pred, gt, w)
    loss.backward()
    assert abs(loss.item() - float(g["mse_loss"])) < 1e-6 * float(g["mse_loss"])
    assert np.abs(pred.grad.cpu().numpy() - g["mse_grad"]).max() < 1e-6 * np.abs(g["mse_grad"]).max()
    pred2 = torch.from_numpy(g["mse_pred"]).cuda().requires_grad_(True)
    loss2 = JointsMSELoss(False)(pred2, gt, w) * 3.0          # upstream gradient is applied
    loss2.backward()
    assert abs(loss2.item() / 3.0 - float(g["mse_loss_now"])) < 1e-6 * float(g["mse_loss_now"])
    assert np.abs(pred2.grad.cpu().numpy() / 3.0 - g["mse_grad_now"]).max() < 1e-6 * np.abs(g["mse_grad_now"]).max()
    prob = torch.from_numpy(g["bce_prob"]).cuda().requires_grad_(True)
    bl = limbs_bce_loss(prob, torch.from_numpy(g["bce_t"]).cuda(), 0.01)
    bl.backward()
    assert abs(bl.item() - 0.01 * float(g["bce_loss"])) < 2e-6 * 0.01 * float(g["bce_loss"])
    err = np.abs(prob.grad.cpu().numpy() - 0.01 * g["bce_grad"])
    assert (err <= 2e-6 * np.abs(0.01 * g["bce_grad"]) + 1e-12).all()
    p = relation_target_vector(gt)
    assert np.abs(p.cpu().numpy() - g["rel_p"]).max() < 1e-6


@pytest.mark.gpu
def test_fused_adam_matches_goldens_and_oracle():
    """Three steps of the flat HIP Adam against torch.optim.Adam fixtures, then against the oracle on odd sizes."""
    from rsgnet_amd.optim import FusedAdam
    g = np.load(GOLD)
    p = torch.nn.Parameter(torch.from_numpy(g["adam_p0"]).cuda())
    opt = FusedAdam([p], lr=1e-3)
    for k in range(3):
        opt.zero_grad()
        p.grad.copy_(torch.from_numpy(g["adam_grads"][k]).cuda())
        opt.step()
        assert np.abs(p.detach().cpu().numpy() - g[f"adam_p{k + 1}"]).max() < 2e-6
    ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in ((7, 3), (33,), (5, 5, 2))]
    ref = [(q.detach().cpu().double().numpy().copy(), 0.0, 0.0) for q in ps]
    opt = FusedAdam(ps, lr=3e-3)
    opt.grad_scale = 0.5
    for step in range(1, 4):
        opt.zero_grad()
        gs = [torch.randn_like(q) for q in ps]
        for q, gg in zip(ps, gs):
            q.grad.copy_(gg)
        opt.step()
        ref = [to.adam_step(r[0], 0.5 * gg.cpu().double().numpy(), r[1], r[2], step, lr=3e-3) for r, gg in zip(ref, gs)]
        for q, r in zip(ps, ref):
            assert np.abs(q.detach().cpu().numpy() - r[0]).max() < 5e-6


@pytest.mark.gpu
def test_fused_adam_checkpoint_round_trip_against_torch_adam():
    """save -> load -> step in both directions with torch.optim.Adam (the reference saves optimizer.state_dict(),
    tools/train.py:223, and resumes with load_state_dict): moments and step counts survive, a parameter whose grad is
    None is skipped exactly as torch skips it, and a torch.optim.Adam checkpoint loads into FusedAdam."""
    from rsgnet_amd.optim import FusedAdam
    torch.manual_seed(3)
    shapes = ((7, 3), (33,), (4, 4), (5, 5, 2), (1,))
    FROZEN = 2                                         # like RSGNet's loc_features: in the index space, never stepped
    init = [torch.randn(s) for s in shapes]

    def mk(tensors):
        out = [torch.nn.Parameter(t.detach().clone().cuda()) for t in tensors]
        out[FROZEN].requires_grad_(False)
        return out

    grads = [[torch.randn(s) for s in shapes] for _ in range(6)]

    def run(opt, ps, steps, skip=None):
        for k in steps:
            for i, (q, g) in enumerate(zip(ps, grads[k])):
                if i == FROZEN:
                    continue
                if skip is not None and i == skip and k % 2 == 1:
                    q.grad = None                      # an unused parameter on odd steps
                elif q.grad is None:
                    q.grad = g.to(q.device).clone()
                else:
                    q.grad.copy_(g)
            opt.step()

    ref_p = mk(init)
    ref = torch.optim.Adam(ref_p, lr=2e-3)
    run(ref, ref_p, range(6), skip=1)

    # (a) FusedAdam alone, same schedule, grad None on the same steps
    ps = mk(init)
    opt = FusedAdam(ps, lr=2e-3)
    run(opt, ps, range(3), skip=1)
    sd = opt.state_dict()
    assert set(sd["state"]) == {0, 1, 3, 4} and sd["param_groups"][0]["params"] == [0, 1, 2, 3, 4] and set(sd["state"][1]) == {"step", "exp_avg", "exp_avg_sq"}
    assert float(sd["state"][1]["step"]) == 2.0 and float(sd["state"][0]["step"]) == 3.0
    # (b) resume in a NEW FusedAdam from that checkpoint
    ps2 = mk(ps)
    opt2 = FusedAdam(ps2, lr=1.0)
    opt2.load_state_dict(sd)
    assert opt2.param_groups[0]["lr"] == 2e-3 and opt2.step_count == 3
    run(opt2, ps2, range(3, 6), skip=1)
    for q, r in zip(ps2, ref_p):
        assert (q.detach() - r.detach()).abs().max().item() < 5e-6
    # (c) the FusedAdam checkpoint loads into torch.optim.Adam and continues identically
    ps3 = mk(ps)
    t3 = torch.optim.Adam(ps3, lr=1.0)
    t3.load_state_dict(sd)
    run(t3, ps3, range(3, 6), skip=1)
    for q, r in zip(ps3, ref_p):
        assert (q.detach() - r.detach()).abs().max().item() < 5e-6
    # (d) a torch.optim.Adam checkpoint loads into FusedAdam
    half_p = mk(init)
    half = torch.optim.Adam(half_p, lr=2e-3)
    run(half, half_p, range(3), skip=1)
    ps4 = mk(half_p)
    opt4 = FusedAdam(ps4, lr=1.0)
    opt4.load_state_dict(half.state_dict())
    run(opt4, ps4, range(3, 6), skip=1)
    for q, r in zip(ps4, ref_p):
        assert (q.detach() - r.detach()).abs().max().item() < 5e-6


@pytest.mark.gpu
def test_fused_adam_resume_under_multisteplr_like_the_reference_trainer():
    """tools/cp_train.py:179-193: optimizer.load_state_dict(checkpoint['optimizer']) and then
    MultiStepLR(optimizer, LR_STEP, LR_FACTOR, last_epoch=checkpoint['epoch']).  torch raises KeyError("param 'initial_lr'
    is not specified ...") unless the loaded param_group still carries the scheduler's `initial_lr` (FusedAdam used to
    drop it).  The same save / resume procedure with torch.optim.Adam is the expected behaviour: learning rates equal at
    every epoch, parameters equal to Adam's rounding."""
    from torch.optim.lr_scheduler import MultiStepLR
    from rsgnet_amd.optim import FusedAdam
    torch.manual_seed(5)
    shapes = ((6, 3), (17,), (2, 3, 3))
    init = [torch.randn(s) for s in shapes]
    grads = [[torch.randn(s) for s in shapes] for _ in range(6)]
    mk = lambda ts: [torch.nn.Parameter(t.detach().clone().cuda()) for t in ts]

    def epoch(opt, ps, k):
        for q, g in zip(ps, grads[k]):
            if q.grad is None:
                q.grad = g.cuda().clone()
            else:
                q.grad.copy_(g)
        opt.step()

    def procedure(make_opt):
        ps = mk(init)
        opt = make_opt(ps)
        sch = MultiStepLR(opt, [2, 4], 0.1)
        lrs = []
        for k in range(3):
            sch.step()                                    # the reference steps the scheduler at the top of an epoch
            epoch(opt, ps, k)
            lrs.append(opt.param_groups[0]["lr"])
        ckpt = dict(epoch=3, optimizer=opt.state_dict(), params=[q.detach().clone() for q in ps])
        assert "initial_lr" in ckpt["optimizer"]["param_groups"][0]
        ps2 = mk(ckpt["params"])                          # a NEW process: optimizer from the config's lr, then the checkpoint
        opt2 = make_opt(ps2)
        opt2.load_state_dict(ckpt["optimizer"])
        assert opt2.param_groups[0]["initial_lr"] == 1e-2
        sch2 = MultiStepLR(opt2, [2, 4], 0.1, last_epoch=ckpt["epoch"])      # KeyError before the fix
        for k in range(3, 6):
            sch2.step()
            epoch(opt2, ps2, k)
            lrs.append(opt2.param_groups[0]["lr"])
        return lrs, ps2

    lr_t, p_t = procedure(lambda ps: torch.optim.Adam(ps, lr=1e-2))
    lr_f, p_f = procedure(lambda ps: FusedAdam(ps, lr=1e-2))
    assert lr_f == pytest.approx(lr_t, rel=1e-12) and len(set(lr_f)) > 1
    for q, r in zip(p_f, p_t):
        assert (q.detach() - r.detach()).abs().max().item() < 5e-6
    with pytest.raises(ValueError):                       # still no weight decay on this path
        bad = FusedAdam(mk(init), lr=1e-2).state_dict()
        bad["param_groups"][0]["weight_decay"] = 1e-4
        FusedAdam(mk(init), lr=1e-2).load_state_dict(bad)


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rsgnet_amd.parallel import DataParallelStep
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    crit = lambda o, t, w: 0.5 * ((o - t) * w).pow(2).mean()
    step = DataParallelStep(net, crit, opt)
    g = torch.Generator().manual_seed(7)
    x, t = torch.randn(4, 3, 8, 6, generator=g), torch.randn(4, 2, 8, 6, generator=g)
    w = torch.ones(4, 2, 1, 1)
    sl = slice(2 * rank, 2 * rank + 2)
    for _ in range(3):
        step(x[sl], t[sl], w[sl])
    q.put((rank, [p.detach().numpy().copy() for p in net.parameters()]))   # arrays, not tensors: no fd hand-over after exit
    dist.destroy_process_group()


def test_data_parallel_step_two_ranks_equals_one_process_gloo():
    """DataParallelStep (forward, criterion, zero_grad, backward, all-reduce/world, step) on two gloo ranks with half the
    batch each ends where one process with the whole batch ends (generic path: a torch optimizer, grads averaged)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(30)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(7)
    x, t = torch.randn(4, 3, 8, 6, generator=g), torch.randn(4, 2, 8, 6, generator=g)
    for _ in range(3):
        loss = 0.5 * (net(x) - t).pow(2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
    for r in (0, 1):
        for a, b in zip(res[r], net.parameters()):
            assert np.abs(a - b.detach().numpy()).max() < 1e-6
