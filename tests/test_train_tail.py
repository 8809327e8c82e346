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
    q.put((rank, bool(torch.equal(g, want)), len(red.buckets), red.grad_scale))
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
