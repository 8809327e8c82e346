#!/usr/bin/env python3
"""GPU box probe for the capture of the RSGNet training head (advisor finding of round 3: the head's ~300 launches per step
are enqueued eagerly because `torch.cuda.make_graphed_callables` ended in a segmentation fault; cause unknown).

  python tools/head_capture_probe.py            runs every stage below in its own child process and prints one line each
  python tools/head_capture_probe.py STAGE      runs one stage in this process

Stages (each after three eager warm-up steps on a side stream, PyTorch's own recipe):
  torch_only   make_graphed_callables on a small PyTorch-only module (is the facility itself sound on this ROCm?)
  fwd          torch.cuda.graph around the head's forward, no autograd
  fwdbwd       torch.cuda.graph around the head's forward AND torch.autograd.grad (the backward's launches come from
               autograd's device thread while the capture was begun on the main thread)
  mgc          torch.cuda.make_graphed_callables on the head (forward and backward graphs, as round 3 tried)
A child that dies prints the native frames of the faulting thread (tools/probes/segv_backtrace.c) and Python's tracebacks
(faulthandler)."""
import ctypes
import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAGES = ("torch_only", "fwd", "fwdbwd", "mgc")


def _install_handlers():
    so = os.path.join(ROOT, "tools", "probes", "segv_backtrace.so")
    src = os.path.join(ROOT, "tools", "probes", "segv_backtrace.c")
    if not os.path.exists(so):
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-g", "-o", so, src])
    faulthandler.enable(all_threads=True)          # Python frames first ...
    ctypes.CDLL(so).install()                      # ... then ours on top: native frames, then the default action


def _head(batch=8):
    import numpy as np
    import torch
    from rsgnet_amd import make_cfg, models
    from rsgnet_amd.train_rsgnet import head_forward
    from rsgnet_amd.train import train_forward as backbone_forward
    from rsgnet_amd.weights import synth_input, synth_state_dict
    cfg = make_cfg("pose_rsgnet", 32)
    model = models.pose_rsgnet.get_pose_net(cfg, is_train=False)
    bn = dict(np.load(os.path.join(ROOT, "tests", "golden", "bnstats_rsgnet_w32_256x192_coco.npz")))
    model.load_state_dict(synth_state_dict(model.state_dict(), 0, bn))
    model = model.to("cuda:0").train()
    x = synth_input(batch, (192, 256), 3).to("cuda:0")
    with torch.no_grad():
        feat = backbone_forward(model, x).detach().clone()
    s = feat.shape[1] * feat.shape[2]
    p = torch.rand(batch, s, device="cuda:0")

    class Head(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, f, p):
            return head_forward(self.m, f, p)

    return Head(model), feat, p


def stage(name):
    import torch
    _install_handlers()
    dev = torch.device("cuda:0")
    if name == "torch_only":
        net = torch.nn.Sequential(torch.nn.Conv2d(32, 32, 3, padding=1), torch.nn.BatchNorm2d(32), torch.nn.ReLU()).to(dev)
        x = torch.randn(8, 32, 64, 48, device=dev, requires_grad=True)
        g = torch.cuda.make_graphed_callables(net, (x,))
        y = g(x)
        y.sum().backward()
        torch.cuda.synchronize()
        print("torch_only: captured, replayed forward and backward", flush=True)
        return
    head, feat, p = _head()
    params = [q for q in head.parameters() if q.requires_grad]
    if name == "fwd":
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(3):
                ref0 = head(feat, p)
                ref = head(feat, p)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g):
            out = head(feat, p)
        g.replay()
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(out, ref))
        print(f"fwd: captured and replayed; outputs equal to eager: {same}", flush=True)
        for k, (a, b, c) in enumerate(zip(out, ref, ref0)):
            print(f"  output {k} {tuple(a.shape)}: max |graph - eager| {(a.float() - b.float()).abs().max().item():.3e}, "
                  f"max |eager - eager| {(c.float() - b.float()).abs().max().item():.3e}", flush=True)
        return
    f = feat.clone().requires_grad_(True)
    if name == "fwdbwd":
        def step():
            out = head(f, p)
            loss = sum(o.float().pow(2).mean() for o in out)
            return loss, torch.autograd.grad(loss, [f] + params, allow_unused=True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                loss_ref, grads_ref = step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss, grads = step()
        g.replay()
        torch.cuda.synchronize()
        same = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(grads, grads_ref))
        _, grads_ref2 = step()
        names = ["feat"] + [n for n, q in head.named_parameters() if q.requires_grad]
        for n, a, b, c in zip(names, grads, grads_ref, grads_ref2):
            if a is None or torch.equal(a, b):
                continue
            den = b.float().norm().item() + 1e-30
            print(f"  d{n}: |graph - eager| / |eager| {(a.float() - b.float()).norm().item() / den:.3e}, "
                  f"|eager - eager| / |eager| {(c.float() - b.float()).norm().item() / den:.3e}", flush=True)
        print(f"fwdbwd: captured and replayed; loss {loss.item():.6f} (eager {loss_ref.item():.6f}); gradients equal to eager: {same}",
              flush=True)
        return
    if name == "mgc":
        gh = torch.cuda.make_graphed_callables(head, (f, p), allow_unused_input=True)
        out = gh(f, p)
        sum(o.float().pow(2).mean() for o in out).backward()
        torch.cuda.synchronize()
        print(f"mgc: captured, replayed forward and backward; |df| {f.grad.float().norm().item():.4e}", flush=True)
        return
    raise SystemExit(f"unknown stage {name}")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        stage(sys.argv[1])
        sys.exit(0)
    rc = {}
    for s in STAGES:
        print(f"---- {s}", flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), s], timeout=600)
        rc[s] = r.returncode
        print(f"---- {s}: exit {r.returncode}", flush=True)
    print(rc)
