#!/usr/bin/env python3
"""Why did RSGNet's outputs differ between the default stream's plan and a side stream's?  (1) rebuilt plans on one stream,
(2) a side stream's plan with nothing else running, (3) two streams overlapping."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import load_case

tag = sys.argv[1] if len(sys.argv) > 1 else "rsgnet_w32_256x192_coco"
cfg, model, sd, x, gold = load_case(tag)
model = model.cuda().eval()
xs = [x.cuda(), torch.flip(x, dims=[3]).contiguous().cuda(), (0.5 * x).cuda()]
names = ["multi", "kpt", "limbs"]


def outs(xi):
    o = model(xi)
    return [t.clone() for t in ((o,) if torch.is_tensor(o) else o) if torch.is_tensor(t)]


def diff(a, b):
    return [f"{names[i] if len(a) > 1 else 'heat'}:{float((p - q).abs().max()):.2e}" for i, (p, q) in enumerate(zip(a, b)) if not torch.equal(p, q)]


with torch.no_grad():
    want = [outs(xi) for xi in xs]
    torch.cuda.synchronize()
    print("(0) same plan again:", [diff(outs(xi), w) for xi, w in zip(xs, want)])
    for rep in range(4):
        model.invalidate_plan()
        got = [outs(xi) for xi in xs]
        torch.cuda.synchronize()
        print(f"(1) rebuilt plan {rep}, default stream:", [diff(g, w) for g, w in zip(got, want)])
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for s_ in (s1, s2):
        with torch.cuda.stream(s_):
            got = [outs(xi) for xi in xs]
        torch.cuda.synchronize()
        print("(2) side stream alone:", [diff(g, w) for g, w in zip(got, want)])
    for rep in range(10):
        got = []
        for k in range(12):
            with torch.cuda.stream((s1, s2)[k % 2]):
                got.append((k % 3, outs(xs[k % 3])))
        torch.cuda.synchronize()
        bad = [(k, w, diff(g, want[w])) for k, (w, g) in enumerate(got) if diff(g, want[w])]
        print(f"(3) two streams overlapping, round {rep}:", bad)
        for k, (w, g) in enumerate(got):
            for i, (p_, q_) in enumerate(zip(g, want[w])):
                if not torch.equal(p_, q_):
                    idx = (p_ != q_).nonzero()
                    d = (p_ - q_).abs()
                    print(f"      launch {k} output {i} shape {tuple(p_.shape)}: {idx.shape[0]} elements differ; images {idx[:, 0].unique().tolist()}, "
                          f"channels {idx[:, 1].unique().tolist()[:20]}, rows {int(idx[:, 2].min())}..{int(idx[:, 2].max())}, cols {int(idx[:, 3].min())}..{int(idx[:, 3].max())}, "
                          f"max {float(d.max()):.3e}, mean over differing {float(d[p_ != q_].mean()):.3e}")
    # which launch differs first: taps are not available across streams; compare the plans' constants instead
    plans = list(model._plans.values())
    print("plans:", len(plans))
