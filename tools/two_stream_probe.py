#!/usr/bin/env python3
"""GPU box: is a batch of 64 faster as two half-batches of 32 whose captured graphs replay on TWO streams (kernel-boundary
gaps and tails of one half overlapping the other) than as one graph of 64?  Prints ms per 64 crops for both."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rsgnet_amd import make_cfg, models  # noqa: E402
from rsgnet_amd.weights import synth_input, synth_state_dict  # noqa: E402


def make(tag="hrnet_w32_256x192"):
    cfg = make_cfg("pose_hrnet", 32)
    m = models.pose_hrnet.get_pose_net(cfg, is_train=False)
    bn = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"bnstats_{tag}.npz")))
    m.load_state_dict(synth_state_dict(m.state_dict(), 0, bn))
    return m.eval().cuda()


def timed(fn, steps=30, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


with torch.no_grad():
    one = make()
    x64 = synth_input(64, (192, 256), 0).cuda()
    print("one graph, batch 64: %.3f ms" % timed(lambda: one(x64)))
    a, b = make(), make()
    xa, xb = x64[:32].contiguous(), x64[32:].contiguous()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        with torch.cuda.stream(sa):
            a(xa)
        with torch.cuda.stream(sb):
            b(xb)
    # the plans capture on first use: do that serially
    a(xa); b(xb); torch.cuda.synchronize()
    print("two graphs of 32 on two streams: %.3f ms" % timed(both))
    print("two graphs of 32 on one stream: %.3f ms" % timed(lambda: (a(xa), b(xb))))
