#!/usr/bin/env python3
"""Bit-identity soak of overlapped forwards: N forwards issued on two alternating streams, three inputs, every output
compared with the one-stream result of the same input.  usage: stream_soak.py <tag> <forwards> [batch multiplier]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import load_case

tag = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w32_256x192"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
mult = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cfg, model, sd, x, gold = load_case(tag)
model = model.cuda().eval()
model.compute_dtype = os.environ.get("SOAK_DTYPE", "fp16")
xb = torch.cat([x] * mult)
xs = [xb.cuda(), torch.flip(xb, dims=[3]).contiguous().cuda(), (0.5 * xb).cuda()]
tens = lambda o: [t for t in ((o,) if torch.is_tensor(o) else o) if torch.is_tensor(t)]
names = ["heat"] if tag.startswith("hrnet") else ["multi", "kpt", "limbs"]
with torch.no_grad():
    want = [[t.clone() for t in tens(model(xi))] for xi in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bad = {}
    done = 0
    while done < n:
        got = []
        for k in range(24):
            with torch.cuda.stream(streams[k % 2]):
                got.append((k % 3, [t.clone() for t in tens(model(xs[k % 3]))]))
        torch.cuda.synchronize()
        for w_, g in got:
            for nm, a, b in zip(names, g, want[w_]):
                if not torch.equal(a, b):
                    bad[nm] = bad.get(nm, 0) + 1
        done += 24
import ctypes
from rsgnet_amd import hip
n_to = ctypes.c_uint32(0)
hip.check(hip.lib().rsg_chain_handoff_timeouts(ctypes.byref(n_to)))
uses = any(p._ho_flags for p in model._plans.values())
print(f"{tag} {model.compute_dtype} batch {xs[0].shape[0]}: {done} overlapped forwards, outputs that differ from the one-stream result: {bad or 'none'}; "
      f"in-launch hand-offs {'in use' if uses else 'not used'}, consumers that gave up waiting: {n_to.value}")
