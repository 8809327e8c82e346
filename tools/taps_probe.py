#!/usr/bin/env python3
"""Why does an eager pass WITHOUT syncs give a garbage stage2 tap?  Variants: clones or not, sync or not; hand-off counter."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from util import load_case  # noqa: E402
from rsgnet_amd import hip  # noqa: E402

cfg, model, sd, x, gold = load_case("hrnet_w32_256x192")
model = model.cuda()
xc = x.cuda()
with torch.no_grad():
    y = model(xc)
plan = model.plan_for(xc)
lib = hip.lib()


def timeouts():
    n = ctypes.c_uint32(0)
    hip.check(lib.rsg_chain_handoff_timeouts(ctypes.byref(n)))
    return n.value


def run(clone_names, sync):
    plan._bind_input(xc)
    st = torch.cuda.current_stream().cuda_stream
    got = {}
    for (name, op), writes in zip(plan.ops, plan._op_writes):
        op(st)
        if sync:
            torch.cuda.synchronize()
        for v in writes:
            if v.name in clone_names:
                got[v.name] = v.tensor(plan.batch).clone()
    torch.cuda.synchronize()
    return got


NAMES4 = {"stage2.0.fuse.0", "layer1.3.conv2+conv3", "stage3.3.fuse.0", "stage4.2.fuse.0"}
first = run(NAMES4, False)            # like the test: the FIRST eager pass after the capture, unsynchronised, with clones
heat_first = plan.heat_slot.static.clone() if hasattr(plan, "heat_slot") else None
print("first unsynced pass:", {k: bool(torch.isfinite(v.float()).all()) for k, v in first.items()})
ref = run(NAMES4, True)
print("vs synced:", {k: bool(torch.equal(first[k], ref[k])) for k in ref})
print("timeouts", timeouts())
for label, names in (("all three", {"stage2.0.fuse.0", "layer1.3.conv2+conv3", "stage3.3.fuse.0"}), ("stage2 only", {"stage2.0.fuse.0"}),
                     ("stage3 only", {"stage3.3.fuse.0"})):
    for rep in range(2):
        got = run(names, False)
        print(label, rep, {k: (bool(torch.isfinite(v).all()), bool(torch.equal(v, ref[k]))) for k, v in got.items()}, "timeouts", timeouts())
out = plan.heat_slot.static if hasattr(plan, "heat_slot") else None
print("final heat finite", bool(torch.isfinite(plan.heat).all()) if hasattr(plan, "heat") else None)
