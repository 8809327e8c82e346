"""rsg_fuse_module (csrc/fuse_module.hip): level 0 of every HighResolutionModule's fuse layers as one launch of row bands,
against the grouped launches it replaces (csrc/group.hip) -- the reference's `fuse_layers` / `HighResolutionModule.forward`
(/root/reference/lib/models/pose_hrnet.py:190-268).  The bar is bit-identity of every 16-bit map either path stores."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from rsgnet_amd import hip
from tests.util import load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _plans(model, batch):
    """(plan with the band launch, plan with the grouped launches) of one model."""
    from rsgnet_amd import switches
    out = []
    for v in ("1", "0"):
        with switches.override("plan", fuse_module=v):
            out.append(model._build_plan(batch, torch.device(DEV)))
    return out


def _fuse_maps(plan):
    return sorted({v.name for w in plan._op_writes for v in w if ".fuse." in v.name and "band" not in v.name})


@pytest.mark.parametrize("tag,dtype,batch", [("hrnet_w32_256x192", "fp16", 3), ("hrnet_w32_256x192", "bf16", 2),
                                             ("rsgnet_w32_256x192_crowdpose", "fp16", 2), ("hrnet_w48_384x288", "fp16", 2),
                                             ("rsgnet_w48_256x192_crowdpose_sub", "fp16", 3)])
def test_fuse_module_equals_the_grouped_launches(tag, dtype, batch):
    """Every fuse map of every module (outputs, and the intermediate maps of the strided chains) and the model's outputs:
    the same bits from the band launch + the remaining grouped levels as from the grouped launches alone.  Batch 3 / 2:
    bands of different pictures share a workgroup index range; 256x192 crops: 4 bands of 16 rows per picture (HRNet-W48's
    widths, 48 / 96 / 192 / 384 channels with the 48-channel maps in 64-channel buffers: bands of 8 rows at 384x288)."""
    from rsgnet_amd.weights import synth_input
    cfg, model, sd, x, gold = load_case(tag)
    model = model.to(DEV)
    model.compute_dtype = dtype
    from tests.util import TAGS
    x = synth_input(batch, TAGS[tag][2], 11).to(DEV)
    new, old = _plans(model, batch)
    names_new = [n for n, _ in new.ops]
    assert any(n.endswith(".fuse.band") for n in names_new), names_new
    assert not any(n.endswith(".fuse.band") for n, _ in old.ops)
    assert len(new.ops) < len(old.ops) + 2          # (the two-branch module's transition conv gets a launch of its own)
    maps = [n for n in _fuse_maps(new) if n in set(_fuse_maps(old))]
    assert len(maps) >= 20, maps
    with torch.no_grad():
        a = new.run_with_taps(x, maps)
        ya = [t.clone() for t in _outputs(new)]
        b = old.run_with_taps(x, maps)
        yb = [t.clone() for t in _outputs(old)]
    torch.cuda.synchronize()
    bad = [n for n in maps if not torch.equal(a[n].view(torch.int16), b[n].view(torch.int16))]
    for n in bad[:6]:
        d = (a[n].float() - b[n].float()).abs()
        print(f"   {n}: {int((d > 0).sum())} of {d.numel()} differ, max {float(d.max()):.3e}, first at "
              f"{[int(v) for v in (d > 0).nonzero()[0]]}, nan {int(torch.isnan(a[n].float()).sum())}")
    assert not bad, bad
    for p, q in zip(ya, yb):
        assert torch.equal(p, q)


def _outputs(plan):
    outs = []
    for v in vars(plan).values():
        if type(v).__name__ == "OutSlot":
            outs.append(v.static)
    assert outs
    return outs


def test_fuse_module_replays_inside_the_graph_and_on_two_streams():
    """The captured graph (what a forward runs) gives the eager pass's bits, twice, and on a side stream's plan."""
    cfg, model, sd, x, gold = load_case("hrnet_w32_256x192")
    model = model.to(DEV)
    x = x.to(DEV)
    with torch.no_grad():
        y1 = model(x).clone()
        y2 = model(x).clone()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            y3 = model(x).clone()
        s.synchronize()
    torch.cuda.synchronize()
    plan = model.plan_for(x)
    assert any(n.endswith(".fuse.band") for n, _ in plan.ops)
    assert torch.equal(y1, y2) and torch.equal(y1, y3)
    from rsgnet_amd import switches
    with switches.override("plan", fuse_module="0"):
        model.invalidate_plan()
        with torch.no_grad():
            y0 = model(x).clone()
    model.invalidate_plan()
    assert torch.equal(y0, y1)
