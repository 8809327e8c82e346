"""Every run-time choice of the package, in THREE environment variables of `name=value` pairs.

Rounds 1-4 grew one RSGNET_HIP_* variable per experiment (61 at the end of round 4).  What is left are three tables:

  RSGNET_HIP_PLAN   choices of the inference plan compiler (rsgnet_amd/plan.py, plan_rsgnet.py): which fused launch replaces
                    which sequence of the reference's operators.  Every entry is an ADOPTED fusion with its un-fused
                    fall-back (the fall-back also serves the shapes the fused kernel does not cover, and is the other side
                    of the bit-identity tests), or a tuning integer.
  RSGNET_HIP_TUNE   knobs of the C library (forwarded to rsg_tune / rsg_wgrad_tune when the library is loaded).
  RSGNET_HIP_TRAIN  choices of the training plan (rsgnet_amd/train.py, train_rsgnet.py).

e.g.  RSGNET_HIP_PLAN="fuse_module=0,chain_pack=1"  RSGNET_HIP_TUNE="fm_prefetch=0"  RSGNET_HIP_TRAIN="head_graph=1".
An unknown name or a value outside the allowed set is an error where the table is read, not a silent default.
DESIGN.md section 9f lists every entry with its default and the last same-visit A/B that decided it.

The remaining variables concern the BUILD and the loader, not behaviour: RSGNET_HIP_LIB (another build of the library, A/B
runs), RSGNET_HIP_BUILD_NAME / _EXTRA_CFLAGS / _DEBUG_BUILD / _NO_VGPR_FORM (rsgnet_amd/build.py: variant builds),
RSGNET_HIP_ALLOW_STALE / _NO_AUTOBUILD / _ALLOW_SCRATCH / _REQUIRE_AUDIT (what the loader / the build tolerate).
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

# name -> (default, allowed values or (lo, hi) for integers)
PLAN: Dict[str, Tuple[str, object]] = {
    # how the launch list is replayed: "capture" = one captured HIP graph on one stream, "dag" = an explicit graph with data
    # edges (slower: r01, r03), "0" = eager launches
    "graph": ("capture", ("capture", "dag", "0")),
    "lanes": ("4", (1, 4)),                   # streams of the eager / dag forms
    "chain": ("1", ("0", "1")),               # LDS-resident BasicBlock chains (0: per-conv launches)
    "group": ("1", ("0", "1")),               # grouped launches (one launch per dependency level / module)
    "chain_handoff": ("1", ("0", "1")),       # both halves of a row-tiled branch in ONE launch (in-launch hand-off)
    "handoff_poll_every": ("1", (1, 1 << 20)),
    "wide_chains": ("1", ("0", "1")),         # HRNet-W48's widths as 12-wave chains
    "chain_slices": ("2", (1, 8)),
    "rowtile_blocks": ("2", (1, 4)),
    "chain_pack": ("auto", ("auto", "0", "1")),   # two images per workgroup in over-subscribed chain launches (side-stream plans)
    "chain_exchange": ("auto", ("auto", "0", "1")),   # row-tiled chains trade their halo inside the launch (side-stream plans, round 5)
    "chain_split64": ("0", (0, 2)),           # rsg_chain_item.split_rows for the 32x24x64 chain (r05: neutral / slower)
    "zip_branches": ("1", ("0", "1")),
    "stem": ("1", ("0", "1")),                # conv1 + conv2 in one launch
    "stem_direct": ("1", ("0", "1")),         # ... reading the fp32 NCHW picture itself
    "stem_tail": ("1", ("0", "1")),           # ... and layer1.0.conv1 on the same launch
    "s2d": ("1", ("0", "1")),
    "l1cat": ("1", ("0", "1")),               # layer1.0: conv3 + projected residual as one K-concat contraction
    "pair": ("1", ("0", "1")),                # Bottleneck conv2 + conv3 in one launch
    "pair_next": ("1", ("0", "1")),           # ... + the next Bottleneck's conv1
    "pair_cat": ("1", ("0", "1")),
    "l1_planes": ("1", ("0", "1")),           # layer1's 256-channel maps as channel-tile planes
    "bottleneck": ("0", ("0", "1")),          # whole Bottleneck in one launch (measured equal: opt-in)
    "trans_dual": ("1", ("0", "1")),          # transition1's two convs in one launch
    "trans_ride": ("1", ("0", "1")),          # the next stage's transition conv inside the last fuse launch
    "fold_sum": ("1", ("0", "1")),            # a module's last fuse sum as the epilogue of the conv that feeds it
    "fuse_module": ("1", ("0", "1")),         # level 0 of a module's fuse layers as one launch of row bands (round 5)
    "fuse_head4": ("1", ("0", "1")),          # the last module's fuse + classifier in one launch
    "fuse_classifier": ("1", ("0", "1")),
    "limbs_ride": ("1", ("0", "1")),
    "head_tail": ("1", ("0", "1")),           # RSGNet: predict_net + final_layer as one launch (rsg_conv_tail1x1, round 5)
}
TRAIN: Dict[str, Tuple[str, object]] = {
    "graph": ("1", ("0", "1")),
    "lanes": ("1", ("0", "1")),
    "fuse_lanes": ("1", ("0", "1")),
    "wgrad_stream": ("2", ("0", "1", "2")),   # weight gradients as graphs of their own on a side stream (r03o)
    "segments": ("every:40", None),           # stage | module | every:N
    "wgrad_pipe": ("1", ("0", "1")),
    "fold_add": ("1", ("0", "1")),
    "head_graph": ("0", ("0", "1")),          # RSGNet's head replayed as HIP graphs (slower here: opt-in, r04)
}
# RSGNET_HIP_TUNE: keys of rsg_tune (csrc/chain.hip) and, prefixed wgrad_, the five integers of rsg_wgrad_tune / rsg_wgrad_set_quad
TUNE_KEYS = ("chain_max_ni", "chain_pitch_pad", "conv_lds_limit", "chain_prefetch", "conv_full_k", "conv_spread",
             "bottleneck_head", "chain_dual", "chain_split_rows", "fm_prefetch", "stem_persist", "dual_waves", "chain_debug", "bottleneck_debug",
             "wgrad_min_px", "wgrad_quad_min_wgs", "wgrad_lds", "wgrad_px_cap", "wgrad_waves", "wgrad_quad")

_VARS = {"plan": ("RSGNET_HIP_PLAN", PLAN), "train": ("RSGNET_HIP_TRAIN", TRAIN)}


def _legacy_names() -> Dict[str, str]:
    """The one-variable-per-switch names of rounds 1-4 -> what replaces them."""
    out = {f"RSGNET_HIP_{k.upper()}": f'RSGNET_HIP_PLAN="{k}=..."' for k in PLAN}
    out.update({f"RSGNET_HIP_TRAIN_{k.upper()}": f'RSGNET_HIP_TRAIN="{k}=..."' for k in TRAIN})
    for k in TUNE_KEYS:
        old = {"conv_lds_limit": "CONV_LDS"}.get(k, k.upper())
        out[f"RSGNET_HIP_{old}"] = f'RSGNET_HIP_TUNE="{k}=..."'
    return out


def check_legacy() -> None:
    """An old variable left in a script would now be ignored in silence -- an A/B that compares a build with itself.  It is an
    error instead."""
    stale = {k: v for k, v in _legacy_names().items() if k in os.environ}
    if stale:
        raise ValueError("environment variables of the one-per-switch scheme (rounds 1-4) are set: " +
                         "; ".join(f"{k} -> {v}" for k, v in sorted(stale.items())))


def _parse(var: str) -> Dict[str, str]:
    raw = os.environ.get(var, "")
    out: Dict[str, str] = {}
    for item in raw.replace(";", ",").split(","):
        item = item.strip()
        if not item:
            continue
        if "=" not in item:
            raise ValueError(f"{var}={raw!r}: {item!r} is not name=value")
        k, v = item.split("=", 1)
        out[k.strip()] = v.strip()
    return out


def _get(kind: str, name: str) -> str:
    check_legacy()
    var, table = _VARS[kind]
    if name not in table:
        raise KeyError(f"rsgnet_amd.switches: no {kind} switch {name!r}")
    given = _parse(var)
    unknown = [k for k in given if k not in table]
    if unknown:
        raise ValueError(f"{var}: unknown name(s) {unknown}; known: {sorted(table)}")
    default, allowed = table[name]
    v = given.get(name, default)
    if isinstance(allowed, tuple) and len(allowed) == 2 and all(isinstance(a, int) for a in allowed):
        try:
            iv = int(v)
        except ValueError:
            raise ValueError(f"{var}: {name}={v!r} is not an integer") from None
        if not allowed[0] <= iv <= allowed[1]:
            raise ValueError(f"{var}: {name}={iv}: allowed range {allowed[0]}..{allowed[1]}")
    elif allowed is not None and v not in allowed:
        raise ValueError(f"{var}: {name}={v!r}; allowed: {allowed}")
    return v


def plan(name: str) -> str:
    return _get("plan", name)


def plan_on(name: str) -> bool:
    return _get("plan", name) != "0"


def plan_int(name: str) -> int:
    return int(_get("plan", name))


def train(name: str) -> str:
    return _get("train", name)


def train_on(name: str) -> bool:
    return _get("train", name) != "0"


def tune_items() -> Dict[str, int]:
    """RSGNET_HIP_TUNE as {key: integer}; validated against TUNE_KEYS (the library validates the values)."""
    check_legacy()
    out: Dict[str, int] = {}
    for k, v in _parse("RSGNET_HIP_TUNE").items():
        if k not in TUNE_KEYS:
            raise ValueError(f"RSGNET_HIP_TUNE: unknown key {k!r}; known: {TUNE_KEYS}")
        try:
            out[k] = int(v)
        except ValueError:
            raise ValueError(f"RSGNET_HIP_TUNE: {k}={v!r} is not an integer") from None
    return out


class override:
    """Context manager for tests and tools: `with switches.override("plan", fuse_module="0"): ...` sets the pairs on top of
    whatever the variable holds and restores it afterwards."""

    def __init__(self, kind: str, **pairs):
        self.var = {"plan": "RSGNET_HIP_PLAN", "train": "RSGNET_HIP_TRAIN", "tune": "RSGNET_HIP_TUNE"}[kind]
        self.pairs = {k: str(v) for k, v in pairs.items()}

    def __enter__(self):
        self.old = os.environ.get(self.var)
        cur = _parse(self.var)
        cur.update(self.pairs)
        os.environ[self.var] = ",".join(f"{k}={v}" for k, v in cur.items())
        return self

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop(self.var, None)
        else:
            os.environ[self.var] = self.old
