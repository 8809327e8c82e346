"""Deterministic synthetic weights, keyed by state_dict name.

Used wherever real checkpoints are unavailable (parity tests, goldens, bench): every tensor is drawn
from ``RandomState(crc32(name) ^ seed)`` so the reference model (in ``oracle/make_goldens.py``), the
CPU oracle and the HIP path can all be given bit-identical fp32 weights without shipping 117 MB.

Unlike torch's default init (identity BatchNorm; SURVEY §4 "important trap") the rule randomises BN
affine terms, and BN running statistics are *calibrated*: ``oracle/make_goldens.py`` runs one
training-mode pass (momentum 1) of the reference over a seeded batch and stores the resulting
running_mean / running_var in ``tests/golden/bnstats_<tag>.npz``.  That keeps activations O(1) through
the ~100 residual blocks (so fp16 storage neither overflows nor underflows) while making BN folding,
eps and channel-order bugs visible.
"""
from __future__ import annotations

import re
import zlib
from typing import Dict, Mapping, Optional

import numpy as np
import torch


def _rs(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


_KEEP = ("type_features", "loc_features", "kt_machine.real_matrix_limb")


_BLOCK_LAST_BN = re.compile(r"(branches\.\d+\.\d+\.bn2|layer1\.\d+\.bn3)$")
CLASSIFIER_STD = 0.04   # x sqrt(1/fan_in): heatmaps come out with |max| ~ 1, like a real Gaussian target


def _init_regime(name: str, shape, r: np.random.RandomState, names) -> np.ndarray:
    """What ``get_pose_net(cfg, is_train=False)`` leaves in the reference (BASELINE configs[0],
    "random-init"): torch-default kaiming-uniform(a=sqrt 5) convs/linears, identity BatchNorm; TRP
    N(0,1e-3) weights incl. its GroupNorm and zero biases (association.py:271-278); KTMachine
    kaiming-normal(fan_out) weights, zero biases, matrix_limb ~ N(0,1) (pose_rsgnet.py:509,581-589)."""
    leaf = name.rsplit(".", 1)[-1]
    stem = name.rsplit(".", 1)[0]
    if leaf == "running_mean":
        return np.zeros(shape)
    if leaf == "running_var":
        return np.ones(shape)
    if stem.startswith("relation_head"):
        return np.zeros(shape) if leaf == "bias" else r.normal(0, 1e-3, shape)
    if name == "kt_machine.matrix_limb":
        return r.normal(0, 1, shape)
    if stem.startswith("kt_machine"):
        return np.zeros(shape) if leaf == "bias" else r.normal(0, np.sqrt(2.0 / shape[0]), shape)
    if (stem + ".running_mean") in names:
        return np.ones(shape) if leaf == "weight" else np.zeros(shape)
    if leaf == "bias":   # only the 1x1 classifier convs on C0=32/48 channels have one
        return r.uniform(-1, 1, shape) / np.sqrt(32.0)
    fan_in = float(np.prod(shape[1:]))   # also torch's convention for ConvTranspose2d's [I,O,k,k]
    b = 1.0 / np.sqrt(fan_in)
    return r.uniform(-b, b, shape)


def synth_state_dict(template: Mapping[str, torch.Tensor], seed: int = 0,
                     bn_stats: Optional[Mapping[str, np.ndarray]] = None,
                     regime: str = "calibrated") -> Dict[str, torch.Tensor]:
    """Fill a state_dict shaped like ``template`` by the name-keyed rule.

    regime ``'calibrated'`` (default) — a stand-in for a *trained* network:
      conv / deconv / linear weights ~ N(0, 2/fan_in); BN/GroupNorm beta ~ N(0,.1), gamma ~ U(.5,1.5)
      except the last BN of every residual block (gamma ~ U(.1,.5): trained ResNets keep the residual
      branch well below the skip path, which keeps the net non-chaotic) and ``predict_contact_net.1``
      (gamma ~ U(.15,.45) so TRP's Gram-matrix sigmoid is not saturated); conv biases ~ N(0,.1);
      ``matrix_limb`` ~ N(0,1); classifier layers ~ N(0, (CLASSIFIER_STD)^2/fan_in) so heatmaps peak
      near 1.  Running stats come from ``bn_stats`` (calibrated, see module docstring) when given.
    regime ``'init'`` — what the reference holds after ``get_pose_net(cfg, is_train=False)``
      (torch default init; identity BatchNorm), the "random-init" case of BASELINE configs[0].
    """
    if regime not in ("calibrated", "init"):
        raise ValueError(regime)
    out: Dict[str, torch.Tensor] = {}
    names = set(template.keys())
    for name, t in template.items():
        if regime == "init" and name not in _KEEP and not name.endswith("num_batches_tracked"):
            v = _init_regime(name, tuple(t.shape), _rs(name, seed), names)
            out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).reshape(tuple(t.shape))
            continue
        shape = tuple(t.shape)
        r = _rs(name, seed)
        leaf = name.rsplit(".", 1)[-1]
        stem = name.rsplit(".", 1)[0]
        # GroupNorm of the TRP: W.1 (plain head) / W.1.1 (RELATION_SUB_SAMPLE head, association.py:254-268)
        is_norm = (stem + ".running_mean") in names or (stem in ("relation_head.W.1", "relation_head.W.1.1") and len(shape) == 1
                                                        and (stem + ".0.weight") not in names)
        if name in _KEEP:
            v = t.detach().cpu().numpy().astype(np.float32)
        elif leaf == "num_batches_tracked":
            out[name] = torch.zeros((), dtype=torch.long)
            continue
        elif leaf in ("running_mean", "running_var"):
            if bn_stats is not None:
                v = np.asarray(bn_stats[name], dtype=np.float32)
            else:
                v = np.zeros(shape, np.float32) if leaf == "running_mean" else np.ones(shape, np.float32)
        elif is_norm and leaf == "weight":
            if stem == "predict_contact_net.1":
                lo, hi = 0.15, 0.45
            elif _BLOCK_LAST_BN.search(stem):
                lo, hi = 0.1, 0.5
            else:
                lo, hi = 0.5, 1.5
            v = r.uniform(lo, hi, shape)
        elif is_norm and leaf == "bias":
            v = r.normal(0.0, 0.1, shape)
        elif name == "kt_machine.matrix_limb":
            v = r.normal(0.0, 1.0, shape)
        elif leaf == "bias":
            v = r.normal(0.0, 0.1, shape)
        elif len(shape) >= 2:
            # Conv2d [O,I,kh,kw] and Linear [O,I]: fan_in = prod(shape[1:]);
            # ConvTranspose2d [I,O,kh,kw] (k4 s2): each output sees I * (k/2)^2 taps.
            if "convtranspose" in name or name == "relation_head.W.0.0.weight":
                fan_in = shape[0] * shape[2] * shape[3] / 4.0
            else:
                fan_in = float(np.prod(shape[1:]))
            gain = CLASSIFIER_STD ** 2 if stem in ("final_layer", "multi_final_layer") else 2.0
            v = r.normal(0.0, np.sqrt(gain / fan_in), shape)
        else:
            raise ValueError(f"no synthetic rule for {name} {shape}")
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).reshape(shape)
    return out


def synth_input(batch: int, image_wh, seed: int = 0) -> torch.Tensor:
    """x ~ N(0,1) fp32 NCHW — the statistics of an ImageNet-normalised crop (tools/test.py:105-107)."""
    r = np.random.RandomState(1000 + seed)
    return torch.from_numpy(r.standard_normal((batch, 3, image_wh[1], image_wh[0])).astype(np.float32))


def gaussian_targets(batch: int, joints: int, hw: tuple, seed: int, sigma: float = 2.0):
    """Seeded training targets shaped like the reference's (lib/dataset/JointsDataset.py generate_target, 'gaussian'):
    one unnormalised Gaussian per joint and a 0/1 visibility weight [B,J,1].  Data for tests, not a restatement."""
    r = np.random.RandomState(seed)
    H, W = hw
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    t = np.zeros((batch, joints, H, W), np.float32)
    w = (r.uniform(size=(batch, joints, 1)) > 0.2).astype(np.float32)
    for b in range(batch):
        for j in range(joints):
            cx, cy = r.uniform(2, W - 3), r.uniform(2, H - 3)
            t[b, j] = np.exp(-((xs - np.round(cx)) ** 2 + (ys - np.round(cy)) ** 2) / (2 * sigma ** 2)) * w[b, j, 0]
    return torch.from_numpy(t), torch.from_numpy(w)


def limb_targets(batch: int, limbs: int, hw: tuple, seed: int):
    """Seeded stand-ins for the data set's limb maps (values in [0,1], a few blobs per channel).  Data, not a restatement."""
    t, _ = gaussian_targets(batch, limbs, hw, seed, sigma=3.0)
    return t.clamp(0, 1)
