"""Why the HIP path sits 7e-3 (fp16) from the fp32 forward on trained-like weights, and what it would take to reach the 1e-3
of `north_star` there -- measured on the CPU with the rounding-aware oracle, which rounds where the
kernels round: the BatchNorm-folded weights once (the MFMA's A operand) and every stored activation (the next conv's B
operand); accumulation is fp32 on both sides.

The error is OPERAND rounding, not storage: a conv of the HIP path multiplies 11-bit-mantissa weights by 11-bit-mantissa
activations, ~70 convs deep.  Keeping more bits in EITHER operand alone leaves the other's rounding, which is of the same
size; both must carry more bits -- on the matrix cores that means splitting each operand into hi + lo 16-bit parts and
running three MFMAs (hi.hi, hi.lo, lo.hi) where the fp16 path runs one, with twice the LDS and HBM bytes per activation.
A "fp16+" mode that widens only the stored branch-0 maps and fuse sums (the verdict's proposal) is the `acts wide` row
restricted to some tensors: it cannot go below the `weights 16-bit` floor.  The 1e-3 bar IS met where BASELINE configs[0]
puts it: the reference's random-init network (tests/test_gpu_e2e.py, 5.9e-4)."""
import numpy as np
import torch

from oracle import pose_oracle as po
from tests.util import load_case


def test_sixteen_bit_operands_set_the_error_floor_on_trained_like_weights():
    cfg, _model, sd, x, gold = load_case("hrnet_w32_256x192")
    ref = gold["heatmaps_crop0"]
    sig = float(np.std(ref))

    def err(emulate, weights, acts):
        ctx = po.Ctx(sd, emulate=emulate)
        ctx.emu_weights, ctx.emu_acts = weights, acts
        y = po.hrnet_forward(sd, cfg, x, ctx=ctx)[0].numpy()
        return float(np.abs(y - ref).max())

    rows = {"fp16 operands (the HIP path)": err("fp16", True, True),
            "fp16 weights, activations wide": err("fp16", True, False),
            "weights wide, fp16 activations": err("fp16", False, True),
            "both wide (fp32)": err("fp16", False, False),
            "bf16 operands": err("bf16", True, True)}
    for k, v in rows.items():
        print(f"   {k:34s}: max-abs {v:.3e} = {v / sig:.4f} sigma of the fp32 golden heat maps (|max| 1.0)")
    base = rows["fp16 operands (the HIP path)"]
    assert 2e-3 < base < 2e-2                                         # the floor the GPU parity tests sit on (7.4e-3 measured)
    assert rows["both wide (fp32)"] < 1e-4                            # the oracle itself: fp32 summation noise
    # widening ONE operand does not reach 1e-3: the other's rounding is of the same size
    assert rows["fp16 weights, activations wide"] > 1e-3 and rows["weights wide, fp16 activations"] > 1e-3
    assert rows["bf16 operands"] > 4 * base                           # 8 mantissa bits: configs[1]/[2]'s format
