#!/usr/bin/env python3
"""Static audit behind DESIGN 9b': per kernel, the packed fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) one of
whose VGPR sources was written by a v_mov_b32 within the previous W instructions -- the sequence of the TRP projection defect.
usage:  for f in rsgnet_amd/csrc/*.hip: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast [per-file flags of
        rsgnet_amd/build.py] -Iinclude -Irsgnet_amd/csrc -S --cuda-device-only -o /tmp/asm/asm_<name>.s $f
        cd /tmp/asm && python tools/audit_packed_fp32.py
End of round 3 (default flags): trp.hip only (trp_w_project_kernel 5-10 instances of 512-1152 packed ops, the backward tail kernels
2-8) -> trp.hip is built without packed fp32 ops; train_tail.hip bce_kernel 10 of 19 (a loss kernel, nothing runs beside it);
conv_igemm / group / chain / dual / stem / fuse_head / elementwise / train / wgrad / crops: none."""
import re, sys, glob, collections
# per kernel: v_pk_{fma,mul,add}_f32 whose VGPR source pair has a half written by v_mov_b32 within the previous W instructions
W = 4
pk = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\s+(.*)")
mov = re.compile(r"^\s*v_mov_b32(?:_e32)?\s+v(\d+),\s*v\d+")
rng = re.compile(r"v\[(\d+):(\d+)\]")
for f in sorted(glob.glob("asm_*.s")):
    kern = None
    stats = collections.Counter(); tot = collections.Counter()
    recent = []          # (vgpr written by v_mov from a vgpr)
    for line in open(f):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = m.group(1); recent = []
            continue
        if kern is None or line.startswith("\t.") or line.strip().startswith(";"):
            continue
        mm = mov.match(line)
        p = pk.match(line)
        if p:
            tot[kern] += 1
            ops = p.group(2).split(",", 1)[1] if "," in p.group(2) else ""
            srcs = set()
            for a, b in rng.findall(ops):
                srcs.update(range(int(a), int(b) + 1))
            if any(r in srcs for r in recent):
                stats[kern] += 1
        recent = (recent + [int(mm.group(1))])[-W:] if mm else (recent + [-1])[-W:]
    for k in tot:
        if stats[k]:
            print(f"{f[4:-2]:12s} {k[:70]:70s} pk_f32 {tot[k]:5d}  fed by a v_mov_b32 within {W} instructions: {stats[k]}")
