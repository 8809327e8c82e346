"""Build librsgnet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import re
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "librsgnet_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: librsgnet_hip.so cannot be built")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    srcs = glob.glob(os.path.join(CSRC, "*")) + [os.path.join(os.path.dirname(PKG), "include", "rsgnet_hip.h")]
    return any(os.path.getmtime(s) > t for s in srcs)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every csrc/*.hip for gfx950 and link them into rsgnet_amd/lib/librsgnet_hip.so."""
    if not force and not stale():
        return LIB
    hipcc = _hipcc()
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    flags = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast"]
    flags += os.environ.get("RSGNET_HIP_EXTRA_CFLAGS", "").split()       # experiment builds (-D switches)
    if os.environ.get("RSGNET_HIP_DEBUG_BUILD") == "1":     # diagnostic library: timing switches that give WRONG results
        flags.append("-DRSG_DEBUG")
    # MFMA accumulators in VGPRs (gfx950 has one unified register file): the default AGPR form costs
    # v_accvgpr moves wherever VALU code touches an accumulator (64 per tile in trp.hip's sigmoid loop,
    # 16 per tile in every conv epilogue).  Measured in one visit: TRP loop -18 registers; HRNet step -1.8 %.
    vgpr_form = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
    extra = {f: vgpr_form for f in ("trp.hip", "chain.hip", "group.hip", "conv_igemm.hip")}
    for f in os.environ.get("RSGNET_HIP_NO_VGPR_FORM", "").split(","):  # experiment switch
        extra.pop(f, None)
    # trp.hip: no packed fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  The W projection dropped FMA terms beside
    # another stream's kernels in a form whose v_pk_fma_f32 operands were assembled by v_mov_b32 (DESIGN 9b'; the build
    # without packed ops never failed); its fp32 per-position kernels are the only ones in the library where the compiler
    # forms that sequence (static audit of every inference source), and they are latency-bound, not VALU-bound
    extra["trp.hip"] = extra.get("trp.hip", []) + ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
    # the crop producer reproduces OpenCV's separately rounded double arithmetic: no fused multiply-adds there
    # (-ffp-contract=fast fuses in the backend and ignores `#pragma clang fp contract(off)`)
    extra["crops.hip"] = ["-ffp-contract=off"]

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc, *flags, *extra.get(os.path.basename(src), []), "-Rpass-analysis=kernel-resource-usage",
               "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        # No kernel may touch scratch.  Register spills are not just slow here: grouped chain launches whose
        # kernel spilled a handful of VGPRs gave different bits from run to run (one wave's pixels wrong, a few
        # per cent of the launches; tools/chain_group_stress.py), and went away with the spills.  RSGNET_HIP_ALLOW_SCRATCH=1
        # turns the check into a warning for experiments.
        name, bad = None, []
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and int(m.group(1)) and name:
                bad.append(f"{name}: {m.group(1)} B/lane")
        if bad:
            msg = f"{os.path.basename(src)}: kernels using scratch memory: " + "; ".join(bad)
            if os.environ.get("RSGNET_HIP_ALLOW_SCRATCH", "0") == "0":
                raise RuntimeError(msg)
            print("warning: " + msg, file=sys.stderr)
        if verbose:
            rest = "\n".join(ln for ln in r.stderr.splitlines() if "-Rpass-analysis" not in ln and "remark:" not in ln)
            if rest.strip():
                print(rest, file=sys.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, srcs))
    tmp = LIB + ".tmp"
    r = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", tmp],
                       capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
