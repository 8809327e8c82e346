"""Build librsgnet_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The build is keyed on what it builds: a hash over every csrc/* file, the public header, the flag table below, the
environment switches that change flags and `hipcc --version` is stored beside the library (`<lib>.buildinfo`, JSON);
`stale()` compares that hash, `info()` returns it with the per-file extra flags (smoke() and bench.py print it), and
`audit_packed_fp32()` disassembles the library's code objects and counts the packed fp32 VALU ops per source file
(the build fails on a hit and tests/test_boundary.py asserts 0 for every source: DESIGN 9b').

    python -m rsgnet_amd.build [--force]
    RSGNET_HIP_BUILD_NAME=base RSGNET_HIP_EXTRA_CFLAGS="-DRSG_CHAIN16=0" python -m rsgnet_amd.build
        -> rsgnet_amd/lib/base/librsgnet_hip.so, a variant for same-visit A/B runs (RSGNET_HIP_LIB, tools/gpu_ab_lib.sh)
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
from typing import Dict, List, Optional

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "librsgnet_hip.so")
HEADER = os.path.join(os.path.dirname(PKG), "include", "rsgnet_hip.h")
ARCH = "gfx950"

# No packed fp32 VALU ops (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) anywhere in the library.  The TRP's W projection dropped
# FMA terms beside another stream's kernels in a form whose packed FMAs read 64-bit operands fed by uniform-address LDS reads
# (DESIGN 9b').  What round 4 established about the cause: four wait states between the v_mov_b32 that assembles an operand
# and the v_pk_fma_f32 that reads it change NOTHING (155 of 600 contended runs differ, as the plain build:
# profiles/r04_trp_nop_discriminator.txt), so it is not that read-after-write window; the builds that do not fail are the one
# without packed ops (-O1) and the one without single-address LDS reads.  Both ingredients are gone from the product: the
# per-position kernels read their tables through the scalar cache, and no source is built with packed fp32 ops -- measured
# free (same-visit A/B 1.260 / 1.400 ms against 1.262 / 1.402, profiles/r04_ab_no_packed_fp32.txt; beside MFMAs packed fp32
# is slower than two scalar ops anyway, MI355X_MICROARCH.md).  audit_packed_fp32() disassembles every code object at build
# time and fails the build on a hit.
_NO_PK_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
BASE_FLAGS = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", *_NO_PK_F32]
# MFMA accumulators in VGPRs (gfx950 has one unified register file): the default AGPR form costs
# v_accvgpr moves wherever VALU code touches an accumulator (64 per tile in trp.hip's sigmoid loop,
# 16 per tile in every conv epilogue).  Measured in one visit: TRP loop -18 registers; HRNet step -1.8 %.
_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
EXTRA_FLAGS: Dict[str, List[str]] = {
    "trp.hip": _VGPR_FORM,
    "chain.hip": _VGPR_FORM,
    "group.hip": _VGPR_FORM,
    "fuse_module.hip": _VGPR_FORM,
    "conv_igemm.hip": _VGPR_FORM,
    # the crop producer reproduces OpenCV's separately rounded double arithmetic: no fused multiply-adds there
    # (-ffp-contract=fast fuses in the backend and ignores `#pragma clang fp contract(off)`)
    "crops.hip": ["-ffp-contract=off"],
}


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: librsgnet_hip.so cannot be built")


def lib_path(name: Optional[str] = None) -> str:
    name = name if name is not None else os.environ.get("RSGNET_HIP_BUILD_NAME", "")
    return os.path.join(LIBDIR, name, "librsgnet_hip.so") if name else LIB


def _flag_table() -> Dict[str, List[str]]:
    """Per-file extra flags after the experiment switches of the environment."""
    extra = {k: list(v) for k, v in EXTRA_FLAGS.items()}
    for f in os.environ.get("RSGNET_HIP_NO_VGPR_FORM", "").split(","):  # experiment switch
        if f in extra:
            extra[f] = [a for a in extra[f] if a not in _VGPR_FORM]
    return extra


def _common_flags() -> List[str]:
    flags = list(BASE_FLAGS)
    flags += os.environ.get("RSGNET_HIP_EXTRA_CFLAGS", "").split()       # experiment builds (-D switches)
    if os.environ.get("RSGNET_HIP_DEBUG_BUILD") == "1":     # diagnostic library: timing switches that give WRONG results
        flags.append("-DRSG_DEBUG")
    return flags


_hipcc_version_cache: Optional[str] = None


def _hipcc_version() -> str:
    global _hipcc_version_cache
    if _hipcc_version_cache is None:
        try:
            r = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True)
            if r.returncode or not r.stdout.strip():     # a compiler that cannot even print its version cannot rebuild either
                raise RuntimeError(f"hipcc --version: rc {r.returncode}")
            _hipcc_version_cache = r.stdout.strip()
        except Exception as e:      # no (working) compiler on this box: a prebuilt library is then judged by its sources alone
            _hipcc_version_cache = f"unavailable ({type(e).__name__})"
    return _hipcc_version_cache


def compiler_available() -> bool:
    return not _hipcc_version().startswith("unavailable")


def source_hash(with_compiler: bool = True) -> str:
    """sha256 over csrc/*, include/rsgnet_hip.h, the flag table and (optionally) `hipcc --version`."""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(CSRC, "*"))) + [HEADER]:
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode())
            with open(path, "rb") as f:
                h.update(f.read())
    h.update(json.dumps({"common": _common_flags(), "extra": _flag_table()}, sort_keys=True).encode())
    if with_compiler:
        h.update(_hipcc_version().encode())
    return h.hexdigest()


def info(name: Optional[str] = None) -> dict:
    """The build record of a library: {hash, sources_hash, common_flags, extra_flags, hipcc}; {} when there is none."""
    p = lib_path(name) + ".buildinfo"
    if not os.path.exists(p):
        return {}
    with open(p) as f:
        return json.load(f)


def describe(name: Optional[str] = None) -> str:
    """One line for logs: which build of which sources is loaded."""
    i = info(name)
    if not i:
        return f"{lib_path(name)}: no build record"
    extra = "; ".join(f"{k}: {' '.join(v)}" for k, v in sorted(i.get("extra_flags", {}).items()))
    return f"librsgnet_hip build {i['hash'][:16]} (sources {i['sources_hash'][:16]}) flags [{' '.join(i.get('common_flags', []))}] per file [{extra}]"


def stale(name: Optional[str] = None) -> bool:
    """True when the library is missing or was built from other sources / flags / compiler than the tree holds now.
    Without a compiler (a box that only runs the prebuilt library) the sources-and-flags hash alone decides."""
    lib = lib_path(name)
    if not os.path.exists(lib):
        return True
    i = info(name)
    if not i:
        return True
    try:
        _hipcc()
    except RuntimeError:
        return i.get("sources_hash") != source_hash(with_compiler=False)
    if not compiler_available():
        return i.get("sources_hash") != source_hash(with_compiler=False)
    return i.get("hash") != source_hash()


def build(force: bool = False, verbose: bool = False, name: Optional[str] = None) -> str:
    """Compile every csrc/*.hip for gfx950 and link them into rsgnet_amd/lib[/name]/librsgnet_hip.so."""
    lib = lib_path(name)
    if not force and not stale(name):
        return lib
    hipcc = _hipcc()
    libdir = os.path.dirname(lib)
    os.makedirs(libdir, exist_ok=True)
    # Several processes may find the library stale at once (torchrun ranks, pytest workers): ONE builds, the others wait on
    # the lock file and find a fresh library when they get it (advisor, round 4: unserialised builds linked each other's
    # half-written objects).
    import fcntl
    with open(os.path.join(libdir, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale(name):
                return lib
            return _build_locked(hipcc, lib, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(hipcc: str, lib: str, verbose: bool) -> str:
    libdir = os.path.dirname(lib)
    objdir = os.path.join(libdir, "obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    flags = _common_flags()
    extra = _flag_table()

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
        kname, bad = None, []
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                kname = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and int(m.group(1)) and kname:
                bad.append(f"{kname}: {m.group(1)} B/lane")
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
    # build step, not a tool: no code object may hold a packed fp32 VALU op (DESIGN 9b')
    audit = audit_packed_fp32(objs)
    hits = {k: v for k, v in audit.items() if v}
    if hits:
        raise RuntimeError(f"packed fp32 VALU ops in the library (built with -packed-fp32-ops): {hits}")
    if not audit:
        # the LLVM tools the audit needs are missing: say so (the record below carries it) or refuse, never pass in silence
        msg = "packed-fp32 audit did NOT run (clang-offload-bundler / llvm-objcopy / llvm-objdump not found)"
        if os.environ.get("RSGNET_HIP_REQUIRE_AUDIT") == "1":
            raise RuntimeError(msg)
        print("warning: " + msg, file=sys.stderr)
    tmp = f"{lib}.tmp.{os.getpid()}"
    r = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", tmp],
                       capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    os.replace(tmp, lib)
    with open(lib + ".buildinfo", "w") as f:
        json.dump({"hash": source_hash(), "sources_hash": source_hash(with_compiler=False), "common_flags": flags,
                   "extra_flags": extra, "hipcc": _hipcc_version().splitlines()[0] if _hipcc_version() else "",
                   "packed_fp32_audit": {"ran": bool(audit), "objects": len(audit), "hits": 0}}, f, indent=1)
    return lib


def _llvm_tool(tool: str) -> Optional[str]:
    for c in (shutil.which(tool), f"/opt/rocm/lib/llvm/bin/{tool}", f"/opt/rocm/llvm/bin/{tool}"):
        if c and os.path.exists(c):
            return c
    return None


def audit_packed_fp32(objs: Optional[List[str]] = None, name: Optional[str] = None) -> Dict[str, int]:
    """{source file: number of v_pk_{fma,mul,add}_f32 in its gfx950 code object}, read from the object files of the last
    build (rsgnet_amd/lib[/name]/obj/*.o).  Needs clang-offload-bundler, llvm-objcopy and llvm-objdump (ROCm's llvm/bin); returns {} when
    they, llvm-objcopy or the objects are missing."""
    if objs is None:
        objs = sorted(glob.glob(os.path.join(os.path.dirname(lib_path(name)), "obj", "*.o")))
    bundler, objdump, objcopy = _llvm_tool("clang-offload-bundler"), _llvm_tool("llvm-objdump"), _llvm_tool("llvm-objcopy")
    if not objs or not bundler or not objdump or not objcopy:
        return {}
    out: Dict[str, int] = {}
    pat = re.compile(r"\bv_pk_(?:fma|mul|add)_f32\b")
    for obj in objs:
        fat, co = obj + ".fatbin", obj + ".gfx950.co"
        try:
            # the host object carries the device code as a clang offload bundle in its .hip_fatbin section
            r = subprocess.run([objcopy, f"--dump-section=.hip_fatbin={fat}", obj], capture_output=True, text=True)
            if r.returncode or not os.path.exists(fat):
                continue            # a host-only source (graph.hip)
            r = subprocess.run([bundler, "--type=o", f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}", f"--input={fat}",
                                f"--output={co}", "--unbundle"], capture_output=True, text=True)
            if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            d = subprocess.run([objdump, "-d", f"--mcpu={ARCH}", co], capture_output=True, text=True)
            if d.returncode == 0:
                out[os.path.basename(obj)[:-2] + ".hip"] = len(pat.findall(d.stdout))
        finally:
            for f in (fat, co):
                if os.path.exists(f):
                    os.remove(f)
    return out


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
    print(describe())
