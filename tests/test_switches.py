"""rsgnet_amd/switches.py: the three name=value tables that replaced one environment variable per experiment."""
import os

import pytest

from rsgnet_amd import switches as sw


def test_defaults_and_overrides(monkeypatch):
    monkeypatch.delenv("RSGNET_HIP_PLAN", raising=False)
    assert sw.plan_on("fuse_module") and not sw.plan_on("bottleneck") and sw.plan("graph") == "capture" and sw.plan_int("lanes") == 4
    monkeypatch.setenv("RSGNET_HIP_PLAN", "fuse_module=0, chain_pack=1;graph=dag")
    assert not sw.plan_on("fuse_module") and sw.plan("chain_pack") == "1" and sw.plan("graph") == "dag"
    with sw.override("plan", fuse_module="1", lanes=2):
        assert sw.plan_on("fuse_module") and sw.plan_int("lanes") == 2 and sw.plan("graph") == "dag"
    assert not sw.plan_on("fuse_module") and os.environ["RSGNET_HIP_PLAN"].startswith("fuse_module=0")
    monkeypatch.delenv("RSGNET_HIP_TRAIN", raising=False)
    assert sw.train("wgrad_stream") == "2" and sw.train("segments") == "every:40" and not sw.train_on("head_graph")


def test_typos_and_bad_values_are_errors(monkeypatch):
    monkeypatch.setenv("RSGNET_HIP_PLAN", "fuse_modul=0")
    with pytest.raises(ValueError):
        sw.plan_on("fuse_module")
    monkeypatch.setenv("RSGNET_HIP_PLAN", "lanes=9")
    with pytest.raises(ValueError):
        sw.plan_int("lanes")
    monkeypatch.setenv("RSGNET_HIP_PLAN", "graph=maybe")
    with pytest.raises(ValueError):
        sw.plan("graph")
    monkeypatch.setenv("RSGNET_HIP_PLAN", "stem")
    with pytest.raises(ValueError):
        sw.plan_on("stem")
    monkeypatch.delenv("RSGNET_HIP_PLAN")
    with pytest.raises(KeyError):
        sw.plan_on("no_such_switch")
    monkeypatch.setenv("RSGNET_HIP_TUNE", "fm_prefetch=0,wgrad_min_px=64")
    assert sw.tune_items() == {"fm_prefetch": 0, "wgrad_min_px": 64}
    monkeypatch.setenv("RSGNET_HIP_TUNE", "fm_prefetchh=0")
    with pytest.raises(ValueError):
        sw.tune_items()


def test_the_package_reads_no_other_behaviour_variable():
    """Every RSGNET_HIP_* name the package reads: the three tables, the library path and the build / loader variables --
    twelve in all (61 at the end of round 4)."""
    import glob
    import re
    pkg = os.path.dirname(os.path.abspath(sw.__file__))
    names = set()
    for f in glob.glob(os.path.join(pkg, "**", "*.py"), recursive=True):
        for m in re.finditer(r'environ[^\n]*?"(RSGNET_HIP_[A-Z0-9_]+)"', open(f).read()):
            names.add(m.group(1))
    allowed = {"RSGNET_HIP_PLAN", "RSGNET_HIP_TUNE", "RSGNET_HIP_TRAIN", "RSGNET_HIP_LIB", "RSGNET_HIP_BUILD_NAME",
               "RSGNET_HIP_EXTRA_CFLAGS", "RSGNET_HIP_DEBUG_BUILD", "RSGNET_HIP_NO_VGPR_FORM", "RSGNET_HIP_ALLOW_STALE",
               "RSGNET_HIP_NO_AUTOBUILD", "RSGNET_HIP_ALLOW_SCRATCH", "RSGNET_HIP_REQUIRE_AUDIT"}
    assert names <= allowed, sorted(names - allowed)
