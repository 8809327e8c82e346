"""Import the reference's own lib/models in the BUILD container (it never travels to the GPU box).

Recipe from SURVEY §8(c): lib/models/__init__.py:17-19 imports modules that do not exist, so a
synthetic ``models`` package is registered and the two model files are imported through it; the cfg is
a dict with attribute access; RSGNet opens ``kpt_word_embs.pkl`` relative to the CWD
(pose_rsgnet.py:675), so construction happens in a scratch directory holding the right pickle.
"""
from __future__ import annotations

import contextlib
import importlib
import io
import os
import shutil
import sys
import tempfile
import types
import warnings

REF = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "lib", "models"))


def _models_pkg():
    sys.dont_write_bytecode = True
    if "models" not in sys.modules or getattr(sys.modules["models"], "__path__", None) != [REF + "/lib/models"]:
        pkg = types.ModuleType("models")
        pkg.__path__ = [REF + "/lib/models"]
        sys.modules["models"] = pkg
    return sys.modules["models"]


def build(cfg, is_train: bool = False):
    """Construct the reference model named by cfg.MODEL.NAME (eval mode unless is_train)."""
    _models_pkg()
    name = cfg["MODEL"]["NAME"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mod = importlib.import_module("models." + name)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="rsgref_")
    try:
        pkl = "kpt_word_embs.pkl" if cfg["MODEL"]["NUM_JOINTS"] == 17 else "cp_kpt_word_embs.pkl"
        shutil.copy(os.path.join(REF, pkl), os.path.join(tmp, "kpt_word_embs.pkl"))
        os.chdir(tmp)
        with contextlib.redirect_stdout(io.StringIO()):
            net = mod.get_pose_net(cfg, is_train=is_train)
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)
    return net if is_train else net.eval()
