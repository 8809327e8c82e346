"""rsgnet_amd — MI355X-native forward path of RSGNet (HRNet backbone + TRP + SGM).

``rsgnet_amd.models`` mirrors the reference's ``lib/models`` constructor surface; the arithmetic runs
in ``librsgnet_hip.so`` (hand-written HIP for gfx950, C ABI in ``include/rsgnet_hip.h``).
"""
from . import models  # noqa: F401
from .config import Cfg, as_cfg, make_cfg  # noqa: F401

__version__ = "0.1.0"
