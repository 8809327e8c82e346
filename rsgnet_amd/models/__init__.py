"""``models`` package of the drop-in: ``models.pose_hrnet`` and ``models.pose_rsgnet``.

The reference resolves its model with ``eval('models.'+cfg.MODEL.NAME+'.get_pose_net')``
(/root/reference/tools/test.py:83); its own ``lib/models/__init__.py:17-19`` imports three modules
that do not exist (SURVEY §0 defect 1).  This package exposes exactly the two in-scope models.
"""
from . import pose_hrnet, pose_rsgnet  # noqa: F401

__all__ = ["pose_hrnet", "pose_rsgnet"]
