"""``tinygfx.g3d.world_objects`` under its upstream module name: the scene objects live in ``pyrayt_amd.g3d.objects``;
this module is that one's public names, so that ``from tinygfx.g3d import world_objects`` /
``import tinygfx.g3d.world_objects as cg`` become ``pyrayt_amd.g3d`` with nothing else to change."""
from .objects import *  # noqa: F401,F403
