"""Render-time materials (the reference's ``tinygfx.g3d.materials``): RGBA colours and the Gooch
shading description the renderers consume."""
from . import color, gooch

__all__ = ["color", "gooch"]
