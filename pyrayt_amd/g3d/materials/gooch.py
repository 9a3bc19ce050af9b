"""Gooch (warm / cool) shading description (``tinygfx/g3d/materials/gooch.py:9-82``).

A material is three colours and two blend weights; the only host arithmetic is the pair of
blended shades (``gooch.py:36-37``), handed to the device as eight floats per surface.  The
per-pixel mix -- unit vector to the light, its cosine with the surface normal, interpolation
between the two shades (``:40-65``) -- runs in the HIP engine (``prt_gooch_shade`` inside a
render, ``prt_gooch_mix`` for a direct ``shade`` call).
"""
import abc
from dataclasses import dataclass, field

import numpy as np

from .._epoch import SceneEpoch
from . import color
from .color import RGBAColor


class Material(abc.ABC):
    """Anything a renderer can ask for pixel colours.  Assigning to an attribute of a material -- a refractive index, a
    Sellmeier coefficient, a colour -- moves the scene objects' change counter (``g3d._epoch.SceneEpoch``), like moving a
    part does: a ``RayTracer`` that holds a compiled copy of a system using the material looks at it again."""

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        SceneEpoch.value += 1

    @abc.abstractmethod
    def shade(self, rays, normals, light_positions):
        """(4,n) RGBA at the (2,4,n) [points, directions] ``rays`` with surface ``normals``."""


@dataclass
class GoochMaterial(Material):
    base_color: RGBAColor = field(default_factory=RGBAColor)
    warm_color: RGBAColor = field(default_factory=RGBAColor)
    cool_color: RGBAColor = field(default_factory=RGBAColor)
    alpha: float = 0.3
    beta: float = 0.3

    def shade_pair(self):
        """(shade_warm, shade_cool), ``gooch.py:36-37``."""
        warm = (1 - self.alpha) * self.warm_color + self.alpha * self.base_color
        cool = (1 - self.beta) * self.cool_color + self.beta * self.base_color
        return np.asarray(warm, dtype=float), np.asarray(cool, dtype=float)

    def shade(self, rays, normals, light_positions):
        from ... import engine

        return engine.gooch_mix(self, rays, normals, light_positions)


def _blue_to_orange(base):
    return GoochMaterial(base_color=base, warm_color=color.ORANGE, cool_color=color.BLUE)


WHITE = _blue_to_orange(color.WHITE)
RED = _blue_to_orange(color.RED)
GREEN = _blue_to_orange(color.GREEN)
BLUE = GoochMaterial(base_color=color.BLUE, warm_color=color.YELLOW, cool_color=color.BLUE, alpha=0.2)
YELLOW = _blue_to_orange(color.YELLOW)
ORANGE = _blue_to_orange(color.ORANGE)
BLACK = _blue_to_orange(color.BLACK)
