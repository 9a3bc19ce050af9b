"""RGBA colours (``tinygfx/g3d/materials/color.py:4-56``): a 4-vector of float64 with named
channels, alpha defaulting to 1."""
import numpy as np


def _channel(index):
    def get(self):
        return self[index]

    def put(self, value):
        self[index] = value

    return property(get, put)


class RGBAColor(np.ndarray):
    def __new__(cls, r=0.0, g=0.0, b=0.0, a=1.0):
        return np.array((r, g, b, a), dtype=float).view(cls)

    r = _channel(0)
    g = _channel(1)
    b = _channel(2)
    a = _channel(3)


WHITE = RGBAColor(1, 1, 1)
BLACK = RGBAColor()
RED = RGBAColor(1, 0, 0)
GREEN = RGBAColor(0, 1, 0)
BLUE = RGBAColor(0, 0, 1)
YELLOW = RGBAColor(1, 1, 0)
ORANGE = RGBAColor(1, 0.5, 0)
