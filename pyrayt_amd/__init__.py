"""pyrayt_amd -- MI355X-native engine for PyRayT's batched ray-propagation hot path.

The package mirrors the part of the reference's Python API that sits on that path
(``pyrayt.RayTracer`` / ``RaySet``, ``pyrayt.materials``, ``pyrayt.components`` and the
``tinygfx.g3d`` scene objects as ``pyrayt_amd.g3d``) and executes the per-generation inner loop
-- ray transform, analytic intersect + CSG, nearest-hit selection, normals, absorb / reflect /
refract shading, dead-ray compaction and record writing -- as hand-written HIP kernels for
gfx950 behind the C-ABI of ``include/prt.h``.  There is no CPU fallback: tracing without the
HIP library or without a GPU raises.
"""
from . import _runtime  # first: asks the HIP runtime for eight hardware queues before torch loads it
from . import g3d
from . import materials
from . import components
from . import utils
from .rayset import RaySet
from .tracer import RayTracer
from .frame import DeviceFrame
from .g3d.objects import pin
from .utils import wavelength_to_rgb

__all__ = ["RayTracer", "RaySet", "DeviceFrame", "pin", "materials", "components", "g3d", "utils",
           "wavelength_to_rgb"]
__version__ = "0.1.0"
