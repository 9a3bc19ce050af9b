"""Orthographic renderers on the intersect kernels (the reference's ``tinygfx/g3d/renderers.py``).

``EdgeRender`` (``renderers.py:11-126``) and ``ShadedRenderer`` (``:129-248``) keep upstream's
constructor / ``render()`` / ``reset()`` surface and return the same ``(v, h, 4)`` float64 RGBA
picture.  Upstream walks INITIALIZE -> PROPAGATE -> INTERACT -> FINISH with whole-array numpy;
here one fused kernel does camera ray -> nearest hit -> Gooch colour per pixel (``prt_render``)
and the edge picture is two small kernels over the surface-id image (``prt_edge_canvas``).
``render_device()`` leaves the picture in HBM.  ``draw()`` (``:251-349``) frames the parts the
same way and hands the picture to matplotlib.

There is no host implementation: without the HIP library or a GPU these raise.
"""
import numpy as np

from . import shapes
from .objects import OrthographicCamera


def _as_list(items):
    return items if hasattr(items, "__iter__") else (items,)


class _Renderer:
    def __init__(self, camera, surfaces):
        self._camera = camera
        self._shapes = _as_list(surfaces)
        self.reset()

    def reset(self):
        self._simulation_complete = False
        self._results = None
        self._hits = None

    def get_results(self):
        return self._results

    @property
    def _hit_distances(self):
        """Per-pixel ray parameter of the last render (host array; +inf where nothing is seen)."""
        return None if self._hits is None else self._hits[0].cpu().numpy()

    @property
    def _hit_surfaces(self):
        """Per-pixel surface id of the last render (host array; -1 where nothing is seen)."""
        return None if self._hits is None else self._hits[1].cpu().numpy()

    def _device(self):
        from .. import engine

        return engine.default_device()

    def render(self):
        """The picture as a host ``(v, h, 4)`` float64 array."""
        self._results = self.render_device().cpu().numpy()
        return self._results

    def render_device(self):
        raise NotImplementedError


class EdgeRender(_Renderer):
    """Outline drawing: black where the surface seen changes between neighbouring pixels."""

    ray_offset_value = 1e-6

    def render_device(self):
        from .. import engine

        self.reset()
        h, v = self._camera.get_resolution()
        scene = engine.DeviceScene.from_components(self._shapes)
        try:
            _, t, surf = scene.render(self._camera, self._device(), light=None, keep_hits=True)
        finally:
            scene.close()
        self._hits = (t, surf)
        rings = max(1, int(max(v, h) / 300))  # renderers.py:106
        canvas = engine.edge_canvas(surf, h, v, rings)
        self._simulation_complete = True
        return canvas


class ShadedRenderer(_Renderer):
    """Gooch-shaded drawing lit from one position."""

    def __init__(self, camera, shapes, light_position):
        self._light = np.asarray(light_position)
        super().__init__(camera, shapes)
        self._surface_lut = tuple(pair for shape in self._shapes for pair in shape.surface_ids)

    def render_device(self):
        from .. import engine

        self.reset()
        scene = engine.DeviceScene.from_components(self._shapes)
        try:
            canvas, t, surf = scene.render(self._camera, self._device(), light=self._light, keep_hits=True)
        finally:
            scene.close()
        self._hits = (t, surf)
        self._simulation_complete = True
        return canvas


def view_of(surfaces, view="xy", bounds=None, resolution=640):
    """Camera, light and matplotlib extent ``draw`` uses for ``surfaces`` (``renderers.py:258-349``).

    "xy" looks down the z axis from 1.5 z_max, "xz" sits at 1.5 y_max; the picture spans 1.5
    times the bounding box, ``resolution`` pixels along its longer side, and the light stands at
    the box's upper corner pushed out three times along the viewing axis."""
    corners = np.hstack([s.bounding_volume.bounding_points[:3] for s in surfaces])
    if bounds is not None:
        mins, maxes = np.asarray(bounds[0]), np.asarray(bounds[1])
    else:
        mins, maxes = np.min(corners, axis=1), np.max(corners, axis=1)
    if view not in ("xy", "xz"):
        return None
    depth, across = (2, 1) if view == "xy" else (1, 2)
    origin = (maxes + mins) / 2
    origin[depth] = 1.5 * maxes[depth]
    h_span, v_span = 1.5 * (maxes[[0, across]] - mins[[0, across]])
    pixels = resolution if h_span > v_span else int(resolution * h_span / v_span)
    camera = OrthographicCamera(pixels, h_span, v_span / h_span)
    if view == "xy":
        camera.rotate_y(90)
    camera.rotate_z(90).move(*origin[:3])
    light = shapes.Point(*maxes)
    light[depth] *= 3 if view == "xy" else -3
    extent = [origin[0] - h_span / 2, origin[0] + h_span / 2,
              origin[across] - v_span / 2, origin[across] + v_span / 2]
    return camera, light, extent


def draw(surfaces, view="xy", axis=None, shaded=True, bounds=None, resolution=640):
    """Render ``surfaces`` in the "xy" or "xz" projection into a matplotlib axis."""
    surfaces = _as_list(surfaces)
    if axis is None:
        import matplotlib.pyplot as plt

        axis = plt.gca()
    framing = view_of(surfaces, view, bounds, resolution)
    if framing is None:  # upstream draws nothing for an unknown view (renderers.py:277-282)
        return
    camera, light, extent = framing
    renderer = ShadedRenderer(camera, surfaces, light_position=light) if shaded else EdgeRender(camera, surfaces)
    axis.imshow(renderer.render(), extent=extent)
    axis.set_axisbelow(True)
