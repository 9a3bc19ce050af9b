"""CPU oracle for the renderer rows: numpy restatement of

    camera grid     tinygfx/g3d/world_objects.py:519-537   OrthographicCamera.generate_rays
    nearest hit     tinygfx/g3d/renderers.py:70-92, 187-209 (_st_propagate of both renderers)
    Gooch shade     tinygfx/g3d/world_objects.py:385-399 + materials/gooch.py:30-65
    shaded canvas   tinygfx/g3d/renderers.py:211-236
    edge canvas     tinygfx/g3d/renderers.py:94-116
    draw() camera   tinygfx/g3d/renderers.py:284-349
    colours         pyrayt/utils.py:5-102 (wavelength_to_rgb)

TEST INFRASTRUCTURE ONLY -- same rule as ``prt_oracle``: imported by ``tests/`` (and smoke /
the bench's cpu leg), never by ``pyrayt_amd``.  Parity is PINNED by ``tests/golden/render.npz``
(outputs of the genuine reference, ``tests/golden/generate_golden.py render``).

Renderer quirk kept on purpose: a component's candidate is ``hits[argmin(where(hits>0, hits,
inf))]`` taken from the *unmasked* list, so a ray with no positive hit on a component offers
that component's smallest (negative) parameter, which then wins the strict-< running minimum.
``draw(view="xz")`` depends on it: its camera sits at +1.5 y_max looking along +y, i.e. away
from the parts, and sees them only through those negative parameters.
"""
import numpy as np

from . import prt_oracle as po

INF = np.inf


def camera_rays(world, h_pixels, v_pixels, h_width, v_width):
    """(2,4,n) rays of the v_pixels x h_pixels grid, row-major, facing +x in camera space."""
    ys = np.linspace(h_width / 2, -h_width / 2, h_pixels)
    zs = np.linspace(v_width / 2, -v_width / 2, v_pixels)
    n = h_pixels * v_pixels
    local = np.zeros((2, 4, n))
    local[0, 1] = np.tile(ys, v_pixels)
    local[0, 2] = np.repeat(zs, h_pixels)
    local[0, 3] = 1.0
    local[1, 0] = 1.0
    rays = np.matmul(np.asarray(world, dtype=float).reshape(4, 4), local)
    rays[1] /= np.linalg.norm(rays[1], axis=0)
    return rays


def nearest_hits(scene, rays):
    """Per pixel (t, surface id) under the renderers' selection rule."""
    rays = np.ascontiguousarray(rays).reshape(2, 4, -1)
    n = rays.shape[-1]
    best_t = np.full(n, INF)
    best_s = np.full(n, -1, dtype=np.int64)
    cols = np.arange(n)
    for root in range(len(scene["roots"])):
        hits, ids = po.component_hits(scene, root, rays)
        row = np.argmin(np.where(hits > 0, hits, INF), axis=0)
        t, s = hits[row, cols], ids[row, cols]
        better = t < best_t
        best_t = np.where(better, t, best_t)
        best_s = np.where(better, s, best_s)
    return best_t, best_s


def gooch_pixels(scene, p, rays, t, shade_warm, shade_cool, light):
    """(4,k) RGBA of the pixels that see primitive ``p``: hit point, world normal, unit vector to
    the single light, warm/cool mix by 0.5 (1 + l.n)."""
    points = rays[0] + t * rays[1]
    with np.errstate(invalid="ignore", divide="ignore"):
        normals = po.world_normals(scene, p, points)
        to_light = np.asarray(light, dtype=float)[:3, None] - points[:3]
        to_light = to_light / np.sqrt((to_light[0] ** 2 + to_light[1] ** 2) + to_light[2] ** 2)
    cosine = (to_light[0] * normals[0] + to_light[1] * normals[1]) + to_light[2] * normals[2]
    mix = 0.5 * (1 + cosine)
    return np.outer(shade_warm, mix) + np.outer(shade_cool, 1 - mix)


def shaded_canvas(scene, gooch, rays, t, surf, light, h_pixels, v_pixels):
    """(v,h,4) image of ShadedRenderer; ``gooch`` is (P,8) = shade_warm | shade_cool per
    primitive.  Pixels that see nothing stay (0,0,0,0)."""
    rays = np.ascontiguousarray(rays).reshape(2, 4, -1)
    canvas = np.zeros((4, rays.shape[-1]))
    for p, sid in enumerate(scene["prim_surface_id"]):
        mask = surf == sid
        if mask.any():
            canvas[:, mask] = gooch_pixels(scene, p, rays[..., mask], t[mask], gooch[p, :4],
                                           gooch[p, 4:], light)
    return canvas.T.reshape(v_pixels, h_pixels, 4)


def edge_mask(surf, h_pixels, v_pixels):
    """Pixels whose surface id differs from the left or upper neighbour (outside = -1), grown by
    max(1, longest side // 300) rings of the 8-neighbourhood."""
    ids = np.asarray(surf).reshape(v_pixels, h_pixels)
    padded = np.full((v_pixels + 1, h_pixels + 1), -1, dtype=ids.dtype)
    padded[1:, 1:] = ids
    edge = (ids != padded[1:, :-1]) | (ids != padded[:-1, 1:])
    for _ in range(max(1, int(max(ids.shape) / 300))):
        grown = np.zeros((v_pixels + 2, h_pixels + 2), dtype=bool)
        for dv in range(3):
            for dh in range(3):
                grown[dv:dv + v_pixels, dh:dh + h_pixels] |= edge
        edge = grown[1:-1, 1:-1]
    return edge


def edge_canvas(surf, h_pixels, v_pixels):
    """(v,h,4) image of EdgeRender: black opaque on edges, white transparent elsewhere."""
    edge = edge_mask(surf, h_pixels, v_pixels)
    canvas = np.empty((v_pixels, h_pixels, 4))
    canvas[..., :3] = ~edge[..., None]
    canvas[..., 3] = edge
    return canvas


def draw_view(corners, view, resolution, bounds=None):
    """Camera + light of ``draw`` for the (3,k) bounding corners of the drawn parts:
    returns (world (4,4), h_pixels, v_pixels, h_width, v_width, light (4,), extent (4,))."""
    if bounds is not None:
        mins, maxes = np.asarray(bounds[0], dtype=float), np.asarray(bounds[1], dtype=float)
    else:
        mins, maxes = np.min(corners, axis=1), np.max(corners, axis=1)
    up = 2 if view == "xy" else 1       # axis the camera sits on
    across = 1 if view == "xy" else 2   # vertical axis of the picture
    origin = (maxes + mins) / 2
    origin[up] = 1.5 * maxes[up]
    h_span = 1.5 * (maxes[0] - mins[0])
    v_span = 1.5 * (maxes[across] - mins[across])
    h_pixels = resolution if h_span > v_span else int(resolution * h_span / v_span)
    aspect = v_span / h_span
    light = np.array((maxes[0], maxes[1], maxes[2], 1.0))
    light[up] *= 3 if view == "xy" else -3
    c, s = np.cos(90 * np.pi / 180.0), np.sin(90 * np.pi / 180.0)  # cos is 6e-17, as upstream
    rot_z = np.array(((c, -s, 0, 0), (s, c, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)))
    rot_y = np.array(((c, 0, s, 0), (0, 1, 0, 0), (-s, 0, c, 0), (0, 0, 0, 1)))
    shift = np.identity(4)
    shift[:3, 3] = origin[:3]
    # camera.rotate_y(90).rotate_z(90).move(origin) for "xy", camera.rotate_z(90).move(origin) for "xz"
    world = shift @ (rot_z @ rot_y if view == "xy" else rot_z)
    extent = np.array((origin[0] - h_span / 2, origin[0] + h_span / 2,
                       origin[across] - v_span / 2, origin[across] + v_span / 2))
    return world, h_pixels, int(aspect * h_pixels), h_span, aspect * h_span, light, extent


def wavelength_to_rgb(wavelength, gamma=0.8):
    """(n,3) display colour of a wavelength in microns: piecewise-linear ramps between
    0.38 / 0.44 / 0.49 / 0.51 / 0.58 / 0.645 / 0.75 um with an intensity roll-off at both ends."""
    w = np.atleast_1d(np.asarray(wavelength, dtype=float))
    rgb = np.empty((3, w.shape[0]))
    zero, one = np.zeros_like(w), np.ones_like(w)
    lo = np.maximum(w, 0.38)
    fade_in = 0.3 + 0.7 * (lo - 0.38) / (0.44 - 0.38)
    hi = np.minimum(w, 0.75)
    fade_out = 0.3 + 0.7 * (0.75 - hi) / (0.75 - 0.645)
    bands = (
        (w < 0.44, (np.abs(-(lo - 0.44) / (0.44 - 0.38) * fade_in) ** gamma, zero, np.abs(fade_in) ** gamma)),
        ((w >= 0.44) & (w < 0.49), (zero, np.abs((w - 0.44) / (0.49 - 0.44)) ** gamma, one)),
        ((w >= 0.49) & (w < 0.51), (zero, one, np.abs((0.51 - w) / (0.51 - 0.49)) ** gamma)),
        ((w >= 0.51) & (w < 0.58), (np.abs((w - 0.51) / (0.58 - 0.51)) ** gamma, one, zero)),
        ((w >= 0.58) & (w < 0.645), (one, np.abs((0.645 - w) / (0.645 - 0.58)) ** gamma, zero)),
        (w >= 0.645, (np.abs(fade_out) ** gamma, zero, zero)),
    )
    for where, colour in bands:
        for k in range(3):
            rgb[k] = np.where(where, colour[k], rgb[k])
    return rgb.T
