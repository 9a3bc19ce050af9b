"""The renderer oracle against fixtures produced by the genuine reference
(tests/golden/render.npz, made by tests/golden/generate_golden.py render)."""
import numpy as np
import pytest

import helpers
from helpers import load, scene_of
from oracle import render_oracle as ro

NAMES = ("spheres", "bench_xy", "bench_xz", "inside")


@pytest.fixture(scope="module")
def fx():
    return load("render.npz")


def view(fx, name):
    pre = name + "__"
    cam = (fx[pre + "cam_world"], int(fx[pre + "cam_pixels"][0]), int(fx[pre + "cam_pixels"][1]),
           float(fx[pre + "cam_span"][0]), float(fx[pre + "cam_span"][1]))
    return scene_of(fx, pre), cam, {k[len(pre):]: v for k, v in fx.items() if k.startswith(pre)}


@pytest.mark.parametrize("name", NAMES)
def test_camera_grid(fx, name):
    _, cam, want = view(fx, name)
    assert np.array_equal(ro.camera_rays(*cam), want["rays"])


@pytest.mark.parametrize("name", NAMES)
def test_nearest_hits(fx, name):
    scene, _, want = view(fx, name)
    t, surf = ro.nearest_hits(scene, want["rays"])
    assert np.array_equal(surf, want["surf"])
    assert np.array_equal(t, want["t"])


def test_fixture_covers_hits_behind_the_camera(fx):
    for name in ("bench_xz", "inside"):
        t, surf = fx[name + "__t"], fx[name + "__surf"]
        assert np.any((surf >= 0) & (t < 0))


@pytest.mark.parametrize("name", NAMES)
def test_shaded_canvas(fx, name):
    scene, cam, want = view(fx, name)
    got = ro.shaded_canvas(scene, want["gooch"], want["rays"], want["t"], want["surf"], want["light"],
                           cam[1], cam[2])
    assert got.shape == want["shaded"].shape
    assert np.array_equal(got, want["shaded"])


@pytest.mark.parametrize("name", NAMES)
def test_edge_canvas(fx, name):
    _, cam, want = view(fx, name)
    assert np.array_equal(ro.edge_canvas(want["surf"], cam[1], cam[2]), want["edges"])


def test_edge_growth_matches_scipy_for_large_images():
    ndimage = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(5)
    ids = np.where(rng.random((650, 700)) < 0.002, 7, -1)
    ids[100:300, 200:450] = 3
    flat = ids.reshape(-1)
    h = np.abs(np.diff(ids, axis=-1, prepend=-1))
    v = np.abs(np.diff(ids, axis=0, prepend=-1))
    want = ndimage.binary_dilation(h + v, ndimage.generate_binary_structure(2, 2), iterations=2)
    assert np.array_equal(ro.edge_mask(flat, 700, 650), want)


def test_wavelength_colours(fx):
    w = fx["utils__wavelengths"]
    assert np.array_equal(ro.wavelength_to_rgb(w), fx["utils__rgb"])
    assert np.array_equal(ro.wavelength_to_rgb(w, gamma=1.7), fx["utils__rgb_gamma"])


@pytest.mark.parametrize("case", range(12))
def test_nearest_hits_of_arbitrary_rays_match_the_reference(case):
    """The renderers' rule for arbitrary rays, degenerate families included (fixture render_rays.npz): which entry it
    selects when none is positive, and whose id it reports for a -inf entry (the reference: the surface's own)."""
    fx = helpers.load("render_rays.npz")
    prefix = f"case{case}__"
    rays = np.ascontiguousarray(fx[prefix + "rays"]).reshape(2, 4, -1)
    with np.errstate(all="ignore"):
        t, surf = ro.nearest_hits(helpers.scene_of(fx, prefix), rays)
    want_t, want_surf = fx[prefix + "t"], fx[prefix + "surf"]
    assert np.array_equal(surf, want_surf)
    finite = np.isfinite(want_t)
    assert np.array_equal(np.isfinite(t), finite) and np.array_equal(np.isneginf(t), np.isneginf(want_t))
    assert np.allclose(t[finite], want_t[finite], rtol=0, atol=helpers.ATOL)
