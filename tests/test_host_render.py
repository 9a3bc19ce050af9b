"""Host side of the renderer rows (no GPU): the scene objects, camera framing, Gooch tables and
plot helpers built by pyrayt_amd equal what the genuine reference built for the same recipes
(tests/golden/render.npz)."""
import numpy as np
import pytest

import helpers
import scenes
from oracle import render_oracle as ro

NAMES = tuple(scenes.RENDER_SCENES)


@pytest.fixture(scope="module")
def fx():
    return helpers.load("render.npz")


@pytest.fixture()
def api():
    import pyrayt_amd.g3d as cg

    cg.CountedObject.reset_ids()
    return scenes.product_api()


@pytest.mark.parametrize("name", NAMES)
def test_snapshot_camera_and_gooch_table_match_reference(fx, api, name):
    from pyrayt_amd.scene import SceneSnapshot

    surfaces, camera, light = scenes.RENDER_SCENES[name](api)
    snap = SceneSnapshot(surfaces)
    got, want = helpers.flat_scene(snap), helpers.scene_of(fx, name + "__")
    for key in helpers.SCENE_KEYS:
        if key in ("prim_material", "mat_kind", "mat_coef"):
            continue  # material slots are a tracer matter; render materials are compared below
        assert np.array_equal(got[key], want[key]), key
    assert np.array_equal(snap.gooch_table(), fx[name + "__gooch"])
    assert np.array_equal(camera.get_world_transform(), fx[name + "__cam_world"])
    assert tuple(camera.get_resolution()) == tuple(fx[name + "__cam_pixels"])
    assert np.array_equal(np.array(camera.get_span()), fx[name + "__cam_span"])
    assert np.array_equal(np.asarray(light, dtype=float), fx[name + "__light"])


@pytest.mark.parametrize("view", ["xy", "xz"])
def test_draw_framing_matches_reference_and_oracle(fx, api, view):
    from pyrayt_amd.g3d import renderers

    surfaces = scenes.optical_bench(api)
    camera, light, extent = renderers.view_of(surfaces, view, resolution=64)
    assert np.array_equal(np.array(extent), fx[f"draw_{view}_shaded__extent"])
    v, h = fx[f"draw_{view}_shaded__image"].shape[:2]
    assert camera.get_resolution() == (h, v)
    corners = np.hstack([s.bounding_volume.bounding_points[:3] for s in surfaces])
    world, hp, vp, hw, vw, spot, ext = ro.draw_view(corners, view, 64)
    assert np.array_equal(world, camera.get_world_transform())
    assert (hp, vp) == camera.get_resolution() and (hw, vw) == camera.get_span()
    assert np.array_equal(spot, np.asarray(light)) and np.array_equal(ext, np.array(extent))


def test_draw_framing_with_bounds(fx, api):
    from pyrayt_amd.g3d import renderers

    camera, light, extent = renderers.view_of(scenes.optical_bench(api), "xy",
                                              bounds=((-3, -2, -1), (4, 2, 1)), resolution=48)
    assert np.array_equal(np.array(extent), fx["draw_bounds__extent"])
    assert camera.get_resolution()[::-1] == fx["draw_bounds__image"].shape[:2]
    assert renderers.view_of(scenes.optical_bench(api), "yz") is None


def test_camera_geometry():
    import pyrayt_amd.g3d as cg

    cam = cg.OrthographicCamera(640, 3.0, 0.3)
    assert cam.get_resolution() == (640, 192) and cam.get_span() == (3.0, 0.3 * 3.0)
    assert cg.OrthographicCamera(10, 10, 1).get_resolution() == (10, 10)  # test_renderers.py:11


def test_colours_and_gooch_presets():
    from pyrayt_amd.g3d.materials import color, gooch

    c = color.RGBAColor(0.1, 0.2, 0.3)
    assert (c.r, c.g, c.b, c.a) == (0.1, 0.2, 0.3, 1.0)
    c.a = 0.5
    assert c[3] == 0.5 and isinstance(0.5 * c, color.RGBAColor)
    assert np.array_equal(color.ORANGE, (1, 0.5, 0, 1)) and np.array_equal(color.BLACK, (0, 0, 0, 1))
    warm, cool = gooch.BLUE.shade_pair()
    assert np.array_equal(warm, (1 - 0.2) * color.YELLOW + 0.2 * color.BLUE)
    assert np.array_equal(cool, (1 - 0.3) * color.BLUE + 0.3 * color.BLUE)
    plain = gooch.GoochMaterial()
    assert np.array_equal(plain.shade_pair()[0], (0, 0, 0, 1)) and plain.alpha == plain.beta == 0.3
    assert gooch.BLACK.warm_color is color.ORANGE and gooch.BLACK.cool_color is color.BLUE


def test_tracer_materials_render_like_upstream():
    import pyrayt_amd as prt
    from pyrayt_amd.g3d.materials import gooch

    assert prt.materials.absorber._base_material is gooch.BLACK      # materials.py:44
    assert prt.materials.mirror._base_material is gooch.BLUE         # :56
    assert prt.materials.glass["BK7"]._base_material is gooch.BLUE   # :68
    assert prt.g3d.Sphere().material is gooch.BLACK                  # world_objects.py:341
    assert isinstance(prt.materials.mirror, gooch.Material)


def test_single_light_only():
    from pyrayt_amd import engine

    assert np.array_equal(engine._light((1, 2, 3, 1)), (1.0, 2.0, 3.0))
    with pytest.raises(ValueError):
        engine._light(np.zeros((3, 2)))
    with pytest.raises(ValueError):
        engine._light((1.0, 2.0))


def test_plot_helpers_match_reference(fx):
    from pyrayt_amd import utils

    w = fx["utils__wavelengths"]
    assert np.array_equal(utils.wavelength_to_rgb(w), fx["utils__rgb"])
    assert np.array_equal(utils.wavelength_to_rgb(w, gamma=1.7), fx["utils__rgb_gamma"])
    assert utils.lensmakers_equation(2, -2, 1.5, 0.25) == fx["utils__lensmakers"][0]
    assert utils.lensmakers_equation(40, -200, 1.62, 5) == fx["utils__lensmakers"][1]


def test_rendering_without_gpu_raises(api):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pyrayt_amd import engine

    surfaces, camera, light = scenes.render_spheres(api)
    with pytest.raises(engine.EngineUnavailable):
        api.cg.renderers.ShadedRenderer(camera, surfaces, light).render()
    with pytest.raises(engine.EngineUnavailable):
        camera.generate_rays()
