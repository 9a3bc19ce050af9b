"""Parity of the HIP engine (through the C-ABI of include/prt.h) with
  (1) the golden vectors produced by the genuine reference, and
  (2) the CPU oracle on seeded inputs the fixtures do not cover,
plus size-independent properties at the north-star size.

Bar (BASELINE.json north_star): intersected-surface index bit-exact; hit point / direction /
refractive index within 1e-6 abs in float64.  helpers.ATOL = 1e-6.
"""
import ctypes

import numpy as np
import pytest

import helpers
import scenes
from oracle import c_oracle
from oracle import prt_oracle as orc
from pyrayt_amd import engine

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SCENE_FIXTURES = ["config1", "config2", "config3", "config4", "config5", "two_mirrors",
                  "tutorial", "mirrors_and_stops", "stopped_lens",
                  # adversarial families (tests/scenes.py adv_*) and upstream's stale cull box
                  "adv_lens", "adv_stop", "adv_prism", "adv_condenser", "adv_still", "adv_short_a", "adv_short_b", "adv_short_c", "adv_bench_a", "adv_bench_b", "adv_bench_c", "stale_box"]
KINDS = ("sphere", "cylinder", "plane", "cube", "paraboloid")
VARIANTS = ("identity", "moved", "rotated", "scaled")


def dev(array):
    return torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64)).to("cuda:0")


FixtureSnapshot = helpers.FixtureSnapshot  # (kept importable from here: the render tests and tools take it from this module)


def device_scene(scene_dict, options=None):
    from pyrayt_amd.engine import DeviceScene

    return DeviceScene(FixtureSnapshot(scene_dict), options=options)


def test_library_loads_on_gpu():
    from pyrayt_amd import engine

    lib = engine.library()
    assert lib.prt_version() == engine.PRT_VERSION
    assert lib.prt_device_count() >= 1


# ---------------------------------------------------------------------------------------------
# golden scenes: whole trace and the stepwise propagate / interact entry points
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", SCENE_FIXTURES)
@pytest.mark.parametrize("flags", [0, 1, 2, 3])
def test_trace_matches_reference(name, flags):
    fx = helpers.load(f"scene_{name}.npz")
    ds = device_scene(helpers.scene_of(fx))
    rows, counts = ds.trace(dev(fx["rays0"]), int(fx["generation_limit"]), flags=flags)
    helpers.assert_frames_match(rows.cpu().numpy().T, fx["frame"], what=f"{name} flags={flags}")
    assert sum(counts) == fx["frame"].shape[0]
    ds.close()


@pytest.mark.parametrize("name", SCENE_FIXTURES)
@pytest.mark.parametrize("knob", ["no_chain", "no_cull", "cull_min", "no_implied", "list_order_groups", "no_intervals", "no_clearance"])
def test_trace_matches_reference_on_every_program_form(name, knob):
    """The same goldens with the scene compiled to the other program forms (prt_scene_options): the step
    interpreter instead of chain steps, no component cull steps, cull steps from two components on, every
    cull box tested exactly, the cull hierarchy in list order."""
    fx = helpers.load(f"scene_{name}.npz")
    ds = device_scene(helpers.scene_of(fx), options={knob: 2 if knob == "cull_min" else 1})
    info = ds.info()
    if knob == "no_chain":
        assert info["chain_steps"] == 0
    if knob == "no_cull":
        assert info["cull_steps"] == 0
    rows, counts = ds.trace(dev(fx["rays0"]), int(fx["generation_limit"]))
    helpers.assert_frames_match(rows.cpu().numpy().T, fx["frame"], what=f"{name} {knob}")
    t, surf = ds.propagate(dev(fx["rays0"]))
    assert np.array_equal(surf.cpu().numpy(), fx["surf_0"]), f"{name} {knob}: surfaces"
    ds.close()


@pytest.mark.parametrize("name", SCENE_FIXTURES)
@pytest.mark.parametrize("variant", ["lanes4", "lanes8,lds", "lanes16", "lds"])
def test_surface_parallel_variants_match_reference(name, variant):
    """The k-lanes-per-ray nearest-hit kernels (K lanes of a wave share a ray, each takes components
    j, j+K, ..., shuffle min-reduce over (t, component order)) and the LDS-staged program fetch give
    the reference's surfaces and frames exactly like the lane-per-ray kernel."""
    lanes = [int(part[5:]) for part in variant.split(",") if part.startswith("lanes")]
    fx = helpers.load(f"scene_{name}.npz")
    ds = device_scene(helpers.scene_of(fx), options={"hit_lanes": lanes[0] if lanes else 0,
                                                     "hit_staged": int("lds" in variant)})
    t, surf = ds.propagate(dev(fx["rays0"]))
    assert np.array_equal(surf.cpu().numpy(), fx["surf_0"]), f"{name} {variant}: surfaces"
    assert np.allclose(t.cpu().numpy(), fx["t_0"], rtol=0, atol=helpers.ATOL), f"{name} {variant}: t"
    rows, counts = ds.trace(dev(fx["rays0"]), int(fx["generation_limit"]), flags=2)
    helpers.assert_frames_match(rows.cpu().numpy().T, fx["frame"], what=f"{name} {variant}")
    assert ds.trace_stats()["variant"] == (3 if "lanes" in variant else 2)
    ds.close()


@pytest.mark.parametrize("name", SCENE_FIXTURES)
def test_stepwise_matches_reference(name):
    """prt_propagate + prt_interact generation by generation against the reference's
    intermediates: t, surface ids (exact) and the ray set after each interaction."""
    fx = helpers.load(f"scene_{name}.npz")
    ds = device_scene(helpers.scene_of(fx))
    limit = int(fx["generation_limit"])
    rays = dev(fx["rays0"])
    blocks = []
    for g in range(int(fx["n_generations"])):
        t, surf = ds.propagate(rays)
        assert np.array_equal(surf.cpu().numpy(), fx[f"surf_{g}"]), f"{name}: surfaces gen {g}"
        assert np.allclose(t.cpu().numpy(), fx[f"t_{g}"], rtol=0, atol=helpers.ATOL), f"{name}: t gen {g}"
        rows, nxt = ds.interact(rays, t, surf, g, limit)
        if rows.shape[1] == 0:  # every ray dead: nothing recorded, the loop ends
            assert g == int(fx["n_generations"]) - 1
            break
        blocks.append(rows.cpu().numpy().T)
        assert np.allclose(nxt.cpu().numpy(), fx[f"next_{g}"], rtol=0, atol=helpers.ATOL,
                           equal_nan=True), f"{name}: state after gen {g}"
        rays = nxt.contiguous()
    helpers.assert_frames_match(np.vstack(blocks), fx["frame"], what=name)
    ds.close()


# ---------------------------------------------------------------------------------------------
# per-object entry points against the golden vectors
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("variant", VARIANTS)
def test_primitive_intersect_and_normals(kind, variant):
    fx = helpers.load("primitives.npz")
    key = f"{kind}_{variant}__"
    ds = device_scene(helpers.scene_of(fx, key))
    hits, ids = ds.intersect(0, dev(fx[key + "rays"][:8]))
    hits = hits.cpu().numpy()
    want = np.where(np.isnan(fx[key + "hits"]), np.inf, fx[key + "hits"])  # NaN == miss
    assert np.array_equal(np.isfinite(hits), np.isfinite(want))
    assert np.allclose(hits, want, rtol=0, atol=helpers.ATOL)
    has = fx[key + "has_hit"]
    normals = ds.world_normals(0, dev(fx[key + "points"])).cpu().numpy()
    assert np.allclose(normals[:, has], fx[key + "normals"][:, has], rtol=0, atol=helpers.ATOL,
                       equal_nan=True)
    ds.close()


@pytest.mark.parametrize("name", ["union_spheres", "intersect_spheres", "difference_spheres",
                                  "plane_minus_cylinder", "cube_chain", "right_nested", "balanced"])
def test_csg_component_intersect(name):
    fx = helpers.load("csg.npz")
    key = name + "__"
    ds = device_scene(helpers.scene_of(fx, key))
    hits, ids = ds.intersect(0, dev(fx[key + "rays"][:8]))
    assert np.array_equal(ids.cpu().numpy(), fx[key + "ids"])
    assert np.allclose(hits.cpu().numpy(), fx[key + "hits"], rtol=0, atol=helpers.ATOL)
    ds.close()


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("material", ["absorber", "mirror", "ideal", "SF5"])
def test_material_trace(kind, material):
    fx = helpers.load("shading.npz")
    key = f"trace_{kind}_{material}__"
    ds = device_scene(helpers.scene_of(fx, key))
    rays = dev(fx[key + "in"])
    ds.material_trace(0, rays)
    assert np.allclose(rays.cpu().numpy(), fx[key + "out"], rtol=0, atol=helpers.ATOL, equal_nan=True)
    ds.close()


# ---------------------------------------------------------------------------------------------
# object API (the reference's own unit tests, run through the HIP engine)
# ---------------------------------------------------------------------------------------------
def test_object_api_known_answers():
    import pyrayt_amd as pyrayt
    from pyrayt_amd import g3d as cg

    # test_world_objects.py:277-282 moved sphere hit t=1
    rays = cg.bundle_of_rays(1)
    rays[1, 0, 0] = 1
    hits, ids = cg.Sphere(1).move_x(3).intersect(rays)
    assert np.allclose(np.sort(hits[:, 0]), (2, 4))
    # test_pyrayt_materials.py:15-21, :29-46 absorber zeroes, mirror flips z
    surf = cg.XYPlane(material=pyrayt.materials.mirror)
    rs = pyrayt.RaySet(4)
    rs.rays[1, 2] = -1
    pyrayt.materials.absorber.trace(surf, rs)
    assert np.all(rs.rays[1] == 0)
    rs.rays[1, 2] = -1
    pyrayt.materials.mirror.trace(surf, rs)
    assert np.allclose(rs.rays[1, 2], 1)
    # :56-71 entering n=1.6 updates the index, leaving resets to 1
    rs = pyrayt.RaySet(2)
    rs.rays[1, 2] = -1
    pyrayt.materials.BasicRefractor(1.6).trace(surf, rs)
    assert np.allclose(rs.index, 1.6)
    rs.rays[1, 2] = 1
    pyrayt.materials.BasicRefractor(1.6).trace(surf, rs)
    assert np.allclose(rs.index, 1.0)
    n = cg.Sphere(2).get_world_normals(np.array([[0.0, 2.0], [0.0, 0.0], [2.0, 0.0], [1.0, 1.0]]))
    assert np.allclose(n[:3].T, ((0, 0, 1), (1, 0, 0)))


def test_raytracer_known_answers():
    """test/test_pyrayt/test_core.py:45-98 and the integration test
    test/integration_tests/int_test_ray_plane_intersection.py:24-54."""
    import pyrayt_amd as pyrayt
    from pyrayt_amd import g3d as cg

    source = pyrayt.components.LineOfRays()
    mirror = cg.XYPlane(material=pyrayt.materials.mirror).rotate_y(-90).move_x(3)
    tracer = pyrayt.RayTracer([source], [mirror])
    tracer.set_rays_per_source(10)
    res = tracer.trace()
    assert res.shape == (10, 15) and np.allclose(res["x1"], 3.0)
    second = cg.XYPlane(material=pyrayt.materials.mirror).rotate_y(90).move_x(-3)
    tracer = pyrayt.RayTracer([source], [mirror, second], generation_limit=10)
    tracer.set_rays_per_source(10)
    res = tracer.trace()
    assert res.shape[0] == 100 and set(res["generation"]) == set(range(10))
    tracer = pyrayt.RayTracer([pyrayt.components.LineOfRays(), pyrayt.components.LineOfRays()], mirror)
    tracer.set_rays_per_source(10)
    res = tracer.trace()
    assert res.shape[0] == 20
    tracer.calculate_source_ids()
    assert set(tracer.get_results()["source_id"]) == {0, 1}

    lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
    focus = scenes.lensmakers_equation(2, -2, 1.5, 0.25)
    src = pyrayt.components.ConeOfRays(cone_angle=6).move_x(-focus)
    baffle = pyrayt.components.baffle((1, 1)).move_x(1)
    tracer = pyrayt.RayTracer(src, [lens, baffle])
    tracer.set_rays_per_source(50)
    tracer.set_generation_limit(100)
    res = tracer.trace()
    assert len(res) == 150
    assert np.allclose(res.loc[res["generation"] == 2]["x1"], 1.0)
    assert list(res.columns) == ["generation", "intensity", "wavelength", "index", "id", "surface",
                                 "x0", "y0", "z0", "x1", "y1", "z1", "x_tilt", "y_tilt", "z_tilt"]
    assert all(str(t) == "float64" for t in res.dtypes)


def test_untracable_surface_raises():
    import pyrayt_amd as pyrayt
    from pyrayt_amd import g3d as cg

    tracer = pyrayt.RayTracer(pyrayt.components.LineOfRays(), cg.Sphere(1).move_x(3))
    with pytest.raises(AttributeError):
        tracer.trace()


# ---------------------------------------------------------------------------------------------
# HIP engine vs oracle on seeded inputs beyond the fixtures
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,args,limit", [
    ("config2", (50_000,), 10), ("config3", (30_000,), 10), ("config4", (4_000,), 10),
    ("config5", (30_000,), 10), ("mirrors_and_stops", (40_000,), 8), ("stopped_lens", (30_000,), 10),
])
def test_trace_matches_oracle(name, args, limit):
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.SCENES[name](scenes.product_api(), *args)
    snap = SceneSnapshot(parts)
    want, want_counts = orc.trace(helpers.flat_scene(snap), rays, limit)
    ds = DeviceScene(snap)
    rows, counts = ds.trace(dev(rays), limit)
    assert counts == want_counts
    helpers.assert_frames_match(rows.cpu().numpy().T, want, what=name)
    ds.close()


def test_empty_ragged_and_tiny_inputs():
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.scene import SceneSnapshot

    parts, rays = scenes.config2(scenes.product_api(), 1000)
    snap = SceneSnapshot(parts)
    ds = DeviceScene(snap)
    flat = helpers.flat_scene(snap)
    for n in (0, 1, 63, 64, 65, 255, 256, 257, 999):
        sub = np.ascontiguousarray(rays[:, :n])
        rows, counts = ds.trace(dev(sub), 10)
        want, want_counts = orc.trace(flat, sub, 10) if n else (np.zeros((0, 15)), [])
        assert counts == want_counts, n
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"n={n}")
    # strided input (a column slice of a wider buffer) and generation_limit 1
    wide = dev(rays)
    rows, counts = ds.trace(wide[:, 100:400], 1)
    want, want_counts = orc.trace(flat, rays[:, 100:400], 1)
    assert counts == want_counts
    helpers.assert_frames_match(rows.cpu().numpy().T, want, what="strided")
    # rays that all miss: nothing recorded
    away = rays.copy()
    away[4] = -1.0
    away[5:7] = 0.0
    rows, counts = ds.trace(dev(away), 10)
    assert rows.shape[1] == 0 and counts == []
    ds.close()


# ---------------------------------------------------------------------------------------------
# north-star size: 1M rays, checked through the reference's summary + invariants
# ---------------------------------------------------------------------------------------------
def test_config2_one_million_rays():
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    fx = helpers.load("config2_1m_summary.npz")
    n = int(fx["n"])
    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), n)
    ds = DeviceScene(SceneSnapshot(parts))
    rows, counts = ds.trace(dev(rays), 10)
    frame = rows.cpu().numpy().T
    assert frame.shape[0] == int(fx["rows"]) == 2999991
    gens = frame[:, 0].astype(np.int64)
    surf = frame[:, 5].astype(np.int64)
    pairs, pair_counts = np.unique(np.stack((gens, surf)), axis=1, return_counts=True)
    assert np.array_equal(pairs, fx["gen_surface_pairs"])
    assert np.array_equal(pair_counts, fx["gen_surface_counts"])
    # the near-axial rays that skip the second lens surface: same ray ids
    detector = surf.max()
    assert np.array_equal(frame[(gens == 1) & (surf == detector), 4].astype(np.int64), fx["q5_ids"])
    assert int((surf * (gens + 1)).sum()) == int(fx["surface_checksum"])
    assert np.allclose(frame.sum(axis=0), fx["column_sums"], rtol=1e-9, atol=1e-3)
    assert np.allclose(frame[fx["sample_index"]], fx["sample_rows"], rtol=0, atol=helpers.ATOL)
    # invariants: generation-major, ids ascending inside a generation, unit tilts,
    # segment end of generation g == segment start of generation g+1 (minus the 1e-6 re-launch)
    assert np.all(np.diff(gens) >= 0)
    for g in range(3):
        ids = frame[gens == g, 4]
        assert np.all(np.diff(ids) > 0)
    assert np.allclose(np.linalg.norm(frame[:, 12:15], axis=1), 1.0, atol=1e-12)
    g0, g1 = frame[gens == 0], frame[gens == 1]
    assert np.allclose(g0[:, 9:12], g1[:, 6:9], atol=2e-6)
    # determinism: a second run is bit-identical
    rows2, _ = ds.trace(dev(rays), 10)
    assert torch.equal(rows, rows2)
    ds.close()


@pytest.mark.skipif(bool(engine.DEFAULT_TRACE_FLAGS & 3), reason="counts the launches of the fused path without upstream's extra generation")
def test_lookback_stall_falls_back_to_three_kernel_path():
    """If the decoupled look-back ever gave up (it relies on in-order workgroup dispatch, which
    is observed but not promised), prt_trace must transparently redo the loop on the path with
    no inter-workgroup dependency and return the same frame."""
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.scene import SceneSnapshot

    parts, rays = scenes.config2(scenes.product_api(), 20_000)
    ds = DeviceScene(SceneSnapshot(parts))
    want, want_counts = ds.trace(dev(rays), 10)
    assert ds.trace_stats()["kernel_launches"] == 4  # one fused launch per generation (+1 empty)
    got, counts = ds.trace(dev(rays), 10, flags=engine.TRACE_TEST_STALL)
    assert ds.trace_stats()["kernel_launches"] == 12  # 3 generations x 4 kernels
    assert counts == want_counts and torch.equal(got, want)
    ds.close()


# ---------------------------------------------------------------------------------------------
# device-side sources (prt_generate_rays) against the reference's source fixtures and the
# host implementations of the same patterns
# ---------------------------------------------------------------------------------------------
def _source_recipes():
    import pyrayt_amd.components as c

    return {
        "line": lambda: c.LineOfRays(spacing=0.1, wavelength=0.5).move_x(-0.5).rotate_y(-3),
        "circle": lambda: c.CircleOfRays(diameter=2.0).move(0.1, 0.2, 0.3),
        "cone": lambda: c.ConeOfRays(6).move_x(-1.9).rotate_z(10),
        "wedge": lambda: c.WedgeOfRays(30, wavelength=0.7).rotate_x(45),
    }


@pytest.mark.parametrize("name", ["line", "circle", "cone", "wedge"])
def test_device_sources_match_reference(name):
    from pyrayt_amd import engine

    fx = helpers.load("sources.npz")
    device = torch.device("cuda", 0)
    for n in (1, 7, 100):
        got = engine.generate_rays([_source_recipes()[name]()], n, device).cpu().numpy()
        assert np.allclose(got, fx[f"{name}_{n}"], rtol=0, atol=1e-14), (name, n)
    # large n against the host implementation of the same pattern, plus a sub-range (a shard)
    src = _source_recipes()[name]()
    want = np.asarray(src.generate_rays(100_003))
    got = engine.generate_rays([src], 100_003, device).cpu().numpy()
    assert np.allclose(got, want, rtol=0, atol=1e-14)
    part = engine.generate_rays([src], 100_003, device, lo=33_333, hi=77_777).cpu().numpy()
    assert np.array_equal(part, got[:, 33_333:77_777])


def test_device_sources_concatenate_like_the_tracer():
    """Several sources: consecutive ids, per-source wavelength, ranges that straddle sources."""
    import pyrayt_amd as pyrayt
    from pyrayt_amd import engine

    srcs = [pyrayt.components.LineOfRays(0.1, wavelength=w).move_x(-0.5).rotate_y(-3)
            for w in np.linspace(0.44, 0.75, 5)]
    tracer = pyrayt.RayTracer(srcs, pyrayt.components.baffle((1, 1)), rays_per_source=1000)
    want = np.asarray(tracer.initial_ray_set())
    device = torch.device("cuda", 0)
    got = engine.generate_rays(srcs, 1000, device).cpu().numpy()
    assert np.allclose(got, want, rtol=0, atol=1e-14)
    assert np.array_equal(got[12], np.arange(5000))
    part = engine.generate_rays(srcs, 1000, device, lo=1500, hi=3200).cpu().numpy()
    assert np.array_equal(part, got[:, 1500:3200])


def test_lamp_on_device_is_lambertian():
    import pyrayt_amd as pyrayt
    from pyrayt_amd import engine

    lamp = pyrayt.components.Lamp(2.0, 3.0, max_angle=60).move_x(1.0)
    rays = engine.generate_rays([lamp], 400_000, torch.device("cuda", 0)).cpu().numpy()
    d = rays[4:7]
    assert np.allclose(np.linalg.norm(d, axis=0), 1.0, atol=1e-12)
    cos_t = d[0]
    assert cos_t.min() >= np.cos(np.radians(60)) - 1e-12
    assert np.allclose(rays[9], 100.0 * cos_t, atol=1e-9)          # intensity = 100 cos(theta)
    # inverse-CDF sampling: cos(theta) uniform on [cos(max), 1]
    assert abs(cos_t.mean() - 0.75) < 2e-3 and abs(np.median(cos_t) - 0.75) < 3e-3
    assert np.allclose(rays[0], 1.0) and abs(rays[1].mean()) < 5e-3 and abs(rays[2].mean()) < 8e-3
    assert rays[1].min() >= -1.0 and rays[1].max() <= 1.0 and rays[2].min() >= -1.5 and rays[2].max() <= 1.5
    assert abs(rays[1].std() - 2.0 / np.sqrt(12)) < 5e-3
    host = np.asarray(lamp.generate_rays(50_000))                    # same law as the host Lamp
    assert abs(host[4].mean() - cos_t.mean()) < 5e-3


def test_tracer_with_device_sources_equals_host_sources():
    import pyrayt_amd as pyrayt

    def build():
        lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
        focus = scenes.lensmakers_equation(2, -2, 1.5, 0.25)
        src = pyrayt.components.ConeOfRays(cone_angle=6).move_x(-focus)
        baffle = pyrayt.components.baffle((1, 1)).move_x(1)
        return pyrayt.RayTracer(src, [lens, baffle], rays_per_source=20_000, generation_limit=100)

    on_device = build()
    frame_d = on_device.trace()
    on_host = build()
    on_host.device_sources = False
    frame_h = on_host.trace()
    assert frame_d.shape == frame_h.shape == (60_000, 15)
    assert np.array_equal(frame_d["id"].to_numpy(), frame_h["id"].to_numpy())
    assert np.array_equal(frame_d["generation"].to_numpy(), frame_h["generation"].to_numpy())
    # surface ids differ by a constant offset only (two scene builds draw different global ids)
    off = frame_d["surface"].to_numpy() - frame_h["surface"].to_numpy()
    assert np.all(off == off[0])
    cols = ["x0", "y0", "z0", "x1", "y1", "z1", "x_tilt", "y_tilt", "z_tilt", "index"]
    assert np.allclose(frame_d[cols].to_numpy(), frame_h[cols].to_numpy(), rtol=0, atol=1e-9)


def test_device_frame_selections_match_pandas():
    """trace_device(): the result stays in HBM; selections / reductions agree with pandas."""
    import pyrayt_amd as pyrayt

    lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
    focus = scenes.lensmakers_equation(2, -2, 1.5, 0.25)
    src = pyrayt.components.ConeOfRays(cone_angle=6).move_x(-focus)
    baffle = pyrayt.components.baffle((1, 1)).move_x(1)
    tracer = pyrayt.RayTracer(src, [lens, baffle], rays_per_source=5000, generation_limit=10)
    dframe = tracer.trace_device()
    assert dframe.rows.is_cuda and dframe.shape == (15000, 15)
    frame = tracer.get_results()                      # lazy conversion
    assert frame.shape == (15000, 15) and list(frame.columns) == list(pyrayt.DeviceFrame.columns)
    det = baffle.get_id()
    on_det = dframe.where(surface=det)
    want = frame.loc[frame["surface"] == det]
    assert len(on_det) == len(want) == 5000
    assert np.array_equal(on_det.to_numpy(), want.to_numpy())
    g1 = dframe.generation(1)
    assert np.array_equal(g1.to_numpy(), frame.loc[frame["generation"] == 1].to_numpy())
    (cy, cz), rms = on_det.spot()
    assert np.isclose(cy, want["y1"].mean()) and np.isclose(cz, want["z1"].mean())
    assert np.isclose(rms, np.sqrt(((want["y1"] - cy) ** 2 + (want["z1"] - cz) ** 2).mean()))
    tracer.calculate_source_ids()
    assert set(tracer.get_results()["source_id"]) == {0}


# ---------------------------------------------------------------------------------------------
# BASELINE configs 3-5 at (per-GPU) full size: size-independent properties + the oracle on a
# subsample.  Rays are independent, so the rows of any subset of rays must be exactly the rows
# the full trace recorded for those ids.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,args,limit", [
    ("config3", (4_000_000,), 10),          # Cooke triplet, 12 primitives, 4M rays
    ("config4", (1_000_000,), 10),          # prism, 8 wavelengths x 1M rays: one GPU's share of the job
    ("config4", (8_000_000,), 4),           # ... and the WHOLE job on one GPU: 8 wavelengths x 8M = 64M rays, 192M rows (23 GB)
    ("config5", (2_000_000,), 10),          # 16M rays / 8 GPUs
    ("config5", (16_000_000,), 10),         # ... and all 16M of them on one GPU (the largest BASELINE job)
])
def test_baseline_configs_at_full_size(name, args, limit):
    from oracle import c_oracle
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.SCENES[name](scenes.product_api(), *args)
    n = rays.shape[1]
    snap = SceneSnapshot(parts)
    ds = DeviceScene(snap)
    rows, counts = ds.trace(dev(rays), limit)
    assert rows.shape[1] == sum(counts) > n
    gens = rows[0]
    ids = rows[4]
    # generation-major, ids strictly ascending inside a generation
    assert bool((gens[1:] >= gens[:-1]).all())
    start = 0
    for g, c in enumerate(counts):
        seg = ids[start:start + c]
        assert bool((gens[start:start + c] == g).all()) and bool((seg[1:] > seg[:-1]).all())
        start += c
    tilt = torch.sqrt(rows[12] ** 2 + rows[13] ** 2 + rows[14] ** 2)
    assert bool(((tilt - 1).abs() < 1e-12).all())
    # subsample: every 64th ray through the C oracle
    pick = np.arange(0, n, 64)
    want, want_counts = c_oracle.trace(helpers.flat_scene(snap), np.ascontiguousarray(rays[:, pick]), limit)
    keep = torch.isin(ids, torch.from_numpy(pick.astype(np.float64)).to(ids.device))
    got = rows[:, keep].cpu().numpy().T
    helpers.assert_frames_match(got, want, what=f"{name} subsample")
    # and tracing the subset alone gives exactly those rows (independence of rays)
    sub_rows, sub_counts = ds.trace(dev(rays[:, pick]), limit)
    assert sub_counts == want_counts and np.array_equal(sub_rows.cpu().numpy().T, got)
    ds.close()


def test_two_scenes_taking_turns_on_one_workspace():
    """A scene leaves its control words in the workspace for its own next trace; a block that another
    scene used in between is re-initialised (the library keeps track of who traced last with an address)."""
    import ctypes

    from pyrayt_amd import engine
    from pyrayt_amd.engine import DeviceScene

    fa, fb = helpers.load("scene_config2.npz"), helpers.load("scene_config5.npz")
    n = 1000
    ra = dev(np.ascontiguousarray(fa["rays0"][:, :n]))
    rb = dev(np.ascontiguousarray(fb["rays0"][:, :n]))
    a, b = device_scene(helpers.scene_of(fa)), device_scene(helpers.scene_of(fb))
    want_a, _ = a.trace(ra, 10)
    want_b, _ = b.trace(rb, 10)
    want_a, want_b = want_a.cpu().numpy().copy(), want_b.cpu().numpy().copy()
    lib = engine.library()
    work = torch.empty(int(lib.prt_trace_workspace_bytes(n)), dtype=torch.uint8, device="cuda:0")
    rows = torch.empty((15, n * 10), dtype=torch.float64, device="cuda:0")
    counts = (ctypes.c_int64 * 10)()
    stream = engine._stream_ptr(torch, rows.device)
    for turn in range(6):
        for scene, rays, want in ((a, ra, want_a), (b, rb, want_b), (a, ra, want_a)):
            total = lib.prt_trace(scene.handle, 0, rays.data_ptr(), n, rays.stride(0), 10, engine.DEFAULT_RAY_OFFSET,
                                  rows.data_ptr(), rows.shape[1], counts, work.data_ptr(), 0, stream)
            assert total == want.shape[1], (turn, total)
            assert np.array_equal(rows[:, :total].cpu().numpy(), want, equal_nan=True)
    a.close()
    b.close()


# ---------------------------------------------------------------------------------------------
# a design loop: the same parts with other numbers go into the scene object that is already there
# ---------------------------------------------------------------------------------------------
def test_scene_update_in_place_equals_a_new_scene():
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    api = scenes.product_api()
    CountedObject.reset_ids()
    parts, rays = scenes.config3(api, 30_000)
    ds = DeviceScene(SceneSnapshot(parts))
    device_rays = dev(rays)
    ds.trace(device_rays, 10)
    ds.trace(device_rays, 10)                                   # hints in place
    for step in range(4):
        parts[1].move_x(0.05).rotate_y(0.3)                       # the flint element wanders
        parts[4].move_x(-0.1)
        snap = SceneSnapshot(parts)
        assert ds.update(snap) is True
        want, want_counts = orc.trace(helpers.flat_scene(snap), rays, 10)
        for turn in range(2):
            rows, counts = ds.trace(device_rays, 10)
            assert counts == want_counts, (step, turn)
            helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"updated scene step {step} turn {turn}")
        fresh = DeviceScene(snap)
        rows_fresh, _ = fresh.trace(device_rays, 10)
        assert np.array_equal(rows_fresh.cpu().numpy(), rows.cpu().numpy(), equal_nan=True)
        t, surf = ds.propagate(device_rays)                       # the per-component programs were updated too
        t_fresh, surf_fresh = fresh.propagate(device_rays)
        assert torch.equal(surf, surf_fresh) and torch.equal(t, t_fresh)
        fresh.close()
    # another shape (one part fewer) does not fit: the scene stays as it was
    assert ds.update(SceneSnapshot(parts[:-1] + [])) is False or len(parts) == 1
    rows_after, _ = ds.trace(device_rays, 10)
    assert np.array_equal(rows_after.cpu().numpy(), rows.cpu().numpy(), equal_nan=True)
    ds.close()


def test_raytracer_keeps_its_scene_when_a_part_moves():
    import pyrayt_amd as pyrayt

    lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
    src = pyrayt.components.ConeOfRays(cone_angle=6).move_x(-1.9)
    det = pyrayt.components.baffle((1, 1)).move_x(1)
    tracer = pyrayt.RayTracer(src, [lens, det], rays_per_source=10_000)
    tracer.trace_device()
    scene = tracer._device_scene()
    spots = []
    for _ in range(3):
        det.move_x(0.05)
        frame = tracer.trace_device()
        assert tracer._device_scene() is scene                   # updated in place, not rebuilt
        spots.append(float(frame.group_stats(surface=det.get_id())["rms_radius"].iloc[0]))
        want = pyrayt.RayTracer(src, [lens, det], rays_per_source=10_000).trace()
        got = tracer.get_results()
        assert np.array_equal(got.to_numpy(), want.to_numpy(), equal_nan=True)
    assert spots[0] != spots[1] != spots[2]                       # the detector really moved


def test_raytracer_looks_at_its_system_again_only_when_something_was_assigned():
    """RayTracer keeps the compiled scene and the generated ray set while the change counter of the scene objects
    (g3d.objects.SceneEpoch) stands still, and picks up every kind of edit the moment it is made: a part moved, a
    source moved, a material replaced, normals flipped, the ray count or the component list changed -- each time
    the frame is that of a tracer built from scratch on the edited system."""
    import pyrayt_amd as pyrayt
    from pyrayt_amd import scene as scene_module

    pyrayt.g3d.objects.CountedObject.reset_ids()
    lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
    src = pyrayt.components.ConeOfRays(cone_angle=6).move_x(-1.9)
    det = pyrayt.components.baffle((1, 1)).move_x(1)
    tracer = pyrayt.RayTracer(src, [lens, det], rays_per_source=5_000)

    def fresh():
        return pyrayt.RayTracer(src, tracer.get_system(), rays_per_source=tracer.get_rays_per_source()).trace()

    def same(a, b):
        return a.shape == b.shape and np.array_equal(a.to_numpy(), b.to_numpy(), equal_nan=True)

    first = tracer.trace()
    snapshots = []
    original = scene_module.SceneSnapshot.__init__

    def counting(self, *args, **kwargs):
        snapshots.append(1)
        original(self, *args, **kwargs)

    scene_module.SceneSnapshot.__init__ = counting
    try:
        rays_before = tracer._ray_cache[1]
        for _ in range(3):
            assert same(tracer.trace(), first)
        assert not snapshots and tracer._ray_cache[1] is rays_before   # nothing was looked at, nothing regenerated
        det.move_x(0.03)
        assert same(tracer.trace(), fresh()) and len(snapshots) >= 1
        src.move_x(-0.02)
        moved = tracer.trace()
        assert tracer._ray_cache[1] is not rays_before and same(moved, fresh()) and not same(moved, first)
        det.material = pyrayt.materials.mirror               # an attribute assigned: seen
        assert same(tracer.trace(), fresh())
        det.material = pyrayt.materials.absorber
        tracer.set_rays_per_source(7_000)
        assert len(tracer.trace()) == len(fresh())
        tracer.set_rays_per_source(5_000)
        extra = pyrayt.components.baffle((1, 1)).move_x(0.6)
        tracer.load_components([lens, extra, det])
        assert same(tracer.trace(), fresh())
        # an array edited in place is not an assignment: upstream's own cached inverse would be stale too
        before = tracer.trace()
        extra._world[0, 3] += 0.05                               # (the part the rays end on)
        extra._object[...] = np.linalg.inv(extra._world)
        count = len(snapshots)
        assert same(tracer.trace(), before) and len(snapshots) == count
        tracer.invalidate()
        assert same(tracer.trace(), fresh()) and not same(tracer.get_results(), before)
    finally:
        scene_module.SceneSnapshot.__init__ = original
    # a material's own numbers edited in place count as well (Material.__setattr__): no trace with the old index
    own = pyrayt.materials.BasicRefractor(1.5)
    lens_b = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1, material=own)
    tracer_b = pyrayt.RayTracer(src, [lens_b, det], rays_per_source=5_000)
    before = tracer_b.trace()
    own._refractive_index = 1.62
    after = tracer_b.trace()
    want = pyrayt.RayTracer(src, [lens_b, det], rays_per_source=5_000).trace()
    assert same(after, want) and not same(after, before)


def test_scene_updates_are_ordered_between_the_traces_around_them():
    """prt_scene_update overwrites the scene tables with one stream-ordered copy -- behind the kernels of the traces
    before it, in front of those after it, with no device-wide synchronisation.  Moving a part before every trace,
    with the traces on different tickets and streams and the previous trace's rows not waited for, every frame must
    be the one a scene compiled from scratch at that position gives."""
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), 200_000, seed=5)
    device_rays = dev(rays)
    ds = DeviceScene(SceneSnapshot(parts))
    streams = ds.ticket_streams(device_rays.device, 3)
    blocks = [torch.empty((15, rays.shape[1] * 10), dtype=torch.float64, device="cuda:0") for _ in range(3)]
    got, positions = [], []
    for step in range(9):
        parts[1].move_x(0.01 * (1 + step % 3))
        positions.append(parts[1].get_world_transform())
        assert ds.update(SceneSnapshot(parts))
        ticket = step % 3
        streams[ticket].wait_stream(torch.cuda.current_stream())
        ds.trace_begin(ticket, device_rays, 10, blocks[ticket], stream=streams[ticket])
        rows, counts = ds.trace_end(ticket)           # (counts are out; the last kernel may still be storing rows)
        torch.cuda.current_stream().wait_stream(streams[ticket])
        got.append((rows.clone(), counts))            # (the clone is ordered behind the trace; the next update is not waited for)
    torch.cuda.synchronize()
    for step, (rows, counts) in enumerate(got):
        parts[1]._world = positions[step]
        parts[1]._object = np.linalg.inv(positions[step])
        reference = DeviceScene(SceneSnapshot(parts))
        want, want_counts = reference.trace(device_rays, 10)
        assert counts == want_counts and torch.equal(rows, want), step
        reference.close()
    # an entry point outside the ticket runtime between two updates: the update falls back to synchronising
    ds.propagate(device_rays)
    parts[1].move_x(0.02)
    assert ds.update(SceneSnapshot(parts))
    rows, counts = ds.trace(device_rays, 10)
    reference = DeviceScene(SceneSnapshot(parts))
    want, want_counts = reference.trace(device_rays, 10)
    assert counts == want_counts and torch.equal(rows, want)
    reference.close()
    ds.close()


# ---------------------------------------------------------------------------------------------
# compact state: rows 3, 7, 8 of the ray state are not carried between generations while they hold
# what RaySet's defaults put there; a ray set that differs anywhere must get all 13 rows
# ---------------------------------------------------------------------------------------------
def _odd_rays(kind, rays):
    odd = rays.copy()
    k = odd.shape[1] // 3
    if kind == "origin_w":
        odd[3, k] = 2.0
    elif kind == "direction_w":
        odd[7, k] = 1e-3
    elif kind == "minus_zero_direction_w":
        odd[7, k] = -0.0
    elif kind == "generation_offset":
        odd[8] = 5.0
    elif kind == "one_generation":
        odd[8, k] = 1.0
    return odd


@pytest.mark.parametrize("kind", ["origin_w", "direction_w", "minus_zero_direction_w", "generation_offset",
                                  "one_generation"])
def test_ray_sets_that_need_all_state_rows(kind):
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.config3(scenes.product_api(), 20_000)
    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    ds = DeviceScene(snap)
    good, good_counts = ds.trace(dev(rays), 10)
    assert ds.telemetry()["full_rows_fallbacks"] == 0        # RaySet defaults: compact state
    want_good, want_good_counts = orc.trace(flat, rays, 10)
    assert good_counts == want_good_counts
    helpers.assert_frames_match(good.cpu().numpy().T, want_good, what="config3 compact")
    odd = _odd_rays(kind, rays)
    want, want_counts = orc.trace(flat, odd, 10)
    for turn in range(2):                                     # the repeat, then straight with all rows
        rows, counts = ds.trace(dev(odd), 10)
        assert counts == want_counts and ds.telemetry()["full_rows_fallbacks"] == 1, (kind, turn)
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"{kind} turn {turn}")
    again, _ = ds.trace(dev(rays), 10)                        # the scene stays on all rows: same frame
    assert np.array_equal(again.cpu().numpy(), good.cpu().numpy())
    ds.close()


def test_compact_and_full_state_rows_give_the_same_frame():
    from pyrayt_amd.engine import DeviceScene

    for name in ("scene_config2.npz", "scene_config3.npz", "scene_config5.npz", "scene_mirrors_and_stops.npz"):
        fx = helpers.load(name)
        rays, limit = dev(fx["rays0"]), int(fx["generation_limit"])
        ds = device_scene(helpers.scene_of(fx))
        compact, counts = ds.trace(rays, limit)
        full, full_counts = ds.trace(rays, limit, flags=engine.TRACE_FULL_ROWS)
        assert counts == full_counts and np.array_equal(compact.cpu().numpy(), full.cpu().numpy(), equal_nan=True)
        helpers.assert_frames_match(compact.cpu().numpy().T, fx["frame"], what=name)
        ds.close()


# ---------------------------------------------------------------------------------------------
# dense-mode hints: a repeated trace launches the generations that were dense last time without the
# look-back; a hint that does not hold must cost a repeat, never a wrong frame
# ---------------------------------------------------------------------------------------------
def test_dense_mode_hints_repeat_and_miss():
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), 50_000)
    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    ds = DeviceScene(snap)
    want, want_counts = orc.trace(flat, rays, 10)
    first, counts = ds.trace(dev(rays), 10)
    assert counts == want_counts and ds.telemetry()["dense_launches"] == 0
    first = first.cpu().numpy().copy()
    again, counts = ds.trace(dev(rays), 10)               # generations 0 and 2 are dense (all carried / none carried)
    tele = ds.telemetry()
    assert tele["dense_launches"] >= 2 and tele["speculation_misses"] == 0
    assert counts == want_counts and np.array_equal(again.cpu().numpy(), first)
    helpers.assert_frames_match(first.T, want, what="config2 hinted")
    # same ray count, but a third of the rays now miss everything: generation 0 is no longer dense
    other = rays.copy()
    other[5, ::3] = 5.0
    other[4:7] /= np.linalg.norm(other[4:7], axis=0)
    want2, want2_counts = orc.trace(flat, other, 10)
    got2, counts2 = ds.trace(dev(other), 10)
    assert ds.telemetry()["speculation_misses"] == 1
    assert counts2 == want2_counts
    helpers.assert_frames_match(got2.cpu().numpy().T, want2, what="config2 after a missed hint")
    before = ds.telemetry()["dense_launches"]
    for _ in range(3):                                    # after a miss the hints rest for two traces ...
        got3, counts3 = ds.trace(dev(other), 10)
        assert counts3 == want2_counts and np.array_equal(got3.cpu().numpy(), got2.cpu().numpy())
    tele = ds.telemetry()                                 # ... then they are back, renewed from the repeats
    assert tele["speculation_misses"] == 1 and tele["dense_launches"] > before
    before = ds.telemetry()["dense_launches"]
    got4, _ = ds.trace(dev(other), 10, flags=engine.TRACE_NO_HINTS)
    assert ds.telemetry()["dense_launches"] == before and np.array_equal(got4.cpu().numpy(), got2.cpu().numpy())
    ds.close()


@pytest.mark.parametrize("n", [5_000, 32_768, 50_000, 16_385 * 4])
def test_dense_mode_miss_in_the_generation_that_ends_the_trace(n):
    """The generation that ends a hinted batch tells the host itself: its tiles check in on counters
    (one level up to 64 tiles, two above) and the last to arrive publishes -- with the verdict of every
    tile, also when the assumption failed in a single one of them."""
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), n)
    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    ds = DeviceScene(snap)
    want, want_counts = orc.trace(flat, rays, 1)
    for _ in range(3):                                    # one generation: it is first and last at once
        got, counts = ds.trace(dev(rays), 1)
        assert counts == want_counts
    helpers.assert_frames_match(got.cpu().numpy().T, want, what="one dense generation")
    assert ds.telemetry()["dense_launches"] == 2 and ds.telemetry()["speculation_misses"] == 0
    for victim in (0, n // 2, n - 1):                     # one ray of one tile misses everything
        other = rays.copy()
        other[4:7, victim] = (0.0, 1.0, 0.0)
        want2, want2_counts = orc.trace(flat, other, 1)
        assert want2_counts == [n - 1]
        ds2 = DeviceScene(snap)
        for _ in range(2):
            ds2.trace(dev(rays), 1)
        got2, counts2 = ds2.trace(dev(other), 1)
        assert ds2.telemetry()["speculation_misses"] == 1 and counts2 == want2_counts
        helpers.assert_frames_match(got2.cpu().numpy().T, want2, what=f"ray {victim} misses")
        ds2.close()
    ds.close()


def test_dense_mode_hints_across_limits_flags_and_sizes():
    """Hints survive what may change between two traces of one scene object: generation limit, the
    keep-absorbed flag, the ray count, a first batch that is shorter than the trace."""
    from pyrayt_amd.engine import DeviceScene

    fx = helpers.load("scene_config3.npz")       # seven generations, two of them lose rays
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    want = fx["frame"]
    for limit, flags in ((10, 0), (10, 0), (3, 0), (10, 0), (10, 1), (10, 1), (10, 0), (2, 2), (10, 0)):
        rows, counts = ds.trace(rays, limit, flags=flags)
        got = rows.cpu().numpy().T
        keep = want[:, 0] < limit                   # rows of the first `limit` generations
        helpers.assert_frames_match(got, want[keep], what=f"config3 limit {limit} flags {flags}")
    assert ds.telemetry()["speculation_misses"] == 0 and ds.telemetry()["dense_launches"] > 0
    # a record block that is too small is reported from the dense generations as from the others
    # (the wrapper then grows it and repeats), and an exact-fit block works
    total = want.shape[0]
    for cap in (rays.shape[1] * 2, total, total - 1):
        rows, counts = ds.trace(rays, 10, rows_cap=cap)
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"config3 rows_cap {cap}")
    block = torch.empty((15, total - 1), dtype=torch.float64, device="cuda:0")
    with pytest.raises(RuntimeError, match="rows_cap"):
        ds.trace(rays, 10, out=block)              # a caller's block is never replaced
    half = dev(np.ascontiguousarray(fx["rays0"][:, ::2]))
    rows, _ = ds.trace(half, 10)                   # another ray count: no hints, then its own
    rows2, _ = ds.trace(half, 10)
    assert np.array_equal(rows.cpu().numpy(), rows2.cpu().numpy())
    ds.close()


# ---------------------------------------------------------------------------------------------
# prt_trace_begin / prt_trace_end: traces in flight together give the frames prt_trace gives
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["config2", "config3", "config5", "mirrors_and_stops", "adv_lens"])
def test_traces_in_flight_equal_synchronous_traces(name):
    fx = helpers.load(f"scene_{name}.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    n = rays.shape[1]
    want, want_counts = ds.trace(rays, limit)
    want = want.cpu().numpy().copy()
    helpers.assert_frames_match(want.T, fx["frame"], what=name)
    blocks = [torch.full((15, n * limit), float("nan"), dtype=torch.float64, device="cuda:0") for _ in range(2)]
    steps = 7
    ds.trace_begin(0, rays, limit, blocks[0])
    for k in range(steps):                       # first trace unhinted, then hinted, one always in flight
        if k + 1 < steps:
            ds.trace_begin((k + 1) & 1, rays, limit, blocks[(k + 1) & 1])
        rows, counts = ds.trace_end(k & 1)
        assert counts == want_counts, (k, counts)
        torch.cuda.synchronize()
        assert np.array_equal(rows.cpu().numpy(), want, equal_nan=True), k
        st = ds.trace_stats()
        fused = not (engine.DEFAULT_TRACE_FLAGS & engine.TRACE_UNFUSED)
        assert st["variant"] == (1 if fused else 2) and st["rows"] == want.shape[1] and st["kernel_ms"] > 0
        blocks[k & 1].fill_(float("nan"))
    ds.close()


def test_tickets_reject_misuse_and_every_flag_goes_through_them():
    from pyrayt_amd import engine as eng

    fx = helpers.load("scene_config3.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    n = rays.shape[1]
    a, b = (torch.empty((15, n * limit), dtype=torch.float64, device="cuda:0") for _ in range(2))
    ds.trace_begin(0, rays, limit, a)
    with pytest.raises(ValueError, match="in flight"):
        ds.trace_begin(0, rays, limit, b)                      # the ticket is taken
    with pytest.raises(ValueError, match="own workspace and record block"):
        ds.trace_begin(1, rays, limit, a)                      # same record block as the trace in flight
    rows, counts = ds.trace_end(0)
    helpers.assert_frames_match(rows.cpu().numpy().T, fx["frame"], what="ticket 0")
    with pytest.raises(ValueError, match="no trace in flight"):
        eng._check(eng.library().prt_trace_end(ds.handle, 0, 0, (ctypes.c_int64 * limit)()))
    # keep-absorbed, the three-kernel path, no hints, all state rows, kernel publish, the stall fallback, sync
    for flags in (1, 2, 3, eng.TRACE_NO_HINTS, eng.TRACE_FULL_ROWS, eng.TRACE_PUBLISH_KERNEL, eng.TRACE_TEST_STALL,
                  eng.TRACE_SYNC, eng.TRACE_NO_HINTS | eng.TRACE_PUBLISH_KERNEL):
        for _ in range(2):
            ds.trace_begin(1, rays, limit, b, flags=flags)
            ds.trace_begin(0, rays, limit, a, flags=flags)
            for ticket, block in ((1, b), (0, a)):
                rows, counts = ds.trace_end(ticket)
                helpers.assert_frames_match(rows.cpu().numpy().T, fx["frame"], what=f"flags {flags} ticket {ticket}")
    # (the stall hook lives in the fused kernels: two flag sets carry it, two traces each, on two tickets)
    assert ds.telemetry()["lookback_fallbacks"] == (0 if engine.DEFAULT_TRACE_FLAGS & engine.TRACE_UNFUSED else 4)
    # a record block one column short is reported by prt_trace_end
    short = torch.empty((15, fx["frame"].shape[0] - 1), dtype=torch.float64, device="cuda:0")
    ds.trace_begin(0, rays, limit, short)
    with pytest.raises(RuntimeError, match="rows_cap"):
        ds.trace_end(0)
    ds.close()


def test_a_missed_hint_in_flight_repeats_only_that_trace():
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    n, limit = 40_000, 10
    parts, rays = scenes.config2(scenes.product_api(), n)
    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    other = rays.copy()
    other[5, ::3] = 5.0
    other[4:7] /= np.linalg.norm(other[4:7], axis=0)
    want_a, counts_a = orc.trace(flat, rays, limit)
    want_b, counts_b = orc.trace(flat, other, limit)
    ds = DeviceScene(snap)
    rays_a, rays_b = dev(rays), dev(other)
    blocks = [torch.empty((15, n * limit), dtype=torch.float64, device="cuda:0") for _ in range(2)]
    for _ in range(2):
        ds.trace(rays_a, limit)                                 # the hints now describe ray set A
    ds.trace_begin(0, rays_b, limit, blocks[0])                 # launched on A's hints: misses
    ds.trace_begin(1, rays_a, limit, blocks[1])                 # launched on A's hints: holds
    rows_b, got_b = ds.trace_end(0)
    rows_a, got_a = ds.trace_end(1)
    assert got_b == counts_b and got_a == counts_a
    helpers.assert_frames_match(rows_b.cpu().numpy().T, want_b, what="missed hint in flight")
    helpers.assert_frames_match(rows_a.cpu().numpy().T, want_a, what="the trace queued behind it")
    assert ds.telemetry()["speculation_misses"] == 1
    ds.close()


def test_a_missed_hint_with_an_exact_fit_record_block_is_repeated_not_reported():
    """A generation launched on a dense-mode hint measures the record block against the ASSUMED offsets;
    when the hint does not hold those are too large, and an exact-fit block must not turn the miss into
    a 'rows_cap too small' (ADVICE round 2): the trace is repeated without hints and fits."""
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    n, limit = 30_000, 10
    parts, rays = scenes.config2(scenes.product_api(), n)
    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    for stride, first in ((3, 0), (7, 5), (n, n - 1)):           # many misses, few, a single ray of the last tile
        other = rays.copy()
        other[4:7, first::stride] = np.array([[0.0], [1.0], [0.0]])
        want, want_counts = orc.trace(flat, other, limit)
        assert sum(want_counts) < 3 * n - 2
        ds = DeviceScene(snap)
        for _ in range(2):
            ds.trace(dev(rays), limit)                          # hints: every generation dense
        block = torch.empty((15, sum(want_counts)), dtype=torch.float64, device="cuda:0")  # fits `other` exactly
        rows, counts = ds.trace(dev(other), limit, out=block)
        assert counts == want_counts and ds.telemetry()["speculation_misses"] == 1
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"exact fit after a miss, stride {stride}")
        ds.close()


# ---------------------------------------------------------------------------------------------
# PRT_TRACE_COUNT_PATHS: the shipping library says how often its shortcuts fall through
# ---------------------------------------------------------------------------------------------
@pytest.mark.skipif(bool(engine.DEFAULT_TRACE_FLAGS & 1), reason="absorbed rays that are carried on have no direction: they count as not well formed")
def test_path_counters_tell_well_formed_rays_from_the_others():
    if engine.DEFAULT_OPTIONS:
        pytest.skip("the expected counts are those of the default scene options")
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    n, limit = 20_000, 10
    parts, rays = scenes.config2(scenes.product_api(), n)
    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    ds = DeviceScene(snap)
    want, want_counts = orc.trace(flat, rays, limit)
    rows, counts = ds.trace(dev(rays), limit, flags=engine.TRACE_COUNT_PATHS)
    helpers.assert_frames_match(rows.cpu().numpy().T, want, what="counted trace")
    tele = ds.telemetry()
    assert tele["counted_traces"] == 1 and ds.trace_stats()["variant"] == 2  # (a counted trace runs on the three-kernel path)
    assert tele["rays_not_well_formed"] == 0                                   # unit directions, w = 1 / 0
    assert tele["implied_box_nodes"] >= 2 * n                                  # both nodes of the lens, generation 0 alone
    assert tele["exact_box_tests"] < tele["implied_box_nodes"] // 10           # most chords through the lens are long
    # the same rays, every fourth one too short, every fifth with an origin w of 2: those take no shortcut
    odd = rays.copy()
    odd[4:7, ::4] *= 0.8
    odd[3, ::5] = 2.0
    want2, want2_counts = orc.trace(flat, odd, limit)
    rows2, counts2 = ds.trace(dev(odd), limit, flags=engine.TRACE_COUNT_PATHS)
    assert counts2 == want2_counts
    helpers.assert_frames_match(rows2.cpu().numpy().T, want2, what="counted trace, odd rays")
    tele2 = ds.telemetry()
    gen0_bad = len(set(range(0, n, 4)) | set(range(0, n, 5)))
    assert tele2["counted_traces"] == 2 and tele2["rays_not_well_formed"] >= gen0_bad
    # (the short ones cross the lens like their unit twins and test the first node's box exactly;
    # what a ray with origin w = 2 meets is another matter)
    assert tele2["exact_box_tests"] - tele["exact_box_tests"] >= n // 4
    # an uncounted trace leaves the counters alone
    ds.trace(dev(odd), limit)
    assert ds.telemetry()["rays_not_well_formed"] == tele2["rays_not_well_formed"]
    ds.close()


def test_traces_in_flight_on_two_streams_equal_synchronous_traces():
    """The two tickets on two HIP streams: their kernels overlap on the device, the frames do not change."""
    fx = helpers.load("scene_config3.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    n = rays.shape[1]
    want, want_counts = ds.trace(rays, limit)
    want = want.cpu().numpy().copy()
    blocks = [torch.full((15, n * limit), float("nan"), dtype=torch.float64, device="cuda:0") for _ in range(2)]
    streams = [torch.cuda.Stream("cuda:0") for _ in range(2)]
    torch.cuda.synchronize()

    def begin(k):
        with torch.cuda.stream(streams[k & 1]):
            ds.trace_begin(k & 1, rays, limit, blocks[k & 1])

    steps = 9
    begin(0)
    for k in range(steps):
        if k + 1 < steps:
            begin(k + 1)
        rows, counts = ds.trace_end(k & 1)
        assert counts == want_counts, (k, counts)
        streams[k & 1].synchronize()
        assert np.array_equal(rows.cpu().numpy(), want, equal_nan=True), k
        with torch.cuda.stream(streams[k & 1]):
            blocks[k & 1].fill_(float("nan"))
    ds.close()


def test_trace_many_overlaps_traces_and_keeps_their_frames():
    """DeviceScene.trace_many: a sequence of ray sets, two or three traces in flight on as many streams."""
    fx = helpers.load("scene_config2.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx))
    base = fx["rays0"]
    sets, want = [], []
    for k in range(7):                          # ray sets that differ (and differ in how their rays die)
        rays = base.copy()
        rays[5, k::5] += 0.02 * k
        rays[4:7] /= np.linalg.norm(rays[4:7], axis=0)
        sets.append(dev(rays))
        rows, counts = ds.trace(sets[-1], limit)
        want.append((rows.cpu().numpy().copy(), counts))
    for depth in (1, 2, 3):
        got = []
        for rows, counts in ds.trace_many(iter(sets), limit, depth=depth):
            got.append((rows.cpu().numpy().copy(), counts))  # (copied before the block is reused)
        assert len(got) == len(want)
        for k, ((rows, counts), (ref_rows, ref_counts)) in enumerate(zip(got, want)):
            assert counts == ref_counts, (depth, k)
            assert np.array_equal(rows, ref_rows, equal_nan=True), (depth, k)
        # the lifetime the docstring states: a result may be HELD while the next one is taken (a design loop
        # comparing frame k with frame k + 1) -- the view handed out is not the block recorded into next
        held = None
        for k, (rows, counts) in enumerate(ds.trace_many(iter(sets), limit, depth=depth)):
            if held is not None:
                assert np.array_equal(held.cpu().numpy(), want[k - 1][0], equal_nan=True), (depth, k, "held frame changed")
            held = rows
    ds.close()


def test_ticket_edges_update_in_flight_bad_ticket_varying_sizes():
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    api = scenes.product_api()
    parts, rays = scenes.config2(api, 30_000)
    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    ds = DeviceScene(snap)
    limit = 10
    block = torch.empty((15, 30_000 * limit), dtype=torch.float64, device="cuda:0")
    with pytest.raises(ValueError, match="ticket out of range"):
        ds.trace_begin(engine.TRACE_TICKETS, dev(rays), limit, block)
    ds.trace_begin(2, dev(rays), limit, block)
    parts[1].move_x(0.01)                                   # a part moves while a trace is in flight:
    with pytest.raises(ValueError, match="in flight"):     # the tables that trace reads are not overwritten
        ds.update(SceneSnapshot(parts))
    rows, counts = ds.trace_end(2)
    want, want_counts = orc.trace(flat, rays, limit)
    assert counts == want_counts
    helpers.assert_frames_match(rows.cpu().numpy().T, want, what="ticket 2")
    moved = SceneSnapshot(parts)
    assert ds.update(moved)                                 # ... and afterwards they are
    # ray sets of different sizes through all four tickets
    sets = [np.ascontiguousarray(rays[:, : 30_000 - 3_001 * k]) for k in range(6)]
    flat2 = helpers.flat_scene(moved)
    got = [(r.cpu().numpy().copy(), c) for r, c in ds.trace_many((dev(s) for s in sets), limit, depth=4)]
    for k, (rows_k, counts_k) in enumerate(got):
        want_k, want_counts_k = orc.trace(flat2, sets[k], limit)
        assert counts_k == want_counts_k, k
        helpers.assert_frames_match(rows_k.T, want_k, what=f"trace_many set {k}")
    ds.close()


def test_four_traces_in_flight_get_their_queues_whichever_import_came_first():
    """This module imported torch before pyrayt_amd: the package set GPU_MAX_HW_QUEUES late, which the runtime honours
    as long as no HIP call preceded it (profiles/r5/queue_probe.txt).  Four ticket streams must come without the
    serialisation warning, and must really overlap; in a process where the setting came too late they warn."""
    import os
    import subprocess
    import sys
    import warnings as _warnings

    from pyrayt_amd import _runtime
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), 4096)
    ds = engine.DeviceScene(SceneSnapshot(parts))
    device = torch.device("cuda", 0)
    if _runtime.HW_QUEUES != "user":
        with _warnings.catch_warnings():
            _warnings.simplefilter("error")
            streams = ds.ticket_streams(device, 4)
        assert len(streams) == 4 and _runtime.queues_overlap(torch, streams, device)
    ds.close()
    # a process whose user asked for two queues is taken at its word: four traces in flight warn
    code = ("import os, sys, warnings; import torch;"
            "sys.path.insert(0, 'tests'); import scenes; from pyrayt_amd import engine; from pyrayt_amd.scene import SceneSnapshot;"
            "parts, _ = scenes.config2(scenes.product_api(), 64); ds = engine.DeviceScene(SceneSnapshot(parts));"
            "warnings.simplefilter('error');\n"
            "try:\n    ds.ticket_streams(torch.device('cuda', 0), 4); print(engine.HW_QUEUES, 'quiet')\n"
            "except RuntimeWarning: print(engine.HW_QUEUES, 'warned')")
    env = dict(os.environ, GPU_MAX_HW_QUEUES="2")
    done = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert done.returncode == 0, done.stderr[-1500:]
    assert done.stdout.split()[-2:] == ["user", "warned"], done.stdout
    # ... and one that made a HIP call before the package could ask is found out by running something: the runtime
    # stays at its default of four queues, the four ticket streams plus the caller's own make five, two of them share
    # a queue and the probe measures twice the single-stream time: the warning fires
    code = code.replace("import torch;", "os.environ.pop('GPU_MAX_HW_QUEUES', None); import torch; torch.cuda.is_available();")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    done = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert done.returncode == 0, done.stderr[-1500:]
    assert done.stdout.split()[-2:] == ["set-late", "warned"], done.stdout


@pytest.mark.parametrize("lens", ["biconvex", "thick_moved", "thick_tilted_scaled"])
def test_rays_around_the_rim_of_a_lens_with_and_without_the_clearance_test(lens):
    """chain_candidate skips the cylinder that cuts a lens to its aperture for a wave whose chords all run inside it by
    a margin (chord_inside_cylinder: a convexity argument).  The rays that could tell a wrong skip apart are the ones
    near the rim and the ones on upstream's degenerate branches: beams parallel to the axis (a = dx^2 + dy^2 <= 1e-8:
    SURVEY Q5, such rays MISS the cylinder they run through), beams a hair off it, rays whose hit points sit within
    1e-12 ... 1e-3 of the aperture radius on either side, rays perpendicular to the axis, zero directions -- mixed
    into every wave with rays from the middle of the beam.  Rows bit for bit against the C oracle, with the test and
    without (scene option no_clearance); every wave of the first holds a ray that forces the cut or none does."""
    from oracle import c_oracle
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    api = scenes.product_api()
    CountedObject.reset_ids()
    if lens == "biconvex":
        part, radius, axis_x = api.components.biconvex_lens(2, 2, 0.25, aperture=1), 0.5, 0.0
    elif lens == "thick_moved":
        part, radius, axis_x = api.components.thick_lens(40, -200, 5, aperture=25.4, material=api.materials.glass["BK7"]).move(3.0, 0.5, -0.25), 12.7, 3.0
    else:
        part = api.components.thick_lens(30, -45, 4, aperture=10.0, material=api.materials.glass["SF2"])
        part.rotate_y(7).rotate_z(-11).scale(1.5, 1.3, 0.8).move(1.0, -2.0, 0.5)
        radius, axis_x = 7.5, 1.0
    detector = api.components.baffle((200, 200)).move_x(axis_x + 60)
    rng = np.random.default_rng(77)
    n = 64 * 600
    rays = scenes.random_rays(n, seed=5, box=1.0)
    centre = np.array([axis_x, 0.5 if lens == "thick_moved" else (-2.0 if lens == "thick_tilted_scaled" else 0.0),
                       -0.25 if lens == "thick_moved" else (0.5 if lens == "thick_tilted_scaled" else 0.0)])
    # every ray starts in front of the lens and aims at a point of the aperture plane at radius rho
    offsets = np.array([0.0, 1e-12, -1e-12, 1e-9, -1e-9, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-3, -1e-3, 1e-2, -1e-2])
    rho = np.where(rng.random(n) < 0.5, radius * rng.random(n) * 0.9, radius * (1 + rng.choice(offsets, n)))
    phi = rng.uniform(0, 2 * np.pi, n)
    target = centre[:, None] + np.stack([np.zeros(n), rho * np.cos(phi), rho * np.sin(phi)])
    kind = rng.integers(0, 6, n)
    start = target.copy()
    start[0] -= rng.uniform(2.0, 50.0, n)                                  # kind 0, 1: parallel to the axis (a == 0: Q5)
    tilt = np.where(kind >= 2, 10.0 ** rng.uniform(-9, -1, n), 0.0)          # kind >= 2: off the axis by 1e-9 ... 1e-1
    start[1] += tilt * (target[0] - start[0]) * np.cos(phi * 3)
    start[2] += tilt * (target[0] - start[0]) * np.sin(phi * 3)
    d = target - start
    d /= np.linalg.norm(d, axis=0)
    rays[0:3], rays[3] = start, 1.0
    rays[4:7], rays[7] = d, 0.0
    side = kind == 5                                                          # perpendicular to the axis, through the rim zone
    rays[4:7, side] = np.stack([np.zeros(side.sum()), np.cos(phi[side]), np.sin(phi[side])])
    rays[0:3, side] = target[:, side] - 3 * radius * rays[4:7, side]
    rays[4:7, ::997] = 0.0                                                    # a few zero directions
    rays[8], rays[9], rays[10], rays[11] = 0.0, 100.0, 0.55, 1.0
    rays[12] = np.arange(n)
    snap = SceneSnapshot([part, detector])
    want, want_counts = c_oracle.trace(helpers.flat_scene(snap), rays, 8)
    assert len(want_counts) >= 3 and want.shape[0] > n
    for options in ({}, {"no_clearance": 1}):
        ds = engine.DeviceScene(snap, options=options)
        rows, counts = ds.trace(dev(rays), 8)
        assert counts == want_counts, options
        assert np.array_equal(rows.cpu().numpy().T, want, equal_nan=True), options
        # the middle of the beam alone: every wave clears its cylinder -- and the rows are those rows
        calm = np.ascontiguousarray(rays[:, (rho < 0.8 * radius) & (kind >= 2) & (kind < 5) & (np.arange(n) % 997 != 0)])
        calm_want, calm_counts = c_oracle.trace(helpers.flat_scene(snap), calm, 8)
        rows, counts = ds.trace(dev(calm), 8)
        assert counts == calm_counts and np.array_equal(rows.cpu().numpy().T, calm_want, equal_nan=True), options
        ds.close()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 255, 256, 257, 5000, 70_001])
def test_lean_state_segments_and_the_ray_sets_that_do_not_qualify(n):
    """Between generations a wave whose rays stay in place hands on intensity, wavelength and id as three numbers (one
    intensity, one wavelength, ids counting up from an integer: interact_store_rows) instead of three rows.  Ray sets
    that qualify, that qualify in some waves only, and that never do -- ids in any order, fractional, negative, beyond
    2^48, wavelengths or intensities that change inside a wave -- must all give the oracle's rows bit for bit; an id that
    carries the tag of a lean segment sends the scene to its 13-row kernels (telemetry: full_rows_fallbacks)."""
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), n, seed=300 + n)
    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    rng = np.random.default_rng(n)
    variants = {"as emitted": rays.copy()}
    v = rays.copy(); v[12] = rng.permutation(n); variants["ids shuffled"] = v
    v = rays.copy(); v[12] += 0.5; variants["fractional ids"] = v
    v = rays.copy(); v[12] -= 7.0; variants["negative ids"] = v
    v = rays.copy(); v[12] += 2.0 ** 48; variants["ids beyond 2^48"] = v
    v = rays.copy(); v[12] += 2.0 ** 48 - 100; variants["ids across 2^48"] = v
    v = rays.copy(); v[12, n // 2] += 1.0; variants["one id out of step"] = v
    v = rays.copy(); v[10] = np.where(np.arange(n) % 64 == 63, 0.5, 0.633); variants["a wavelength per wave edge"] = v
    v = rays.copy(); v[10] = 0.45 + 0.01 * (np.arange(n) // 64 % 7); variants["one wavelength per wave"] = v
    v = rays.copy(); v[9, -1] = 7.0; variants["last intensity differs"] = v
    v = rays.copy(); v[9] = -0.0; v[9, ::3] = 0.0; variants["signed zero intensities"] = v
    v = rays.copy(); v[10, 0] = np.nan; variants["a NaN wavelength"] = v
    ds = engine.DeviceScene(snap)
    for name, block in variants.items():
        want, want_counts = orc.trace(flat, block, 10)
        for _ in range(2):  # (a first trace by look-back, a repeat on the hints: both write lean segments where they may)
            rows, counts = ds.trace(dev(block), 10)
            assert counts == want_counts, (name, n)
            assert np.array_equal(rows.cpu().numpy().T, want, equal_nan=True), (name, n)
    assert ds.telemetry()["full_rows_fallbacks"] == 0
    ds.close()
    # an id that looks like the head of a lean segment: the compact kernels must not hand it on
    tagged = rays.copy()
    tagged[12, n // 3] = np.frombuffer(np.array([(0x7ffb << 48) | 12345], dtype=np.uint64).tobytes(), dtype=np.float64)[0]
    want, want_counts = orc.trace(flat, tagged, 10)
    ds = engine.DeviceScene(snap)
    rows, counts = ds.trace(dev(tagged), 10)
    assert counts == want_counts and np.array_equal(rows.cpu().numpy().T.view(np.uint64), want.view(np.uint64))
    # (tools/run_matrix.sh: a scene forced onto the three-kernel path or onto all 13 rows has no compact state to leave)
    forced = ds.trace_flags & (engine.TRACE_UNFUSED | engine.TRACE_FULL_ROWS | engine.TRACE_COUNT_PATHS)
    assert ds.telemetry()["full_rows_fallbacks"] == (0 if forced else 1)
    ds.close()


def test_batch_reports_how_long_the_device_was_busy_with_it():
    """PRT_TRACE_BUSY / prt_trace_batch_busy (bench.py's roofline): every job of a batch bracketed by its own pair of HIP
    events on its own stream, the intervals merged by the library.  The union can never exceed first-start-to-last-end
    nor the sum of the intervals; with two traces in flight the sum exceeds the union (they overlap); a job without
    rays and the unfused path contribute nothing; the rows are the rows of a batch without the flag."""
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), 200_000)
    ds = engine.DeviceScene(SceneSnapshot(parts))
    full = dev(rays)
    sets = [full, full[:, :0], full[:, :150_000].contiguous(), full, full[:, :70_000].contiguous(), full]
    plain = engine.TraceBatch(ds, sets, 10, depth=2)
    plain.run()
    want = [(r.cpu().numpy().copy(), c) for r, c in plain.results()]
    timed = engine.TraceBatch(ds, sets, 10, depth=2, flags=engine.TRACE_BUSY | engine.TRACE_NO_TIMING)
    if ds.trace_flags & (engine.TRACE_UNFUSED | engine.TRACE_COUNT_PATHS):  # (tools/run_matrix.sh: the three-kernel path brackets nothing)
        timed.run()
        assert timed.busy()["traces"] == 0
        ds.close()
        return
    for _ in range(3):
        timed.run()
        busy = timed.busy()
        assert busy["traces"] == 5  # (the empty job launched nothing)
        assert 0 < busy["union_ms"] <= busy["span_ms"] * (1 + 1e-6) and busy["union_ms"] <= busy["sum_ms"] * (1 + 1e-6)
        assert busy["sum_ms"] > 1.05 * busy["union_ms"], busy  # two traces in flight do overlap
        for (rows, counts), (want_rows, want_counts) in zip(timed.results(), want):
            assert counts == want_counts and np.array_equal(rows.cpu().numpy(), want_rows)
    one = engine.TraceBatch(ds, [full, full, full], 10, depth=1, flags=engine.TRACE_BUSY)
    one.run()
    alone = one.busy()
    assert alone["traces"] == 3 and abs(alone["sum_ms"] - alone["union_ms"]) <= 1e-3 * alone["sum_ms"] + 1e-4  # one at a time
    slow = engine.TraceBatch(ds, [full], 10, depth=1, flags=engine.TRACE_BUSY | engine.TRACE_UNFUSED)
    slow.run()
    assert slow.busy()["traces"] == 0
    ds.close()


def test_trace_many_abandoned_midway_frees_its_tickets():
    fx = helpers.load("scene_config2.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    stream = ds.trace_many((rays for _ in range(6)), limit, depth=3)
    first_rows, first_counts = next(stream)
    want = first_rows.cpu().numpy().copy()
    stream.close()                                    # two more traces were in flight: they are collected
    rows, counts = ds.trace(rays, limit)              # ... and ticket 0 is free for an ordinary trace
    assert counts == first_counts and np.array_equal(rows.cpu().numpy(), want, equal_nan=True)
    ds.close()


def test_one_million_rays_overlapped_equal_the_synchronous_trace():
    """BASELINE config 2 at its full size through trace_many (two and three traces in flight on as many
    streams): every frame is the synchronous trace's, bit for bit (compared on the device)."""
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), 1_000_000, seed=1234)
    ds = DeviceScene(SceneSnapshot(parts))
    device_rays = dev(rays)
    want, want_counts = ds.trace(device_rays, 10)
    want = want.clone()
    assert want_counts == [1_000_000, 1_000_000, 999_991]
    for depth in (2, 3):
        seen = 0
        for rows, counts in ds.trace_many((device_rays for _ in range(2 * depth + 1)), 10, depth=depth):
            assert counts == want_counts
            assert torch.equal(rows, want)
            seen += 1
        assert seen == 2 * depth + 1
    ds.close()


def test_trace_batch_is_trace_many_in_one_library_call():
    """prt_trace_batch / DeviceScene.trace_batch: ray sets of different sizes (an empty one among them), one to
    four traces in flight; every frame is the synchronous trace's, bit for bit."""
    fx = helpers.load("scene_config2.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx))
    base = fx["rays0"]
    sets, want = [], []
    for k in range(9):
        rays = base[:, : base.shape[1] - 37 * k].copy() if k != 4 else base[:, :0].copy()
        rays[5, k::5] += 0.02 * k
        if rays.shape[1]:
            rays[4:7] /= np.linalg.norm(rays[4:7], axis=0)
        sets.append(dev(rays))
        rows, counts = ds.trace(sets[-1], limit)
        want.append((rows.cpu().numpy().copy(), counts))
    for depth in (1, 2, 3, 4):
        got = ds.trace_batch(sets, limit, depth=depth)
        assert len(got) == len(want)
        for k, ((rows, counts), (ref_rows, ref_counts)) in enumerate(zip(got, want)):
            assert counts == ref_counts, (depth, k)
            assert np.array_equal(rows.cpu().numpy(), ref_rows, equal_nan=True), (depth, k)
    assert ds.trace_batch([], limit) == []
    ds.close()


def test_a_prepared_batch_runs_again_and_may_reuse_its_record_blocks():
    """TraceBatch: the same table traced twice; with as many record blocks as traces in flight only the
    last `depth` frames remain, and those are right."""
    fx = helpers.load("scene_config2.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    n = rays.shape[1]
    want, want_counts = ds.trace(rays, limit)
    want = want.clone()
    depth = 3
    outs = [torch.empty((15, n * limit), dtype=torch.float64, device="cuda:0") for _ in range(depth)]
    batch = engine.TraceBatch(ds, [rays] * 8, limit, depth=depth, outs=outs)
    for _ in range(2):
        for out in outs:
            out.fill_(-1.0)
        assert batch.run() == 8 * want.shape[1]
        found = batch.results()
        for rows, counts in found[-depth:]:
            assert counts == want_counts and torch.equal(rows, want)
    with pytest.raises(ValueError, match="record block per trace in flight"):
        engine.TraceBatch(ds, [rays] * 8, limit, depth=3, outs=outs[:2])
    ds.close()


def test_a_batch_reports_the_first_error_and_leaves_the_tickets_free():
    fx = helpers.load("scene_config2.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    n = rays.shape[1]
    want, want_counts = ds.trace(rays, limit)
    want = want.clone()
    outs = [torch.empty((15, n * limit), dtype=torch.float64, device="cuda:0") for _ in range(6)]
    outs[2] = torch.empty((15, n), dtype=torch.float64, device="cuda:0")  # one generation's worth: too small
    batch = engine.TraceBatch(ds, [rays] * 6, limit, depth=2, outs=outs)
    with pytest.raises(RuntimeError, match="rows_cap too small"):
        batch.run()
    totals = batch.jobs["total"]
    assert totals[0] == totals[1] == want.shape[1] and totals[2] == engine.ERR_ROWS_CAP
    assert totals[4] == totals[5] == 0                      # never started
    rows, counts = ds.trace(rays, limit)                    # every ticket was collected
    assert counts == want_counts and torch.equal(rows, want)
    got = ds.trace_batch([rays] * 3, limit, depth=3)
    assert all(c == want_counts and torch.equal(r, want) for r, c in got)
    ds.close()


def test_a_ticket_keeps_no_stale_workspace_after_a_larger_trace_grew_it():
    """trace_begin caches its arguments per ticket; the ticket's workspace is replaced when a larger ray set
    comes (here through a batch): the cached call must not keep the freed block's address."""
    fx = helpers.load("scene_config2.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx))
    base = fx["rays0"]
    small, large = dev(base[:, :2000].copy()), dev(np.tile(base, (1, 3)))
    want, want_counts = ds.trace(small, limit)
    want = want.clone()
    block = torch.empty((15, 2000 * limit), dtype=torch.float64, device="cuda:0")
    for _ in range(2):                                   # the second call comes from the cache
        ds.trace_begin(0, small, limit, block)
        rows, counts = ds.trace_end(0)
        assert counts == want_counts and torch.equal(rows, want)
    before = ds._ticket_work[0]
    ds.trace_batch([large, large], limit, depth=2)
    assert ds._ticket_work[0] is not before              # grown: the old block went back to the allocator
    del before
    filler = [torch.full((1 << 20,), 7, dtype=torch.uint8, device="cuda:0") for _ in range(64)]  # ... and is reused
    ds.trace_begin(0, small, limit, block)
    rows, counts = ds.trace_end(0)
    assert counts == want_counts and torch.equal(rows, want)
    assert all(int(f.min()) == 7 for f in filler)         # nobody wrote through a stale pointer
    ds.close()


@pytest.mark.parametrize("flags", [engine.TRACE_UNFUSED, engine.TRACE_NO_HINTS, engine.TRACE_FULL_ROWS | engine.TRACE_SYNC,
                                   engine.TRACE_KEEP_ABSORBED, engine.TRACE_COUNT_PATHS])
def test_trace_batch_under_the_trace_flags(flags):
    """The batch entry hands its flags to every trace: the three-kernel path (collected synchronously in
    prt_trace_end), no hints, all 13 state rows, upstream's bookkeeping of absorbed rays, the counting build."""
    fx = helpers.load("scene_mirrors_and_stops.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    want, want_counts = ds.trace(rays, limit)
    want = want.clone()
    got = ds.trace_batch([rays] * 5, limit, depth=2, flags=flags)
    torch.cuda.synchronize()
    for rows, counts in got:
        assert counts == want_counts
        assert torch.equal(rows, want)
    helpers.assert_frames_match(want.cpu().numpy().T, fx["frame"], what="mirrors_and_stops")
    ds.close()


# ---------------------------------------------------------------------------------------------
# per-tile records: a generation that loses rays runs, in the next trace of the same ticket / workspace / ray
# count, on the record of where every tile's rows went last time; every tile checks its counts against it
# ---------------------------------------------------------------------------------------------
def lossy_scene_and_rays(n=40_000, seed=5):
    """Config 2 with a cone wide enough that rays miss the lens: every generation loses rays."""
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), n, seed=seed)
    rng = np.random.default_rng(seed)
    wide = rng.choice(n, n // 50, replace=False)
    rays[5, wide] += rng.uniform(0.3, 0.6, len(wide))            # these miss the lens (and the detector)
    rays[4:7] /= np.linalg.norm(rays[4:7], axis=0)
    return SceneSnapshot(parts), rays


def test_a_ray_buffer_that_loses_rays_in_numbers_is_replayed_and_refilled():
    """Generations that lose rays in numbers compact by look-back, trace after trace (the per-tile records that once
    served the replay of such a buffer were retired in round 6: worth under 2 %, profiles/r6/ab_round6.txt): the
    same buffer traced again and again, then refilled with rays whose losses sit in other tiles -- equal totals or
    not --, always gives the oracle's frame, and no trace is repeated for it."""
    from pyrayt_amd.engine import DeviceScene

    snap, rays = lossy_scene_and_rays(seed=5)
    _, other = lossy_scene_and_rays(seed=6)
    flat = helpers.flat_scene(snap)
    ds = DeviceScene(snap)
    buf = dev(rays)
    block = torch.empty((15, rays.shape[1] * 10), dtype=torch.float64, device="cuda:0")
    want, want_counts = c_oracle.trace(flat, rays, 10)
    assert want_counts[0] < rays.shape[1]                          # (generation 0 already loses rays)
    for k in range(4):
        rows, counts = ds.trace(buf, 10, out=block)
        assert counts == want_counts
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"replayed, trace {k}")
    # same lossy rays, one of them swapped with a surviving one: equal totals, two tiles with other counts
    lost = int(np.nonzero(np.abs(rays[5]) > 0.25)[0][0])
    kept = int(np.nonzero(np.abs(rays[5]) < 0.05)[0][-1])
    swapped = rays.copy()
    swapped[:, [lost, kept]] = swapped[:, [kept, lost]]
    swapped[12] = rays[12]                                       # (ids stay in order)
    for changed in (swapped, other):
        want, want_counts = c_oracle.trace(flat, changed, 10)
        buf.copy_(torch.from_numpy(changed))
        for k in range(3):
            rows, counts = ds.trace(buf, 10, out=block)
            assert counts == want_counts
            helpers.assert_frames_match(rows.cpu().numpy().T, want, what="refilled buffer")
    assert ds.telemetry()["speculation_misses"] == 0
    ds.close()


# ---------------------------------------------------------------------------------------------
# sparse loss (hint mode 4): a generation that records every ray and absorbs a few of them runs dense with those rays
# kept the way upstream carries them (direction zeroed, dropped a generation later) instead of compacting them away
# ---------------------------------------------------------------------------------------------
def stop_before_lens(n=30_000, seed=3, ring=200, astray=0, with_hole=False):
    """A small absorbing plate in front of a lens, a detector behind it: collimated rays, nearly all past the plate,
    `ring` of them onto it (absorbed in generation 0: a sparse loss), `astray` past the plate and past everything
    else (not recorded at all).  with_hole: the plate is an aperture() around the beam -- its hole stock has no
    material, so the trace publishes its counts behind each batch and keeps no per-tile records."""
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    api = scenes.product_api()
    c = api.components
    stop = c.aperture((10, 10), 6).move_x(-5) if with_hole else c.baffle((1, 1)).move(-5, 3.5, 0)
    lens = c.thick_lens(60, -60, 3, aperture=8, material=api.materials.glass["BK7"])
    det = c.baffle((6, 6)).move_x(40)
    rng = np.random.default_rng(seed)
    radius = 2.9 * np.sqrt(rng.random(n))
    phi = 2 * np.pi * rng.random(n)
    rays = scenes.blank_rays(n, 0.55)
    rays[0] = -20.0
    rays[1], rays[2] = radius * np.cos(phi), radius * np.sin(phi)
    where = rng.choice(n, ring + astray, replace=False)
    rays[1, where[:ring]] = rng.uniform(3.1, 3.9, ring)         # onto the plate
    rays[2, where[:ring]] = rng.uniform(-0.4, 0.4, ring)
    rays[1, where[ring:]] = rng.uniform(6.5, 9.0, astray)       # beside the plate, the lens and the detector
    rays[4] = 1.0
    return SceneSnapshot([stop, lens, det]), rays


@pytest.mark.skipif(bool(engine.DEFAULT_OPTIONS) or bool(engine.DEFAULT_TRACE_FLAGS), reason="counts launches of the default path")
@pytest.mark.parametrize("with_hole", [False, True])
def test_sparse_loss_generations_run_dense_with_their_absorbed_rays_kept(with_hole):
    from pyrayt_amd.engine import DeviceScene

    snap, rays = stop_before_lens(with_hole=with_hole)
    flat = helpers.flat_scene(snap)
    want, want_counts = c_oracle.trace(flat, rays, 10)
    n = rays.shape[1]
    assert want_counts[0] == n and want_counts[1] == n - 200     # every ray recorded, the ring absorbed by the plate
    ds = DeviceScene(snap)
    block = torch.empty((15, n * 10), dtype=torch.float64, device="cuda:0")
    buffers = [dev(rays) for _ in range(10)]                     # (all alive: ten addresses)
    rows, counts = ds.trace(buffers.pop(), 10, out=block)       # first trace: no hints
    assert counts == want_counts and ds.telemetry()["sparse_keep_launches"] == 0
    for k in range(3):                                           # another buffer every time: no per-tile records
        rows, counts = ds.trace(buffers[k], 10, out=block)
        assert counts == want_counts
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"absorbed rays kept, trace {k}")
    told = ds.telemetry()
    assert told["sparse_keep_launches"] == 3 and told["speculation_misses"] == 0
    # other rays, lost elsewhere and in other numbers: the hint holds for them as well
    _, other = stop_before_lens(seed=4, ring=350, with_hole=with_hole)
    want_other, counts_other = c_oracle.trace(flat, other, 10)
    other_buffer = dev(other)
    rows, counts = ds.trace(other_buffer, 10, out=block)
    assert counts == counts_other
    helpers.assert_frames_match(rows.cpu().numpy().T, want_other, what="other rays on the same hint")
    assert ds.telemetry()["sparse_keep_launches"] == 4 and ds.telemetry()["speculation_misses"] == 0
    # the same buffer again and again: nothing but the hints is kept between traces (the per-tile records of earlier
    # rounds are retired), so a replayed buffer is served like any other
    buf = dev(rays)
    for k in range(4):
        rows, counts = ds.trace(buf, 10, out=block)
        assert counts == want_counts
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"same buffer, trace {k}")
    assert ds.telemetry()["sparse_keep_launches"] == 8 and ds.telemetry()["speculation_misses"] == 0
    # not with the flag; upstream's bookkeeping for the whole trace is the other flag and has its own hints
    before = ds.telemetry()["sparse_keep_launches"]
    for flags in (engine.TRACE_NO_SPARSE_KEEP, engine.TRACE_KEEP_ABSORBED, engine.TRACE_NO_HINTS, engine.TRACE_UNFUSED):
        for k in range(2):
            rows, counts = ds.trace(buffers[3 + k], 10, out=block, flags=flags)
            assert counts == want_counts
            helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"flags {flags}")
    assert ds.telemetry()["sparse_keep_launches"] == before
    ds.close()


@pytest.mark.skipif(bool(engine.DEFAULT_OPTIONS) or bool(engine.DEFAULT_TRACE_FLAGS), reason="counts misses of the default path")
def test_a_sparse_loss_hint_that_does_not_hold_repeats_the_trace():
    """Rays that are not recorded at all in a generation launched dense with its absorbed rays kept: the tiles
    that hold them refute the hint, the trace is repeated without hints, the frame is the oracle's."""
    from pyrayt_amd.engine import DeviceScene

    snap, rays = stop_before_lens()
    flat = helpers.flat_scene(snap)
    ds = DeviceScene(snap)
    n = rays.shape[1]
    block = torch.empty((15, n * 10), dtype=torch.float64, device="cuda:0")
    buffers = [dev(rays) for _ in range(2)]
    for buffer in buffers:
        ds.trace(buffer, 10, out=block)
    assert ds.telemetry()["sparse_keep_launches"] == 1
    _, astray = stop_before_lens(seed=8, astray=7)
    want, want_counts = c_oracle.trace(flat, astray, 10)
    assert want_counts[0] == n - 7
    astray_buffer = dev(astray)
    rows, counts = ds.trace(astray_buffer, 10, out=block)
    assert counts == want_counts
    helpers.assert_frames_match(rows.cpu().numpy().T, want, what="after a sparse-loss hint that did not hold")
    assert ds.telemetry()["speculation_misses"] == 1
    # the first ray set again, and in flight on two tickets
    want, want_counts = c_oracle.trace(flat, rays, 10)
    got = ds.trace_batch([dev(rays), dev(astray), dev(rays)], 10, depth=2)
    torch.cuda.synchronize()
    assert got[0][1] == want_counts and got[2][1] == want_counts
    helpers.assert_frames_match(got[0][0].cpu().numpy().T, want, what="in flight, ticket 0")
    helpers.assert_frames_match(got[2][0].cpu().numpy().T, want, what="in flight, ticket 0 again")
    ds.close()


def plate_between_two_lenses(n=30_000, seed=3, ring=200, astray=0):
    """stop_before_lens with a weak lens in front: the plate absorbs its rays in generation 2, so that the sparse loss
    has a generation in front of it (which empties the dead list) and one behind it that loses nothing."""
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    api = scenes.product_api()
    c = api.components
    first = c.thick_lens(200, -200, 2, aperture=8.4, material=api.materials.glass["BK7"]).move_x(-12)
    stop = c.baffle((1, 1)).move(-5, 3.5, 0)
    lens = c.thick_lens(60, -60, 3, aperture=8, material=api.materials.glass["BK7"])
    det = c.baffle((6, 6)).move_x(40)
    trap = c.baffle((0.6, 0.6)).move(0, -3.1, 0)                # inside the second lens, off the beam: for `astray`
    rng = np.random.default_rng(seed)
    radius = 0.3 + 2.6 * np.sqrt(rng.random(n))                 # (no near-axial rays: those have quirks of their own)
    phi = 2 * np.pi * rng.random(n)
    rays = scenes.blank_rays(n, 0.55)
    rays[0] = -20.0
    rays[1], rays[2] = radius * np.cos(phi), radius * np.sin(phi)
    where = rng.choice(n, ring + astray, replace=False)
    rays[1, where[:ring]] = rng.uniform(3.3, 3.9, ring)         # onto the plate
    rays[2, where[:ring]] = rng.uniform(-0.4, 0.4, ring)
    rays[1, where[ring:]] = rng.uniform(-3.35, -3.25, astray)   # into the second lens and onto the trap inside it:
    rays[2, where[ring:]] = rng.uniform(-0.1, 0.1, astray)      # absorbed a generation after the plate's rays
    rays[4] = 1.0
    return SceneSnapshot([first, stop, lens, det, trap]), rays


def near_axial_config2(n=20_000, seed=5, odd=60):
    """Config 2 with a handful of near-axial rays: they miss the lens's back surface (an upstream quirk) and end on the
    detector a generation early -- generation 1 records every ray and absorbs those few, generation 2 absorbs the rest."""
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), n, seed=seed)
    rng = np.random.default_rng(seed + 100)
    idx = rng.choice(n, odd, replace=False)
    angle, phi = rng.uniform(0, 3e-4, odd), rng.uniform(0, 2 * np.pi, odd)
    rays[4, idx], rays[5, idx], rays[6, idx] = np.cos(angle), np.sin(angle) * np.cos(phi), np.sin(angle) * np.sin(phi)
    return SceneSnapshot(parts), rays


@pytest.mark.skipif(bool(engine.DEFAULT_OPTIONS) or bool(engine.DEFAULT_TRACE_FLAGS), reason="counts launches of the default path")
@pytest.mark.parametrize("which", ["carried", "ends"])
def test_the_generation_behind_a_sparse_loss_runs_on_its_dead_list(which):
    """Hint modes 5 / 6: behind a generation that kept its absorbed rays, a generation that loses none of its own takes
    its tiles' offsets from the list of kept rays -- no generation of such a trace compacts by look-back."""
    from pyrayt_amd.engine import DeviceScene

    if which == "carried":
        snap, rays = plate_between_two_lenses()
        _, other = plate_between_two_lenses(seed=4, ring=350)
        shape = lambda counts, n: len(counts) == 5 and counts[2] == n and counts[3] < n and counts[4] == counts[3]
    else:
        snap, rays = near_axial_config2()
        _, other = near_axial_config2(seed=6, odd=25)
        shape = lambda counts, n: len(counts) == 3 and counts[1] == n and counts[2] < n
    flat = helpers.flat_scene(snap)
    want, want_counts = c_oracle.trace(flat, rays, 10)
    n = rays.shape[1]
    assert shape(want_counts, n), want_counts
    ds = DeviceScene(snap)
    block = torch.empty((15, n * 10), dtype=torch.float64, device="cuda:0")
    buffers = [dev(rays) for _ in range(8)]
    generations = len(want_counts)
    rows, counts = ds.trace(buffers[0], 10, out=block)           # first trace: look-backs
    assert counts == want_counts and ds.telemetry()["dense_launches"] == 0
    for k in range(1, 4):                                        # from other buffers: every generation dense
        before = ds.telemetry()
        rows, counts = ds.trace(buffers[k], 10, out=block)
        assert counts == want_counts
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"dead list, trace {k}")
        told = ds.telemetry()
        assert told["dense_launches"] - before["dense_launches"] == generations, (told, before)
        assert told["sparse_keep_launches"] - before["sparse_keep_launches"] == 1 and told["speculation_misses"] == 0
    # other rays, lost in other tiles and other numbers
    want_other, counts_other = c_oracle.trace(flat, other, 10)
    assert shape(counts_other, n)
    other_buffer = dev(other)
    rows, counts = ds.trace(other_buffer, 10, out=block)
    assert counts == counts_other and ds.telemetry()["speculation_misses"] == 0
    helpers.assert_frames_match(rows.cpu().numpy().T, want_other, what="other rays on the same hints")
    # in flight on two tickets and two streams, and under the flags that switch the forms off
    got = ds.trace_batch([buffers[4], other_buffer, buffers[5]], 10, depth=2)
    torch.cuda.synchronize()
    for (rows, counts), (frame, frame_counts) in zip(got, ((want, want_counts), (want_other, counts_other), (want, want_counts))):
        assert counts == frame_counts
        helpers.assert_frames_match(rows.cpu().numpy().T, frame, what="in flight")
    for flags in (engine.TRACE_NO_SPARSE_KEEP, engine.TRACE_NO_HINTS, engine.TRACE_KEEP_ABSORBED):
        for k in (6, 7):
            rows, counts = ds.trace(buffers[k], 10, out=block, flags=flags)
            assert counts == want_counts
            helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"flags {flags}")
    # the same buffer again and again
    for k in range(4):
        rows, counts = ds.trace(buffers[7], 10, out=block)
        assert counts == want_counts
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"same buffer, trace {k}")
    assert ds.telemetry()["speculation_misses"] == 0
    ds.close()


@pytest.mark.skipif(bool(engine.DEFAULT_OPTIONS) or bool(engine.DEFAULT_TRACE_FLAGS), reason="counts misses of the default path")
def test_a_dead_list_hint_that_does_not_hold_repeats_the_trace():
    """Rays absorbed in the generation launched on the dead list (by a trap inside the second lens): their tiles' counts
    refute the hint, the trace is repeated, the frame is the oracle's."""
    from pyrayt_amd.engine import DeviceScene

    snap, rays = plate_between_two_lenses()
    flat = helpers.flat_scene(snap)
    ds = DeviceScene(snap)
    n = rays.shape[1]
    block = torch.empty((15, n * 10), dtype=torch.float64, device="cuda:0")
    buffers = [dev(rays) for _ in range(3)]
    for buffer in buffers[:2]:
        ds.trace(buffer, 10, out=block)
    assert ds.telemetry()["sparse_keep_launches"] == 1
    _, astray = plate_between_two_lenses(seed=8, astray=5)
    want, want_counts = c_oracle.trace(flat, astray, 10)
    assert want_counts[3] == n - 200 and want_counts[4] == n - 205   # (recorded in generation 3, by the trap)
    astray_buffer = dev(astray)
    rows, counts = ds.trace(astray_buffer, 10, out=block)
    assert counts == want_counts
    helpers.assert_frames_match(rows.cpu().numpy().T, want, what="after a dead-list hint that did not hold")
    assert ds.telemetry()["speculation_misses"] == 1
    want, want_counts = c_oracle.trace(flat, rays, 10)
    rows, counts = ds.trace(buffers[2], 10, out=block)
    assert counts == want_counts
    helpers.assert_frames_match(rows.cpu().numpy().T, want, what="the first rays again")
    ds.close()


@pytest.mark.skipif(bool(engine.DEFAULT_OPTIONS) or bool(engine.DEFAULT_TRACE_FLAGS), reason="counts misses of the default path")
def test_a_dead_list_that_overflows_costs_one_repeat_and_rests():
    """The dead-list forms are adopted for a generation that loses a few rays (here 60: the list is worth reading when it
    is short).  Then a ray set loses them by the thousand, in more tiles than the list has entries (1020): the
    generation launched on it cannot take its offsets from it, every tile says so, the trace is repeated -- and from
    then on that generation compacts."""
    from pyrayt_amd.engine import DeviceScene

    snap, few = near_axial_config2(n=600_000, odd=60)
    _, many = near_axial_config2(n=600_000, odd=2500)
    flat = helpers.flat_scene(snap)
    want_few, counts_few = c_oracle.trace(flat, few, 10)
    want, want_counts = c_oracle.trace(flat, many, 10)
    n = many.shape[1]
    lost = n - want_counts[2]
    assert want_counts[1] == n and 1500 < lost and lost * 64 <= n and 0 < n - counts_few[2] <= 256
    ds = DeviceScene(snap)
    block = torch.empty((15, n * 3), dtype=torch.float64, device="cuda:0")
    few_buffers = [dev(few) for _ in range(3)]
    for k, buffer in enumerate(few_buffers):
        rows, counts = ds.trace(buffer, 10, out=block)
        assert counts == counts_few, k
    told = ds.telemetry()
    assert told["sparse_keep_launches"] == 2 and told["dense_launches"] == 2 * 3 and told["speculation_misses"] == 0
    helpers.assert_frames_match(rows.cpu().numpy().T, want_few, what="a short dead list")
    buffers = [dev(many) for _ in range(3)]
    for k, buffer in enumerate(buffers):
        rows, counts = ds.trace(buffer, 10, out=block)
        assert counts == want_counts, k
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"overflowing dead list, trace {k}")
        told = ds.telemetry()
        assert told["speculation_misses"] == 1, (k, told)        # (the first of them found out)
    assert told["sparse_keep_launches"] == 2 + 3                 # generation 1 keeps its rays all the same
    ds.close()


@pytest.mark.skipif(bool(engine.DEFAULT_OPTIONS) or bool(engine.DEFAULT_TRACE_FLAGS), reason="counts launches of the default path")
def test_a_generation_that_keeps_too_many_rays_goes_back_to_compacting():
    """The sparse-loss form is sticky (how many rays a launch kept cannot be told from its counts), but the generation
    behind sees them arrive dead: when they are more than a sixteenth of its rays the form is dropped, and taken up
    again when the loss is sparse again."""
    from pyrayt_amd.engine import DeviceScene

    snap, few = stop_before_lens(ring=200)
    _, many = stop_before_lens(seed=4, ring=3000)
    flat = helpers.flat_scene(snap)
    frames = {"few": c_oracle.trace(flat, few, 10), "many": c_oracle.trace(flat, many, 10)}
    n = few.shape[1]
    assert frames["many"][1][1] == n - 3000
    ds = DeviceScene(snap)
    block = torch.empty((15, n * 10), dtype=torch.float64, device="cuda:0")
    kept = []
    buffers = []                                                  # (all alive: another address every time)
    for which in ("few", "few", "many", "many", "few", "few"):
        buffers.append(dev(few if which == "few" else many))
        rows, counts = ds.trace(buffers[-1], 10, out=block)
        want, want_counts = frames[which]
        assert counts == want_counts, which
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"{which}, trace {len(kept)}")
        kept.append(ds.telemetry()["sparse_keep_launches"])
    # first trace: no hints; second: kept; third: still kept (3000 rays of them); fourth: dropped; fifth: compacts and
    # sees a sparse loss again; sixth: kept
    assert kept == [0, 1, 2, 2, 2, 3], kept
    assert ds.telemetry()["speculation_misses"] == 0
    ds.close()
