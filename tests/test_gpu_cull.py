"""Scenes of many components: the trace program carries a cull step per component (a ray that
cannot reach a component's box before its current nearest hit skips the component, per wave).
The step must never change a result: HIP engine vs the C oracle (which has no such step), surface
ids exact, on scenes built to stress it -- a lens train traversed in both directions, rays born
inside boxes, incoherent rays over a grid of parts, empty and unioned solids, touching parts."""
import os

import numpy as np
import pytest

import helpers
import scenes
from oracle import c_oracle
from test_gpu_fuzz import random_component

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def check(parts, rays, limit, expect_culls=True, options=None):
    from pyrayt_amd import engine
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.scene import SceneSnapshot

    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    ds = DeviceScene(snap, options=options)
    info = ds.info()
    in_force = dict(engine.DEFAULT_OPTIONS, **(options or {}))  # (tools/run_matrix.sh runs the suite with the steps compiled out)
    if expect_culls and not in_force.get("no_cull"):
        # one cull step per component, plus the steps over groups of components from eight components on
        grouped = len(parts) >= 8 and not in_force.get("no_groups")
        assert info["cull_steps"] >= len(parts) and (info["cull_steps"] > len(parts)) == grouped, info
    elif not expect_culls:
        assert info["cull_steps"] == 0, info
    device_rays = torch.from_numpy(np.ascontiguousarray(rays)).to("cuda:0")
    t, surf = ds.propagate(device_rays)
    want_t, want_surf = c_oracle.propagate(flat, rays)
    assert np.array_equal(surf.cpu().numpy(), want_surf)
    assert np.allclose(t.cpu().numpy(), want_t, rtol=0, atol=helpers.ATOL)
    want, want_counts = c_oracle.trace(flat, rays, limit)
    for flags in (0, 2):
        rows, counts = ds.trace(device_rays, limit, flags=flags)
        assert counts == want_counts, flags
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"flags {flags}")
    ds.close()
    return want_counts


@pytest.fixture()
def api():
    import pyrayt_amd.g3d as cg

    cg.CountedObject.reset_ids()
    return scenes.product_api()


def test_lens_train_both_directions(api):
    c = api.components
    parts = [c.biconvex_lens(4, 4, 0.25, aperture=1).move_x(1.0 * k) for k in range(6)]
    parts.append(c.plane_mirror(0.1, aperture=(2.0, 2.0)).move_x(6.5))      # sends the beam back
    parts.append(c.baffle((3, 3)).move_x(-4))                                # behind the source
    rays = np.hstack((scenes.cone_rays(6000, (-3.0, 0.0, 0.0), 3.0, 17),
                      scenes.random_rays(2192, 18, box=7.0, degenerate=True)))
    rays[12] = np.arange(rays.shape[1])
    counts = check(parts, rays, 40)
    assert len(counts) > 20  # forward through six lenses, off the mirror, and back again


@pytest.mark.parametrize("groups", [True, False])
def test_long_lens_train_with_run_cull_steps(api, groups):
    """33 components: the hierarchy of cull steps over groups of components (and the flat form,
    options.no_groups) against the C oracle, rays entering from both ends and from the side."""
    options = None if groups else {"no_groups": 1}
    c = api.components
    parts = [c.biconvex_lens(4, 4, 0.25, aperture=1).move_x(1.0 * k) for k in range(32)]
    parts.append(c.baffle((2, 2)).move_x(33.0))
    rays = np.hstack((scenes.cone_rays(5000, (-3.0, 0.0, 0.0), 3.0, 41),
                      scenes.cone_rays(3000, (12.5, 0.0, 0.0), 30.0, 42),
                      scenes.random_rays(4288, 43, box=20.0, degenerate=True)))
    rays[4, 5000:6500] *= -1.0  # part of the mid-train cone runs backwards
    rays[12] = np.arange(rays.shape[1])
    counts = check(parts, rays, 70, options=options)
    assert len(counts) > 60


def test_rays_born_inside_boxes_and_nested_parts(api):
    cg, m = api.cg, api.materials
    shell = cg.csg.difference(cg.Sphere(3.0, material=m.glass["BK7"]), cg.Sphere(2.5, material=m.glass["BK7"]))
    core = cg.Sphere(0.8, material=m.mirror).move(0.3, 0.2, -0.1)
    bar = cg.Cuboid.from_sides(0.3, 4.0, 0.3, material=m.glass["SF5"]).rotate_z(30)
    plate = api.components.baffle((8, 8)).move_x(3.5)
    rays = scenes.random_rays(8192, 23, box=2.4, degenerate=True)
    check([shell, core, bar, plate], rays, 8)


def test_incoherent_rays_over_a_grid_of_parts(api):
    cg, m = api.cg, api.materials
    parts = []
    for ix in range(4):
        for iy in range(4):
            kind = (ix + iy) % 3
            if kind == 0:
                part = cg.Sphere(0.35, material=m.mirror)
            elif kind == 1:
                part = api.components.biconvex_lens(1.5, 1.5, 0.2, aperture=0.7).rotate_z(15 * ix)
            else:
                part = cg.Cuboid.from_sides(0.5, 0.4, 0.6, material=m.glass["SF2"]).rotate_x(20 * iy)
            parts.append(part.move(1.2 * ix - 1.8, 1.2 * iy - 1.8, 0.3 * (ix - iy)))
    rays = scenes.random_rays(16384, 29, box=3.5, degenerate=True)
    check(parts, rays, 6)


def test_empty_unioned_and_touching_solids(api):
    cg, m = api.cg, api.materials
    nothing = cg.csg.intersect(cg.Sphere(1.0, material=m.mirror), cg.Sphere(1.0, material=m.mirror).move_x(5))
    kissing = cg.csg.intersect(cg.Sphere(1.0, material=m.mirror).move_y(3),
                               cg.Sphere(1.0, material=m.mirror).move(2.0, 3.0, 0.0))  # touch in one point
    dumbbell = cg.csg.union(cg.Sphere(0.6, material=m.glass["ideal"]).move_z(-2),
                            cg.Sphere(0.6, material=m.glass["ideal"]).move_z(2.2))   # disjoint union
    stack_a = cg.Cuboid.from_sides(1, 1, 1, material=m.mirror).move(-3, 0, 0)
    stack_b = cg.Cuboid.from_sides(1, 1, 1, material=m.absorber).move(-2, 0, 0)       # shares a face with a
    rays = scenes.random_rays(8192, 31, box=4.0, degenerate=True)
    # a beam along the shared face plane and through the touching point
    rays[0:3, 100:164] = np.array([[-2.5], [-3.0], [0.0]]) + np.random.default_rng(3).uniform(-0.2, 0.2, (3, 64)) * [[0], [0], [1]]
    rays[4:7, 100:164] = [[0.0], [1.0], [0.0]]
    check([nothing, kissing, dumbbell, stack_a, stack_b], rays, 6)


@pytest.mark.parametrize("seed", range(6))
def test_random_scenes_of_many_components(api, seed):
    rng = np.random.default_rng(7000 + seed)
    parts = []
    for _ in range(int(rng.integers(6, 14))):
        comp = random_component(rng, api.cg, api.materials, depth=int(rng.integers(0, 3)))
        parts.append(comp.move(*rng.uniform(-4.0, 4.0, 3)))
    rays = scenes.random_rays(12_000, seed=7100 + seed, box=5.0, wavelength=0.55)
    check(parts, rays, 6)


def test_two_components_carry_no_cull_steps(api):
    parts, rays = scenes.config2(api, 2048)
    check(parts, rays, 10, expect_culls=False)


# ---- groups formed by position: components listed in any order ------------------------------------------
def test_shuffled_lens_train_groups_by_position(api):
    """32 lenses + detector listed in random order: the cull-step hierarchy is built over positions, the
    program visits components out of list order, ids and frames still equal the C oracle's."""
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.scene import SceneSnapshot

    c = api.components
    order = np.random.default_rng(11).permutation(32)
    parts = [c.biconvex_lens(4, 4, 0.25, aperture=1).move_x(1.0 * k) for k in order]
    parts.insert(7, c.baffle((2, 2)).move_x(33.0))
    from pyrayt_amd import engine
    if not engine.DEFAULT_OPTIONS:  # (tools/run_matrix.sh runs the suite under other scene options)
        assert DeviceScene(SceneSnapshot(parts)).info()["spatial_groups"] == 1
    rays = np.hstack((scenes.cone_rays(5000, (-3.0, 0.0, 0.0), 3.0, 41),
                      scenes.cone_rays(3000, (12.5, 0.0, 0.0), 30.0, 42),
                      scenes.random_rays(4288, 43, box=20.0, degenerate=True)))
    rays[4, 5000:6500] *= -1.0
    rays[12] = np.arange(rays.shape[1])
    counts = check(parts, rays, 70)
    assert len(counts) > 60
    check(parts, rays, 12, options={"list_order_groups": 1})


def test_ties_between_components_resolve_in_list_order_whatever_the_program_order(api):
    """Coincident surfaces in different components: the reference keeps the component that comes first in
    the LIST (strict '<' running minimum, pyrayt/_pyrayt.py:380-386).  With groups formed by position the
    program meets them in another order and has to compare (t, list index)."""
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.scene import SceneSnapshot

    cg, m, c = api.cg, api.materials, api.components
    rng = np.random.default_rng(23)
    parts = []
    # pairs of identical surfaces at identical places, the twins far apart in the list; materials differ,
    # so a wrong winner changes the frame as well as the surface id
    spots = [(float(x), float(y)) for x in (-6, -2, 2, 6) for y in (-3, 3)]
    twins = []
    for k, (x, y) in enumerate(spots):
        first = cg.Sphere(0.8, material=m.mirror).move(x, y, 0.0)
        second = cg.Sphere(0.8, material=m.absorber if k % 2 else m.glass["BK7"]).move(x, y, 0.0)
        parts.append(first)
        twins.append(second)
    plates = [c.baffle((3, 3)).move_x(9.0), c.baffle((3, 3)).move_x(9.0)]   # the same plane twice
    order = rng.permutation(len(twins))
    parts = parts[:3] + [plates[1]] + parts[3:] + [twins[k] for k in order] + [plates[0]]
    from pyrayt_amd import engine
    if not engine.DEFAULT_OPTIONS:
        assert DeviceScene(SceneSnapshot(parts)).info()["spatial_groups"] == 1
    rays = scenes.random_rays(16384, 77, box=7.0, degenerate=True)
    aim = np.array([spots[k % len(spots)] + (0.0,) for k in range(4000)]).T + rng.normal(0, 0.3, (3, 4000))
    rays[4:7, 200:4200] = aim - rays[0:3, 200:4200]
    rays[4:7, 200:4200] /= np.linalg.norm(rays[4:7, 200:4200], axis=0)
    rays[12] = np.arange(rays.shape[1])
    check(parts, rays, 6)


@pytest.mark.parametrize("seed", range(8))
def test_random_scenes_listed_in_random_order(api, seed):
    rng = np.random.default_rng(9000 + seed)
    parts = []
    for _ in range(int(rng.integers(9, 20))):
        comp = random_component(rng, api.cg, api.materials, depth=int(rng.integers(0, 3)))
        parts.append(comp.scale(*rng.uniform(0.4, 0.9, 3)).move(*rng.uniform(-6.0, 6.0, 3)))
    rays = scenes.random_rays(12_000, seed=9100 + seed, box=7.0, wavelength=0.55)
    check(parts, rays, 6)
    check(parts, rays, 3, options={"hit_lanes": 8})
