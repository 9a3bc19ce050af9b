"""Host-side scene logic (CPU): transforms, ids, bounding boxes, snapshots, sources, RaySet.

The snapshot built by pyrayt_amd's own scene classes must equal the one extracted from the
reference's objects (stored in the fixtures) -- same matrices, same cull boxes, same surface
ids in the same order -- since the snapshot is all the device ever sees of a scene.
"""
import numpy as np
import pytest

import helpers
import scenes
from pyrayt_amd import RaySet, components, g3d as cg, materials
from pyrayt_amd.g3d.objects import CountedObject
from pyrayt_amd.scene import SceneSnapshot

SCENE_ARGS = {
    "config1": (1000,), "config2": (2048,), "config3": (2048,), "config4": (256,),
    "config5": (2048,), "two_mirrors": (10,), "tutorial": (10,), "mirrors_and_stops": (4096,),
    "stopped_lens": (2048,),
    # rays on the engine's thresholds, and upstream's stale cull box of a right-nested tree
    "adv_lens": (), "adv_stop": (), "adv_prism": (), "adv_condenser": (), "adv_still": (), "adv_short_a": (), "adv_short_b": (), "adv_short_c": (), "adv_bench_a": (), "adv_bench_b": (), "adv_bench_c": (), "stale_box": (3000,),
}


@pytest.mark.parametrize("name", sorted(SCENE_ARGS))
def test_snapshot_equals_reference(name):
    fx = helpers.load(f"scene_{name}.npz")
    CountedObject.reset_ids()
    parts, rays = scenes.SCENES[name](scenes.product_api(), *SCENE_ARGS[name])
    snap = SceneSnapshot(parts)
    mine, ref = helpers.flat_scene(snap), helpers.scene_of(fx)
    for key in ("prim_type", "prim_material", "prim_normal_scale", "prim_surface_id", "node_op",
                "node_left", "node_right", "node_prim", "roots", "mat_kind"):
        assert np.array_equal(mine[key], ref[key]), key
    for key in ("prim_params", "prim_minv", "node_aabb", "mat_coef"):
        assert np.allclose(mine[key], ref[key], rtol=0, atol=1e-13), key
    assert np.array_equal([sid for sid, _ in snap.surface_lut], fx["lut_ids"])
    # the seeded / source-generated input rays are the ones the reference traced
    assert np.allclose(rays, fx["rays0"], rtol=0, atol=1e-15)


def test_ids_are_global_and_increasing():
    """test/test_tinygfx/test_g3d/test_world_objects.py:11-16 and SURVEY Q9."""
    CountedObject.reset_ids()
    lens = components.biconvex_lens(2, 2, 0.25, aperture=1)
    assert [sid for sid, _ in lens.surface_ids] == [1, 2, 0]  # spheres, then the aperture stock
    assert lens.get_id() == 4
    src = components.ConeOfRays(6)
    assert src.get_id() == 5
    assert components.baffle((1, 1)).get_id() == 6


def test_transform_algebra():
    """world_objects tests :60-188 -- move / rotate / scale compose on the left and the object
    matrix is the inverse."""
    s = cg.Sphere(1).move(1, 2, 3).rotate_z(90).scale(2, 2, 2)
    assert np.allclose(s.get_position()[:3], (-4, 2, 6))
    assert np.allclose(s.get_world_transform() @ s.get_object_transform(), np.identity(4))
    assert np.allclose(cg.Sphere(1).rotate_x(90).get_orientation()[:3], (0, -1, 0))
    assert np.allclose(cg.Sphere(1).rotate_y(90).get_orientation()[:3], (1, 0, 0))
    with pytest.raises(ValueError):
        cg.Sphere(1).scale(-1, 1, 1)
    with pytest.raises(ValueError):
        cg.Sphere(1).rotate_x(1, units="grad")
    with pytest.raises(ValueError):
        cg.Paraboloid(-1, 1)


def test_bounding_boxes():
    """world_objects tests :307-365 and csg tests :46-55,:105-114,:158-172."""
    s = cg.Sphere(1).scale(2, 1, 1).move_x(3)
    assert np.allclose(s.bounding_box.axis_spans, ((1, 5), (-1, 1), (-1, 1)))
    left, right = cg.Sphere(1), cg.Sphere(1)
    union = cg.csg.union(left, right)
    right.move_y(-1)
    assert np.allclose(union.bounding_box.axis_spans, ((-1, 1), (-2, 1), (-1, 1)))
    union.move_x(3)
    assert np.allclose(union.bounding_box.axis_spans, ((2, 4), (-2, 1), (-1, 1)))
    assert left.get_position()[0] == 3 and right.get_position()[0] == 3
    a, b = cg.Sphere(1), cg.Sphere(1)
    inter = cg.csg.intersect(a, b)
    b.move_x(1)
    assert np.allclose(inter.bounding_box.axis_spans, ((0, 1), (-1, 1), (-1, 1)))
    c, d = cg.Sphere(1), cg.Sphere(1).move_y(-1)
    diff = cg.csg.difference(c, d)
    d.move_y(-1)
    assert np.allclose(diff.bounding_box.axis_spans, ((-1, 1), (-1, 1), (-1, 1)))
    assert d._normal_scale == -1 and c._normal_scale == 1


def test_csg_surface_ids_nest():
    """test_csg.py:15-27."""
    l, r = cg.Sphere(1), cg.Sphere(1)
    node = cg.csg.CSGSurface(l, r, cg.csg.Operation.UNION)
    assert [sid for sid, _ in node.surface_ids] == [l.get_id(), r.get_id()]
    assert len(cg.csg.CSGSurface(cg.Sphere(1), node, cg.csg.Operation.UNION).surface_ids) == 3
    with pytest.raises(ValueError):
        cg.csg.CSGSurface(l, r, 7)


def test_sources_match_reference():
    fx = helpers.load("sources.npz")
    c = components
    recipes = {
        "line": lambda: c.LineOfRays(spacing=0.1, wavelength=0.5).move_x(-0.5).rotate_y(-3),
        "circle": lambda: c.CircleOfRays(diameter=2.0).move(0.1, 0.2, 0.3),
        "cone": lambda: c.ConeOfRays(6).move_x(-1.9).rotate_z(10),
        "wedge": lambda: c.WedgeOfRays(30, wavelength=0.7).rotate_x(45),
    }
    for name, make in recipes.items():
        for n in (1, 7, 100):
            got = np.asarray(make().generate_rays(n))
            assert np.allclose(got, fx[f"{name}_{n}"], rtol=0, atol=1e-15), (name, n)
    # test/test_pyrayt/test_components/test_sources.py:48-58
    assert np.allclose(c.LineOfRays(1).generate_rays(3).rays[0, 1], (-0.5, 0, 0.5))
    lamp = c.StaticLamp(1, 1)
    assert lamp.generate_rays(16) is lamp.generate_rays(16)
    assert np.allclose(np.linalg.norm(c.Lamp(1, 2, 45).generate_rays(64).rays[1], axis=0), 1)


def test_rayset_views():
    """test/test_pyrayt/test_core.py:14-36."""
    rs = RaySet(1000)
    assert rs.rays.shape == (2, 4, 1000) and rs.metadata.shape == (5, 1000)
    assert np.all(rs.rays[0, 3] == 1) and np.all(rs.intensity == 100) and np.all(rs.index == 1)
    assert np.allclose(rs.wavelength, 0.633) and np.array_equal(rs.id, np.arange(1000))
    for j, field in enumerate(RaySet.fields):
        rs.metadata[j] = j
        assert np.allclose(getattr(rs, field), j)
        setattr(rs, field, j + 1)
        assert np.allclose(rs.metadata[j], j + 1)
    rs.generation[:10] = 7
    assert np.allclose(rs.metadata[0, :10], 7)


def test_lens_full_thickness():
    """test/test_pyrayt/test_components/test_components.py:8-38."""
    f = components._lens_full_thickness
    assert np.allclose(f(np.inf, np.inf, 1.0, 1.0), (1.0, 0.0))
    assert np.allclose(f(1.0, -1.0, 1.0, 1.0), (1.0, 0.0))  # biconvex: no extra thickness
    sag = 1 - np.sqrt(1 - 0.25)
    assert np.allclose(f(-1.0, 1.0, 1.0, 1.0), (1.0 + 2 * sag, 0.0))
    assert np.allclose(f(-1.0, np.inf, 1.0, 1.0), (1.0 + sag, -sag))


def test_glass_helpers():
    assert np.isclose(materials.SellmeierRefractor(b1=1, c1=1).index_at(2.0), np.sqrt(7 / 3))
    assert 60 < materials.glass["BK7"].abbe() < 68  # BK7: ~64.2
    assert materials.glass["ideal"].index_at(np.array([0.5, 0.6])).tolist() == [1.5, 1.5]


def test_object_group_and_pin():
    """test/test_tinygfx/test_g3d/test_world_objects.py:191-240 (groups) and the pin context
    manager of pyrayt/_pyrayt.py:539-575."""
    import pyrayt_amd as pyrayt

    group = cg.ObjectGroup()
    a, b = cg.WorldObject(), cg.WorldObject()
    group.append(a)
    group.append(b)
    assert len(group) == 2 and list(group) == [a, b] and hasattr(group, "__iter__")
    a.move(1, 0, 0)
    b.move(-1, 0, 0)
    group.scale_all(2)
    assert np.allclose(a.get_position(), (2, 0, 0, 1)) and np.allclose(b.get_position(), (-2, 0, 0, 1))
    group.rotate_z(90)
    assert np.allclose(a.get_position(), (0, 2, 0, 1)) and np.allclose(b.get_position(), (0, -2, 0, 1))
    sub = cg.ObjectGroup()
    c = cg.WorldObject().move(1, 0, 0)
    sub.append(c)
    group.append(sub)
    group.move_x(3)
    assert np.allclose(sub.get_position(), (3, 0, 0, 1)) and np.allclose(c.get_position(), (4, 0, 0, 1))

    lens = components.biconvex_lens(2, 2, 0.25, aperture=1)
    before = SceneSnapshot([lens]).prims["minv"].copy()
    with pyrayt.pin(lens) as (pinned,):
        pinned.move_x(100).rotate_y(30)
        assert np.allclose(lens.get_position()[:3], (100 * np.cos(np.radians(30)), 0, -100 * np.sin(np.radians(30))))
    assert np.allclose(lens.get_position(), (0, 0, 0, 1))
    assert np.allclose(SceneSnapshot([lens]).prims["minv"], before, atol=1e-12)


def test_compiled_programs_carry_cull_steps_only_for_three_or_more_components():
    """prt_scene_create / prt_scene_info are host-only: the scene compiler runs without a GPU."""
    import scenes
    from pyrayt_amd import engine
    from pyrayt_amd.g3d.objects import CountedObject

    api = scenes.product_api()
    expected = {"config2": (2, 4, 0), "config3": (5, 12, 5), "stopped_lens": (3, 5, 3)}
    for name, (components, prims, culls) in expected.items():
        CountedObject.reset_ids()
        parts, _ = scenes.SCENES[name](api, 8)
        ds = engine.DeviceScene.from_components(parts)
        info = ds.info()
        ds.close()
        assert (info["components"], info["primitives"], info["cull_steps"]) == (components, prims, culls), name
        # trace programs: every part factory builds a left-deep chain of two or three leaves, which
        # compiles to one chain record (three step slots); a bare surface is one leaf step (+ culls)
        csg_parts = sum(1 for p in parts if hasattr(p, "children"))
        assert info["chain_steps"] == csg_parts, name
        assert info["trace_steps"] == 3 * csg_parts + (components - csg_parts) + culls, name
        # render programs: one step per leaf and per CSG node plus a root step per CSG component
        # ... behind one line-of-sight cull step per component
        assert info["render_steps"] == (2 * prims - components) + csg_parts + components, name


def test_coordinate_helpers_like_upstream():
    """tinygfx/g3d/primitives.py:18-122: bundle_of_rays / bundle_rays, Point / Vector / Ray."""
    import numpy as np
    import pyrayt_amd.g3d as cg

    p, v = cg.Point(1, 2, 3), cg.Vector(0, 3, 4)
    assert (p.x, p.y, p.z, p.w) == (1, 2, 3, 1) and v.w == 0 and p.dtype == np.float64
    assert isinstance(p, cg.HomogeneousCoordinate) and isinstance(p, np.ndarray)
    assert np.allclose(v.normalize(), (0, 0.6, 0.8, 0)) and v.w == 0
    p.z = 9
    assert p[2] == 9
    ray = cg.Ray()
    assert np.array_equal(ray, ((0, 0, 0, 1), (1, 0, 0, 0)))
    ray.direction = cg.Vector(0, 0, 1)
    ray.origin = cg.Point(1, 1, 1)
    assert ray.origin.x == 1 and ray.direction.z == 1
    block = cg.bundle_rays([ray, cg.Ray()])
    assert block.shape == (2, 4, 2) and np.array_equal(block[:, :, 0], ray)
    assert np.array_equal(cg.bundle_of_rays(3)[0], np.array([[0, 0, 0], [0, 0, 0], [0, 0, 0], [1, 1, 1]]))


def test_group_is_a_list_and_small_world_object_helpers():
    """world_objects.py:15-23 (bounding_box), :156-160 (get_quaternion), :283-295 (ObjectGroup is a
    UserList), :315-317 (attach_to)."""
    import numpy as np
    import pyrayt_amd.g3d as cg

    a, b, c = cg.Sphere(1), cg.Sphere(2), cg.Cuboid()
    group = cg.ObjectGroup([a, b])
    group.append(c)
    group.insert(0, cg.XYPlane())
    assert len(group) == 4 and group.index(a) == 1 and group.count(b) == 1
    group.move_x(2).rotate_z(90)
    assert np.allclose(c.get_position(), (0, 2, 0, 1)) and np.allclose(group.get_position(), (0, 2, 0, 1))
    assert group.pop() is c and len(group) == 3
    group.reverse()
    assert group[0] is b
    quat = cg.Sphere().rotate_z(90).get_quaternion()
    assert np.allclose(quat, (0, 0, np.sqrt(0.5), np.sqrt(0.5)))
    box = cg.bounding_box(np.array([[0.0, 1, 2], [3, -1, 2], [0, 0, 5], [1, 1, 1]]))
    assert box.axis_spans.tolist() == [[0.0, 2.0], [-1.0, 3.0], [0.0, 5.0]]
    part = cg.Sphere()
    part.attach_to(group)
    assert part._parent is group


def test_scene_update_keeps_the_object_for_the_same_shape_only():
    """prt_scene_update is host-side work until a device holds the tables: same parts with other numbers
    go into the existing scene, another shape is refused and leaves it untouched."""
    import scenes
    from pyrayt_amd import engine
    from pyrayt_amd.g3d.objects import CountedObject

    api = scenes.product_api()
    CountedObject.reset_ids()
    parts, _ = scenes.SCENES["config3"](api, 8)
    ds = engine.DeviceScene.from_components(parts)
    before = ds.info()
    parts[1].move_x(0.25).rotate_y(1.0)
    moved = SceneSnapshot(parts)
    assert ds.update(moved) is True and ds.snapshot is moved and ds.info() == before
    assert ds.update(SceneSnapshot(parts[:-1])) is False        # one component fewer: other tables
    assert ds.snapshot is moved and ds.info() == before
    assert ds.update(SceneSnapshot(parts + [api.components.baffle((1, 1)).move_x(90)])) is False
    ds.close()


def test_cull_step_hierarchy_follows_space_when_the_list_does_not():
    """Scenes of eight or more components get a hierarchy of cull steps over groups of components.  Listed
    along the axis, runs of consecutive components are tight groups and are kept; listed in any other
    order, the groups are formed by position (host-only: prt_scene_info says which)."""
    import numpy as np

    import scenes
    from pyrayt_amd import engine
    from pyrayt_amd.g3d.objects import CountedObject

    api = scenes.product_api()

    def train(order):
        CountedObject.reset_ids()
        lenses = [api.components.biconvex_lens(4, 4, 0.25, aperture=1).move_x(1.0 * k) for k in order]
        return lenses + [api.components.baffle((2, 2)).move_x(len(order) + 1.0)]

    in_order = engine.DeviceScene.from_components(train(range(32))).info()
    assert in_order["spatial_groups"] == 0 and in_order["cull_steps"] > 33
    assert in_order["both_directions"] == 1   # a grouped program is stored forwards and in mirror image
    shuffled = np.random.default_rng(5).permutation(32)
    by_position = engine.DeviceScene.from_components(train(shuffled)).info()
    assert by_position["spatial_groups"] == 1
    assert by_position["cull_steps"] > 33                               # one per component plus the group steps
    kept = engine.DeviceScene.from_components(train(shuffled), options={"list_order_groups": 1}).info()
    assert kept["spatial_groups"] == 0 and kept["cull_steps"] == in_order["cull_steps"]
    flat = engine.DeviceScene.from_components(train(shuffled), options={"no_groups": 1}).info()
    assert flat["spatial_groups"] == 0 and flat["cull_steps"] == 33
    few = engine.DeviceScene.from_components(train(np.random.default_rng(6).permutation(6))).info()
    assert few["spatial_groups"] == 0 and few["cull_steps"] == 7         # below eight components: no groups at all
    assert few["both_directions"] == 0 and flat["both_directions"] == 0


def test_scene_options_are_validated_and_versioned():
    """prt_scene_options: zeros are the defaults, unknown names and bad values are rejected on the host, and
    a caller compiled against a shorter struct is served (struct_size says how much of it there is)."""
    import ctypes

    import numpy as np

    import scenes
    from pyrayt_amd import engine
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, _ = scenes.config3(scenes.product_api(), 8)
    snap = SceneSnapshot(parts)
    assert engine.DeviceScene(snap).info()["cull_steps"] == 5
    assert engine.DeviceScene(snap, options={"no_cull": 1}).info()["cull_steps"] == 0
    assert engine.DeviceScene(snap, options={"cull_min": 6}).info()["cull_steps"] == 0
    assert engine.DeviceScene(snap, options={"no_chain": 1}).info()["chain_steps"] == 0
    with pytest.raises(ValueError, match="unknown scene option"):
        engine.DeviceScene(snap, options={"no_such_knob": 1})
    with pytest.raises(ValueError, match="hit_lanes"):
        engine.DeviceScene(snap, options={"hit_lanes": 3})
    # a struct that stops after `no_cull` (12 bytes): the rest counts as zero
    lib = engine.library()
    short = np.zeros(3, dtype=np.int32)
    short[0], short[2] = 12, 1
    handle = ctypes.c_void_p()
    prims, nodes, roots, mats = (np.ascontiguousarray(a) for a in (snap.prims, snap.nodes, snap.roots, snap.materials))
    rc = lib.prt_scene_create(prims.ctypes.data, len(prims), nodes.ctypes.data, len(nodes), roots.ctypes.data, len(roots),
                              mats.ctypes.data, len(mats), short.ctypes.data, ctypes.byref(handle))
    assert rc == 0
    info = (ctypes.c_int64 * 10)()
    assert lib.prt_scene_info(handle, info) == 0 and info[4] == 0  # no cull steps
    lib.prt_scene_destroy(handle)
    short[0] = 0  # struct_size not set
    assert lib.prt_scene_create(prims.ctypes.data, len(prims), nodes.ctypes.data, len(nodes), roots.ctypes.data, len(roots),
                                mats.ctypes.data, len(mats), short.ctypes.data, ctypes.byref(handle)) == -1


def test_every_compiled_program_passes_the_library_own_walk():
    """prt_scene_create walks the trace program it compiled the way the kernels will (every jump lands on a
    step of its own region, no chain record is cut, each direction yields one candidate per component) and
    refuses a scene whose program fails: crowds of random parts at random places, under every option."""
    import numpy as np

    import scenes
    from pyrayt_amd import engine
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    api = scenes.product_api()
    switches = ("no_chain", "no_cull", "no_groups", "no_implied", "list_order_groups", "one_direction", "no_intervals", "no_clearance")
    seen_groups = seen_mirror = 0
    for seed in range(120):
        rng = np.random.default_rng(9000 + seed)
        CountedObject.reset_ids()
        parts = []
        for _ in range(int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 33, 40]))):
            if rng.random() < 0.5:
                part = api.components.biconvex_lens(4, 4, 0.25, aperture=1)
            else:
                part = scenes.random_component(rng, api.cg, api.materials, depth=int(rng.integers(0, 4)))
            spread = 10.0 ** rng.uniform(0.0, 1.5)
            part.move(*(rng.uniform(-spread, spread, 3) * np.array([1.0, rng.random() < 0.5, rng.random() < 0.5])))
            parts.append(part)
        options = {name: 1 for name in switches if rng.random() < 0.2}
        if rng.random() < 0.2:
            options["cull_min"] = int(rng.integers(1, 12))
        scene = engine.DeviceScene(SceneSnapshot(parts), options=options)     # raises if the walk fails
        info = scene.info()
        assert info["components"] == len(parts) and info["trace_steps"] >= len(parts)
        seen_groups += info["spatial_groups"]
        seen_mirror += info["both_directions"]
        scene.close()
    assert seen_groups > 5 and seen_mirror > 10      # the forms the walk is there for did come up


def test_upstream_module_names_resolve():
    """``tinygfx.g3d.world_objects`` and the 2-D helpers of ``tinygfx.g3d.primitives`` (``primitives.py:163-217,
    605-618``) exist under their upstream names; values checked against upstream's in this container."""
    import numpy as np

    import pyrayt_amd
    from pyrayt_amd.g3d import objects, primitives, world_objects

    assert world_objects.Sphere is objects.Sphere and world_objects.TracerSurface is objects.TracerSurface
    assert pyrayt_amd.wavelength_to_rgb is pyrayt_amd.utils.wavelength_to_rgb
    points = np.array([[0.0, 0.6, 0.8, -0.75, 0.2], [0.0, 0.6, 0.0, 0.35, -0.36]])
    assert primitives.Disk(0.8).point_in_shape(points).tolist() == [True, False, True, False, True]
    assert primitives.Disk.from_diameter(1.6).point_in_shape(points[:, 2]) == True  # noqa: E712  (on the edge)
    assert primitives.Rectangle(1.5, 0.7).point_in_shape(points).tolist() == [True, False, False, True, False]
    assert primitives.Rectangle(1.5, 0.7).point_in_shape(np.array([0.2, 0.3])) is True
    left, right = primitives.overlap(np.array([1.0, 5, 3, 9]), np.array([4.0, 2, 12, 8]))
    assert left.tolist() == [5, 3, 9] and right.tolist() == [4, 2, 8]
    assert primitives.overlap(np.zeros((2, 2)), np.zeros((2, 2))) is None
    assert issubclass(primitives.Disk, primitives.Shape2D)
