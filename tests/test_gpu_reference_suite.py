"""The reference's own unit / integration tests for the path, re-stated against pyrayt_amd's API so
that a maintainer can read them side by side (same scenarios, same analytic expectations, same
tolerances -- np.allclose defaults unless the reference states otherwise):

    test/integration_tests/int_test_thick_lenses.py:8-113      six thick-lens families
    test/test_tinygfx/test_g3d/test_csg.py:38-209              two offset unit spheres under each operation
    test/test_pyrayt/test_pyrayt_materials.py:48-169           refraction, TIR, Sellmeier
    test/test_tinygfx/test_g3d/test_primitives.py              representative known answers
    docs/source/tutorial.rst:184-231                           the printed tutorial frame

Everything below runs on the HIP engine (component.intersect / Material.trace / RayTracer.trace)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture()
def pyrayt():
    import pyrayt_amd

    pyrayt_amd.g3d.CountedObject.reset_ids()
    return pyrayt_amd


# ---------------------------------------------------------------------------------------------
# int_test_thick_lenses.py
# ---------------------------------------------------------------------------------------------
class ThickLensBench:
    focus, aperture, thickness = 5, 1, 0.1

    def __init__(self, pyrayt, focus=None):
        self.pyrayt = pyrayt
        self.focus = self.focus if focus is None else focus
        self.baffle = pyrayt.components.baffle((2 * self.aperture, 2 * self.aperture)).move_x(self.focus)
        self.source = pyrayt.components.LineOfRays(0.5 * self.aperture).move_x(-1)

    def on_baffle(self, lens):
        results = self.pyrayt.RayTracer(self.source, [lens, self.baffle]).trace()
        rays = results.loc[results["surface"] == self.baffle.get_id()]
        return results, np.asarray(rays["x_tilt"]), np.asarray(rays["y_tilt"]), np.asarray(rays["y0"])


def test_planar_lens(pyrayt):  # :15-24
    bench = ThickLensBench(pyrayt)
    results, *_ = bench.on_baffle(pyrayt.components.thick_lens(np.inf, np.inf, bench.thickness, aperture=1))
    assert np.allclose(results["x_tilt"], 1.0) and np.allclose(results["y_tilt"], 0.0)
    assert np.allclose(results["z_tilt"], 0.0)


def test_positive_meniscus_lens(pyrayt):  # :26-50: the power has the right sign
    r_lens, thickness = 1, 1
    focus = ((0.5 ** 2) / 1.5 * (thickness / r_lens ** 2)) ** -1
    bench = ThickLensBench(pyrayt, focus)
    _, x_tilt, y_tilt, y_0 = bench.on_baffle(pyrayt.components.thick_lens(r_lens, r_lens, thickness, aperture=1))
    assert len(y_0) and np.all(-focus * y_tilt / x_tilt * y_0 > 0)


@pytest.mark.parametrize("radii,sign,rtol", [
    (lambda f: (f, -f), -1, 0.01),            # biconvex :52-65
    (lambda f: (np.inf, -f / 2), -1, 0.01),   # plano-convex :67-81
    (lambda f: (-f, f), +1, 0.01),            # biconcave :83-97
    (lambda f: (np.inf, f / 2), +1, 0.02),    # plano-concave :99-113
])
def test_focusing_and_diverging_lenses(pyrayt, radii, sign, rtol):
    bench = ThickLensBench(pyrayt)
    r1, r2 = radii(bench.focus)
    _, x_tilt, y_tilt, y_0 = bench.on_baffle(pyrayt.components.thick_lens(r1, r2, bench.thickness, aperture=1))
    assert len(y_0) == 10
    assert np.allclose(bench.focus * y_tilt / x_tilt, sign * y_0, rtol=rtol)


# ---------------------------------------------------------------------------------------------
# test_csg.py: two unit spheres, the right one moved by -1 in y
# ---------------------------------------------------------------------------------------------
def sweep(cg, n_rays):
    y_vals = np.linspace(-2, 2, n_rays)
    rays = cg.bundle_of_rays(n_rays)
    rays[1, 0] = 1
    rays[0, 0] = -5
    rays[0, 1] = y_vals
    return y_vals, rays


def test_csg_union(pyrayt):  # :38-92
    cg = pyrayt.g3d
    left, right = cg.Sphere(1), cg.Sphere(1)
    node = cg.csg.CSGSurface(left, right, cg.csg.Operation.UNION)
    assert np.allclose(node.bounding_box.axis_spans, np.array(((-1, -1, -1), (1, 1, 1))).T)
    right.move_y(-1)
    assert np.allclose(node.bounding_box.axis_spans, np.array(((-1, -2, -1), (1, 1, 1))).T)
    y_vals, rays = sweep(cg, 11)
    hits, surfaces = node.intersect(rays)
    assert np.all(np.isinf(hits[2:]))
    missed = np.all(np.isinf(hits), axis=0)
    assert not np.any(missed[(y_vals > -2) & (y_vals < 1)])
    assert np.allclose(hits, np.sort(hits, axis=0))
    r_hits, _ = right.intersect(rays)
    l_hits, _ = left.intersect(rays)
    low, high = y_vals < -0.5, y_vals > -0.5
    assert np.allclose(hits[:2, low], r_hits[:, low])
    assert np.all(surfaces[:2, low & ~missed] == right.get_id())
    assert np.allclose(hits[:2, high], l_hits[:, high])
    assert np.all(surfaces[:2, high & ~missed] == left.get_id())


def test_csg_intersect(pyrayt):  # :99-150
    cg = pyrayt.g3d
    left, right = cg.Sphere(1), cg.Sphere(1)
    node = cg.csg.CSGSurface(left, right, cg.csg.Operation.INTERSECT)
    right.move_x(1)
    assert np.allclose(node.bounding_box.axis_spans, np.array(((0, -1, -1), (1, 1, 1))).T)
    right.move_x(-1).move_y(-1)
    y_vals, rays = sweep(cg, 11)
    hits, surfaces = node.intersect(rays)
    assert np.all(np.isinf(hits[2:]))
    missed = np.all(np.isinf(hits), axis=0)
    assert not np.any(missed[(y_vals > -1) & (y_vals < 0)])
    assert np.allclose(hits, np.sort(hits, axis=0))
    r_hits, _ = right.intersect(rays)
    l_hits, _ = left.intersect(rays)
    low, high = y_vals < -0.5, y_vals > -0.5
    assert np.allclose(hits[:2, low], l_hits[:, low])
    assert np.all(surfaces[:2, low & ~missed] == left.get_id())
    assert np.allclose(hits[:2, high], r_hits[:, high])
    assert np.all(surfaces[:2, high & ~missed] == right.get_id())


def test_csg_difference(pyrayt):  # :152-209
    cg = pyrayt.g3d
    left, right = cg.Sphere(1), cg.Sphere(1).move_y(-1)
    node = cg.csg.CSGSurface(left, right, cg.csg.Operation.DIFFERENCE)
    box = np.array(((-1, -1, -1), (1, 1, 1))).T
    assert np.allclose(node.bounding_box.axis_spans, box)
    right.move_y(-1)   # moving the subtracted solid does not move the box
    assert np.allclose(node.bounding_box.axis_spans, box)
    right.move_y(1)
    y_vals, rays = sweep(cg, 101)
    hits, surfaces = node.intersect(rays)
    assert np.all(np.isinf(hits[2:, y_vals > 0]))
    bitten = (y_vals < 0) & (y_vals > -0.5)
    assert not np.any(np.isinf(hits[2:, bitten]))
    missed = np.all(np.isinf(hits), axis=0)
    assert np.all(missed[(y_vals < -0.5) | (y_vals > 1)])
    assert np.allclose(hits, np.sort(hits, axis=0))
    l_hits, _ = left.intersect(rays)
    r_hits, _ = right.intersect(rays)
    assert np.allclose(hits[:2, y_vals > 0], l_hits[:, y_vals > 0])
    assert np.all(surfaces[:2, (y_vals > 0) & ~missed] == left.get_id())
    assert np.allclose(hits[[[0], [3]], bitten], l_hits[:, bitten])
    assert np.all(surfaces[[[0], [3]], bitten] == left.get_id())
    assert np.allclose(hits[1:3, bitten], r_hits[:, bitten])
    assert np.all(surfaces[1:3, bitten] == right.get_id())


# ---------------------------------------------------------------------------------------------
# test_pyrayt_materials.py
# ---------------------------------------------------------------------------------------------
def test_basic_refractor(pyrayt):  # :48-110
    index = 1.6
    material, surface = pyrayt.materials.BasicRefractor(index), pyrayt.g3d.XYPlane()

    def angle(rs):
        return np.arctan(np.abs(rs.rays[1, 1] / rs.rays[1, 2]))

    rays = pyrayt.RaySet(10)
    rays.rays[1, 2] = 1.0
    rays.index = 20
    assert np.allclose(material.trace(surface, rays).index, 1.0)          # exiting resets to 1
    rays = pyrayt.RaySet(10)
    rays.rays[1, 1], rays.rays[1, 2] = 1, -1                               # into the medium at 45 deg
    assert np.allclose(angle(material.trace(surface, rays)), np.arcsin(np.sin(np.pi / 4) / index))
    rays = pyrayt.RaySet(10)
    rays.rays[1, 1], rays.rays[1, 2] = np.sin(0.1), np.cos(0.1)           # leaving, near normal
    rays.index = index
    assert np.allclose(angle(material.trace(surface, rays)), np.arcsin(np.sin(0.1) * index))
    rays = pyrayt.RaySet(10)
    rays.rays[1, 1], rays.rays[1, 2] = 1, 1                                # leaving at 45 deg: TIR
    rays.index = index
    out = material.trace(surface, rays)
    assert np.allclose(angle(out), np.pi / 4) and np.allclose(out.index, index)


def test_sellmeier_refractor(pyrayt):  # :112-169
    for coeff in ([1, 0, 0, 1, 0, 0], [0, 1, 0, 0, 1, 0], [0, 0, 1, 0, 0, 1]):
        material = pyrayt.materials.SellmeierRefractor(*coeff)
        assert material.index_at(2.0) == pytest.approx(np.sqrt(7 / 3), abs=1e-7)
        assert np.allclose(material.index_at(np.full(100, 2.0)), np.sqrt(7 / 3))
    material, surface = pyrayt.materials.SellmeierRefractor(b1=1, c1=1), pyrayt.g3d.XYPlane()
    rays = pyrayt.RaySet(2)
    rays.wavelength[:] = 2.0
    rays.rays[1, 2] = -1.0
    assert np.allclose(material.trace(surface, rays).index, np.sqrt(7 / 3))
    rays = pyrayt.RaySet(10)
    rays.wavelength[:] = 2.0
    rays.rays[1, 2], rays.rays[1, 1] = -1.0, 1.0
    out = material.trace(surface, rays)
    assert np.allclose(np.arctan(np.abs(out.rays[1, 1] / out.rays[1, 2])),
                       np.arcsin(np.sqrt(3 / 7) * np.sqrt(2) / 2))


# ---------------------------------------------------------------------------------------------
# test_primitives.py -- the known answers SURVEY.md section 4 lists
# ---------------------------------------------------------------------------------------------
def one_ray(cg, origin, direction):
    rays = cg.bundle_of_rays(1)
    rays[0, :3, 0] = origin
    rays[1, :3, 0] = direction
    return rays


def test_primitive_known_answers(pyrayt):
    cg = pyrayt.g3d
    root2 = np.sqrt(2)
    hits = lambda surface, o, d: surface.intersect(one_ray(cg, o, d))[0][:, 0]  # noqa: E731
    assert np.allclose(hits(cg.Sphere(1), (0, 0, 0), (1, 0, 0)), (-1, 1))                   # :126-137
    assert np.allclose(hits(cg.Sphere(1), (-1, 0, 1), (1, 0, 0)), (1, 1))                   # tangent :160-163
    assert np.allclose(hits(cg.Paraboloid(1, 3), (0, 0, -1), (0, 0, 1)), (1, 4))            # linear :204-210
    assert np.all(np.isinf(hits(cg.Paraboloid(1, 3), (10, 0, -1), (0, 0, 1))))              # beside it
    assert np.allclose(hits(cg.XYPlane(2, 2), (-1, 0, 1), (1, 0, -1) / root2), root2)       # :295-297
    assert np.all(np.isinf(hits(cg.XYPlane(2, 2), (5, 0, 1), (0, 0, -1))))                  # off the patch :309-313
    assert np.allclose(hits(cg.Cuboid(), (-2, 0, 0), (1, 0, 0)), (1, 3))                    # :375-384
    assert np.allclose(hits(cg.Cuboid(), (-2, -2, 0), (1, 1, 0) / root2), root2 * np.array((1, 3)))
    assert np.allclose(hits(cg.Cylinder(1, -1, 1), (-2, 0, 0), (1, 0, 0)), (1, 3))          # sidewall :504-513
    assert np.allclose(hits(cg.Cylinder(1, -1, 1), (0, 0, -2), (0, 0, 1)), (1, 3))          # cap to cap
    assert np.all(np.isinf(hits(cg.Cylinder(1, -1, 1), (-2, 3, 0), (1, 0, 0))))             # miss :548-560
    corner = np.array(((1.0,), (1.0,), (1.0,), (1.0,)))
    assert np.allclose(cg.Cuboid().get_world_normals(corner)[:3, 0], np.ones(3) / np.sqrt(3))  # :468-477
    near_face = np.array(((1.0 + 1e-9,), (0.2,), (0.3,), (1.0,)))
    assert np.allclose(cg.Cuboid().get_world_normals(near_face)[:3, 0], (1, 0, 0))          # :460-466


# ---------------------------------------------------------------------------------------------
# the tutorial's printed frame
# ---------------------------------------------------------------------------------------------
def test_tutorial_frame(pyrayt):  # docs/source/tutorial.rst:184-231
    lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
    source = pyrayt.components.ConeOfRays(10).move_x(-2.04)
    baffle = pyrayt.components.baffle((1, 1)).move_x(1)
    frame = pyrayt.RayTracer(source, [lens, baffle], rays_per_source=10).trace()
    assert len(frame) == 30
    ray0 = frame.loc[frame["id"] == 0].sort_values("generation")
    assert np.allclose(ray0[["x_tilt", "y_tilt", "z_tilt"]].to_numpy(),
                       ((0.984808, 0, 0.173648), (0.998415, -1.99e-17, 0.056272), (0.999988, 9.6e-20, -0.004965)),
                       atol=1e-6)
    assert np.allclose(ray0["x1"], (-0.095388, 0.093505, 1.0), atol=1e-6)
    assert np.allclose(ray0["index"], (1.0, 1.5, 1.0))
    assert list(ray0["surface"]) == [1.0, 2.0, 6.0]
