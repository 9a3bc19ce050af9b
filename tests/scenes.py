"""Scene recipes shared by the golden-fixture generator (run against the reference) and the
parity tests (run against pyrayt_amd).

Every recipe takes an ``api`` namespace with the attributes ``components``, ``materials``,
``cg`` (the g3d module) and ``RaySet`` and builds its scene through names that exist, with
the same meaning, in both the reference and this package -- so the two sides construct the
same objects in the same order (and therefore draw the same surface ids).

Each recipe returns ``(components, rays)`` where ``rays`` is a (13,n) float64 array (the
initial RaySet) built from a seeded generator, never from a global RNG.
"""
from types import SimpleNamespace

import numpy as np


def reference_api():
    """The reference modules (only importable where /root/reference exists)."""
    import pyrayt
    import pyrayt.components
    import pyrayt.materials
    import tinygfx.g3d as cg

    return SimpleNamespace(
        components=pyrayt.components, materials=pyrayt.materials, cg=cg, RaySet=pyrayt.RaySet
    )


def product_api():
    import pyrayt_amd
    import pyrayt_amd.components
    import pyrayt_amd.g3d as cg
    import pyrayt_amd.materials

    return SimpleNamespace(
        components=pyrayt_amd.components,
        materials=pyrayt_amd.materials,
        cg=cg,
        RaySet=pyrayt_amd.RaySet,
    )


def lensmakers_equation(r1, r2, n_lens, thickness):
    p = (n_lens - 1) * (1 / r1 - 1 / r2 + (n_lens - 1) * thickness / (n_lens * r1 * r2))
    return 1 / p


# ---------------------------------------------------------------------------------------------
# ray generators (plain arrays; layout of pyrayt/_pyrayt.py:13-44)
# ---------------------------------------------------------------------------------------------
def blank_rays(n, wavelength=0.633):
    rays = np.zeros((13, n))
    rays[3] = 1.0
    rays[9] = 100.0
    rays[10] = wavelength
    rays[11] = 1.0
    rays[12] = np.arange(n)
    return rays


def cone_rays(n, origin, half_angle_deg, seed, wavelength=0.633):
    """Point source filling a cone about +x uniformly in solid angle."""
    rng = np.random.default_rng(seed)
    u = rng.random(n)
    phi = 2 * np.pi * rng.random(n)
    cos_t = 1 - u * (1 - np.cos(np.radians(half_angle_deg)))
    sin_t = np.sqrt(1 - cos_t ** 2)
    rays = blank_rays(n, wavelength)
    rays[0], rays[1], rays[2] = origin
    rays[4] = cos_t
    rays[5] = sin_t * np.cos(phi)
    rays[6] = sin_t * np.sin(phi)
    return rays


def random_rays(n, seed, box=3.0, wavelength=0.633, degenerate=True):
    """Origins uniform in a cube of half-side ``box``, directions uniform on the sphere, plus
    (optionally) the degenerate families of SURVEY.md appendix B item 1 spliced over the first
    rays: axis-parallel directions, components of magnitude 1e-8 * {0.5, 1, 2}, zero vector."""
    rng = np.random.default_rng(seed)
    rays = blank_rays(n, wavelength)
    rays[0:3] = rng.uniform(-box, box, (3, n))
    d = rng.normal(size=(3, n))
    d /= np.linalg.norm(d, axis=0)
    rays[4:7] = d
    if degenerate and n >= 64:
        k = 0
        for axis in range(3):
            for sign in (1.0, -1.0):
                for _ in range(4):
                    rays[4:7, k] = 0.0
                    rays[4 + axis, k] = sign
                    k += 1
        for axis in range(3):
            for tiny in (0.5e-8, 1e-8, 2e-8, -1e-8):
                other = rng.normal(size=3)
                other[axis] = 0
                other /= np.linalg.norm(other)
                rays[4:7, k] = other
                rays[4 + axis, k] = tiny
                k += 1
        # half of the remaining rays are aimed at the central region so that objects get hit
        m = (n - k) // 2
        target = rng.uniform(-0.6 * box, 0.6 * box, (3, m))
        aim = target - rays[0:3, k : k + m]
        rays[4:7, k : k + m] = aim / np.linalg.norm(aim, axis=0)
        rays[4:7, n - 1] = 0.0  # an absorbed (zero-direction) ray
    return rays


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs
# ---------------------------------------------------------------------------------------------
def config1(api, n):
    """examples/convex_collimator.py: biconvex lens, ConeOfRays(6) at -f, baffle at x=1.
    Rays come from the api's own ConeOfRays source (deterministic)."""
    lens = api.components.biconvex_lens(2, 2, 0.25, aperture=1)
    focus = lensmakers_equation(2, -2, 1.5, 0.25)
    source = api.components.ConeOfRays(cone_angle=6).move_x(-focus)
    baffle = api.components.baffle((1, 1)).move_x(1)
    rays = np.array(source.generate_rays(n))
    rays[12] = np.arange(n)
    return [lens, baffle], rays


def config2(api, n, seed=1234):
    """Single biconvex glass lens + detector plane, seeded 6 degree cone at -f."""
    lens = api.components.biconvex_lens(2, 2, 0.25, aperture=1)
    focus = lensmakers_equation(2, -2, 1.5, 0.25)
    baffle = api.components.baffle((1, 1)).move_x(1)
    return [lens, baffle], cone_rays(n, (-focus, 0.0, 0.0), 6.0, seed)


def config3(api, n, seed=7):
    """Cooke-style triplet + aperture stop + detector: 12 primitives in 5 components."""
    glass = api.materials.glass
    c = api.components
    l1 = c.thick_lens(40, -200, 5, aperture=25.4, material=glass["BK7"])
    l2 = c.thick_lens(-45, 45, 2, aperture=25.4, material=glass["SF2"]).move_x(10)
    l3 = c.thick_lens(200, -40, 5, aperture=25.4, material=glass["BK7"]).move_x(20)
    stop = c.aperture((25.4, 25.4), 12).move_x(14.5)
    det = c.baffle((25.4, 25.4)).move_x(70)
    return [l1, l2, l3, stop, det], cone_rays(n, (-60.0, 0.0, 0.0), 4.0, seed, wavelength=0.55)


def config4(api, n_per_wavelength, n_wavelengths=8):
    """examples/chromatic_dispersion.py: BK7 prism + baffle, one LineOfRays per wavelength."""
    prism = api.components.equilateral_prism(1, 1).move_x(0.25)
    baffle = api.components.baffle((1, 1)).rotate_y(90).move(1, 0, -0.5)
    blocks = []
    for wavelength in np.linspace(0.44, 0.75, n_wavelengths):
        src = api.components.LineOfRays(spacing=0.1, wavelength=wavelength).move_x(-0.5).rotate_y(-3)
        blocks.append(np.array(src.generate_rays(n_per_wavelength)))
    rays = np.hstack(blocks)
    rays[12] = np.arange(rays.shape[1])
    return [prism, baffle], rays


def config5(api, n, seed=11):
    """Plano-parabolic 'aspheric' condenser: Paraboloid & Cylinder in BK7, plus a baffle."""
    cg, glass = api.cg, api.materials.glass["BK7"]
    body = cg.csg.intersect(
        cg.Paraboloid(2.0, 1.0, material=glass),
        cg.Cylinder(1.5, -0.25, 0.75, material=glass),
    ).rotate_y(90)
    det = api.components.baffle((6, 6)).move_x(6)
    return [body, det], cone_rays(n, (-4.0, 0.0, 0.0), 14.0, seed, wavelength=0.59)


# ---------------------------------------------------------------------------------------------
# systems the reference's own tests pin
# ---------------------------------------------------------------------------------------------
def two_mirrors(api, n=10):
    """test/test_pyrayt/test_core.py:54-66: two facing plane mirrors, rays bounce forever."""
    cg, m = api.cg, api.materials.mirror
    src = api.components.LineOfRays()
    first = cg.XYPlane(material=m).rotate_y(-90).move_x(3)
    second = cg.XYPlane(material=m).rotate_y(90).move_x(-3)
    return [first, second], np.array(src.generate_rays(n))


def tutorial(api, n=10):
    """docs/source/tutorial.rst:184-231: biconvex lens, ConeOfRays(10) at x=-2.04, baffle."""
    lens = api.components.biconvex_lens(2, 2, 0.25, aperture=1)
    src = api.components.ConeOfRays(10).move_x(-2.04)  # draws id 5, the baffle then gets 6
    baffle = api.components.baffle((1, 1)).move_x(1)
    return [lens, baffle], np.array(src.generate_rays(n))


def mirrors_and_stops(api, n, seed=3):
    """Mixed materials and every primitive kind as a top-level or CSG member: spherical and
    parabolic mirrors, a plane mirror slab, a union and a difference of spheres in glass."""
    cg, c, matl = api.cg, api.components, api.materials
    sm = c.spherical_mirror(6.0, 0.5, aperture=2.0).move_x(4).rotate_y(180).move_x(8)
    pm = c.parabolic_mirror(3.0, 0.5, aperture=1.5).move_x(-6)
    slab = c.plane_mirror(0.2, aperture=(2.0, 2.0)).rotate_z(30).move(0, 3, 0)
    blob = cg.csg.union(
        cg.Sphere(0.8, material=matl.glass["SF5"]),
        cg.Sphere(0.6, material=matl.glass["SF5"]).move_x(0.7),
    ).move(0, -2.5, 0.3)
    shell = cg.csg.difference(
        cg.Sphere(1.0, material=matl.glass["ideal"]),
        cg.Sphere(0.7, material=matl.glass["ideal"]).move_z(0.2),
    ).scale(1.0, 1.5, 0.8).move(0.5, 0.2, -2.5)
    box = cg.Cuboid.from_sides(1.0, 2.0, 0.5, material=matl.mirror).rotate_x(20).move(-2, -1, 2)
    return [sm, pm, slab, blob, shell, box], random_rays(n, seed, box=5.0, wavelength=0.5)


def stopped_lens(api, n, seed=21):
    """A lens behind an aperture stop that really clips the beam, then a detector.  The stop is
    plane-minus-cylinder, whose Plane child reports its hit twice (t, t): the outcome for
    clipped rays depends on the argsort being stable (true for the numpy the reference locks)."""
    c = api.components
    stop = c.aperture((3.0, 3.0), 0.5).move_x(-0.5)
    lens = c.plano_convex_lens(1.5, 0.3, aperture=1.2, material=api.materials.glass["SF2"])
    det = c.baffle((4, 4)).move_x(2.5)
    return [stop, lens, det], cone_rays(n, (-3.0, 0.0, 0.0), 12.0, seed, wavelength=0.48)


# ---------------------------------------------------------------------------------------------
# renderer views (tinygfx/g3d/renderers.py): (surfaces, camera, light position)
# ---------------------------------------------------------------------------------------------
def draw_camera(api, surfaces, view, resolution):
    """The camera and light ``renderers.draw`` sets up for a view (renderers.py:284-349),
    restated so that stepwise fixtures can be taken with the same geometry."""
    cg = api.cg
    corners = np.hstack([s.bounding_volume.bounding_points[:3] for s in surfaces])
    mins, maxes = np.min(corners, axis=1), np.max(corners, axis=1)
    origin = (maxes + mins) / 2
    light = cg.Point(*maxes)
    if view == "xy":
        origin[2] = 1.5 * maxes[2]
        h_span, v_span = 1.5 * (maxes[:2] - mins[:2])
        light[2] *= 3
    else:
        origin[1] = 1.5 * maxes[1]
        h_span, v_span = 1.5 * (maxes[[0, 2]] - mins[[0, 2]])
        light[1] *= -3
    resolution = resolution if h_span > v_span else int(resolution * h_span / v_span)
    camera = cg.OrthographicCamera(resolution, h_span, v_span / h_span)
    if view == "xy":
        camera.rotate_y(90).rotate_z(90).move(*origin[:3])
    else:
        camera.rotate_z(90).move(*origin[:3])
    return camera, np.asarray(light, dtype=float)


def render_spheres(api):
    """test/test_tinygfx/test_g3d/test_renderers.py:19-24 with coloured Gooch materials."""
    cg = api.cg
    gooch = cg.materials.gooch
    surfaces = (
        cg.Sphere(1, material=gooch.WHITE).move_x(3).move_y(0.5),
        cg.Sphere(1, material=gooch.RED).move_x(3).move_y(-0.5),
    )
    return surfaces, cg.OrthographicCamera(40, 10, 1), np.array((0.0, 10.0, 10.0))


def optical_bench(api):
    """Lens, stop, curved mirror and detector: every tracer material, CSG parts and a bare plane."""
    c = api.components
    lens = c.biconvex_lens(2, 2, 0.25, aperture=1)
    stop = c.aperture((1.5, 1.5), 0.4).move_x(0.6)
    mirror = c.spherical_mirror(4.0, 0.3, aperture=1.2).rotate_z(200).move(2.5, 0.4, 0)
    prism = c.equilateral_prism(0.8, 0.6).move(-1.5, -0.5, 0.1)
    detector = c.baffle((1, 1)).rotate_z(10).move_x(1.2)
    return [lens, stop, mirror, prism, detector]


def render_bench_xy(api, resolution=96):
    surfaces = optical_bench(api)
    camera, light = draw_camera(api, surfaces, "xy", resolution)
    return surfaces, camera, light


def render_bench_xz(api, resolution=96):
    surfaces = optical_bench(api)
    camera, light = draw_camera(api, surfaces, "xz", resolution)
    return surfaces, camera, light


def render_inside(api):
    """A camera in the middle of the scene: some of its rays have all their hits behind them,
    which the renderers' argmin-over-masked / gather-unmasked rule turns into a negative
    nearest hit (renderers.py:79-86)."""
    cg = api.cg
    gooch = cg.materials.gooch
    behind = cg.Sphere(1.2, material=gooch.GREEN).move(-3, 0.4, 0.2)
    ahead = cg.Cuboid.from_sides(1, 1.5, 1, material=gooch.YELLOW).rotate_z(25).move(3, -0.9, 0.3)
    lens_behind = api.components.biconvex_lens(3, 3, 0.4, aperture=1.6).move(-5, -0.8, -0.5)
    shell = cg.csg.difference(
        cg.Sphere(1.0, material=gooch.ORANGE), cg.Sphere(0.8, material=gooch.BLUE).move_x(-0.5)
    ).move(4, 1.2, -0.4)
    around = cg.Cylinder(0.9, -0.3, 0.3).rotate_y(90).move(0, 0.2, 1.0)
    camera = cg.OrthographicCamera(48, 6, 0.75).rotate_z(4).rotate_y(-3)
    return [behind, ahead, lens_behind, shell, around], camera, np.array((2.0, -4.0, 9.0, 1.0))


RENDER_SCENES = {
    "spheres": render_spheres,
    "bench_xy": render_bench_xy,
    "bench_xz": render_bench_xz,
    "inside": render_inside,
}


SCENES = {
    "stopped_lens": stopped_lens,
    "config1": config1,
    "config2": config2,
    "config3": config3,
    "config4": config4,
    "config5": config5,
    "two_mirrors": two_mirrors,
    "tutorial": tutorial,
    "mirrors_and_stops": mirrors_and_stops,
}
