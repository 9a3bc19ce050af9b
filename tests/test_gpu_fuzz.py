"""Randomised scenes: HIP engine vs the C oracle on seeded random CSG trees, transforms and
materials -- shapes the part factories never build (unions, right-nested and balanced trees,
anisotropic scales, every primitive as any child).  Surface ids must agree exactly, values to
1e-6 (they agree far tighter; the assertion keeps the north-star tolerance)."""
import numpy as np
import pytest

import helpers
import scenes
from oracle import c_oracle

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def random_surface(rng, cg, matl):
    material = [matl.absorber, matl.mirror, matl.glass["ideal"], matl.glass["BK7"], matl.glass["SF2"]][
        rng.integers(0, 5)]
    kind = rng.integers(0, 5)
    if kind == 0:
        s = cg.Sphere(rng.uniform(0.4, 1.2), material=material)
    elif kind == 1:
        s = cg.Cylinder(rng.uniform(0.3, 0.9), -rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), material=material)
    elif kind == 2:
        s = cg.XYPlane(rng.uniform(1.0, 3.0), rng.uniform(1.0, 3.0), material=material)
    elif kind == 3:
        s = cg.Cuboid.from_sides(*rng.uniform(0.5, 1.8, 3), material=material)
    else:
        s = cg.Paraboloid(rng.uniform(0.3, 1.0), rng.uniform(0.5, 1.5), material=material)
    if rng.random() < 0.5:
        s.scale(*rng.uniform(0.6, 1.5, 3))
    s.rotate_x(rng.uniform(-180, 180)).rotate_y(rng.uniform(-180, 180)).rotate_z(rng.uniform(-180, 180))
    s.move(*rng.uniform(-0.6, 0.6, 3))
    return s


def random_component(rng, cg, matl, depth):
    if depth == 0 or rng.random() < 0.25:
        return random_surface(rng, cg, matl)
    op = [cg.csg.union, cg.csg.intersect, cg.csg.difference][rng.integers(0, 3)]
    left = random_component(rng, cg, matl, depth - 1)
    right = random_component(rng, cg, matl, depth - 1)
    node = op(left, right)
    if rng.random() < 0.5:
        node.rotate_z(rng.uniform(-90, 90)).move(*rng.uniform(-0.3, 0.3, 3))
    return node


def _seeds():
    # the everyday tier, and the 500-seed tier for gpurun sessions: PRT_FUZZ_SEEDS=500 pytest -m gpu ...
    import os

    return range(int(os.environ.get("PRT_FUZZ_SEEDS", "24")))


@pytest.mark.parametrize("seed", _seeds())
def test_random_scene(seed):
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    api = scenes.product_api()
    rng = np.random.default_rng(1000 + seed)
    CountedObject.reset_ids()
    parts = []
    for _ in range(rng.integers(1, 5)):
        comp = random_component(rng, api.cg, api.materials, depth=int(rng.integers(0, 4)))
        comp.move(*rng.uniform(-2.0, 2.0, 3))
        parts.append(comp)
    rays = scenes.random_rays(20_000, seed=5000 + seed, box=4.0, wavelength=0.55)
    rays[10] = rng.uniform(0.4, 0.8, rays.shape[1])
    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    ds = DeviceScene(snap)
    device_rays = torch.from_numpy(rays).to("cuda:0")
    # nearest hit of every ray
    t, surf = ds.propagate(device_rays)
    want_t, want_surf = c_oracle.propagate(flat, rays)
    assert np.array_equal(surf.cpu().numpy(), want_surf)
    assert np.allclose(t.cpu().numpy(), want_t, rtol=0, atol=helpers.ATOL)
    # whole trace, reference-faithful bookkeeping (absorbed rays carried) and the default
    want, want_counts = c_oracle.trace(flat, rays, 6)
    for flags in (0, 1, 2):
        rows, counts = ds.trace(device_rays, 6, flags=flags)
        assert counts == want_counts, (seed, flags)
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"seed {seed} flags {flags}")
    ds.close()
