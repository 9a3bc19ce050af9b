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


random_surface, random_component = scenes.random_surface, scenes.random_component  # (shared with the fuzz-seed fixtures)


def _seeds():
    # the everyday tier, and the soak tiers for gpurun sessions: PRT_FUZZ_SEEDS=60000 [PRT_FUZZ_FIRST=60000] pytest -m gpu -n 6 ...
    import os

    first = int(os.environ.get("PRT_FUZZ_FIRST", "0"))
    return range(first, first + int(os.environ.get("PRT_FUZZ_SEEDS", "200")))


def build_random_scene(seed):
    """(parts, rays, rng, short, odd) of fuzz seed `seed` (also used by tools/diag_fuzz.py)."""
    from pyrayt_amd.g3d.objects import CountedObject

    api = scenes.product_api()
    rng = np.random.default_rng(1000 + seed)
    CountedObject.reset_ids()
    parts = []
    for _ in range(rng.integers(1, 5)):
        comp = random_component(rng, api.cg, api.materials, depth=int(rng.integers(0, 4)))
        comp.move(*rng.uniform(-2.0, 2.0, 3))
        parts.append(comp)
    if seed % 7 == 3:  # a crowd of small parts: cull steps over runs of components, with rays from everywhere
        crowd = np.random.default_rng(31_000 + seed)
        for _ in range(int(crowd.integers(5, 10))):
            comp = random_component(crowd, api.cg, api.materials, depth=int(crowd.integers(0, 3)))
            comp.scale(*crowd.uniform(0.3, 0.7, 3)).move(*crowd.uniform(-3.0, 3.0, 3))
            parts.append(comp)
    if seed % 11 == 5:  # parts of any size: upstream's absolute thresholds meet object-space directions of any length
        giant = np.random.default_rng(57_000 + seed)
        for comp in parts[: int(giant.integers(1, 3))]:
            comp.scale(*(10.0 ** giant.uniform(-2.0, 2.5, 3)))
    rays = scenes.random_rays(20_000, seed=5000 + seed, box=4.0, wavelength=0.55)
    rays[10] = rng.uniform(0.4, 0.8, rays.shape[1])
    # directions of any length: upstream never normalises what it is given, and its isclose() branches
    # (absolute thresholds) fire for short directions whether or not they are parallel to anything
    short = rng.choice(rays.shape[1] - 1000, size=600, replace=False) + 500
    rays[4:7, short] *= 10.0 ** rng.uniform(-9.0, 1.0, size=600)
    # ... and lengths at the edges of the band in which a ray counts as well formed (the one gate of every
    # shortcut: |d|^2 >= 0.81, prt_device.hpp well_formed): a hair inside, on, and a hair outside (and around
    # 1.1, which an earlier form of the gate also excluded)
    edge_rng = np.random.default_rng(77_000 + seed)  # (its own stream: the families above stay as they were)
    edge = edge_rng.choice(2000, size=90, replace=False) + 16_000
    scale = np.concatenate((np.repeat([0.9, 1.1], 15) * (1.0 + edge_rng.integers(-4, 5, 30) * 2.0 ** -52),
                            edge_rng.uniform(0.85, 1.15, 30), edge_rng.choice([0.9, 1.1], 30) + edge_rng.normal(0, 1e-9, 30)))
    rays[4:7, edge] *= scale
    # rays that start inside some medium (their index row is what Snell's law takes for n1), of any brightness
    inside = rng.choice(18_000, size=1500, replace=False)
    rays[11, inside] = rng.uniform(1.0, 2.0, 1500)
    rays[9, inside[:300]] = 10.0 ** rng.uniform(-12.0, 3.0, 300)
    # ... nor does it insist on homogeneous coordinates being 1 / 0: the 4x4 transforms take whatever is there
    odd = rng.choice(400, size=40, replace=False) + 19_000
    rays[3, odd[:20]] = rng.uniform(0.3, 3.0, 20) * rng.choice([-1.0, 1.0], 20)
    rays[7, odd[20:]] = 10.0 ** rng.uniform(-6.0, -1.0, 20) * rng.choice([-1.0, 1.0], 20)
    return parts, rays, rng, short, odd


@pytest.mark.parametrize("seed", _seeds())
def test_random_scene(seed):
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.scene import SceneSnapshot

    parts, rays, rng, short, odd = build_random_scene(seed)
    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    ds = DeviceScene(snap)
    device_rays = torch.from_numpy(rays).to("cuda:0")
    # nearest hit of every ray
    t, surf = ds.propagate(device_rays)
    want_t, want_surf = c_oracle.propagate(flat, rays)
    assert np.array_equal(surf.cpu().numpy(), want_surf)
    assert np.allclose(t.cpu().numpy(), want_t, rtol=0, atol=helpers.ATOL)
    # every component's own hit list (component.intersect(), csg.py:118-160) on a slice that holds all
    # the degenerate families: values where finite, ids where the list has an entry
    from oracle import prt_oracle

    sub = np.concatenate([np.arange(0, 600), short[:300], odd, [rays.shape[1] - 1]])
    rays8 = np.ascontiguousarray(rays[:8, sub])
    for root in range(len(flat["roots"])):
        hits, ids = ds.intersect(root, torch.from_numpy(rays8).to("cuda:0"))
        want_hits, want_ids = prt_oracle.component_hits(flat, root, rays8.reshape(2, 4, -1))
        got_hits, got_ids = hits.cpu().numpy(), ids.cpu().numpy()
        want_hits = np.where(np.isnan(want_hits), np.inf, want_hits)       # upstream's NaN = miss
        finite = np.isfinite(want_hits)
        assert np.array_equal(np.isfinite(got_hits), finite), (seed, root)
        assert np.allclose(got_hits[finite], want_hits[finite], rtol=1e-12, atol=helpers.ATOL), (seed, root)
        assert np.array_equal(np.isneginf(got_hits), np.isneginf(want_hits)), (seed, root)
        assert np.array_equal(got_ids[finite], np.asarray(want_ids)[finite]), (seed, root)
    # whole trace, reference-faithful bookkeeping (absorbed rays carried) and the default; every fifth
    # seed on a ray count around a wave / tile / grid boundary (the last rays: the degenerate ones are there)
    if seed % 5 == 1:
        n = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 511, 513, 1025, 4097]))
        rays = np.ascontiguousarray(rays[:, -n:])
        device_rays = torch.from_numpy(rays).to("cuda:0")
    want, want_counts = c_oracle.trace(flat, rays, 6)
    for flags in (0, 1, 2):
        rows, counts = ds.trace(device_rays, 6, flags=flags)
        assert counts == want_counts, (seed, flags)
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"seed {seed} flags {flags}")
    ds.close()


# ---------------------------------------------------------------------------------------------
# optical benches from the part factories: the shapes the engine has specialised paths for (chain
# steps per lens / prism / mirror, component cull steps from three parts, cull steps over runs of
# parts from eight), traced repeatedly so that the second and third trace run on the dense-mode
# hints of the one before
# ---------------------------------------------------------------------------------------------
random_part = scenes.random_part  # (shared with the fuzz-seed fixtures)


@pytest.mark.parametrize("seed", _seeds())
def test_random_bench(seed):
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    api = scenes.product_api()
    rng = np.random.default_rng(77_000 + seed)
    CountedObject.reset_ids()
    n_parts = int(rng.integers(1, 6)) if rng.random() < 0.8 else int(rng.integers(8, 13))
    parts, x = [], 0.0
    for _ in range(n_parts):
        parts.append(random_part(rng, api.components, api.materials).move_x(x))
        x += float(rng.uniform(0.5, 1.8))
    parts.append(api.components.baffle((3.0, 3.0)).move_x(x + 0.5))
    n = int(rng.choice([3_000, 20_000, 33_333]))
    rays = scenes.cone_rays(n, (-1.5, 0.0, 0.0), float(rng.uniform(2.0, 12.0)), 6000 + seed,
                            wavelength=float(rng.uniform(0.45, 0.7)))
    # a few rays of the families that are not optics any more but that upstream traces all the same: directions
    # of any length, homogeneous coordinates other than 1 / 0 (every third seed, so that most benches keep
    # the compact state form and the rest exercise the fall-back to all 13 rows)
    odd = rng.choice(n, size=60, replace=False)
    rays[4:7, odd[:40]] *= 10.0 ** rng.uniform(-9.0, 1.0, size=40)
    if seed % 3 == 0:
        rays[3, odd[40:50]] = rng.uniform(0.3, 3.0, 10)
        rays[7, odd[50:]] = 10.0 ** rng.uniform(-6.0, -1.0, 10)
    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    limit = int(rng.integers(3, 14))
    want, want_counts = c_oracle.trace(flat, rays, limit)
    ds = DeviceScene(snap)
    device_rays = torch.from_numpy(rays).to("cuda:0")
    twin = device_rays.clone()
    # every trace after the first runs on the hints of the one before (among them sparse-loss generations with their
    # absorbed rays kept and the generations behind them on dead lists), whichever buffer the rays come from
    for turn, buffer in enumerate((device_rays, twin, device_rays, device_rays, device_rays)):
        rows, counts = ds.trace(buffer, limit)
        assert counts == want_counts, (seed, turn)
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"bench seed {seed} turn {turn}")
    assert ds.telemetry()["speculation_misses"] == 0
    rows, counts = ds.trace(device_rays, limit, flags=2)       # and the three-kernel path
    assert counts == want_counts
    helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"bench seed {seed} unfused")
    ds.close()


@pytest.mark.parametrize("seed", _seeds())
def test_random_scene_traced_again_from_the_same_buffers(seed):
    """The fuzz scenes lose rays in every generation: traced again with the same buffers, then with some rays changed
    in place -- rays that change places, a block replaced -- so that some tiles hold what they held and others do not:
    whatever the hints of the trace before are worth then, the frame is the oracle's."""
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.scene import SceneSnapshot

    parts, rays, rng, short, odd = build_random_scene(seed)
    rays = np.ascontiguousarray(rays[:, : int(rng.choice([700, 4096, 20_000]))])
    n = rays.shape[1]
    snap = SceneSnapshot(parts)
    flat = helpers.flat_scene(snap)
    ds = DeviceScene(snap)
    buf = torch.from_numpy(rays).to("cuda:0")
    block = torch.empty((15, n * 6), dtype=torch.float64, device="cuda:0")
    want, want_counts = c_oracle.trace(flat, rays, 6)
    for k in range(2):
        rows, counts = ds.trace(buf, 6, out=block)
        assert counts == want_counts, (seed, k)
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"seed {seed} pass {k}")
    from pyrayt_amd import engine

    if not engine.DEFAULT_OPTIONS and not engine.DEFAULT_TRACE_FLAGS and want_counts:
        assert ds.telemetry()["speculation_misses"] == 0, seed  # (the same rays again refute no hint)
    changed = rays.copy()
    kind = seed % 3
    if kind == 0:      # two rays change places (ids stay in order): equal totals, at most two tiles differ
        a, b = rng.choice(n, 2, replace=False)
        changed[:12, [a, b]] = changed[:12, [b, a]]
    elif kind == 1:    # a block of rays replaced by copies of one ray
        start = int(rng.integers(0, n - 64))
        changed[:12, start:start + 64] = changed[:12, [int(rng.integers(0, n))]]
    # (kind 2: nothing changes -- a third pass on the records)
    want, want_counts = c_oracle.trace(flat, changed, 6)
    buf.copy_(torch.from_numpy(changed))
    for k in range(2):
        rows, counts = ds.trace(buf, 6, out=block)
        assert counts == want_counts, (seed, "changed", k)
        helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"seed {seed} changed pass {k}")
    ds.close()


# ---------------------------------------------------------------------------------------------
# record plans (round 6): the PLAN instantiations of the generation kernel on the fuzz scenes -- a filtered trace
# must be the full trace's frame filtered, BIT FOR BIT (same arithmetic, other stores), the fused sums those of the
# frame oracle on it; repeated, so that the plan's own dense-mode hints (and their kept-absorbed-rays form) run too
# ---------------------------------------------------------------------------------------------
def _check_plans(ds, device_rays, limit, rng, seed):
    from oracle import frame_oracle
    from pyrayt_amd import engine

    if engine.DEFAULT_TRACE_FLAGS & (engine.TRACE_UNFUSED | engine.TRACE_COUNT_PATHS):
        pytest.skip("record plans need the fused path")

    full, full_counts = ds.trace(device_rays, limit, plan=None)
    full = full.clone()
    frame = full.cpu().numpy()
    ids = np.unique(frame[5]).astype(np.int64) if frame.shape[1] else np.array([7], dtype=np.int64)
    picks = [tuple(int(v) for v in rng.choice(ids, size=min(len(ids), int(rng.integers(1, 4))), replace=False)), ()]
    n = device_rays.shape[1]
    for surfaces in picks:
        keep = np.isin(frame[5].astype(np.int64), surfaces) if surfaces else np.ones(frame.shape[1], dtype=bool)
        want = frame[:, keep]
        rps, groups = (max(n // 3, 1), 4) if seed % 2 else (None, 1)
        for rows_on in (True, False):
            if not rows_on and not surfaces and seed % 4:
                continue
            plan = engine.RecordPlan(surfaces=surfaces, rows=rows_on, stats=True, rays_per_source=rps, n_groups=groups,
                                     mean_square=("y1", 0.125, None), generation_limit=limit)
            for turn in range(3):
                rows, counts = ds.trace(device_rays, limit, plan=plan)
                torch.cuda.synchronize()
                if rows_on:
                    assert np.array_equal(rows.cpu().numpy(), want, equal_nan=True), (seed, surfaces, turn)
                    assert sum(counts) == want.shape[1]
                else:
                    assert rows.shape[1] == 0
                got = plan.sums.cpu().numpy()
                # (rows with a NaN in a summed column poison that sum, in the oracle as in the kernel: compared as such)
                for g in range(limit):
                    ref = frame_oracle.reduce_sums(want, None, float(g), rps, groups)
                    ok = np.isclose(got[g, :, :9], ref, rtol=1e-10, atol=1e-9, equal_nan=True) | ~np.isfinite(ref)
                    assert ok.all(), (seed, surfaces, rows_on, turn, g, got[g, :, :9], ref)
    rows, counts = ds.trace(device_rays, limit, plan=None)
    assert counts == full_counts and np.array_equal(rows.cpu().numpy(), frame, equal_nan=True), seed


@pytest.mark.parametrize("seed", list(_seeds())[::4])
def test_random_scene_under_record_plans(seed):
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.scene import SceneSnapshot

    parts, rays, rng, short, odd = build_random_scene(seed)
    rays = np.ascontiguousarray(rays[:, : int(rng.choice([700, 4096, 20_000]))])
    ds = DeviceScene(SceneSnapshot(parts))
    _check_plans(ds, torch.from_numpy(rays).to("cuda:0"), 6, rng, seed)
    ds.close()


@pytest.mark.parametrize("seed", list(_seeds())[::4])
def test_random_bench_under_record_plans(seed):
    from pyrayt_amd.engine import DeviceScene
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    api = scenes.product_api()
    rng = np.random.default_rng(99_000 + seed)
    CountedObject.reset_ids()
    parts, x = [], 0.0
    for _ in range(int(rng.integers(1, 6))):
        parts.append(random_part(rng, api.components, api.materials).move_x(x))
        x += float(rng.uniform(0.5, 1.8))
    parts.append(api.components.baffle((3.0, 3.0)).move_x(x + 0.5))
    n = int(rng.choice([3_000, 20_000]))
    rays = scenes.cone_rays(n, (-1.5, 0.0, 0.0), float(rng.uniform(2.0, 12.0)), 8000 + seed,
                            wavelength=float(rng.uniform(0.45, 0.7)))
    ds = DeviceScene(SceneSnapshot(parts))
    _check_plans(ds, torch.from_numpy(rays).to("cuda:0"), int(rng.integers(3, 10)), rng, seed)
    ds.close()
