"""The C restatement (oracle/prt_oracle.c) pinned to the same golden vectors as the numpy one,
and the two oracles cross-checked on seeded scenes the fixtures do not contain.  CPU only."""
import os
import subprocess

import numpy as np
import pytest

import helpers
import scenes
from oracle import c_oracle, prt_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(c_oracle.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)


@pytest.mark.parametrize("name", ["config1", "config2", "config3", "config4", "config5",
                                  "two_mirrors", "tutorial", "mirrors_and_stops", "stopped_lens",
                                  "adv_lens", "adv_stop", "adv_prism", "adv_condenser", "adv_still", "adv_short_a", "adv_short_b", "adv_short_c", "adv_bench_a", "adv_bench_b", "adv_bench_c", "stale_box"])
def test_trace_matches_reference(name):
    fx = helpers.load(f"scene_{name}.npz")
    frame, counts = c_oracle.trace(helpers.scene_of(fx), fx["rays0"], int(fx["generation_limit"]))
    helpers.assert_frames_match(frame, fx["frame"], what=name)
    t, surf = c_oracle.propagate(helpers.scene_of(fx), fx["rays0"])
    assert np.array_equal(surf, fx["surf_0"])
    assert np.allclose(t, fx["t_0"], rtol=0, atol=helpers.ATOL)


@pytest.mark.parametrize("name", ["union_spheres", "intersect_spheres", "difference_spheres",
                                  "plane_minus_cylinder", "cube_chain", "right_nested", "balanced"])
def test_csg_nearest_hit(name):
    """First positive finite entry of the reference's component hit list == C propagate."""
    fx = helpers.load("csg.npz")
    key = name + "__"
    hits, ids = fx[key + "hits"], fx[key + "ids"]
    masked = np.where((hits > 0) & np.isfinite(hits), hits, np.inf)
    row = np.argmin(masked, axis=0)
    cols = np.arange(hits.shape[1])
    want_t = masked[row, cols]
    want_s = np.where(np.isfinite(want_t), ids[row, cols], -1)
    t, surf = c_oracle.propagate(helpers.scene_of(fx, key), fx[key + "rays"])
    assert np.array_equal(surf, want_s)
    assert np.allclose(t, want_t, rtol=0, atol=helpers.ATOL)


@pytest.mark.parametrize("name,args,limit", [("config3", (5000,), 10), ("mirrors_and_stops", (20000,), 8),
                                             ("stopped_lens", (8000,), 10), ("config5", (6000,), 10)])
def test_two_oracles_agree(name, args, limit):
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.SCENES[name](scenes.product_api(), *args)
    flat = helpers.flat_scene(SceneSnapshot(parts))
    a, ca = prt_oracle.trace(flat, rays, limit)
    b, cb = c_oracle.trace(flat, rays, limit)
    assert ca == cb
    helpers.assert_frames_match(b, a, atol=1e-12, what=name)


def test_untracable_surface():
    from pyrayt_amd import components, g3d as cg
    from pyrayt_amd.scene import SceneSnapshot

    flat = helpers.flat_scene(SceneSnapshot([cg.Sphere(1).move_x(3)]))
    rays = np.asarray(components.LineOfRays().generate_rays(4))
    with pytest.raises(AttributeError):
        c_oracle.trace(flat, rays, 5)
    with pytest.raises(AttributeError):
        prt_oracle.trace(flat, rays, 5)


def test_all_cores_baseline_worker_runs():
    """oracle/cpu_bench.py (bench.py's `cpu_baseline_all_cores` leg): P processes over contiguous id ranges of
    the seeded job give the same number of rows as one process, and a JSON line."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = []
    for procs in (1, 3):
        done = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", "--rays", "6000", "--procs", str(procs),
                               "--repeat", "2", "--limit", "10"], cwd=root, capture_output=True, text=True, timeout=300)
        assert done.returncode == 0, done.stderr[-1000:]
        lines.append(json.loads(done.stdout.strip().splitlines()[-1]))
    assert lines[0]["rows"] == lines[1]["rows"] > 2 * 2 * 6000
    assert lines[1]["procs"] == 3 and lines[1]["rows_per_s"] > 0
