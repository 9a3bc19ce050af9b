"""Record plans (include/prt.h prt_record_plan, round 6): a trace that stores only the rows of some surfaces, and / or
accumulates the frame reductions in its generation kernels instead of storing rows.

What is pinned here, against the genuine reference's golden frames (tests/golden/scene_*.npz):
  * a filtered trace's frame is ``frame.loc[frame.surface.isin(ids)]`` of the reference's frame, row for row --
    surface / generation / id exact, the rest to the parity bar (it is bit-identical in practice);
  * the fused sums equal the frame oracle's sums of the reference frame (oracle/frame_oracle.py, itself pinned
    against pandas) to 1e-12 relative, and the tables built from them equal DeviceFrame.group_stats / mean_square;
  * a plan changes nothing for the traces without one, before or after;
  * plans of different tickets in flight together, repeated traces (the plan's own dense-mode hints), rotating ray sets.
"""
import numpy as np
import pytest

import helpers
import scenes
from oracle import frame_oracle
from pyrayt_amd import engine
from pyrayt_amd.frame import DeviceFrame, SinkStats

# (tools/run_matrix.sh runs the suites under every trace flag: a plan's traces run on the fused path only)
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(bool(engine.DEFAULT_TRACE_FLAGS & (engine.TRACE_UNFUSED | engine.TRACE_COUNT_PATHS)),
                                                  reason="record plans need the fused path")]

torch = pytest.importorskip("torch")

SCENE_FIXTURES = ["config1", "config2", "config3", "config4", "config5", "two_mirrors", "tutorial", "mirrors_and_stops",
                  "stopped_lens", "adv_lens", "adv_stop", "adv_prism", "adv_condenser", "adv_still", "adv_short_a",
                  "adv_bench_a", "adv_bench_b", "stale_box"]


def dev(array):
    return torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64)).to("cuda:0")


def device_scene(scene_dict, options=None):
    return engine.DeviceScene(helpers.FixtureSnapshot(scene_dict), options=options)


def surface_choices(frame):
    """Surface sets worth filtering a reference frame by: the surface the last generation hits most (the "imager"),
    the most frequent surface overall, a pair, every surface of the frame, and one that is not in the scene."""
    surf = frame[:, 5].astype(np.int64)
    if surf.size == 0:
        return [(999,)]
    ids, counts = np.unique(surf, return_counts=True)
    last = surf[frame[:, 0] == frame[:, 0].max()]
    last_ids, last_counts = np.unique(last, return_counts=True)
    picks = [(int(last_ids[np.argmax(last_counts)]),), (int(ids[np.argmax(counts)]),), tuple(int(v) for v in ids[:2]),
             tuple(int(v) for v in ids[:8]), (999_999,)]
    out = []
    for p in picks:
        if p not in out:
            out.append(p)
    return out


def filtered(frame, ids):
    keep = np.isin(frame[:, 5].astype(np.int64), np.asarray(ids, dtype=np.int64))
    return frame[keep]


def counts_of(frame, limit):
    counts = [int((frame[:, 0] == g).sum()) for g in range(limit)]
    while counts and counts[-1] == 0:
        counts.pop()
    return counts


@pytest.mark.parametrize("name", SCENE_FIXTURES)
def test_filtered_frame_is_the_reference_frame_filtered(name):
    fx = helpers.load(f"scene_{name}.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    for ids in surface_choices(fx["frame"]):
        want = filtered(fx["frame"], ids)
        plan = engine.RecordPlan(surfaces=ids, rows=True, generation_limit=limit)
        for attempt in range(3):  # (a first trace, then two on the plan's own hints)
            rows, counts = ds.trace(rays, limit, plan=plan)
            helpers.assert_frames_match(rows.cpu().numpy().T, want, what=f"{name} surfaces={ids} attempt {attempt}")
            assert counts == counts_of(want, limit), (name, ids, attempt)
    # ... and without a plan the scene traces as ever
    rows, counts = ds.trace(rays, limit, plan=None)
    helpers.assert_frames_match(rows.cpu().numpy().T, fx["frame"], what=f"{name} after the plans")
    ds.close()


@pytest.mark.parametrize("name", SCENE_FIXTURES)
@pytest.mark.parametrize("store_rows", [False, True])
def test_fused_sums_equal_the_frame_oracle_on_the_reference_frame(name, store_rows):
    fx = helpers.load(f"scene_{name}.npz")
    limit = int(fx["generation_limit"])
    frame = fx["frame"]
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    n = rays.shape[1]
    for ids in surface_choices(frame)[:3] + [()]:
        for rays_per_source, n_groups in ((None, 1), (max(n // 3, 1), 4)):
            pivots_host = np.array([[1e-3 * (g + 1), -2e-3, 0.5 + g] for g in range(n_groups)])
            for pivots in (None, dev(pivots_host)):
                plan = engine.RecordPlan(surfaces=ids, rows=store_rows, stats=True, rays_per_source=rays_per_source,
                                         n_groups=n_groups, pivots=pivots, mean_square=("axis_intercept", 0.25, None),
                                         generation_limit=limit)
                for attempt in range(2):
                    rows, counts = ds.trace(rays, limit, plan=plan)
                    torch.cuda.synchronize()
                    got = plan.sums.cpu().numpy()
                    sel = filtered(frame, ids) if ids else frame
                    if store_rows:
                        helpers.assert_frames_match(rows.cpu().numpy().T, sel, what=f"{name} {ids} rows beside the sums")
                    else:
                        assert rows.shape[1] == 0 and sum(counts) == 0
                    for g in range(limit):
                        want = frame_oracle.reduce_sums(sel.T, None, float(g), rays_per_source, n_groups,
                                                        None if pivots is None else pivots_host)
                        scale = np.maximum(np.abs(want), 1.0)
                        assert np.all(np.abs(got[g, :, :9] - want) <= 1e-12 * scale * max(sel.shape[0], 1) ** 0.5 + 1e-300), (
                            name, ids, g, got[g, :, :9], want)
                        # the mean-square sums of the same rows: count of finite values, sum v, sum v^2
                        rows_g = sel[sel[:, 0] == g]
                        with np.errstate(all="ignore"):
                            v = rows_g[:, 6] - rows_g[:, 12] * rows_g[:, 7] / rows_g[:, 13] - 0.25
                        group = (np.floor(rows_g[:, 4] / rays_per_source).astype(int) if rays_per_source
                                 else np.zeros(len(rows_g), int))
                        for k in range(n_groups):
                            vk = v[(group == k) & np.isfinite(v)]
                            ms = np.array([len(vk), vk.sum(), (vk * vk).sum()])
                            assert np.allclose(got[g, k, 9:], ms, rtol=1e-11, atol=1e-12 * max(len(vk), 1)), (name, ids, g, k)
    ds.close()


@pytest.mark.parametrize("name", ["config2", "config3", "config4", "tutorial"])
def test_tables_from_the_fused_sums_equal_the_frame_reductions(name):
    """SinkStats.group_stats / mean_square == DeviceFrame.group_stats / mean_square of the full trace's frame: per
    generation, for the last generation and for all of them, per source."""
    fx = helpers.load(f"scene_{name}.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    n = rays.shape[1]
    rows, counts = ds.trace(rays, limit, plan=None)
    full = DeviceFrame(rows.clone(), counts)
    imager = surface_choices(fx["frame"])[0][0]
    rps, groups = max(n // 2, 1), 2
    for transform, quantity, about in ((None, "axis_intercept", 0.7), ("sin", "y_tilt", 0.01), (None, "x1", 0.0)):
        plan = engine.RecordPlan(surfaces=(imager,), rows=False, stats=True, rays_per_source=rps, n_groups=groups,
                                 mean_square=(quantity, about, transform), generation_limit=limit)
        ds.trace(rays, limit, plan=plan)
        torch.cuda.synchronize()
        stats = SinkStats(plan.sums)
        for generation in [None, "last"] + list(range(len(counts))):
            number = None if generation is None else (stats.last_generation_number() if generation == "last" else generation)
            want = full.group_stats(surface=imager, generation=number, rays_per_source=rps, n_groups=groups)
            got = stats.group_stats(generation)
            assert np.array_equal(got["count"].to_numpy(), want["count"].to_numpy()), (name, generation)
            for column in ("y", "z", "rms_radius", "focus", "focus_std", "wavelength", "intensity"):
                assert np.allclose(got[column].to_numpy(), want[column].to_numpy(), rtol=1e-9, atol=1e-12, equal_nan=True), (
                    name, generation, column, got[column].to_numpy(), want[column].to_numpy())
            want_ms = full.mean_square(quantity, about=about, transform=transform, surface=imager,
                                       generation=number, rays_per_source=rps, n_groups=groups)
            got_ms = stats.mean_square(generation, per_source=True)
            assert np.array_equal(got_ms["count"].to_numpy(), want_ms["count"].to_numpy()), (name, generation, quantity)
            assert np.allclose(got_ms["mean_square"].to_numpy(), want_ms["mean_square"].to_numpy(), rtol=1e-10, atol=1e-300,
                               equal_nan=True), (name, generation, quantity)
    ds.close()


def test_plans_of_two_tickets_in_flight_together():
    """Ticket 0 stores the imager's rows, ticket 1 only sums, ticket 2 has no plan: three traces in flight on three
    streams, each gives what it gives alone."""
    fx = helpers.load("scene_config3.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    imager = surface_choices(fx["frame"])[0]
    want_rows = filtered(fx["frame"], imager)
    plan_rows = engine.RecordPlan(surfaces=imager, rows=True, generation_limit=limit)
    plan_sums = engine.RecordPlan(surfaces=imager, rows=False, stats=True, generation_limit=limit)
    ds.set_plan(0, plan_rows)
    ds.set_plan(1, plan_sums)
    streams = ds.ticket_streams(rays.device, 3)
    outs = [torch.empty((15, rays.shape[1] * limit), dtype=torch.float64, device="cuda:0") for _ in range(3)]
    for rounds in range(3):
        for ticket in range(3):
            streams[ticket].wait_stream(torch.cuda.current_stream())
            ds.trace_begin(ticket, rays, limit, outs[ticket], stream=streams[ticket])
        results = [ds.trace_end(ticket) for ticket in range(3)]
        torch.cuda.synchronize()
        helpers.assert_frames_match(results[0][0].cpu().numpy().T, want_rows, what="ticket 0 (rows of the imager)")
        assert results[1][0].shape[1] == 0
        want = frame_oracle.reduce_sums(want_rows.T, None, None, None, 1)
        got = plan_sums.sums.cpu().numpy().sum(axis=0)[:, :9]
        assert np.allclose(got, want, rtol=1e-11, atol=1e-12), (got, want)
        helpers.assert_frames_match(results[2][0].cpu().numpy().T, fx["frame"], what="ticket 2 (no plan)")
    ds.close()


def test_sums_only_plan_at_the_north_star_size_runs_dense_and_matches_the_reference_summary():
    """BASELINE config 2 at 1M rays under a sums-only plan on the detector: after the first trace every generation
    launches dense (the plan keeps its absorbed rays, so nothing is compacted for the 9 near-axial rays), rotating
    ray sets included, and the counts are the reference's (tests/golden/config2_1m_summary*.npz)."""
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays0 = scenes.config2(scenes.product_api(), 1_000_000, seed=1234)
    ds = engine.DeviceScene(SceneSnapshot(parts))
    detector = parts[1].get_id()
    plan = engine.RecordPlan(surfaces=(detector,), rows=False, stats=True, generation_limit=10)
    sets = [dev(rays0)]
    for seed in (1235, 1236):
        CountedObject.reset_ids()
        _, more = scenes.config2(scenes.product_api(), 1_000_000, seed=seed)
        sets.append(dev(more))
    full_rows, full_counts = ds.trace(sets[0], 10, plan=None)
    full = DeviceFrame(full_rows.clone(), full_counts)
    want = full.group_stats(surface=detector)
    before = ds.telemetry()
    for k in range(9):
        ds.trace(sets[k % 3], 10, plan=plan)
    torch.cuda.synchronize()
    ds.trace(sets[0], 10, plan=plan)
    torch.cuda.synchronize()
    stats = SinkStats(plan.sums)
    counts = plan.sums[:, 0, 0].cpu().numpy()
    assert counts[:3].tolist() == [0.0, 9.0, 999_991.0] and not counts[3:].any()
    got = stats.group_stats(None)
    for column in ("y", "z", "rms_radius", "focus", "focus_std", "wavelength", "intensity"):
        assert np.allclose(got[column].to_numpy(), want[column].to_numpy(), rtol=1e-9, atol=1e-12, equal_nan=True), column
    after = ds.telemetry()
    # 10 traces x 3 generations under the plan, of which only the very first trace's ran without hints
    if not engine.DEFAULT_TRACE_FLAGS & engine.TRACE_NO_HINTS:   # (tools/run_matrix.sh also runs this suite without hints)
        assert after["plan_dense_launches"] - before["plan_dense_launches"] >= 27
    ds.close()


def test_rows_plan_at_the_north_star_size():
    """1M rays, rows of the detector only: 9 + 999 991 rows, equal to the detector rows of the full trace."""
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays0 = scenes.config2(scenes.product_api(), 1_000_000, seed=1234)
    ds = engine.DeviceScene(SceneSnapshot(parts))
    detector = parts[1].get_id()
    rays = dev(rays0)
    full_rows, full_counts = ds.trace(rays, 10, plan=None)
    keep = full_rows[5] == float(detector)
    want = full_rows[:, keep].clone()
    plan = engine.RecordPlan(surfaces=(detector,), rows=True, generation_limit=10)
    for attempt in range(4):
        rows, counts = ds.trace(rays, 10, plan=plan)
        assert counts == [0, 9, 999_991], counts
        assert torch.equal(rows, want), attempt
    ds.close()


def test_raytracer_record_only_and_trace_stats():
    """The front end: record_only() gives the frame's .loc[] cut, trace_stats() the spot table, and a plain trace()
    afterwards is the whole frame again."""
    import pyrayt_amd as pyrayt

    pyrayt.g3d.objects.CountedObject.reset_ids()
    lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
    src = pyrayt.components.ConeOfRays(cone_angle=6).move_x(-1.9)
    det = pyrayt.components.baffle((1, 1)).move_x(1)
    tracer = pyrayt.RayTracer(src, [lens, det], rays_per_source=20_000)
    whole = tracer.trace()
    device_whole = tracer.device_frame
    want_table = device_whole.group_stats(surface=det.get_id())
    want_ms = device_whole.mean_square("y_tilt", about=0.01, transform="sin", generation="last")
    cut = whole.loc[whole["surface"] == det.get_id()].reset_index(drop=True)
    tracer.record_only(det)
    got = tracer.trace()
    assert got.shape == cut.shape and np.array_equal(got.to_numpy(), cut.to_numpy(), equal_nan=True)
    stats = tracer.trace_stats(surface=det, mean_square=("y_tilt", 0.01, "sin"))
    table = stats.group_stats()
    for column in table.columns:
        assert np.allclose(table[column].to_numpy(), want_table[column].to_numpy(), rtol=1e-9, atol=1e-12, equal_nan=True), column
    # (the highest generation that counted a row of the detector is the frame's last one here)
    assert np.isclose(stats.mean_square("last"), want_ms, rtol=1e-10)
    tracer.record_only(det, columns=("y1", "z1"))                   # the spot diagram of cells 11 / 19: two columns of the cut
    spot = tracer.trace()
    assert list(spot.columns) == ["y1", "z1"] and np.array_equal(spot.to_numpy(), cut[["y1", "z1"]].to_numpy())
    tracer.record_only()
    again = tracer.trace()
    assert again.shape == whole.shape and np.array_equal(again.to_numpy(), whole.to_numpy(), equal_nan=True)


@pytest.mark.parametrize("name", ["config2", "config3", "tutorial", "adv_prism"])
def test_a_column_list_writes_those_columns_and_no_others(name):
    """RecordPlan(columns=...): the rows a plan keeps write the listed columns only -- equal to the reference frame's,
    the other rows of the record block untouched (a block filled with a sentinel keeps it) -- with and without a
    surface filter, and the DeviceFrame hands out, and brings to the host, those columns alone."""
    fx = helpers.load(f"scene_{name}.npz")
    limit = int(fx["generation_limit"])
    frame = fx["frame"]
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    cap = rays.shape[1] * limit
    names = ("y1", "z1", "surface")
    index = [DeviceFrame(torch.zeros((15, 0))).columns.index(c) for c in names]
    for ids in (surface_choices(frame)[0], ()):
        want = filtered(frame, ids) if ids else frame
        plan = engine.RecordPlan(surfaces=ids, rows=True, columns=names, generation_limit=limit)
        for attempt in range(2):
            block = torch.full((15, cap), -7.25, dtype=torch.float64, device="cuda:0")
            rows, counts = ds.trace(rays, limit, out=block, plan=plan)
            torch.cuda.synchronize()
            assert rows.shape[1] == want.shape[0] and counts == counts_of(want, limit)
            got = rows.cpu().numpy()
            for k in range(15):
                if k in index:
                    assert np.array_equal(got[k], want[:, k], equal_nan=True), (name, ids, k)
                else:
                    assert np.all(got[k] == -7.25), (name, ids, k)
            assert torch.all(block[:, rows.shape[1]:] == -7.25)          # nothing behind the rows either
        view = DeviceFrame(rows, counts, plan.columns)
        assert view.columns == tuple(sorted(names, key=lambda c: DeviceFrame(torch.zeros((15, 0))).columns.index(c)))
        table = view.to_pandas()
        assert list(table.columns) == list(view.columns) and table.shape == (want.shape[0], 3)
        assert np.array_equal(table["y1"].to_numpy(), want[:, 10], equal_nan=True)
        with pytest.raises(KeyError):
            view["x0"]
        with pytest.raises(KeyError):
            view.group_stats()
    rows, counts = ds.trace(rays, limit, plan=None)
    helpers.assert_frames_match(rows.cpu().numpy().T, frame, what=f"{name} after the column plans")
    ds.close()


def test_plan_arguments_are_checked():
    fx = helpers.load("scene_config2.npz")
    ds = device_scene(helpers.scene_of(fx))
    rays = dev(fx["rays0"])
    plan = engine.RecordPlan(surfaces=(1,), rows=True, generation_limit=3)
    with pytest.raises(ValueError):
        ds.trace(rays, 10, plan=plan)                       # the trace's limit exceeds the plan's
    plan = engine.RecordPlan(surfaces=(1,), rows=True, generation_limit=10)
    with pytest.raises(ValueError):
        ds.trace(rays, 10, plan=plan, flags=engine.TRACE_UNFUSED)
    with pytest.raises(ValueError):
        engine.RecordPlan(surfaces=tuple(range(9)))
    rows, counts = ds.trace(rays, 10, plan=None)
    helpers.assert_frames_match(rows.cpu().numpy().T, fx["frame"], what="after the refused plans")
    ds.close()


def test_example_best_focus_agrees_with_the_detector_rows():
    """examples/best_focus.py: a golden-section search on the fused sums lands on a focus whose spot radius is the
    one the notebook's arithmetic gives on the detector's rows (which are the only rows the last trace kept)."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "best_focus.py")
    spec = importlib.util.spec_from_file_location("best_focus_example", path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    out = module.main(rays=20_000, verbose=False)
    assert out["surfaces"] == {out["detector"]} and out["rows"] == 20_000
    assert 45.0 < out["x"] < 115.0
    assert abs(out["radius"] - out["from_rows"]) <= 1e-9 * max(1.0, out["from_rows"]) + 1e-12
    # a focus: the spot there is far smaller than the 16 mm line of rays that went in (RMS 4.6 mm)
    assert out["radius"] < 0.5
