"""Host side of the record plans, on the CPU: the tables SinkStats builds from the fused sums against pandas on the
reference's golden frames (the sums themselves come from the numpy frame oracle here; on the GPU box
tests/test_gpu_record_plan.py checks the kernels' sums against the same oracle), the RecordPlan record, and the scaling
model of bench.py."""
import numpy as np
import pandas as pd
import pytest

import helpers
from oracle import frame_oracle
from pyrayt_amd import engine
from pyrayt_amd.frame import COLUMNS, SinkStats


def sums_of(frame, surface, limit, rays_per_source, n_groups, pivots=None, ms=None):
    """(limit, n_groups, 12): per generation the nine sums of the frame oracle and the three mean-square sums."""
    out = np.zeros((limit, n_groups, engine.SINK_STATS))
    rows = frame[frame[:, 5] == surface] if surface is not None else frame
    for g in range(limit):
        out[g, :, :9] = frame_oracle.reduce_sums(rows.T, None, float(g), rays_per_source, n_groups, pivots)
        if ms is not None:
            column, about = ms
            sel = rows[rows[:, 0] == g]
            group = np.floor(sel[:, 4] / rays_per_source).astype(int) if rays_per_source else np.zeros(len(sel), int)
            v = sel[:, COLUMNS.index(column)] - about
            for k in range(n_groups):
                vk = v[(group == k) & np.isfinite(v)]
                out[g, k, 9:] = (len(vk), vk.sum(), (vk * vk).sum())
    return out


@pytest.mark.parametrize("name", ["config2", "config3", "config4", "tutorial", "mirrors_and_stops"])
def test_tables_from_fused_sums_equal_pandas_on_the_reference_frame(name):
    fx = helpers.load(f"scene_{name}.npz")
    frame, limit = fx["frame"], int(fx["generation_limit"])
    table = pd.DataFrame(frame, columns=COLUMNS)
    last = table.loc[table["generation"] == table["generation"].max()]
    imager = float(last["surface"].mode().iloc[0])
    n = fx["rays0"].shape[1]
    rays_per_source, n_groups = max(n // 2, 1), 2
    pivots = np.array([[0.01, -0.02, 1.5], [0.0, 0.03, 0.5]])
    for piv in (None, pivots):
        stats = SinkStats(sums_of(frame, imager, limit, rays_per_source, n_groups, piv, ms=("y_tilt", 0.02)), piv)
        for generation in (None, "last", 0, limit - 1):
            rows = table.loc[table["surface"] == imager]
            if generation == "last":
                number = int(rows["generation"].max()) if len(rows) else None
                assert stats.last_generation_number() == number
                rows = rows.loc[rows["generation"] == number] if number is not None else rows
            elif generation is not None:
                rows = rows.loc[rows["generation"] == generation]
            got = stats.group_stats(generation)
            got_ms = stats.mean_square(generation, per_source=True)
            source = (rows["id"] // rays_per_source).astype(int)
            for k in range(n_groups):
                part = rows.loc[source == k]
                assert int(got["count"].iloc[k]) == len(part)
                if len(part) == 0:
                    assert np.isnan(got["y"].iloc[k]) and np.isnan(got["rms_radius"].iloc[k])
                    continue
                y, z = part["y1"].to_numpy(), part["z1"].to_numpy()
                assert np.isclose(got["y"].iloc[k], y.mean(), rtol=1e-10, atol=1e-12)
                assert np.isclose(got["z"].iloc[k], z.mean(), rtol=1e-10, atol=1e-12)
                radius = np.sqrt(np.mean((y - y.mean()) ** 2 + (z - z.mean()) ** 2))
                assert np.isclose(got["rms_radius"].iloc[k], radius, rtol=1e-8, atol=1e-11)
                with np.errstate(all="ignore"):
                    focus = (part["x0"] - part["x_tilt"] * part["y0"] / part["y_tilt"]).to_numpy()
                focus = focus[np.isfinite(focus)]
                if len(focus):
                    assert np.isclose(got["focus"].iloc[k], focus.mean(), rtol=1e-9, atol=1e-10)
                    assert np.isclose(got["focus_std"].iloc[k], focus.std(), rtol=1e-6, atol=1e-9)
                assert np.isclose(got["wavelength"].iloc[k], part["wavelength"].mean(), rtol=1e-12)
                v = part["y_tilt"].to_numpy() - 0.02
                assert int(got_ms["count"].iloc[k]) == len(v)
                assert np.isclose(got_ms["mean_square"].iloc[k], np.mean(v * v), rtol=1e-10, atol=1e-300)


def test_record_plan_record_and_its_checks():
    plan = engine.RecordPlan(surfaces=(3, 7), rows=False, stats=False, generation_limit=6)
    assert plan.key() != engine.RecordPlan(surfaces=(3,), rows=False, generation_limit=6).key()
    with pytest.raises(ValueError):
        engine.RecordPlan(surfaces=tuple(range(9)))
    with pytest.raises(ValueError):
        engine.RecordPlan(n_groups=3)                       # groups need rays_per_source
    with pytest.raises(ValueError):
        engine.RecordPlan(mean_square=("no_such_column", 0.0, None))
    ms = engine.RecordPlan(mean_square=("axis_intercept", 1.25, "sin"))
    assert ms.ms == (engine.AXIS_INTERCEPT, 1.25, 1)

    class FakeTorch:  # (record() only allocates when the plan sums: a plan without stats needs no device)
        pass

    rec = plan.record(FakeTorch, None)
    assert rec["struct_size"][0] == engine.PLAN_DTYPE.itemsize and rec["n_surfaces"][0] == 2
    assert rec["surfaces"][0, :2].tolist() == [3, 7] and rec["store_rows"][0] == 0 and rec["n_groups"][0] == 0
    assert rec["ms_quantity"][0] == -1 and rec["generation_limit"][0] == 6


def test_scaling_model_of_the_bench_line():
    import bench

    one = bench.scaling_model(1_000_000, 1, 3.0e6)
    assert one["shard_rays"] == 1_000_000 and one["overlapped"]["speedup_over_1_gpu"] == 1.0
    assert one["gather"]["bytes_into_each_gpu"] == 0.0
    last = one
    for world in (2, 4, 8):
        model = bench.scaling_model(1_000_000, world, 3.0e6)
        assert model["shard_rays"] == 1_000_000 // world
        assert model["overlapped"]["ms_per_step"] < last["overlapped"]["ms_per_step"]
        assert 1.0 < model["overlapped"]["speedup_over_1_gpu"] < world            # sub-linear: small shards are latency-bound
        assert model["synchronous"]["speedup_over_1_gpu"] < model["overlapped"]["speedup_over_1_gpu"]
        assert np.isclose(model["gather"]["bytes_into_each_gpu"], 3.0e6 * 120 * (world - 1) / world)
        ring, direct = model["gather"]["ring_ms_one_link"], model["gather"]["direct_ms_all_links"]
        assert ring[0] < ring[1] and direct[0] < direct[1] and np.isclose(ring[0], 7 * direct[0])
        last = model
    between = bench.scaling_model(1_000_000, 3, 3.0e6)                               # a shard size the table does not hold
    assert bench.scaling_model(1_000_000, 4, 3.0e6)["overlapped"]["ms_per_step"] < between["overlapped"]["ms_per_step"] \
        < bench.scaling_model(1_000_000, 2, 3.0e6)["overlapped"]["ms_per_step"]
