"""Result sink on the device: the grouped reductions of
``prt_frame_reduce`` against pandas on the *reference's own* frames (the golden fixtures), i.e.
what examples/lens_design.ipynb computes from ``tracer.trace()``'s DataFrame (cells 11-16):
rows of one surface / generation, grouped by source id, spot centroid / rms radius and the
x-axis intercept ``x0 - x_tilt * y0 / y_tilt``."""
import numpy as np
import pandas as pd
import pytest

import helpers

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

COLUMNS = ("generation", "intensity", "wavelength", "index", "id", "surface",
           "x0", "y0", "z0", "x1", "y1", "z1", "x_tilt", "y_tilt", "z_tilt")


def _frames(name):
    from pyrayt_amd.frame import DeviceFrame

    fx = helpers.load(f"scene_{name}.npz")
    golden = fx["frame"]                                    # (R, 15) float64, the reference's rows
    device = DeviceFrame(torch.from_numpy(np.ascontiguousarray(golden.T)).to("cuda:0"))
    return pd.DataFrame(golden, columns=COLUMNS), device, fx


def _pandas_stats(frame, rays_per_source):
    with np.errstate(all="ignore"):
        focus = frame["x0"] - frame["x_tilt"] * frame["y0"] / frame["y_tilt"]
    work = frame.assign(focus=focus.where(np.isfinite(focus)),
                        source_id=(frame["id"] // rays_per_source).astype(int) if rays_per_source else 0)
    out = {}
    for sid, rows in work.groupby("source_id"):
        cy, cz = rows["y1"].mean(), rows["z1"].mean()
        out[sid] = dict(count=len(rows), y=cy, z=cz,
                        rms_radius=np.sqrt(((rows["y1"] - cy) ** 2 + (rows["z1"] - cz) ** 2).mean()),
                        wavelength=rows["wavelength"].mean(), intensity=rows["intensity"].mean(),
                        focus=rows["focus"].mean(), focus_std=rows["focus"].std(ddof=0),
                        n_focus=int(rows["focus"].notna().sum()))
    return out


@pytest.mark.parametrize("name,rays_per_source", [("config4", 256), ("config3", None), ("mirrors_and_stops", 512),
                                                  ("config2", 100)])
def test_group_stats_match_pandas_on_the_reference_frame(name, rays_per_source):
    frame, device, fx = _frames(name)
    detector = float(frame["surface"].iloc[-1])             # the surface the last recorded row ended on
    for surface, generation in ((detector, None), (None, float(frame["generation"].max())), (None, None),
                                (detector, float(frame["generation"].max()))):
        sel = frame
        if surface is not None:
            sel = sel.loc[sel["surface"] == surface]
        if generation is not None:
            sel = sel.loc[sel["generation"] == generation]
        want = _pandas_stats(sel, rays_per_source)
        n_groups = int(frame["id"].max() // rays_per_source) + 1 if rays_per_source else None
        got = device.group_stats(surface=surface, generation=generation, rays_per_source=rays_per_source,
                                 n_groups=n_groups)
        assert int(got["count"].sum()) == len(sel)
        for sid, w in want.items():
            g = got.loc[sid]
            assert int(g["count"]) == w["count"]
            for key in ("y", "z", "wavelength", "intensity"):
                assert np.isclose(g[key], w[key], rtol=1e-12, atol=1e-12), (name, sid, key)
            assert np.isclose(g["rms_radius"], w["rms_radius"], rtol=1e-9, atol=1e-12), (name, sid)
            # rays without an axis intercept (y_tilt = 0: 0 / 0) are skipped, as pandas skips NaN
            if w["n_focus"] == 0:
                assert np.isnan(g["focus"]) and np.isnan(g["focus_std"]), (name, sid)
            elif abs(w["focus"]) < 1e6:
                assert np.isclose(g["focus"], w["focus"], rtol=1e-9, atol=1e-9), (name, sid, "focus")
                assert np.isclose(g["focus_std"], w["focus_std"], rtol=1e-6, atol=1e-9 * (1 + abs(w["focus"]))), (name, sid)
        empty = got.loc[got["count"] == 0]
        assert empty[["y", "z", "rms_radius"]].isna().all().all()


def test_trace_keeps_the_device_frame_and_converts_lazily():
    """RayTracer.trace() is trace_device() + to_pandas(): the block stays in HBM for the reductions."""
    import pyrayt_amd as pyrayt
    import scenes

    lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
    focus = scenes.lensmakers_equation(2, -2, 1.5, 0.25)
    sources = [pyrayt.components.ConeOfRays(cone_angle=a).move_x(-focus) for a in (2, 4, 6)]
    baffle = pyrayt.components.baffle((1, 1)).move_x(1)
    tracer = pyrayt.RayTracer(sources, [lens, baffle], rays_per_source=400, generation_limit=10)
    frame = tracer.trace()
    assert tracer.device_frame is not None and tracer.device_frame.rows.is_cuda
    assert np.array_equal(tracer.device_frame.to_numpy(), frame.to_numpy())
    stats = tracer.device_frame.group_stats(surface=baffle.get_id(), rays_per_source=400)
    tracer.calculate_source_ids()
    on_det = frame.loc[frame["surface"] == baffle.get_id()]
    for sid, rows in on_det.groupby("source_id"):
        assert int(stats.loc[sid, "count"]) == len(rows)
        assert np.isclose(stats.loc[sid, "y"], rows["y1"].mean(), atol=1e-12)
        rms = np.sqrt(((rows["y1"] - rows["y1"].mean()) ** 2 + (rows["z1"] - rows["z1"].mean()) ** 2).mean())
        assert np.isclose(stats.loc[sid, "rms_radius"], rms, rtol=1e-9)
    # a second trace of the unchanged system re-uses the compiled scene ...
    first_key, first_scene = tracer._scene_cache
    before = tracer.trace()
    assert tracer._scene_cache[1] is first_scene and tracer._scene_cache[0] == first_key
    # ... and a moved part goes into the same scene object (prt_scene_update): new tables, nothing rebuilt
    lens.move_x(0.01)
    after = tracer.trace()
    assert tracer._scene_cache[1] is first_scene and tracer._scene_cache[0] != first_key
    assert not np.array_equal(before.to_numpy(), after.to_numpy())


@pytest.mark.parametrize("name,rays_per_source", [("config4", 256), ("config3", None), ("mirrors_and_stops", 512),
                                                  ("config2", 100)])
def test_last_generation_and_mean_squares_match_pandas_on_the_reference_frame(name, rays_per_source):
    """examples/lens_design.ipynb cells 12 / 15 / 20 / 28 / 32 on the reference's own frames: "the imager's rays are the
    rows of the highest generation", the coma metric np.mean(np.square(np.sin(y_tilt) - np.sin(angle))) and the focus
    error np.mean(np.square(intercept - focus)) -- computed by pandas / numpy as the notebook writes them and by
    DeviceFrame.last_generation / mean_square (prt_frame_mean_square) on the device."""
    frame, device, fx = _frames(name)
    top = np.max(frame["generation"])
    imager_rays = frame.loc[frame["generation"] == top]                      # cells 12, 15, 20
    last = device.last_generation()
    assert np.array_equal(last.to_numpy(), imager_rays.to_numpy())
    assert device.last_generation_number() == int(top)
    counted = type(device)(device.rows, [int((frame["generation"] == g).sum()) for g in range(int(top) + 1)])
    assert np.array_equal(counted.last_generation().to_numpy(), imager_rays.to_numpy())  # the slice, no kernel
    angle = 10.0
    coma = np.mean(np.square(np.sin(imager_rays["y_tilt"]) - np.sin(angle * np.pi / 180)))       # cell 20
    got = device.mean_square("y_tilt", about=np.sin(angle * np.pi / 180), transform="sin", generation="last")
    assert np.isclose(got, coma, rtol=1e-12, atol=1e-300), (got, coma)
    assert np.isclose(last.mean_square("y_tilt", about=np.sin(angle * np.pi / 180), transform="sin"), coma, rtol=1e-12)
    with np.errstate(all="ignore"):
        intercept = -imager_rays["x_tilt"] * imager_rays["y0"] / imager_rays["y_tilt"] + imager_rays["x0"]  # cells 12, 15
    finite = intercept[np.isfinite(intercept)]
    if len(finite) and np.abs(finite).max() < 1e6:
        system_focus = float(np.median(finite))
        want = np.mean(np.square(finite - system_focus))                                       # cells 28, 32
        got = device.mean_square("axis_intercept", about=system_focus, generation="last")
        assert np.isclose(got, want, rtol=1e-9, atol=1e-18), (got, want)
        assert torch.allclose(last.axis_intercept().cpu()[np.isfinite(intercept.to_numpy())],
                              torch.from_numpy(finite.to_numpy()), rtol=1e-12, atol=1e-12)
    # grouped by source, on the detector's rows, a plain column
    if rays_per_source:
        detector = float(frame["surface"].iloc[-1])
        sel = frame.loc[frame["surface"] == detector]
        by_source = device.mean_square("y1", about=0.25, surface=detector, rays_per_source=rays_per_source)
        work = sel.assign(source_id=(sel["id"] // rays_per_source).astype(int))
        for sid, rows in work.groupby("source_id"):
            assert int(by_source.loc[sid, "count"]) == len(rows)
            assert np.isclose(by_source.loc[sid, "mean_square"], np.mean(np.square(rows["y1"] - 0.25)), rtol=1e-12, atol=1e-300)
            assert np.isclose(by_source.loc[sid, "mean"], np.mean(rows["y1"] - 0.25), rtol=1e-9, atol=1e-15)
    empty = type(device)(device.rows[:, :0])
    assert empty.last_generation_number() is None and np.isnan(empty.mean_square("y1"))


def test_the_notebooks_coma_metric_from_a_trace():
    """Cell 20 end to end on the product API: trace, take the last generation, mean square of sin(y_tilt) - sin(angle)."""
    import pyrayt_amd as pyrayt

    lens = pyrayt.components.thick_lens(r1=51.5, r2=-51.5, thickness=5, aperture=25.4, material=pyrayt.materials.glass["BK7"])
    imager = pyrayt.components.baffle((25.4, 25.4)).move_x(50)
    angle = 10.0
    source = pyrayt.components.LineOfRays(2 * 0.25 * 25.4).rotate_x(90).move_x(-10).rotate_z(angle)
    tracer = pyrayt.RayTracer(source, [lens, imager], rays_per_source=11)
    results = tracer.trace()
    ray_set = results.loc[results["generation"] == np.max(results["generation"])]
    want = np.mean(np.square((np.sin(ray_set["y_tilt"]) - np.sin(angle * np.pi / 180))))
    got = tracer.device_frame.mean_square("y_tilt", about=np.sin(angle * np.pi / 180), transform="sin", generation="last")
    assert np.isclose(got, want, rtol=1e-12)
