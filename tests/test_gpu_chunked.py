"""Chunked traces (include/prt.h "Chunked traces", round 6): a repeated trace whose generations all run dense is issued as
two chunks on two streams, and must give the rows of the one-chain trace, bit for bit, at the same places.

Every golden scene is traced several times with chunking forced on (scene option chunks = 2: any trace of two tiles or
more whose hints qualify): the first trace of a scene runs as one chain (no hints yet), the later ones as two chunks
wherever the scene's generations are dense -- `trace_stats()["variant"]` says which it was -- and every frame is the
reference's.  Then the north-star size, rotating ray sets (a hint that does not hold in one chunk repeats the trace as
one chain), traces in flight together, and the interplay with scene updates and record plans."""
import numpy as np
import pytest

import helpers
import scenes
from pyrayt_amd import engine

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SCENE_FIXTURES = ["config1", "config2", "config3", "config4", "config5", "two_mirrors", "tutorial", "mirrors_and_stops",
                  "stopped_lens", "adv_lens", "adv_stop", "adv_prism", "adv_condenser", "adv_still", "adv_short_a",
                  "adv_short_b", "adv_short_c", "adv_bench_a", "adv_bench_b", "adv_bench_c", "stale_box"]
CHUNKED = 4  # PRT_VARIANT_CHUNKED


def dev(array):
    return torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64)).to("cuda:0")


def device_scene(scene_dict, **options):
    return engine.DeviceScene(helpers.FixtureSnapshot(scene_dict), options=options)


@pytest.mark.parametrize("name", SCENE_FIXTURES)
@pytest.mark.parametrize("flags", [0, engine.TRACE_KEEP_ABSORBED])
def test_chunked_traces_of_the_golden_scenes_equal_the_reference(name, flags):
    fx = helpers.load(f"scene_{name}.npz")
    limit = int(fx["generation_limit"])
    ds = device_scene(helpers.scene_of(fx), chunks=2)
    rays = dev(fx["rays0"])
    variants = []
    for attempt in range(5):
        rows, counts = ds.trace(rays, limit, flags=flags)
        helpers.assert_frames_match(rows.cpu().numpy().T, fx["frame"], what=f"{name} attempt {attempt}")
        assert sum(counts) == fx["frame"].shape[0]
        variants.append(ds.trace_stats()["variant"])
    assert variants[0] == 1                       # a first trace has no hints: one chain
    # ... and the same with chunking off gives the same rows (the A/B partner of everything above)
    plain = device_scene(helpers.scene_of(fx), chunks=1)
    for attempt in range(3):
        rows, counts = plain.trace(rays, limit, flags=flags)
        helpers.assert_frames_match(rows.cpu().numpy().T, fx["frame"], what=f"{name} unchunked {attempt}")
        assert plain.trace_stats()["variant"] == 1
    plain.close()
    ds.close()


def test_the_baseline_scenes_do_run_chunked():
    """The scenes the chunks are for -- one lens / prism / condenser and a detector: every generation dense -- run as
    two chunks from their second or third trace on (the sparse-loss forms need one more trace to settle)."""
    for name in ("config1", "config2", "config4", "config5"):
        fx = helpers.load(f"scene_{name}.npz")
        ds = device_scene(helpers.scene_of(fx), chunks=2)
        rays = dev(fx["rays0"])
        variants = []
        for _ in range(6):
            ds.trace(rays, int(fx["generation_limit"]))
            variants.append(ds.trace_stats()["variant"])
        assert CHUNKED in variants[1:], (name, variants)
        assert variants[-1] == CHUNKED, (name, variants)
        ds.close()


def test_one_million_rays_chunked_equal_the_one_chain_trace():
    """BASELINE config 2 at 1M rays, the default options: from the third trace on a blocking trace() runs as two
    chunks; its rows are those of the one-chain trace, rotating ray sets included (a ray set whose near-axial rays sit
    elsewhere refutes nothing: the forms a chunked trace runs on cover it)."""
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    sets = []
    for seed in (1234, 1235, 1236):
        CountedObject.reset_ids()
        parts, rays = scenes.config2(scenes.product_api(), 1_000_000, seed=seed)
        sets.append(dev(rays))
    plain = engine.DeviceScene(SceneSnapshot(parts), options={"chunks": 1})
    want = []
    for rays in sets:
        rows, counts = plain.trace(rays, 10)
        want.append((rows.clone(), counts))
    plain.close()
    ds = engine.DeviceScene(SceneSnapshot(parts))
    chunked = 0
    for k in range(12):
        rows, counts = ds.trace(sets[k % 3], 10)
        assert counts == want[k % 3][1], (k, counts)
        assert torch.equal(rows, want[k % 3][0]), k
        chunked += ds.trace_stats()["variant"] == CHUNKED
    assert chunked >= 8, chunked
    assert ds.telemetry()["speculation_misses"] <= 1
    # a per-call opt-out, and the scene keeps tracing as one chain while another trace is in flight
    rows, counts = ds.trace(sets[0], 10, flags=engine.TRACE_NO_CHUNKS)
    assert ds.trace_stats()["variant"] == 1 and torch.equal(rows, want[0][0])
    ds.close()


def test_a_ray_set_that_refutes_a_chunk_is_traced_again_as_one_chain():
    """Hints learnt on one ray set, then a ray set that loses rays where the first lost none: a tile of one chunk
    refutes its dense form, the library repeats the trace as one chain, the rows are right, and the ticket goes on."""
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), 300_000, seed=3)
    ds = engine.DeviceScene(SceneSnapshot(parts))
    plain = engine.DeviceScene(SceneSnapshot(parts), options={"chunks": 1})
    inside = dev(rays)
    # the same rays with a stretch of the SECOND half aimed past the lens: they miss everything in generation 0
    stray = rays.copy()
    stray[4:7, 200_000:200_300] = np.array([[0.0], [1.0], [0.0]])
    stray = dev(stray)
    for _ in range(4):
        ds.trace(inside, 10)
    assert ds.trace_stats()["variant"] == CHUNKED
    misses = ds.telemetry()["speculation_misses"]
    rows, counts = ds.trace(stray, 10)
    want, want_counts = plain.trace(stray, 10)
    assert counts == want_counts and torch.equal(rows, want)
    assert ds.telemetry()["speculation_misses"] == misses + 1 and ds.trace_stats()["variant"] == 1
    rows, counts = ds.trace(inside, 10)
    want, want_counts = plain.trace(inside, 10)
    assert counts == want_counts and torch.equal(rows, want)
    ds.close()
    plain.close()


def test_chunked_traces_on_several_tickets_and_around_scene_updates():
    """chunks = 2 lets traces in flight together run chunked as well (four chains on four streams), and a scene
    update between chunked traces is ordered in front of BOTH chunks of the next one."""
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), 120_000, seed=9)
    device_rays = dev(rays)
    ds = engine.DeviceScene(SceneSnapshot(parts), options={"chunks": 2})
    streams = ds.ticket_streams(device_rays.device, 2)
    blocks = [torch.empty((15, rays.shape[1] * 10), dtype=torch.float64, device="cuda:0") for _ in range(2)]
    for step in range(8):
        parts[1].move_x(0.01)
        assert ds.update(SceneSnapshot(parts))
        reference = engine.DeviceScene(SceneSnapshot(parts), options={"chunks": 1})
        want, want_counts = reference.trace(device_rays, 10)
        for ticket in range(2):
            streams[ticket].wait_stream(torch.cuda.current_stream())
            ds.trace_begin(ticket, device_rays, 10, blocks[ticket], stream=streams[ticket])
        for ticket in range(2):
            rows, counts = ds.trace_end(ticket)
            torch.cuda.current_stream().wait_stream(streams[ticket])
            assert counts == want_counts and torch.equal(rows, want), (step, ticket)
        reference.close()
    assert ds.trace_stats()["variant"] == CHUNKED
    # a record plan takes the ticket off the chunked path and puts it back when it is removed
    plan = engine.RecordPlan(surfaces=(parts[1].get_id(),), rows=True, generation_limit=10)
    rows, counts = ds.trace(device_rays, 10, plan=plan)
    assert ds.trace_stats()["variant"] == 1 and counts[-1] == rows.shape[1] - sum(counts[:-1])
    for _ in range(3):
        rows, counts = ds.trace(device_rays, 10, plan=None)
    assert ds.trace_stats()["variant"] == CHUNKED and torch.equal(rows, want)
    ds.close()
