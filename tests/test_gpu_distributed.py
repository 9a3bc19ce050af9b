"""Multi-rank path on the GPU: real HIP traces per shard, frames re-assembled by the library.

A gpurun box has one GPU, and RCCL refuses two ranks on one device, so the two-rank tests let the
blocks travel through a gloo group and order them with the library's placement kernel
(``prt_place_rows`` -- the same kernel ``prt_allgather_rows`` runs behind its RCCL all-gathers);
the RCCL leg itself (communicator bootstrap, count all-gather, fifteen grouped all-gathers) runs
with a one-rank communicator.  The assembled frame must equal the single-rank frame bit for bit,
row order included (pyrayt/_pyrayt.py:168-186)."""
import os
import socket
import time

import numpy as np
import pytest
import torch

import scenes

pytestmark = pytest.mark.gpu

LIMIT = 10


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene_and_rays(name, n):
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.SCENES[name](scenes.product_api(), n)
    return SceneSnapshot(parts), rays


def _worker(rank, world, port, name, n, mode, result_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyrayt_amd import distributed as pdist
        from pyrayt_amd import engine

        torch.cuda.set_device(0)
        snap, rays = _scene_and_rays(name, n)
        group = pdist.resolve_group(None)
        lo, hi = pdist.shard_bounds(rays.shape[1], group)
        scene = engine.DeviceScene(snap)
        shard = torch.from_numpy(np.ascontiguousarray(rays[:, lo:hi])).to("cuda:0")
        rows, counts = scene.trace(shard, LIMIT)
        assert scene.trace_stats()["variant"] == 1
        full, full_counts = pdist.assemble_rows(rows, counts, LIMIT, group, mode)
        assert full.is_cuda
        np.save(os.path.join(result_dir, f"rows_{rank}.npy"), full.cpu().numpy())
        np.save(os.path.join(result_dir, f"counts_{rank}.npy"), np.array(full_counts))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,n,world,mode", [("config2", 20011, 2, "all"), ("stopped_lens", 6007, 3, "all"),
                                               ("config3", 9001, 2, "root")])
def test_sharded_hip_trace_assembles_to_the_single_rank_frame(tmp_path, name, n, world, mode):
    import torch.multiprocessing as mp

    from pyrayt_amd import engine

    mp.start_processes(_worker, args=(world, _free_port(), name, n, mode, str(tmp_path)), nprocs=world,
                       join=True, start_method="spawn")
    snap, rays = _scene_and_rays(name, n)
    rows, counts = engine.DeviceScene(snap).trace(torch.from_numpy(rays).to("cuda:0"), LIMIT)
    want = rows.cpu().numpy()
    for rank in range(world):
        got = np.load(tmp_path / f"rows_{rank}.npy")
        assert np.load(tmp_path / f"counts_{rank}.npy").tolist() == counts
        if mode == "all" or rank == 0:
            assert got.shape == want.shape
            assert np.array_equal(got, want, equal_nan=True)  # sharded == unsharded, same row order
        else:
            assert got.shape[1] == 0


def test_rccl_leg_with_a_one_rank_communicator():
    """prt_comm_create / prt_allgather_counts / prt_allgather_rows over RCCL itself."""
    from pyrayt_amd import distributed as pdist
    from pyrayt_amd import engine

    snap, rays = _scene_and_rays("config2", 5000)
    rows, counts = engine.DeviceScene(snap).trace(torch.from_numpy(rays).to("cuda:0"), LIMIT)
    comm = pdist.LibraryComm(0, 1, 0, pdist.LibraryComm.unique_id())
    try:
        matrix = comm.gather_counts(counts, LIMIT)
        assert matrix.shape == (1, LIMIT) and matrix[0, : len(counts)].tolist() == counts
        out, merged = pdist.assemble_rows(rows, counts, LIMIT, None, "all", comm=comm)
        assert merged == counts
        assert torch.equal(out, rows)
        # two frames a caller keeps do not alias (as on one GPU); a loop may ask for the communicator's own block
        again, _ = pdist.assemble_rows(rows, counts, LIMIT, None, "all", comm=comm)
        assert again.data_ptr() != out.data_ptr() and torch.equal(again, rows)
        kept, _ = pdist.assemble_rows(rows, counts, LIMIT, None, "all", comm=comm, reuse=True)
        kept_again, _ = pdist.assemble_rows(rows, counts, LIMIT, None, "all", comm=comm, reuse=True)
        assert kept.data_ptr() == kept_again.data_ptr() and torch.equal(kept_again, rows)
    finally:
        comm.close()


def test_communicator_reports_what_rccl_sees_and_gives_up_on_a_missing_rank():
    """prt_comm_info = ncclCommCount / ncclCommUserRank (bench.py prints it as config.rccl_ranks); a communicator whose
    other rank never arrives ends with a TimeoutError that says so, not with a hang (the attempt stays blocked on a helper
    thread, so this runs in a process of its own that exits right away)."""
    import subprocess
    import sys

    from pyrayt_amd import distributed as pdist

    comm = pdist.LibraryComm(0, 1, 0, pdist.LibraryComm.unique_id())
    try:
        assert comm.info() == {"ranks": 1, "rank": 0, "device": 0}
    finally:
        comm.close()
    code = ("import os, sys, torch\n"
            "from pyrayt_amd import distributed as pdist\n"
            "try:\n"
            "    pdist.LibraryComm(0, 2, 0, pdist.LibraryComm.unique_id(), timeout=5)\n"
            "    print('created')\n"
            "except TimeoutError as exc:\n"
            "    print('timeout:', 'ncclCommInitRank did not return' in str(exc))\n"
            "sys.stdout.flush(); os._exit(0)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    done = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=300)
    assert done.returncode == 0 and "timeout: True" in done.stdout, (done.stdout, done.stderr[-1500:])


def test_gather_pipelined_behind_the_next_trace_gives_the_same_frames():
    """pyrayt_amd.distributed.trace_and_gather: frame k is re-assembled on a communication stream while trace k + 1
    runs.  With a one-rank RCCL communicator every assembled frame must be the trace's own rows, for ray sets that
    differ from step to step (two frames are handed out from two blocks in turn)."""
    from pyrayt_amd import distributed as pdist
    from pyrayt_amd import engine
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    sets = []
    for seed in range(7):
        CountedObject.reset_ids()
        parts, rays = scenes.config2(scenes.product_api(), 20_000 + 1000 * (seed % 3), seed=40 + seed)
        sets.append(torch.from_numpy(rays).to("cuda:0"))
    scene = engine.DeviceScene(SceneSnapshot(parts))
    want = []
    for rays in sets:
        rows, counts = scene.trace(rays, LIMIT)
        want.append((rows.cpu().numpy().copy(), counts))
    comm = pdist.LibraryComm(0, 1, 0, pdist.LibraryComm.unique_id())
    try:
        for depth in (1, 2, 3):
            got = 0
            for k, (frame, counts) in enumerate(pdist.trace_and_gather(scene, iter(sets), LIMIT, comm, depth=depth)):
                assert counts == want[k][1], (depth, k)
                assert np.array_equal(frame.cpu().numpy(), want[k][0]), (depth, k)
                got += 1
            assert got == len(sets)
    finally:
        comm.close()


def test_gather_pipelined_abandoned_midway_frees_its_tickets():
    """A consumer that stops iterating trace_and_gather (or an error inside it) must not leave traces in flight: the
    generator's finally collects them, so the tickets are free for the next trace of the scene and the record blocks
    are not released under running kernels (mirrors test_trace_many_abandoned_midway_frees_its_tickets)."""
    from pyrayt_amd import distributed as pdist
    from pyrayt_amd import engine
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, rays = scenes.config2(scenes.product_api(), 30_000, seed=77)
    rays = torch.from_numpy(rays).to("cuda:0")
    scene = engine.DeviceScene(SceneSnapshot(parts))
    want_rows, want_counts = scene.trace(rays, LIMIT)
    want = want_rows.cpu().numpy().copy()
    comm = pdist.LibraryComm(0, 1, 0, pdist.LibraryComm.unique_id())
    try:
        for depth in (2, 3):
            stream = pdist.trace_and_gather(scene, (rays for _ in range(8)), LIMIT, comm, depth=depth)
            frame, counts = next(stream)
            assert counts == want_counts and np.array_equal(frame.cpu().numpy(), want)
            stream.close()                               # depth traces were in flight: collected by the generator
            for ticket in range(depth):                  # every ticket is free again ...
                out = torch.empty((15, rays.shape[1] * LIMIT), dtype=torch.float64, device="cuda:0")
                scene.trace_begin(ticket, rays, LIMIT, out)
                rows, counts = scene.trace_end(ticket)
                torch.cuda.synchronize()
                assert counts == want_counts and np.array_equal(rows.cpu().numpy(), want)
            rows, counts = scene.trace(rays, LIMIT)      # ... and so is the blocking call

        # an error raised inside the loop (the gather of frame 1 fails) also leaves no ticket behind
        calls = {"n": 0}
        real = comm.gather_rows

        def failing(*args, **kwargs):
            calls["n"] += 1
            if calls["n"] == 2:
                raise RuntimeError("injected")
            return real(*args, **kwargs)

        comm.gather_rows = failing
        with pytest.raises(RuntimeError, match="injected"):
            for _ in pdist.trace_and_gather(scene, (rays for _ in range(6)), LIMIT, comm, depth=2):
                pass
        comm.gather_rows = real
        rows, counts = scene.trace(rays, LIMIT)
        assert counts == want_counts and np.array_equal(rows.cpu().numpy(), want)
    finally:
        comm.close()
    scene.close()


def test_placement_kernel_against_the_indexed_copy():
    """prt_place_rows == the torch placement on a ragged count matrix with empty segments."""
    from pyrayt_amd import distributed as pdist

    matrix = torch.tensor([[5, 0, 3, 0], [0, 0, 4, 1], [2, 7, 0, 0]])
    world, limit = matrix.shape
    widest = int(matrix.sum(dim=1).max())
    gen = torch.Generator().manual_seed(3)
    blocks = torch.rand((world, 15, widest), generator=gen, dtype=torch.float64)
    _, _, total = pdist.placement(matrix)
    want = pdist._place_with_torch(list(blocks.unbind(0)), matrix, total)
    got = pdist._place_on_device(blocks.to("cuda:0"), matrix, limit, total)
    assert torch.equal(got.cpu(), want)


# ---- real multi-rank RCCL: switches itself on when the box shows two or more devices -----------------
def _rccl_worker(rank, world, port, name, n, mode, result_dir):
    """One rank per device, backend nccl (= RCCL): real HIP trace of the shard, then the library's own
    exchange -- prt_allgather_counts + prt_allgather_rows over a communicator bootstrapped through the
    torch.distributed group (LibraryComm.from_group)."""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    import datetime

    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device,
                            timeout=datetime.timedelta(seconds=120))
    try:
        from pyrayt_amd import distributed as pdist
        from pyrayt_amd import engine

        snap, rays = _scene_and_rays(name, n)
        group = pdist.resolve_group(None)
        lo, hi = pdist.shard_bounds(rays.shape[1], group)
        scene = engine.DeviceScene(snap)
        shard = torch.from_numpy(np.ascontiguousarray(rays[:, lo:hi])).to(device)
        rows, counts = scene.trace(shard, LIMIT)
        comm = pdist.LibraryComm.from_group(group, device)
        try:
            for repeat in range(2):  # the second pass runs on the communicator's reused buffers
                full, full_counts = pdist.assemble_rows(rows, counts, LIMIT, group, mode, comm=comm, reuse=True)
            assert full.is_cuda and full.device == device
            torch.cuda.synchronize(device)
        finally:
            comm.close()
        np.save(os.path.join(result_dir, f"rows_{rank}.npy"), full.cpu().numpy())
        np.save(os.path.join(result_dir, f"counts_{rank}.npy"), np.array(full_counts))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two or more GPUs (one RCCL rank per device)")
@pytest.mark.parametrize("name,n,mode", [("config2", 20011, "all"), ("config3", 9001, "root"),
                                         ("stopped_lens", 6007, "all")])
def test_rccl_allgather_rows_across_real_devices(tmp_path, name, n, mode):
    """The first execution of prt_allgather_rows with more than one rank: one rank per device of the box, assembled frame bit-identical to the single-rank frame on every rank ("all") or on
    rank 0 ("root")."""
    import torch.multiprocessing as mp

    from pyrayt_amd import engine

    world = torch.cuda.device_count()  # every device of the box: the BASELINE curve ends at 8
    ctx = mp.start_processes(_rccl_worker, args=(world, _free_port(), name, n, mode, str(tmp_path)), nprocs=world,
                             join=False, start_method="spawn")
    deadline = time.time() + 300  # (a rank that dies leaves the others waiting in a collective: do not hang the suite)
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for proc in ctx.processes:
                proc.terminate()
            pytest.fail("the RCCL ranks did not finish within 300 s")
    snap, rays = _scene_and_rays(name, n)
    rows, counts = engine.DeviceScene(snap).trace(torch.from_numpy(rays).to("cuda:0"), LIMIT)
    want = rows.cpu().numpy()
    for rank in range(world):
        got = np.load(tmp_path / f"rows_{rank}.npy")
        assert np.load(tmp_path / f"counts_{rank}.npy").tolist() == counts
        if mode == "all" or rank == 0:
            assert got.shape == want.shape
            assert np.array_equal(got, want, equal_nan=True)
        else:
            assert got.shape[1] == 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two or more GPUs")
def test_bench_launches_its_own_ranks_on_real_devices():
    """`python bench.py --gpus 2` with no launcher in front of it: one JSON line, n_gpus 2, gather over RCCL."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    done = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                           "--rays", "100000", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, done.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert "ms" in line["gather"] and "RCCL" in line["gather"]["transport"], line["gather"]


def test_bench_launches_its_own_ranks_sharing_one_gpu():
    """The same without a second device: the ranks share GPU 0 and talk over gloo (plumbing check of the
    launcher + sharded bench path; RCCL refuses two ranks per device)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PRT_DIST_BACKEND"] = "gloo"
    done = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                           "--rays", "100000", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, done.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["rays_per_gpu"] == 50000
    assert "ms" in line["gather"], line["gather"]


def test_bench_checks_the_shards_against_the_reference_before_it_times_them():
    """The north-star job itself (1M rays) on two ranks sharing GPU 0 over gloo: before anything is timed every ray set's
    frame is re-assembled on rank 0 and compared with the reference's summary of that run (all four seeds); the line
    carries the verdict.  (RCCL refuses two ranks per device; the re-assembly then goes through torch.distributed + the
    placement kernel -- the checks are the same.)"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PRT_DIST_BACKEND"] = "gloo"
    done = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                           "--reps", "2", "--side-steps", "0", "--no-cpu-baseline"], capture_output=True, text=True, env=env,
                          timeout=900)
    assert done.returncode == 0, done.stderr[-2000:]
    line = json.loads([ln for ln in done.stdout.splitlines() if ln.strip()][-1])
    assert line["n_gpus"] == 2 and line["verified"] is True, line.get("verification")
    assert line["verification"]["seeds"] == [1234, 1235, 1236, 1237]
    assert "BEFORE the timed region" in line["verification"]["of"]
    assert line["config"]["rays_per_gpu"] == 500000 and line["repetitions"]["n"] == 2


# ---- sharded result sink: statistics of the whole frame from the rows every rank kept ----------------------------
def _stats_of(frame, detector, rays_per_source, **how):
    return frame.group_stats(surface=detector, rays_per_source=rays_per_source, **how)[
        ["count", "y", "z", "rms_radius", "focus", "focus_std", "wavelength", "intensity"]].to_numpy(dtype=float)


def _stats_worker(rank, world, port, name, n, rays_per_source, result_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyrayt_amd import distributed as pdist
        from pyrayt_amd import engine
        from pyrayt_amd.frame import DeviceFrame

        torch.cuda.set_device(0)
        snap, rays = _scene_and_rays(name, n)
        group = pdist.resolve_group(None)
        lo, hi = pdist.shard_bounds(rays.shape[1], group)
        scene = engine.DeviceScene(snap)
        rows, counts = scene.trace(torch.from_numpy(np.ascontiguousarray(rays[:, lo:hi])).to("cuda:0"), LIMIT)
        detector = float(snap.prims["surface_id"][-1])
        got = _stats_of(DeviceFrame(rows, counts), detector, rays_per_source, group=group)  # the rows stay where they are
        np.save(os.path.join(result_dir, f"stats_{rank}.npy"), got)
        # the notebook's merit functions on the sharded frame: mean squares of the whole frame from every rank's rows
        frame = DeviceFrame(rows, counts)
        coma = frame.mean_square("y_tilt", about=0.05, transform="sin", generation="last", group=group)
        per_source = frame.mean_square("axis_intercept", about=1.0, surface=detector, rays_per_source=rays_per_source,
                                       group=group) if rays_per_source else None
        np.save(os.path.join(result_dir, f"ms_{rank}.npy"),
                np.concatenate([[coma], per_source.to_numpy(dtype=float).ravel() if per_source is not None else []]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,n,rays_per_source,world", [("config2", 20011, 2500, 2), ("config3", 9001, None, 3),
                                                          ("config2", 40000, 5000, 8)])
def test_sharded_group_stats_on_the_gpu(tmp_path, name, n, rays_per_source, world):
    """2, 3 and 8 ranks sharing the one GPU (gloo adds the sums): every rank gets the single-rank statistics."""
    import torch.multiprocessing as mp

    from pyrayt_amd import engine
    from pyrayt_amd.frame import DeviceFrame

    mp.start_processes(_stats_worker, args=(world, _free_port(), name, n, rays_per_source, str(tmp_path)), nprocs=world,
                       join=True, start_method="spawn")
    snap, rays = _scene_and_rays(name, n)
    rows, counts = engine.DeviceScene(snap).trace(torch.from_numpy(rays).to("cuda:0"), LIMIT)
    want = _stats_of(DeviceFrame(rows, counts), float(snap.prims["surface_id"][-1]), rays_per_source)
    assert np.isfinite(want[:, 1]).any()
    whole = DeviceFrame(rows, counts)
    detector = float(snap.prims["surface_id"][-1])
    want_ms = [whole.mean_square("y_tilt", about=0.05, transform="sin", generation="last")]
    if rays_per_source:
        want_ms += whole.mean_square("axis_intercept", about=1.0, surface=detector,
                                     rays_per_source=rays_per_source).to_numpy(dtype=float).ravel().tolist()
    for rank in range(world):
        got = np.load(tmp_path / f"stats_{rank}.npy")
        assert got.shape == want.shape and np.array_equal(got[:, 0], want[:, 0])
        assert np.allclose(got, want, rtol=0, atol=1e-12, equal_nan=True), (rank, np.nanmax(np.abs(got - want)))
        got_ms = np.load(tmp_path / f"ms_{rank}.npy")
        assert np.allclose(got_ms, want_ms, rtol=1e-10, atol=1e-300, equal_nan=True), (rank, got_ms, want_ms)


def _plan_worker(rank, world, port, n, result_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pyrayt_amd as pyrayt

        torch.cuda.set_device(0)
        pyrayt.g3d.objects.CountedObject.reset_ids()
        lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
        sources = [pyrayt.components.ConeOfRays(cone_angle=6, wavelength=w).move_x(-1.9) for w in (0.5, 0.633)]
        det = pyrayt.components.baffle((1, 1)).move_x(1)
        tracer = pyrayt.RayTracer(sources, [lens, det], rays_per_source=n)          # the world group, gather "all"
        stats = tracer.trace_stats(surface=det, rays_per_source=True, mean_square=("axis_intercept", 1.0, None))
        table = stats.group_stats("last")
        np.save(os.path.join(result_dir, f"table_{rank}.npy"), table.to_numpy(dtype=float))
        np.save(os.path.join(result_dir, f"ms_{rank}.npy"), stats.mean_square(None, per_source=True).to_numpy(dtype=float))
        tracer.record_only(det)
        frame = tracer.trace()                                                        # every rank: the whole cut
        np.save(os.path.join(result_dir, f"rows_{rank}.npy"), frame.to_numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_record_plans_of_a_sharded_tracer(tmp_path, world):
    """RayTracer.trace_stats / record_only with the rays sharded over ranks (gloo, sharing the one GPU): the sums of the
    shards add up to the single-rank tables, and the filtered frames re-assemble to the single-rank cut."""
    import torch.multiprocessing as mp

    import pyrayt_amd as pyrayt

    n = 6007
    mp.start_processes(_plan_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    pyrayt.g3d.objects.CountedObject.reset_ids()
    lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
    sources = [pyrayt.components.ConeOfRays(cone_angle=6, wavelength=w).move_x(-1.9) for w in (0.5, 0.633)]
    det = pyrayt.components.baffle((1, 1)).move_x(1)
    tracer = pyrayt.RayTracer(sources, [lens, det], rays_per_source=n)
    whole = tracer.trace()
    frame = tracer.device_frame
    want_table = frame.group_stats(surface=det.get_id(), generation=frame.last_generation_number(), rays_per_source=n,
                                   n_groups=2).to_numpy(dtype=float)
    want_ms = frame.mean_square("axis_intercept", about=1.0, surface=det.get_id(), rays_per_source=n,
                                n_groups=2).to_numpy(dtype=float)
    want_rows = whole.loc[whole["surface"] == det.get_id()].to_numpy()
    for rank in range(world):
        assert np.allclose(np.load(tmp_path / f"table_{rank}.npy"), want_table, rtol=1e-9, atol=1e-12, equal_nan=True), rank
        assert np.allclose(np.load(tmp_path / f"ms_{rank}.npy"), want_ms, rtol=1e-10, atol=1e-300, equal_nan=True), rank
        assert np.array_equal(np.load(tmp_path / f"rows_{rank}.npy"), want_rows, equal_nan=True), rank


def test_sharded_group_stats_over_a_one_rank_rccl_communicator():
    """prt_frame_stats_sharded itself (two ncclAllReduce inside the library) with the one rank a 1-GPU box allows."""
    from pyrayt_amd import distributed as pdist
    from pyrayt_amd import engine
    from pyrayt_amd.frame import DeviceFrame

    snap, rays = _scene_and_rays("config2", 30000)
    rows, counts = engine.DeviceScene(snap).trace(torch.from_numpy(rays).to("cuda:0"), LIMIT)
    frame = DeviceFrame(rows, counts)
    detector = float(snap.prims["surface_id"][-1])
    comm = pdist.LibraryComm(0, 1, 0, pdist.LibraryComm.unique_id())
    try:
        want = _stats_of(frame, detector, 3000)
        got = _stats_of(frame, detector, 3000, comm=comm, n_groups=10)
        # (the reduction adds with atomics: two runs agree to rounding, not bit for bit)
        assert np.array_equal(got[:, 0], want[:, 0]) and np.allclose(got, want, rtol=0, atol=1e-12, equal_nan=True)
        with pytest.raises(ValueError):
            frame.group_stats(surface=detector, rays_per_source=3000, comm=comm)  # how many groups is the caller's to say
    finally:
        comm.close()


def _rccl_stats_worker(rank, world, port, name, n, rays_per_source, result_dir):
    import datetime

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=datetime.timedelta(seconds=120))
    try:
        from pyrayt_amd import distributed as pdist
        from pyrayt_amd import engine
        from pyrayt_amd.frame import DeviceFrame

        snap, rays = _scene_and_rays(name, n)
        group = pdist.resolve_group(None)
        lo, hi = pdist.shard_bounds(rays.shape[1], group)
        rows, counts = engine.DeviceScene(snap).trace(torch.from_numpy(np.ascontiguousarray(rays[:, lo:hi])).to(device), LIMIT)
        detector = float(snap.prims["surface_id"][-1])
        comm = pdist.LibraryComm.from_group(group, device)
        try:
            got = _stats_of(DeviceFrame(rows, counts), detector, rays_per_source, group=group, comm=comm)
        finally:
            comm.close()
        np.save(os.path.join(result_dir, f"stats_{rank}.npy"), got)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two or more GPUs (one RCCL rank per device)")
def test_rccl_sharded_group_stats_across_real_devices(tmp_path):
    """prt_frame_stats_sharded over every device of the box (one rank each): all ranks get the single-rank statistics."""
    import torch.multiprocessing as mp

    from pyrayt_amd import engine
    from pyrayt_amd.frame import DeviceFrame

    world = torch.cuda.device_count()
    name, n, rays_per_source = "config2", 40000, 5000
    ctx = mp.start_processes(_rccl_stats_worker, args=(world, _free_port(), name, n, rays_per_source, str(tmp_path)),
                             nprocs=world, join=False, start_method="spawn")
    deadline = time.time() + 300
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for proc in ctx.processes:
                proc.terminate()
            pytest.fail("the RCCL ranks did not finish within 300 s")
    snap, rays = _scene_and_rays(name, n)
    rows, counts = engine.DeviceScene(snap).trace(torch.from_numpy(rays).to("cuda:0"), LIMIT)
    want = _stats_of(DeviceFrame(rows, counts), float(snap.prims["surface_id"][-1]), rays_per_source)
    for rank in range(world):
        got = np.load(tmp_path / f"stats_{rank}.npy")
        assert np.allclose(got, want, rtol=0, atol=1e-12, equal_nan=True), rank


# ---- the BASELINE partitions at world 8 with real HIP traces: eight ranks sharing the one GPU ---------------------
@pytest.mark.parametrize("name,n,mode", [("config4", 8 * 512, "all"), ("config5", 16001, "all")])
def test_baseline_partitions_at_world_eight_sharing_one_gpu(tmp_path, name, n, mode):
    """BASELINE config 4 (8 wavelengths, one per rank: contiguous id shards give exactly that) and config 5
    (generation_limit 10) over 8 ranks; the blocks travel over gloo, prt_place_rows orders them on the device."""
    import torch.multiprocessing as mp

    from pyrayt_amd import engine

    world = 8
    per_call = n // 8 if name == "config4" else n
    mp.start_processes(_worker, args=(world, _free_port(), name, per_call, mode, str(tmp_path)), nprocs=world,
                       join=True, start_method="spawn")
    snap, rays = _scene_and_rays(name, per_call)
    if name == "config4":
        from pyrayt_amd import distributed as pdist

        for rank in range(world):
            lo, hi = pdist.shard_bounds(rays.shape[1], rank=rank, world=world)
            assert np.all(rays[10, lo:hi] == np.linspace(0.44, 0.75, 8)[rank])  # one source / wavelength per rank
    rows, counts = engine.DeviceScene(snap).trace(torch.from_numpy(rays).to("cuda:0"), LIMIT)
    want = rows.cpu().numpy()
    for rank in range(world):
        assert np.load(tmp_path / f"counts_{rank}.npy").tolist() == counts
        assert np.array_equal(np.load(tmp_path / f"rows_{rank}.npy"), want, equal_nan=True)
