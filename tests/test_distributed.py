"""Multi-process path on CPU: world_size-2 (and 3) gloo groups exercising the sharding and the
row re-assembly of pyrayt_amd.distributed -- the only collective on the path.  The per-rank
rows come from the oracle (used here as the checker's data source; the product's trace itself
needs a GPU and is covered by the -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
import scenes
from pyrayt_amd import distributed as pdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import prt_oracle
        from pyrayt_amd.g3d.objects import CountedObject
        from pyrayt_amd.scene import SceneSnapshot

        CountedObject.reset_ids()
        parts, rays = scenes.stopped_lens(scenes.product_api(), 3001)  # odd count: ragged shards
        flat = helpers.flat_scene(SceneSnapshot(parts))
        group = pdist.resolve_group(None)
        lo, hi = pdist.shard_bounds(rays.shape[1], group)
        rows, counts = prt_oracle.trace(flat, rays[:, lo:hi], 10)
        local = torch.from_numpy(np.ascontiguousarray(rows.T))  # (15, R) like the engine
        full, full_counts = pdist.assemble_rows(local, counts, 10, group, mode)
        np.save(os.path.join(result_dir, f"rows_{mode}_{rank}.npy"), full.numpy())
        np.save(os.path.join(result_dir, f"counts_{mode}_{rank}.npy"), np.array(full_counts))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["all", "root"])
def test_sharded_trace_reassembles_reference_order(tmp_path, world, mode):
    from oracle import prt_oracle
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    CountedObject.reset_ids()
    parts, rays = scenes.stopped_lens(scenes.product_api(), 3001)
    want, want_counts = prt_oracle.trace(helpers.flat_scene(SceneSnapshot(parts)), rays, 10)
    for rank in range(world):
        got = np.load(tmp_path / f"rows_{mode}_{rank}.npy").T
        counts = np.load(tmp_path / f"counts_{mode}_{rank}.npy").tolist()
        assert counts == want_counts
        if mode == "all" or rank == 0:
            # sharded == unsharded, bit for bit and in the same row order
            assert np.array_equal(got, want, equal_nan=True)
        else:
            assert got.shape[0] == 0


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 1000, 1_000_003):
        for world in (1, 2, 3, 8):
            edges = [pdist.shard_bounds(n, rank=r, world=world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_placement_is_generation_major_rank_major():
    counts = torch.tensor([[3, 2, 0], [1, 4, 2]])
    dest, local, total = pdist.placement(counts)
    assert total == 12
    assert dest.tolist() == [[0, 4, 10], [3, 6, 10]]
    assert local.tolist() == [[0, 3, 5], [0, 1, 5]]


def test_single_process_passthrough():
    rows = torch.arange(30.0).reshape(15, 2)
    out, counts = pdist.assemble_rows(rows, [2], 10, None, "all")
    assert out is rows and counts == [2]
    assert pdist.resolve_group(None) is None


# ---------------------------------------------------------------------------------------------
# sharded result sink: DeviceFrame.group_stats(group=...) on the rows each rank kept (gather "none")
# ---------------------------------------------------------------------------------------------
def _install_cpu_frame_steps(monkeypatch_target):
    """The three device steps of the sharded statistics replaced by the numpy restatement (oracle/frame_oracle.py):
    what is under test on a box without a GPU is the composition -- which sums are added across ranks, about
    which pivots the second pass runs, how many groups there are -- not the kernels (tests/test_gpu_frame.py and
    tests/test_gpu_distributed.py run those)."""
    from oracle import frame_oracle

    def reduce_pass(self, surface, generation, rays_per_source, n_groups, pivots):
        return torch.from_numpy(frame_oracle.reduce_sums(self.rows.numpy(), surface, generation, rays_per_source, n_groups,
                                                         None if pivots is None else pivots.numpy()))

    monkeypatch_target._reduce_pass = reduce_pass
    monkeypatch_target._pivots = lambda self, sums, n_groups: torch.from_numpy(frame_oracle.pivots_of(sums.numpy()))
    monkeypatch_target._finish = lambda self, sums, pivots, n_groups: torch.from_numpy(frame_oracle.finish(sums.numpy(), pivots.numpy()))


def _stats_worker(rank, world, port, name, rays_per_source, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyrayt_amd.frame import DeviceFrame

        _install_cpu_frame_steps(DeviceFrame)
        fx = helpers.load(f"scene_{name}.npz")
        frame = fx["frame"]  # the reference's own rows
        ids = frame[:, 4]
        lo, hi = pdist.shard_bounds(int(ids.max()) + 1, pdist.resolve_group(None))
        mine = frame[(ids >= lo) & (ids < hi)]  # what this rank's trace of its id range would have recorded
        local = DeviceFrame(torch.from_numpy(np.ascontiguousarray(mine.T)))
        detector = float(frame[-1, 5])
        got = local.group_stats(surface=detector, rays_per_source=rays_per_source, group=pdist.resolve_group(None))
        got.to_pickle(os.path.join(result_dir, f"stats_{rank}.pkl"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,rays_per_source,world", [("config4", 256, 2), ("config4", 256, 3), ("config2", None, 3),
                                                        ("mirrors_and_stops", 512, 2)])
def test_sharded_group_stats_equal_the_whole_frames(tmp_path, name, rays_per_source, world):
    """Each rank reduces the rows of its own id range; the per-group sums are added across ranks (one all-reduce per
    pass); every rank ends up with the statistics pandas computes on the whole reference frame."""
    import pandas as pd

    from oracle import frame_oracle

    mp.spawn(_stats_worker, args=(world, _free_port(), name, rays_per_source, str(tmp_path)), nprocs=world, join=True)
    fx = helpers.load(f"scene_{name}.npz")
    frame = fx["frame"]
    detector = float(frame[-1, 5])
    n_groups = int(frame[:, 4].max() // rays_per_source) + 1 if rays_per_source else 1
    # the restatement itself against pandas (pins the oracle), then the sharded composition against it
    sums = frame_oracle.reduce_sums(frame.T, detector, None, rays_per_source, n_groups)
    piv = frame_oracle.pivots_of(sums)
    whole = frame_oracle.finish(frame_oracle.reduce_sums(frame.T, detector, None, rays_per_source, n_groups, piv), piv)
    cols = ("generation", "intensity", "wavelength", "index", "id", "surface", "x0", "y0", "z0", "x1", "y1", "z1",
            "x_tilt", "y_tilt", "z_tilt")
    df = pd.DataFrame(frame, columns=cols)
    sel = df.loc[df["surface"] == detector]
    keys = (sel["id"] // rays_per_source).astype(int) if rays_per_source else np.zeros(len(sel), dtype=int)
    for sid, rows in sel.groupby(keys.values if hasattr(keys, "values") else keys):
        cy, cz = rows["y1"].mean(), rows["z1"].mean()
        assert whole[sid, 0] == len(rows)
        assert abs(whole[sid, 1] - cy) < 1e-12 and abs(whole[sid, 2] - cz) < 1e-12
        assert abs(whole[sid, 3] - np.sqrt(((rows["y1"] - cy) ** 2 + (rows["z1"] - cz) ** 2).mean())) < 1e-12
        assert abs(whole[sid, 6] - rows["wavelength"].mean()) < 1e-12
    for rank in range(world):
        got = pd.read_pickle(tmp_path / f"stats_{rank}.pkl")
        assert len(got) == n_groups
        block = got[["count", "y", "z", "rms_radius", "focus", "focus_std", "wavelength", "intensity"]].to_numpy(dtype=float)
        assert np.array_equal(block[:, 0], whole[:, 0])
        assert np.allclose(block, whole, rtol=0, atol=1e-12, equal_nan=True), (rank, np.nanmax(np.abs(block - whole)))


# ---------------------------------------------------------------------------------------------
# the BASELINE multi-GPU partitions at world 8 (configs 4 and 5), rows from the oracle
# ---------------------------------------------------------------------------------------------
def _world8_worker(rank, world, port, name, args, limit, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import prt_oracle
        from pyrayt_amd.g3d.objects import CountedObject
        from pyrayt_amd.scene import SceneSnapshot

        CountedObject.reset_ids()
        parts, rays = scenes.SCENES[name](scenes.product_api(), *args)
        flat = helpers.flat_scene(SceneSnapshot(parts))
        group = pdist.resolve_group(None)
        lo, hi = pdist.shard_bounds(rays.shape[1], group)
        if name == "config4":
            # BASELINE config 4: "8 wavelengths, sharded 8 x MI355X" -- with contiguous id shards every rank gets
            # exactly one source, i.e. one wavelength
            assert hi - lo == args[0] and len(np.unique(rays[10, lo:hi])) == 1
            assert rays[10, lo] == np.linspace(0.44, 0.75, 8)[rank]
        rows, counts = prt_oracle.trace(flat, rays[:, lo:hi], limit)
        local = torch.from_numpy(np.ascontiguousarray(rows.T))
        full, full_counts = pdist.assemble_rows(local, counts, limit, group, "all")
        if rank in (0, world - 1):
            np.save(os.path.join(result_dir, f"rows_{rank}.npy"), full.numpy())
        np.save(os.path.join(result_dir, f"counts_{rank}.npy"), np.array(full_counts))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,args,limit", [("config4", (96,), 10), ("config5", (1601,), 10)])
def test_baseline_partitions_at_world_eight(tmp_path, name, args, limit):
    """BASELINE configs 4 (one wavelength per GPU) and 5 (16M rays over 8 GPUs, generation_limit 10) at their
    world size, scaled down in rays: sharded == unsharded, bit for bit, in the reference's row order."""
    from oracle import prt_oracle
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    world = 8
    mp.spawn(_world8_worker, args=(world, _free_port(), name, args, limit, str(tmp_path)), nprocs=world, join=True)
    CountedObject.reset_ids()
    parts, rays = scenes.SCENES[name](scenes.product_api(), *args)
    want, want_counts = prt_oracle.trace(helpers.flat_scene(SceneSnapshot(parts)), rays, limit)
    for rank in range(world):
        assert np.load(tmp_path / f"counts_{rank}.npy").tolist() == want_counts
    for rank in (0, world - 1):
        assert np.array_equal(np.load(tmp_path / f"rows_{rank}.npy").T, want, equal_nan=True)
