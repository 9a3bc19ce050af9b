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
