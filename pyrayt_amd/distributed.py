"""Ray data-parallelism across the GPUs of a node: one process per GPU.

Rays are independent (no cross-ray term anywhere on the path), so the
trace itself needs no communication: rank r takes the contiguous id range
[r*n/G, (r+1)*n/G) and runs its own generation loop.  The only exchange is the re-assembly of
the result rows in the reference's order -- generation-major, and inside a generation
ascending ray id, which with contiguous shards and order-preserving compaction is simply
rank-major.

The exchange lives in the HIP library (``csrc/prt_gather.hpp``): an all-gather of the small count
matrix, fifteen grouped RCCL all-gathers straight out of the record block, and one placement
kernel (``prt_allgather_rows``).  This module is the thin caller:

* ``LibraryComm`` wraps the library's RCCL communicator; it is bootstrapped through an existing
  ``torch.distributed`` group (rank 0's 128-byte id is broadcast over it).
* ``assemble_rows`` picks the path: GPU tensors + a ``LibraryComm`` -> RCCL over xGMI entirely
  inside the library; GPU tensors + a gloo group (several ranks sharing one GPU in the tests) ->
  the blocks travel through ``torch.distributed`` and the library's placement kernel
  (``prt_place_rows``) orders them on the device; CPU tensors (the gloo tests without a GPU) ->
  ``torch.distributed`` + an indexed copy.

The reference has no counterpart: it is a single Python thread (``pyrayt/_pyrayt.py:329-339``).
"""
import ctypes

import torch


def resolve_group(group=None):
    """The process group to shard over, or None when running single-process."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return None
    if group is None:
        group = dist.group.WORLD
    return group if dist.get_world_size(group) > 1 else None


def shard_bounds(n, group=None, rank=None, world=None):
    """[lo, hi) of this rank's contiguous slice of n rays."""
    if rank is None or world is None:
        if group is None:
            return 0, n
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
    return (rank * n) // world, ((rank + 1) * n) // world


def placement(count_matrix):
    """Destination offset of every (rank, generation) block in the assembled frame.

    count_matrix: int64 (G, L) rows recorded by rank r in generation g.  Returns
    (dest (G, L), local (G, L), total): block (r, g) occupies assembled rows
    [dest[r,g], dest[r,g] + count[r,g]) and local rows [local[r,g], ...) of rank r."""
    c = count_matrix.to(torch.int64)
    per_generation = c.sum(dim=0)
    generation_start = torch.cumsum(per_generation, 0) - per_generation
    rank_start = torch.cumsum(c, 0) - c
    dest = generation_start.unsqueeze(0) + rank_start
    local = torch.cumsum(c, 1) - c
    return dest, local, int(per_generation.sum())


class LibraryComm:
    """The HIP library's RCCL communicator (``prt_comm``), one per process / GPU.

    ``LibraryComm.from_group(group, device)`` bootstraps it over an initialised
    ``torch.distributed`` group of any backend: rank 0 draws the id, everybody receives it."""

    def __init__(self, device, world, rank, unique_id):
        from . import engine

        self._lib = engine.library()
        self.world, self.rank = int(world), int(rank)
        self.device = torch.device("cuda", device) if isinstance(device, int) else device
        self._handle = ctypes.c_void_p()
        engine._check(self._lib.prt_comm_create(self.device.index or 0, self.world, self.rank,
                                                ctypes.c_char_p(bytes(unique_id)), ctypes.byref(self._handle)))
        self._work = None

    @staticmethod
    def unique_id():
        from . import engine

        buf = ctypes.create_string_buffer(128)
        engine._check(engine.library().prt_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_group(cls, group, device):
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        src = dist.get_global_rank(group, 0) if hasattr(dist, "get_global_rank") else 0
        dist.broadcast_object_list(box, src=src, group=group)
        return cls(device, world, rank, box[0])

    def close(self):
        if self._handle:
            self._lib.prt_comm_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gather_counts(self, counts, limit):
        """(G, limit) int64 count matrix on the host; synchronises the current stream."""
        from . import engine

        mine = (ctypes.c_int64 * limit)(*([int(c) for c in counts] + [0] * (limit - len(counts))))
        everyone = (ctypes.c_int64 * (limit * self.world))()
        engine._check(self._lib.prt_allgather_counts(self._handle, mine, limit, everyone,
                                                     engine._stream_ptr(torch, self.device)))
        return torch.tensor(list(everyone), dtype=torch.int64).view(self.world, limit)

    def gather_rows(self, rows, matrix_host, limit, out=None, reuse=False):
        """All ranks' rows in reference order, (15, total) on this rank's GPU.  Stream-ordered.

        rows: this rank's record block (a (15, R) view of it is enough: the all-gather reads `widest`
        elements of every column, which the (15, cap) block the trace recorded into always has -- only a
        narrower or strided tensor is copied into a block of that width first).
        out: a (15, >= total) block to assemble into (a loop hands the previous frame's block back).
        reuse: without `out`, assemble into the communicator's own block, kept while the size stays the same
        -- for a loop that is done with the previous frame by then; the default is a new block per call, as
        a single-GPU trace returns one (two frames a caller keeps never alias)."""
        from . import engine

        per_rank = matrix_host.sum(dim=1)
        widest, total = int(per_rank.max()), int(per_rank.sum())
        if rows.stride(0) < widest or rows.stride(1) != 1:
            # the all-gather sends `widest` elements of every column: give it a block that wide
            block = torch.empty((rows.shape[0], max(widest, 1)), dtype=rows.dtype, device=rows.device)
            block[:, : rows.shape[1]] = rows
            rows = block
        need = int(self._lib.prt_allgather_workspace_bytes(self.world, limit, widest))
        if self._work is None or self._work.numel() < need:
            self._work = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is None and reuse:
            kept = getattr(self, "_out", None)
            if kept is None or kept.shape != (rows.shape[0], total) or kept.dtype != rows.dtype:
                self._out = kept = torch.empty((rows.shape[0], total), dtype=rows.dtype, device=self.device)
            out = kept
        elif out is None:
            out = torch.empty((rows.shape[0], total), dtype=rows.dtype, device=self.device)
        else:
            assert out.is_cuda and out.shape[0] == rows.shape[0] and out.shape[1] >= total and out.stride(1) == 1
        flat = (ctypes.c_int64 * (limit * self.world))(*[int(v) for v in matrix_host.reshape(-1)])
        engine._check(self._lib.prt_allgather_rows(self._handle, rows.data_ptr(), rows.stride(0), flat, limit,
                                                   out.data_ptr(), max(out.stride(0), 1), self._work.data_ptr(),
                                                   engine._stream_ptr(torch, self.device)))
        return out[:, :total]


def _merged_counts(matrix_host):
    merged = [int(v) for v in matrix_host.sum(dim=0)]
    while merged and merged[-1] == 0:
        merged.pop()
    return merged


def _place_on_device(blocks, matrix_host, limit, total):
    """Order gathered blocks with the library's placement kernel.  blocks: (G, 15, widest) CUDA."""
    from . import engine

    lib = engine.library()
    world, cols, widest = blocks.shape
    dev = blocks.device
    out = torch.empty((cols, total), dtype=blocks.dtype, device=dev)
    work = torch.empty(int(lib.prt_place_workspace_bytes(world, limit)), dtype=torch.uint8, device=dev)
    flat = (ctypes.c_int64 * (limit * world))(*[int(v) for v in matrix_host.reshape(-1)])
    engine._check(lib.prt_place_rows(dev.index or 0, blocks.data_ptr(), cols * widest, widest, world, flat,
                                     limit, out.data_ptr(), max(total, 1), work.data_ptr(),
                                     engine._stream_ptr(torch, dev)))
    torch.cuda.current_stream(dev).synchronize()  # `work` and `blocks` die with this frame
    return out


def _place_with_torch(blocks, matrix_host, total):
    """CPU tensors (gloo tests without a GPU): an indexed copy per rank."""
    dest, local, _ = placement(matrix_host)
    out = torch.empty((blocks[0].shape[0], total), dtype=blocks[0].dtype, device=blocks[0].device)
    for r, block in enumerate(blocks):
        reps = matrix_host[r]
        count = int(reps.sum())
        if count == 0:
            continue
        shift = torch.repeat_interleave(dest[r] - local[r], reps)
        out[:, shift + torch.arange(count)] = block[:, :count]
    return out


def assemble_rows(rows, counts, generation_limit, group=None, gather="all", comm=None, count_matrix=None, out=None,
                  reuse=False):
    """Re-assemble per-rank record blocks into the reference's row order.

    rows: (15, R_local) tensor, generation-major; counts: rows per generation (list).
    Returns (rows, rows-per-generation list).  gather: "all" | "root" | "none".
    comm: a ``LibraryComm`` -> the whole exchange runs inside the HIP library over RCCL.
    count_matrix: the (G, limit) rows-per-generation matrix if the caller already holds it (a repeated
    trace whose counts did not change): the count all-gather and its host synchronisation are skipped.
    out / reuse: where the frame is assembled (``LibraryComm.gather_rows``)."""
    if (group is None and comm is None) or gather == "none":
        return rows, list(counts)
    limit = int(generation_limit)
    if comm is not None:
        matrix_host = comm.gather_counts(counts, limit) if count_matrix is None else count_matrix
        out = comm.gather_rows(rows, matrix_host, limit, out=out, reuse=reuse)
        if gather == "root" and comm.rank != 0:
            return rows[:, :0], _merged_counts(matrix_host)
        return out, _merged_counts(matrix_host)
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    on_gpu = rows.is_cuda
    # the transport of this path is torch.distributed on whatever device its backend serves
    backend_dev = rows.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.zeros(limit, dtype=torch.int64)
    if counts:
        mine[: len(counts)] = torch.tensor(list(counts), dtype=torch.int64)
    mine = mine.to(backend_dev)
    matrix = torch.empty(world * limit, dtype=torch.int64, device=backend_dev)
    dist.all_gather_into_tensor(matrix, mine, group=group)
    matrix_host = matrix.view(world, limit).cpu()
    _, _, total = placement(matrix_host)
    widest = int(matrix_host.sum(dim=1).max())
    merged_counts = _merged_counts(matrix_host)

    padded = torch.zeros((rows.shape[0], widest), dtype=rows.dtype, device=backend_dev)
    padded[:, : rows.shape[1]] = rows.to(backend_dev)
    if gather == "all":
        everything = torch.empty((world * padded.shape[0], widest), dtype=rows.dtype, device=backend_dev)
        dist.all_gather_into_tensor(everything, padded, group=group)
        blocks = everything.view(world, padded.shape[0], widest)
    elif gather == "root":
        root = dist.get_global_rank(group, 0) if hasattr(dist, "get_global_rank") else 0
        pieces = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
        dist.gather(padded, pieces, dst=root, group=group)
        if rank != 0:
            return rows[:, :0], merged_counts
        blocks = torch.stack(pieces)
    else:
        raise ValueError(f"unknown gather mode {gather!r}")
    if on_gpu:
        return _place_on_device(blocks.to(rows.device).contiguous(), matrix_host, limit, total), merged_counts
    return _place_with_torch(list(blocks.unbind(0)), matrix_host, total), merged_counts
