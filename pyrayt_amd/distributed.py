"""Ray data-parallelism across the GPUs of a node: one process per GPU, torch.distributed.

Rays are independent (no cross-ray term anywhere on the path), so the
trace itself needs no communication: rank r takes the contiguous id range
[r*n/G, (r+1)*n/G) and runs its own generation loop.  The only exchange is the re-assembly of
the result rows in the reference's order -- generation-major, and inside a generation
ascending ray id, which with contiguous shards and order-preserving compaction is simply
rank-major.  That is one all-gather of a small count matrix plus one all-gather of the
(padded) row blocks; with backend "nccl" this is RCCL over xGMI, with "gloo" it runs on CPU
tensors (used by the tests).

The reference has no counterpart: it is a single Python thread (``pyrayt/_pyrayt.py:329-339``).
"""
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


def _scatter_rank(out, block, dest_row, local_row, count_row):
    """Copy one rank's generation-major rows into their assembled positions."""
    reps = count_row.to(block.device)
    total = int(reps.sum())
    if total == 0:
        return
    shift = torch.repeat_interleave((dest_row - local_row).to(block.device), reps)
    index = shift + torch.arange(total, device=block.device)
    out[:, index] = block[:, :total]


def assemble_rows(rows, counts, generation_limit, group=None, gather="all"):
    """Re-assemble per-rank record blocks into the reference's row order.

    rows: (15, R_local) tensor, generation-major; counts: rows per generation (list).
    Returns (rows, rows-per-generation list).  gather: "all" | "root" | "none"."""
    if group is None or gather == "none":
        return rows, list(counts)
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = rows.device
    limit = int(generation_limit)
    mine = torch.zeros(limit, dtype=torch.int64, device=dev)
    if counts:
        mine[: len(counts)] = torch.tensor(list(counts), dtype=torch.int64, device=dev)
    # outputs are the rank-major concatenation along dim 0 (the form every backend accepts)
    matrix = torch.empty(world * limit, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(matrix, mine, group=group)
    matrix_host = matrix.view(world, limit).cpu()
    dest, local, total = placement(matrix_host)
    per_rank = matrix_host.sum(dim=1)
    widest = int(per_rank.max())
    merged_counts = [int(v) for v in matrix_host.sum(dim=0)]
    while merged_counts and merged_counts[-1] == 0:
        merged_counts.pop()

    padded = torch.zeros((rows.shape[0], widest), dtype=rows.dtype, device=dev)
    padded[:, : rows.shape[1]] = rows
    if gather == "all":
        everything = torch.empty((world * padded.shape[0], widest), dtype=rows.dtype, device=dev)
        dist.all_gather_into_tensor(everything, padded, group=group)
        blocks = list(everything.view(world, padded.shape[0], widest).unbind(0))
    elif gather == "root":
        root = dist.get_global_rank(group, 0) if hasattr(dist, "get_global_rank") else 0
        blocks = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
        dist.gather(padded, blocks, dst=root, group=group)
        if rank != 0:
            return rows[:, :0], merged_counts
    else:
        raise ValueError(f"unknown gather mode {gather!r}")

    out = torch.empty((rows.shape[0], total), dtype=rows.dtype, device=dev)
    for r in range(world):
        _scatter_rank(out, blocks[r], dest[r], local[r], matrix_host[r])
    return out, merged_counts
