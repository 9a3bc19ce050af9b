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

    INIT_TIMEOUT_S = 180.0  # how long ncclCommInitRank may take before the rank gives up (all ranks must arrive)

    def __init__(self, device, world, rank, unique_id, timeout=None):
        """timeout (seconds, default INIT_TIMEOUT_S): ncclCommInitRank blocks until every rank of the communicator has
        called it -- a rank that died, was never started or sits on another id makes the others wait for ever.  The
        call therefore runs on a helper thread and is abandoned with a clear error when the time is up (the process is
        then expected to exit: the thread cannot be cancelled)."""
        import threading

        from . import engine

        self._lib = engine.library()
        self.world, self.rank = int(world), int(rank)
        self.device = torch.device("cuda", device) if isinstance(device, int) else device
        self._handle = ctypes.c_void_p()
        self._work = None
        self._work_streams = set()
        limit = self.INIT_TIMEOUT_S if timeout is None else float(timeout)
        handle, outcome = ctypes.c_void_p(), {}

        def create():
            try:
                outcome["rc"] = self._lib.prt_comm_create(self.device.index or 0, self.world, self.rank,
                                                          ctypes.c_char_p(bytes(unique_id)), ctypes.byref(handle))
                outcome["message"] = self._lib.prt_last_error().decode("utf-8", "replace")
                if outcome.get("abandoned") and outcome["rc"] >= 0 and handle:
                    # the caller gave up waiting and is gone: a communicator completed this late belongs to nobody
                    self._lib.prt_comm_destroy(handle)
            except BaseException as exc:  # noqa: BLE001
                outcome["exc"] = exc

        worker = threading.Thread(target=create, name="prt_comm_create", daemon=True)
        worker.start()
        worker.join(limit)
        if worker.is_alive():
            outcome["abandoned"] = True
            raise TimeoutError(f"rank {self.rank} of {self.world}: ncclCommInitRank did not return within {limit:.0f} s -- not "
                               "every rank of the communicator called it (a rank that failed earlier, fewer processes than "
                               "WORLD_SIZE, ranks holding different unique ids, or no path between the GPUs: check "
                               "HSA_ENABLE_IPC_MODE_LEGACY=0 and NCCL_DEBUG=INFO)")
        if "exc" in outcome:
            raise outcome["exc"]
        if outcome["rc"] < 0:  # (the error text is thread-local to the helper: carried over by hand)
            raise RuntimeError(f"libprt_hip error {outcome['rc']}: {outcome['message']}")
        self._handle = handle

    def info(self):
        """What RCCL reports about the communicator (``prt_comm_info``): ranks in it, this rank, device."""
        from . import engine

        out = (ctypes.c_int * 3)()
        engine._check(self._lib.prt_comm_info(self._handle, out))
        return {"ranks": int(out[0]), "rank": int(out[1]), "device": int(out[2])}

    @staticmethod
    def unique_id():
        from . import engine

        buf = ctypes.create_string_buffer(128)
        engine._check(engine.library().prt_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_group(cls, group, device, timeout=None):
        """Bootstrap over an initialised ``torch.distributed`` group (any backend): rank 0 draws the 128-byte id,
        everybody receives it, every rank creates -- within ``timeout`` seconds (see ``__init__``) or not at all."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        src = dist.get_global_rank(group, 0) if hasattr(dist, "get_global_rank") else 0
        dist.broadcast_object_list(box, src=src, group=group)
        return cls(device, world, rank, box[0], timeout=timeout)

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
            self._work_streams = set()
        # (the staging block is kept across calls, and a communicator may be used from several streams -- the caller's,
        # trace_and_gather's communication stream: the allocator must not hand the block's memory out again while a
        # stream other than the one it was allocated under still works in it)
        stream_now = torch.cuda.current_stream(self.device)
        if stream_now.cuda_stream not in self._work_streams:
            self._work_streams.add(stream_now.cuda_stream)
            self._work.record_stream(stream_now)
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


def trace_and_gather(scene, ray_sets, generation_limit, comm, depth=2, flags=0, ray_offset=1e-6):
    """Trace a sequence of (this rank's shards of) ray sets and re-assemble every frame on every rank, with the
    re-assembly of frame k running behind trace k + 1: traces go out on the scene's ticket streams (``depth`` in
    flight), the RCCL all-gathers and the placement kernel of ``prt_allgather_rows`` on a communication stream of their
    own.  The communication stream waits for the consumer only where it must: the gather of frame k waits for an event
    the generator records on the current stream when it is resumed after handing out frame k - 1 (by then the consumer
    is done with frame k - 2, whose block frame k is assembled into), not for everything enqueued on that stream.
    A consumer that stops early -- or an error on the way -- leaves nothing behind: the traces in flight are collected
    and the current stream waits for the ticket streams and the communication stream before the blocks are released.  At N = 8 the gather of the north-star job moves 315 MB into every GPU and takes several times as long as the
    trace; pipelined, a loop that wants whole frames pays max(trace, gather) per step instead of their sum.

    Yields ``(frame, rows_per_generation)`` per ray set, in order: ``frame`` is the whole (15, total) frame in reference
    order on this rank's GPU, complete on the CURRENT stream when it is handed out (the generator makes the current
    stream wait for the communication stream).  Frames are assembled into two blocks used in turn: frame k is
    overwritten when frame k + 2 is asked for.  Collective: every rank iterates, with ray sets of the same count.
    The one host synchronisation per step is the all-gather of the small count matrix (the ranks must agree on the
    frame's layout)."""
    from . import engine

    torch_mod = torch
    limit = int(generation_limit)
    depth = max(1, min(int(depth), engine.TRACE_TICKETS))
    device = comm.device
    streams = scene.ticket_streams(device, depth)
    comm_stream = getattr(comm, "_stream", None)
    if comm_stream is None:
        comm_stream = comm._stream = torch_mod.cuda.Stream(device)
    blocks = [None] * (depth + 1)   # record blocks of the traces: one more than are in flight (a gather may still read one)
    gathered = [None] * (depth + 1)  # event per record block: the gather that read it last has finished
    frames = [None, None]
    pending = []                     # (lane, slot) of traces begun and not yet collected
    current = torch_mod.cuda.current_stream(device)
    source = iter(ray_sets)
    begun = 0

    def begin_next():
        nonlocal begun
        try:
            rays = next(source)
        except StopIteration:
            return False
        lane, slot = begun % depth, begun % len(blocks)
        need = max(rays.shape[1], 1) * limit
        if blocks[slot] is None or blocks[slot].shape[1] < need:
            blocks[slot] = torch_mod.empty((engine.RECORD_COLS, need), dtype=torch_mod.float64, device=device)
        streams[lane].wait_stream(current)              # whatever produced the ray set
        if gathered[slot] is not None:
            streams[lane].wait_event(gathered[slot])    # the gather that last read this record block
        scene.trace_begin(lane, rays, limit, blocks[slot], ray_offset=ray_offset, flags=flags, stream=streams[lane])
        pending.append((lane, slot))
        begun += 1
        return True

    consumer_done = [None, None]     # per frame block: the consumer on `current` has moved past the frame it held
    try:
        for _ in range(depth):
            if not begin_next():
                break
        k = 0
        while pending:
            lane, slot = pending[0]
            rows, counts = scene.trace_end(lane)
            pending.pop(0)
            begin_next()                                    # trace k + depth goes out before frame k is assembled
            which = k % 2
            with torch_mod.cuda.stream(comm_stream):
                comm_stream.wait_stream(streams[lane])      # the rows are ordered on the ticket's stream
                # (host sync of the communication stream only: the count matrix does not wait for what the consumer
                # enqueued on the current stream for frame k - 1 -- only the frame block about to be reused does)
                matrix = comm.gather_counts(counts, limit)
                per_rank = matrix.sum(dim=1)
                total = int(per_rank.sum())
                if consumer_done[which] is not None:
                    comm_stream.wait_event(consumer_done[which])   # whoever still read frame k - 2 out of this block
                if frames[which] is None or frames[which].shape[1] < total:
                    frames[which] = torch_mod.empty((engine.RECORD_COLS, max(total, 1)), dtype=torch_mod.float64,
                                                    device=device)
                    frames[which].record_stream(current)    # (allocated under comm_stream, read on the caller's stream)
                frame = comm.gather_rows(blocks[slot], matrix, limit, out=frames[which])
                done = torch_mod.cuda.Event()
                done.record(comm_stream)
                gathered[slot] = done
            current.wait_event(done)
            yield frame, _merged_counts(matrix)
            # the generator resumes: the consumer is done with the frame BEFORE this one (it may hold frame k while it
            # takes frame k + 1); what it enqueued on the current stream up to here covers its reads of that frame
            mark = torch_mod.cuda.Event()
            mark.record(current)
            consumer_done[1 - which] = mark
            k += 1
    finally:
        # A consumer that stops early, or an error in gather_counts / gather_rows / trace_end: the traces still in
        # flight are collected -- their kernels may be writing into `blocks`, which go back to the allocator when this
        # generator is closed, and their tickets stay `active` in the library otherwise (every later trace of the scene
        # would be refused) -- and the caller's stream waits for everything that touched the blocks and frames.
        for lane, _slot in pending:
            try:
                scene.trace_end(lane)
            except Exception:  # noqa: BLE001
                pass
        for lane in range(depth):
            current.wait_stream(streams[lane])
        current.wait_stream(comm_stream)
