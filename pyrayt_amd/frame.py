"""DeviceFrame: the trace result kept columnar in HBM.

``RayTracer.trace()`` returns a pandas DataFrame like the reference
(``pyrayt/_pyrayt.py:147-186``); for a 1M-ray trace that is a 360 MB device-to-host copy which
costs two orders of magnitude more than the trace itself.  ``RayTracer.trace_device()`` returns
this view instead: the engine's (15, R) record block, one contiguous row per column, with the
handful of selections the reference's examples make on the frame (``results.loc[results[
"surface"] == id]``, per-generation slices, spot statistics: ``examples/lens_design.ipynb``)
done on the device, so that only what is looked at crosses PCIe.  The grouped reductions
(``group_stats``: per-source / per-wavelength spot and focus statistics, the notebook's cells
11-16) are one HIP kernel over the column block (``prt_frame_reduce``, ``csrc/prt_frame.hpp``);
row selections that return a new frame are torch indexing -- plumbing around the result.
"""

import numpy as np
import pandas as pd

COLUMNS = ("generation", "intensity", "wavelength", "index", "id", "surface",
           "x0", "y0", "z0", "x1", "y1", "z1", "x_tilt", "y_tilt", "z_tilt")
_INDEX = {name: k for k, name in enumerate(COLUMNS)}


class DeviceFrame:
    def __init__(self, rows, rows_per_generation=None, columns=None):
        """rows: (15, R) tensor (device or host), generation-major.  columns: indices of the columns the trace wrote
        (a record plan with a column list, ``engine.RecordPlan(columns=...)``; None: all fifteen) -- the other rows of
        the block hold whatever was there and are never handed out."""
        assert rows.shape[0] == len(COLUMNS)
        self.rows = rows
        self.rows_per_generation = list(rows_per_generation or [])
        self.written = None if columns is None or len(columns) == len(COLUMNS) else tuple(sorted(columns))
        if self.written is not None:
            self.columns = tuple(COLUMNS[k] for k in self.written)  # (names of the columns THIS frame holds)

    # --- shape / access -----------------------------------------------------------------------
    columns = COLUMNS

    def __len__(self):
        return int(self.rows.shape[1])

    @property
    def shape(self):
        return (len(self), len(self.columns))

    def _need(self, *names):
        if self.written is not None:
            missing = [name for name in names if _INDEX[name] not in self.written]
            if missing:
                raise KeyError(f"this frame was recorded without the column(s) {missing} (RecordPlan(columns=...))")

    def __getitem__(self, column):
        """One column as a 1-D tensor view (no copy)."""
        self._need(column)
        return self.rows[_INDEX[column]]

    # --- selections ------------------------------------------------------------------------------
    def generation(self, g):
        """Rows of generation g: a contiguous slice (rows are generation-major), no kernel."""
        if g < len(self.rows_per_generation):
            start = sum(self.rows_per_generation[:g])
            return DeviceFrame(self.rows[:, start:start + self.rows_per_generation[g]],
                               [0] * g + [self.rows_per_generation[g]], self.written)
        return self.where(generation=g)

    def last_generation(self):
        """Rows of the highest generation in the frame -- the notebook's way to say "the rays that reached the imager"
        when the imager is not at hand (``examples/lens_design.ipynb`` cells 12, 15, 20:
        ``results.loc[results['generation'] == np.max(results['generation'])]``).  Rows are generation-major, so
        this is the frame's last slice: no kernel, no copy."""
        number = self.last_generation_number()
        return self if number is None else self.generation(number)

    def last_generation_number(self):
        """The highest generation that recorded rows (None for an empty frame)."""
        counts = self.rows_per_generation
        if counts:
            working = [g for g, k in enumerate(counts) if k]
            return working[-1] if working else None
        if len(self) == 0:
            return None
        return int(float(self["generation"].max()))

    def where(self, **equals):
        """Rows whose named columns equal the given values, e.g. where(surface=6, generation=2)."""
        mask = None
        for name, value in equals.items():
            m = self[name] == float(value)
            mask = m if mask is None else (mask & m)
        if mask is None:
            return self
        return DeviceFrame(self.rows[:, mask], None, self.written)

    def select(self, mask):
        return DeviceFrame(self.rows[:, mask], None, self.written)

    # --- reductions the notebook does on the frame -------------------------------------------------
    def group_stats(self, surface=None, generation=None, rays_per_source=None, n_groups=None, group=None, comm=None):
        """Per-source statistics of the rows that hit ``surface`` and / or belong to ``generation``
        (``examples/lens_design.ipynb`` cells 11-16: ``results.loc[results['surface'] == id]``
        grouped by ``source_id = id // rays_per_source``, ``_pyrayt.py:349-354``).

        Returns a DataFrame indexed by source id with columns ``count``, ``y`` / ``z`` (spot centroid
        of the end points), ``rms_radius`` (about that centroid), ``focus`` / ``focus_std`` (mean and
        spread of the x-axis intercepts ``x0 - x_tilt * y0 / y_tilt``, the notebook's paraxial-focus
        estimate), ``wavelength`` and ``intensity`` (means).  One library call (``prt_frame_stats``): the
        HIP reduction kernel runs twice, the second pass about the first pass's per-group means so that
        the second moments are well conditioned, and the final arithmetic happens on the device too.
        Without ``rays_per_source`` everything is one group.

        ``group`` (a ``torch.distributed`` group) or ``comm`` (a ``pyrayt_amd.distributed.LibraryComm``): this
        frame holds one rank's rows of a sharded trace (``RayTracer(..., gather="none")``) and the statistics
        wanted are those of the WHOLE frame.  Every rank reduces its own rows; the per-group sums -- nine doubles a
        group -- are added across the ranks (RCCL all-reduce inside the library with ``comm``; ``torch.distributed``
        with a group of another backend), once per pass.  Every rank gets the full statistics and the rows stay
        where they are: the alternative, re-assembling the frame, moves 315 MB into every GPU for a 1M-ray trace.
        Collective: every rank of the group calls it, with the same arguments (``n_groups`` included, or None)."""
        self._need("generation", "surface", "id", "intensity", "wavelength", "x0", "y0", "x_tilt", "y_tilt", "y1", "z1")
        sharded = group is not None or comm is not None
        if rays_per_source:
            if n_groups is None:
                top = float(self["id"].max()) if len(self) else -1.0
                if sharded:
                    top = _all_reduce_max(top, group, comm, self.rows.device)
                n_groups = max(1, int(top // rays_per_source) + 1)
        else:
            n_groups = 1
        stats = (self._stats_sharded(surface, generation, rays_per_source, n_groups, group, comm) if sharded
                 else self._stats(surface, generation, rays_per_source, n_groups))
        frame = pd.DataFrame({
            "count": stats[:, 0].astype(np.int64), "y": stats[:, 1], "z": stats[:, 2], "rms_radius": stats[:, 3],
            "focus": stats[:, 4], "focus_std": stats[:, 5], "wavelength": stats[:, 6], "intensity": stats[:, 7],
        })
        frame.index.name = "source_id"
        return frame

    def _stats(self, surface, generation, rays_per_source, n_groups):
        """``prt_frame_stats``: both reduction passes and the final arithmetic on the device, one
        (n_groups, 8) block brought to the host."""
        import torch

        from . import engine

        rows = self.rows
        if rows.stride(1) != 1:
            rows = rows.contiguous()
        dev = rows.device
        lib = engine.library()
        out = torch.empty((n_groups, 8), dtype=torch.float64, device=dev)
        work = torch.empty(int(lib.prt_frame_stats_workspace_bytes(n_groups)), dtype=torch.uint8, device=dev)
        nan = float("nan")
        engine._check(lib.prt_frame_stats(
            dev.index or 0, rows.data_ptr(), rows.stride(0), rows.shape[1],
            nan if surface is None else float(surface), nan if generation is None else float(generation),
            float(rays_per_source or 0), n_groups, out.data_ptr(), work.data_ptr(), engine._stream_ptr(torch, dev)))
        return out.cpu().numpy()

    def _reduce_pass(self, surface, generation, rays_per_source, n_groups, pivots):
        """One pass of ``prt_frame_reduce`` over this frame's rows: the (n_groups, 9) sums, on the device."""
        import torch

        from . import engine

        rows = self.rows if self.rows.stride(1) == 1 else self.rows.contiguous()
        dev = rows.device
        sums = torch.empty((n_groups, 9), dtype=torch.float64, device=dev)
        nan = float("nan")
        engine._check(engine.library().prt_frame_reduce(
            dev.index or 0, rows.data_ptr(), max(rows.stride(0), 1), rows.shape[1],
            nan if surface is None else float(surface), nan if generation is None else float(generation),
            float(rays_per_source or 0), n_groups, pivots.data_ptr() if pivots is not None else None,
            sums.data_ptr(), engine._stream_ptr(torch, dev)))
        return sums

    def _stats_sharded(self, surface, generation, rays_per_source, n_groups, group, comm):
        """The whole frame's statistics from this rank's rows (see ``group_stats``)."""
        import torch

        from . import engine

        lib = engine.library()
        nan = float("nan")
        if comm is not None:  # one library call: both passes, two ncclAllReduce of (n_groups, 9) doubles
            rows = self.rows if self.rows.stride(1) == 1 else self.rows.contiguous()
            dev = rows.device
            out = torch.empty((n_groups, 8), dtype=torch.float64, device=dev)
            work = torch.empty(int(lib.prt_frame_stats_workspace_bytes(n_groups)), dtype=torch.uint8, device=dev)
            engine._check(lib.prt_frame_stats_sharded(
                comm._handle, rows.data_ptr(), max(rows.stride(0), 1), rows.shape[1],
                nan if surface is None else float(surface), nan if generation is None else float(generation),
                float(rays_per_source or 0), n_groups, out.data_ptr(), work.data_ptr(), engine._stream_ptr(torch, dev)))
            return out.cpu().numpy()
        # another transport adds the sums: the passes and the two small steps are separate library calls
        sums = _all_reduce_sum(self._reduce_pass(surface, generation, rays_per_source, n_groups, None), group)
        pivots = self._pivots(sums, n_groups)
        sums = _all_reduce_sum(self._reduce_pass(surface, generation, rays_per_source, n_groups, pivots), group)
        return self._finish(sums, pivots, n_groups).cpu().numpy()

    def _pivots(self, sums, n_groups):
        """``prt_frame_pivots``: per group the means a second pass accumulates about, (n_groups, 3) on the device."""
        import torch

        from . import engine

        pivots = torch.empty((n_groups, 3), dtype=torch.float64, device=sums.device)
        engine._check(engine.library().prt_frame_pivots(sums.device.index or 0, sums.data_ptr(), n_groups, pivots.data_ptr(),
                                                        engine._stream_ptr(torch, sums.device)))
        return pivots

    def _finish(self, sums, pivots, n_groups):
        """``prt_frame_finish``: a second pass's sums and its pivots -> the (n_groups, 8) statistics, on the device."""
        import torch

        from . import engine

        out = torch.empty((n_groups, 8), dtype=torch.float64, device=sums.device)
        engine._check(engine.library().prt_frame_finish(sums.device.index or 0, sums.data_ptr(), pivots.data_ptr(), n_groups,
                                                        out.data_ptr(), engine._stream_ptr(torch, sums.device)))
        return out

    def mean_square(self, quantity, about=0.0, transform=None, surface=None, generation=None, rays_per_source=None,
                    n_groups=None, group=None):
        """``np.mean(np.square(f(rows) - about))`` -- the shape of every merit function in the lens-design notebook --
        as one HIP reduction over the column block (``prt_frame_mean_square``), nothing but the result crossing PCIe.

        quantity: a column name, or ``"axis_intercept"`` for ``x0 - x_tilt * y0 / y_tilt`` (the notebook's paraxial
        focus, cells 12 / 15); transform: None or ``"sin"``; surface / generation filter the rows as in
        ``group_stats`` (``generation="last"``: the highest generation, cells 12 / 15 / 20).  Examples:
        the coma metric of cell 20, ``np.mean(np.square(np.sin(ray_set['y_tilt']) - np.sin(angle)))``, is
        ``frame.mean_square("y_tilt", about=np.sin(angle), transform="sin", generation="last")``; the focus error
        of cells 28 / 32 is ``frame.mean_square("axis_intercept", about=system_focus, generation="last")``.
        Rows whose value is not finite are skipped, like pandas' mean skips NaN.

        Returns a float, or with ``rays_per_source`` a DataFrame indexed by source id (``count``, ``mean`` of
        f(rows) - about, ``mean_square``).  ``group`` (a ``torch.distributed`` group): this frame is one rank's
        share of a sharded trace; the three sums per source are added across the ranks before dividing."""
        import torch

        from . import engine

        column = 15 if quantity == "axis_intercept" else _INDEX[quantity]
        how = {None: 0, "sin": 1}[transform]
        self._need("generation", "surface", "id", *(("x0", "y0", "x_tilt", "y_tilt") if column == 15 else (quantity,)))
        if generation == "last":
            generation = self.last_generation_number()
            if group is not None:
                generation = int(_all_reduce_max(-1.0 if generation is None else float(generation), group, None,
                                                 self.rows.device))
            if generation is None or generation < 0:
                generation = 0
        if rays_per_source:
            if n_groups is None:
                top = float(self["id"].max()) if len(self) else -1.0
                if group is not None:
                    top = _all_reduce_max(top, group, None, self.rows.device)
                n_groups = max(1, int(top // rays_per_source) + 1)
        else:
            n_groups = 1
        rows = self.rows if self.rows.stride(1) == 1 else self.rows.contiguous()
        dev = rows.device
        sums = torch.empty((n_groups, 3), dtype=torch.float64, device=dev)
        nan = float("nan")
        engine._check(engine.library().prt_frame_mean_square(
            dev.index or 0, rows.data_ptr(), max(rows.stride(0), 1), rows.shape[1],
            nan if surface is None else float(surface), nan if generation is None else float(generation),
            float(rays_per_source or 0), n_groups, column, how, float(about), sums.data_ptr(),
            engine._stream_ptr(torch, dev)))
        if group is not None:
            sums = _all_reduce_sum(sums, group)
        sums = sums.cpu().numpy()
        with np.errstate(invalid="ignore", divide="ignore"):
            mean, mean_square = sums[:, 1] / sums[:, 0], sums[:, 2] / sums[:, 0]
        if not rays_per_source:
            return float(mean_square[0])
        frame = pd.DataFrame({"count": sums[:, 0].astype(np.int64), "mean": mean, "mean_square": mean_square})
        frame.index.name = "source_id"
        return frame

    def axis_intercept(self):
        """x where each ray's line crosses the optical (x) axis in the xy plane, from the segment's start point as the
        notebook writes it (cells 12, 15): ``x0 - x_tilt * y0 / y_tilt``."""
        return self["x0"] - self["x_tilt"] * self["y0"] / self["y_tilt"]

    def spot(self, plane=("y1", "z1")):
        """(centroid, rms radius) of the end points in a transverse plane."""
        if tuple(plane) == ("y1", "z1") and getattr(self.rows, "is_cuda", False) and len(self):
            stats = self.group_stats().iloc[0]
            return (float(stats["y"]), float(stats["z"])), float(stats["rms_radius"])
        a, b = self[plane[0]], self[plane[1]]
        ca, cb = a.mean(), b.mean()
        rms = (((a - ca) ** 2 + (b - cb) ** 2).mean()) ** 0.5
        return (float(ca), float(cb)), float(rms)

    def axis_crossing(self):
        """x where each ray of this frame crosses the optical (x) axis in the xy plane:
        x1 - y1 * x_tilt / y_tilt (the paraxial-focus estimate of the lens-design notebook)."""
        return self["x1"] - self["y1"] * self["x_tilt"] / self["y_tilt"]

    # --- export -------------------------------------------------------------------------------------
    def to_numpy(self):
        """(R, columns) float64 view of a host copy (one D2H transfer; a frame recorded with a column list brings only
        those columns across)."""
        from . import engine

        if self.written is not None:
            return engine.to_host(self.rows[list(self.written)]).T  # (the written columns gathered on the device first)
        return engine.to_host(self.rows).T  # a strided view crosses PCIe as it is: no device-side repack

    def to_pandas(self):
        values = self.to_numpy()
        if values.shape[0] == 0:
            return pd.DataFrame(columns=self.columns, dtype="float64")
        return pd.DataFrame(values, columns=self.columns, copy=False)


class SinkStats:
    """The sums a trace under a ``RecordPlan(stats=True)`` accumulated in its generation kernels -- per generation and
    per group what ``prt_frame_reduce`` / ``prt_frame_mean_square`` would have read back out of the stored frame --
    turned into the tables ``DeviceFrame.group_stats`` / ``mean_square`` return.  One small device-to-host copy (96
    bytes per generation and group); the frame itself was never written."""

    def __init__(self, sums, pivots=None, about=0.0):
        """sums: (generations, n_groups, 12) tensor or array; pivots: (n_groups, 3) the sums were taken about."""
        host = sums.detach().cpu().numpy() if hasattr(sums, "detach") else np.asarray(sums)
        self.sums = np.array(host, dtype=float)
        piv = None if pivots is None else (pivots.detach().cpu().numpy() if hasattr(pivots, "detach") else np.asarray(pivots))
        self.pivots = np.zeros((self.sums.shape[1], 3)) if piv is None else np.array(piv, dtype=float)

    def last_generation_number(self):
        """The highest generation that counted a row (None: no row passed)."""
        working = np.nonzero(self.sums[:, :, 0].sum(axis=1) > 0)[0]
        return int(working[-1]) if len(working) else None

    def _block(self, generation):
        if generation is None:
            return self.sums.sum(axis=0)   # (additive over generations: one set of pivots)
        if generation == "last":
            generation = self.last_generation_number()
            if generation is None:
                return np.zeros(self.sums.shape[1:])
        if not 0 <= int(generation) < self.sums.shape[0]:
            return np.zeros(self.sums.shape[1:])
        return self.sums[int(generation)]

    def group_stats(self, generation=None):
        """The table of ``DeviceFrame.group_stats`` (count, y, z, rms_radius, focus, focus_std, wavelength, intensity per
        source) for the rows the plan let pass, of one generation (a number, or "last": the highest that counted a
        row) or of all of them (None).  Same arithmetic as ``k_frame_finish`` (csrc/prt_frame.hpp)."""
        frame = pd.DataFrame(self.values(generation))
        frame.index.name = "source_id"
        return frame

    def values(self, generation=None):
        """``group_stats`` as a dict of numpy arrays (one entry per source): what a merit function reads in a loop,
        without the 0.2 ms a DataFrame takes to build."""
        s = self._block(generation)
        count, with_focus = s[:, 0], s[:, 8]
        with np.errstate(invalid="ignore", divide="ignore"):
            safe, safe_f = np.where(count > 0, count, 1.0), np.where(with_focus > 0, with_focus, 1.0)
            dy, dz, df = s[:, 1] / safe, s[:, 2] / safe, s[:, 4] / safe_f
            var_r = np.maximum(s[:, 3] / safe - dy * dy - dz * dz, 0.0)
            var_f = np.maximum(s[:, 5] / safe_f - df * df, 0.0)
            nan = np.nan
            return {
                "count": count.astype(np.int64),
                "y": np.where(count > 0, self.pivots[:, 0] + dy, nan), "z": np.where(count > 0, self.pivots[:, 1] + dz, nan),
                "rms_radius": np.where(count > 0, np.sqrt(var_r), nan),
                "focus": np.where(with_focus > 0, self.pivots[:, 2] + df, nan),
                "focus_std": np.where(with_focus > 0, np.sqrt(var_f), nan),
                "wavelength": np.where(count > 0, s[:, 6] / safe, nan), "intensity": np.where(count > 0, s[:, 7] / safe, nan),
            }

    def mean_square(self, generation=None, per_source=False):
        """``np.mean(np.square(f(rows) - about))`` of the plan's ``mean_square`` quantity (``DeviceFrame.mean_square``):
        a float, or per source a DataFrame (count, mean, mean_square)."""
        s = self._block(generation)
        with np.errstate(invalid="ignore", divide="ignore"):
            mean, mean_square = s[:, 10] / s[:, 9], s[:, 11] / s[:, 9]
        if not per_source:
            return float(mean_square[0])
        frame = pd.DataFrame({"count": s[:, 9].astype(np.int64), "mean": mean, "mean_square": mean_square})
        frame.index.name = "source_id"
        return frame


def _all_reduce_sum(tensor, group):
    """Sum of a small device tensor over the ranks of a torch.distributed group (through the host for a backend
    that does not take device tensors)."""
    import torch.distributed as dist

    if dist.get_backend(group) == "nccl":
        dist.all_reduce(tensor, group=group)
        return tensor
    host = tensor.cpu()
    dist.all_reduce(host, group=group)
    return host.to(tensor.device)


def _all_reduce_max(value, group, comm, device):
    """Largest of a host number over the ranks (the highest ray id: how many source groups there are)."""
    import torch
    import torch.distributed as dist

    if group is None:  # a bare library communicator: the count is the caller's to give
        raise ValueError("group_stats over a LibraryComm needs n_groups (or pass the torch.distributed group as well)")
    on = device if dist.get_backend(group) == "nccl" else torch.device("cpu")
    box = torch.tensor([value], dtype=torch.float64, device=on)
    dist.all_reduce(box, op=dist.ReduceOp.MAX, group=group)
    return float(box[0])
