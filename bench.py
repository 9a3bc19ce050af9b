#!/usr/bin/env python3
"""bench.py -- ray-surface intersections/sec on the north-star workload (BASELINE.json config 2).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no launcher (no WORLD_SIZE in the environment) the process that was started
becomes a launcher itself: before anything touches the GPU it picks a free port and starts N rank
processes of this same file (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), passes rank 0's JSON line
through as its only stdout and exits non-zero if any rank fails (`spawn_ranks`).

One *step* = one pass of the hot path (every generation: intersect + nearest hit + shade +
compaction + record rows) over the job's rays, already resident in HBM.  The metric counts
result rows (one row = one ray segment resolved to its nearest surface and shaded), aggregated
over all ranks, divided by the max-over-ranks wall time of the K timed steps (barrier +
synchronize on both sides).

Scaling is strong (BASELINE.json: "1M-ray biconvex lens, 1/2/4/8 GPU"): ONE seeded 1M-ray job;
rank r traces the contiguous id range [r n/G, (r+1) n/G) of the same ray set (rays are
independent, the trace needs no collective).  `--scaling weak` gives every rank its own 1M rays
instead (secondary figure).  The re-assembly of the result frame (RCCL all-gathers + a placement
kernel inside the library, pyrayt_amd/csrc/prt_gather.hpp) is timed separately after the timed
region and reported under "gather" and "value_with_gather" -- it is not part of `value`.

The timed steps ROTATE through four distinct seeded ray sets of the workload (config 2: seeds 1234 ... 1237, all
resident in HBM before the timed region), so that no step re-traces its predecessor's rays: the per-tile
compaction records of the previous trace never apply and every step reads its input from HBM.  They are issued with
2-3 traces in flight, each ticket of the library on its own HIP stream (prt_trace_batch: the loop over
prt_trace_begin / prt_trace_end, run by the library; --python-loop runs it from here): the host enqueues ahead and the
kernels of different traces overlap on the device.

ONE regime describes the line.  The region of K steps (barrier + synchronize on both sides) is timed --reps times
(default 5); `value` / `ms_per_step` are those of the MEDIAN repetition, `repetitions` carries all of them with min
and max.  `roofline` prices that same repetition: every trace of the region is bracketed by a pair of HIP events of
its own on its own stream (PRT_TRACE_BUSY), and behind the region the library merges the intervals
(prt_trace_batch_busy): `kernel_ms_per_step` = the time the device had at least one of the region's traces in
flight, per step -- never more than `ms_per_step` -- and `avg_launch_ms` = that time per generation launch.  The
kernel's duration when it has the device to itself (what `rocprofv3 --stats` averages when the steps run on one
stream) is `roofline.one_stream`, measured right behind the timed region, with `value_one_stream` next to it.

After the timed region the rows of the LAST timed step of EACH of the four ray sets are compared with the reference's
own summary of that run (tests/golden/config2_1m_summary*.npz, written by the genuine reference: rows per generation x
surface, the ids of the near-axial rays of SURVEY Q5, a checksum of the surface column, column sums, sampled rows to
1e-6) -> "verified" / `verification.seeds`; a false verdict exits non-zero.  With --gpus N the ranks' shards are
checked together (counts and sums are additive), BEFORE the timed region.

Other kinds of step, untimed side runs: `value_replay` (the same ray set again and again: round 3's headline),
`value_synchronous` (one blocking prt_trace at a time: what one RayTracer.trace(), pyrayt/_pyrayt.py:329-339, costs in
a running design loop), `value_cold` (a fresh scene's first trace), `value_no_hints`, `value_first_trace`.

rank 0 prints ONE JSON line.  `roofline` is for the generation kernel(s): algorithmic bytes (328 B per ray alive at
generation entry that is recorded and goes on) over the busy time above; the bytes the kernel
actually moves (it carries 10 of the 13 state rows between generations) are reported next to it as
`moved_bytes_per_launch` / `moved_frac`.  `cpu_baseline` is the numpy oracle (a port of the reference's path,
validated against it) timed on this host, rank 0, N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    "config2": "BASELINE config 2: biconvex_lens(2,2,0.25,aperture=1) + detector baffle, seeded 6 degree cone at -f",
    "config3": "[informational] BASELINE config 3: Cooke-style triplet (3 thick lenses) + aperture stop + detector, "
               "seeded 4 degree cone",
    "config4": "[informational] BASELINE config 4: equilateral BK7 prism + detector, 8 wavelengths of LineOfRays",
    "config5": "[informational] BASELINE config 5: paraboloid & cylinder CSG condenser + baffle",
}
RAYS_PER_GPU = 1_000_000
GENERATION_LIMIT = 10
BYTES_PER_RAY_GENERATION = 328  # 104 B state read + 104 B state write + 120 B record row
# what the kernel moves: between generations the state goes without its rows 3, 7, 8 (w = 1, w = +0,
# generation number: the same in every ray, include/prt.h "compact state"); generation 0 reads the
# caller's 13 rows
# ... and where a wave's rays stay in place, with one intensity, one wavelength and ids counting up (ray sets as sources
# emit them: every BASELINE workload), those three rows travel as three numbers per wave ("lean segments", round 5): 56 B
STATE_BYTES, STATE_BYTES_LEAN, STATE_BYTES_FULL, ROW_BYTES = 80, 56, 104, 120
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
COPY_GBS = 6300.0               # what a bare copy with the kernel's row pattern reaches (tools/ubench/copy_f64: 5.7-6.4 TB/s)


# What one rank does at the shard sizes of the 1/2/4/8 curve, measured on ONE MI355X (profiles/r6/shard_scaling.txt:
# config 2, the library's own loop with 2-4 traces in flight, and one blocking prt_trace at a time) -- the stated
# expectation a first multi-GPU record can be read against (no N > 1 run has happened on hardware yet).
SHARD_MS_PER_STEP = {1_000_000: {"overlapped": 0.1357, "synchronous": 0.1760},
                     500_000: {"overlapped": 0.0705, "synchronous": 0.1155},
                     250_000: {"overlapped": 0.0393, "synchronous": 0.0777},
                     125_000: {"overlapped": 0.0232, "synchronous": 0.0623}}
XGMI_LINK_GBS, XGMI_LINKS = 153.0, 7  # 7 links x ~153 GB/s per GPU (~76 per direction if that figure is bidirectional)


def scaling_model(n_job, world, rows_per_step):
    """Expected step time and aggregate rate of `world` ranks on an `n_job`-ray job, from single-GPU measurements of a
    rank's shard, and what re-assembling the frame on every GPU would add (bytes into each GPU; bound by one link for
    a ring schedule, by all seven for a direct one)."""
    shard = max(1, n_job // max(world, 1))
    sizes = sorted(SHARD_MS_PER_STEP)
    lo = max([k for k in sizes if k <= shard] or [sizes[0]])
    hi = min([k for k in sizes if k >= shard] or [sizes[-1]])
    model = {"shard_rays": shard, "from": "profiles/r6/shard_scaling.txt (one GPU tracing a rank's shard; config 2)"}
    for mode in ("overlapped", "synchronous"):
        a, b = SHARD_MS_PER_STEP[lo][mode], SHARD_MS_PER_STEP[hi][mode]
        ms = a if hi == lo else a + (b - a) * (shard - lo) / (hi - lo)
        model[mode] = {"ms_per_step": ms, "value": rows_per_step / (ms * 1e-3),
                       "speedup_over_1_gpu": SHARD_MS_PER_STEP[1_000_000][mode] / ms * (n_job / 1_000_000)}
    into_each = rows_per_step * 120.0 * (world - 1) / max(world, 1)
    model["gather"] = {"bytes_into_each_gpu": into_each,
                       "ring_ms_one_link": [into_each / (XGMI_LINK_GBS * 1e9) * 1e3, into_each / (XGMI_LINK_GBS / 2 * 1e9) * 1e3],
                       "direct_ms_all_links": [into_each / (XGMI_LINK_GBS * XGMI_LINKS * 1e9) * 1e3,
                                               into_each / (XGMI_LINK_GBS / 2 * XGMI_LINKS * 1e9) * 1e3],
                       "what": "every rank receives the other ranks' rows (ncclAllGather per record column): link-bound if "
                               "RCCL schedules a ring, seven links in parallel if it pushes directly; each pair = the "
                               "153 GB/s figure taken per direction / as bidirectional.  `value` (no gather in the timed "
                               "region) is what the model's `overlapped` line predicts; value_with_gather adds this"}
    model["note"] = ("strong scaling of one 1M-ray job: at N = 8 a rank's shard is 125k rays, less than one round of workgroups "
                     "(1280 x 256 rays resident), so a step is a chain of three launches each about as long as one wave lives; "
                     "the expectation is ~5.8x at N = 8 with four chains in flight, ~2.8x with one blocking trace at a time")
    return model


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5,
                    help="how many times the region of --steps steps is timed: value / ms_per_step are the median "
                         "repetition's, min and max are reported next to them")
    ap.add_argument("--spinup-ms", type=float, default=60.0,
                    help="untimed traces before the warmup steps until this much wall time has passed: a GPU "
                         "that has been idle runs its first ~10 ms of work at lower clocks (measured: 7 %% on "
                         "the kernel), which a short --steps/--warmup would otherwise be quoted on")
    ap.add_argument("--rays", type=int, default=RAYS_PER_GPU,
                    help="rays of the job (strong scaling) / per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--flags", type=int, default=0, help="PRT_TRACE_* flags")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config4", "config5"],
                    help="informational runs of the other BASELINE scenes; the bench line is config2")
    ap.add_argument("--generation-limit", type=int, default=GENERATION_LIMIT,
                    help="experiments only: the north-star workload uses 10")
    ap.add_argument("--cpu-rays", type=int, default=1_000_000,
                    help="rays of the same workload timed on the CPU oracle")
    ap.add_argument("--options", default="",
                    help="A/B runs only: prt_scene_options for the scene, e.g. no_tail=1,no_cull=1 (the bench line uses none)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="time synchronous traces (prt_trace) instead of keeping one trace in flight "
                         "(prt_trace_begin / prt_trace_end)")
    ap.add_argument("--streams", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="traces in flight, each on its own HIP stream (their kernels overlap on the device); "
                         "1 = one stream with one trace ahead; 0 (default) = as many as keep their next states "
                         "(80 B per ray) within ~200 MB of the Infinity Cache, at most 3")
    ap.add_argument("--python-loop", action="store_true",
                    help="issue the overlapped steps from a Python loop over prt_trace_begin / prt_trace_end "
                         "(DeviceScene.trace_many's form) instead of one prt_trace_batch call per region")
    ap.add_argument("--ray-sets", type=int, default=4,
                    help="distinct seeded ray sets the timed steps rotate through (1 = re-trace the same rays: the "
                         "value_replay form)")
    ap.add_argument("--side-steps", type=int, default=40,
                    help="steps of each untimed side measurement (synchronous / no hints / changing ray count)")
    return ap.parse_args()


def summary_file(seed):
    """The reference's own summary of the 1M-ray config-2 job for one seed of the rotation (tests/golden/generate_golden.py)."""
    name = "config2_1m_summary.npz" if seed == 1234 else f"config2_1m_summary_seed{seed}.npz"
    return os.path.join(ROOT, "tests", "golden", name)


def summary_checks(frame, want):
    """Rows of a whole 1M-ray config-2 trace, (R, 15) on the host, against the reference's summary of that run:
    rows per generation x surface, the ids of the near-axial rays of SURVEY Q5, a checksum of the surface column,
    column sums, sampled rows to 1e-6."""
    import numpy as np

    gens, surf = frame[:, 0].astype(np.int64), frame[:, 5].astype(np.int64)
    checks = {"rows": frame.shape[0] == int(want["rows"])}
    if checks["rows"]:
        pairs, pair_counts = np.unique(np.stack((gens, surf)), axis=1, return_counts=True)
        checks["rows_per_generation_x_surface"] = bool(np.array_equal(pairs, want["gen_surface_pairs"]) and
                                                       np.array_equal(pair_counts, want["gen_surface_counts"]))
    if not checks.get("rows_per_generation_x_surface", False):
        # say WHICH generation x surface differs: the first thing anybody debugging a first multi-GPU run wants to know
        pairs, pair_counts = np.unique(np.stack((gens, surf)), axis=1, return_counts=True)
        got = {tuple(p): int(c) for p, c in zip(pairs.T.tolist(), pair_counts)}
        ref = {tuple(p): int(c) for p, c in zip(np.asarray(want["gen_surface_pairs"]).T.tolist(), want["gen_surface_counts"])}
        for key in sorted(set(got) | set(ref)):
            if got.get(key, 0) != ref.get(key, 0):
                print(f"bench.py: generation {key[0]} x surface {key[1]}: {got.get(key, 0)} rows, the reference has "
                      f"{ref.get(key, 0)}", file=sys.stderr, flush=True)
    if checks["rows"]:
        checks["q5_ids"] = bool(np.array_equal(frame[(gens == 1) & (surf == surf.max()), 4].astype(np.int64), want["q5_ids"]))
        checks["surface_checksum"] = int((surf * (gens + 1)).sum()) == int(want["surface_checksum"])
        checks["column_sums"] = bool(np.allclose(frame.sum(axis=0), want["column_sums"], rtol=1e-9, atol=1e-3))
        checks["sample_rows_1e-6"] = bool(np.allclose(frame[want["sample_index"]], want["sample_rows"], rtol=0, atol=1e-6))
    return checks


def _free_port():
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def spawn_ranks(world, argv, worker=None, poll_s=0.05):
    """Start `world` rank processes of this benchmark (one per GPU) and wait for them.

    Runs in a parent that has not touched the GPU (and never will: no torch import, no library
    load) -- a process that has initialised HIP must not be replaced or forked on this pool, so the
    ranks are fresh interpreters.  Rank 0 inherits this process's stdout (its one JSON line is the
    launcher's only stdout); the other ranks' stdout goes to stderr.  Returns the exit code: 0 when
    every rank exited 0, otherwise the first non-zero code seen (the remaining ranks are stopped by
    PID).  `worker` (tests): the command to run instead of [python, bench.py].
    Libraries write to stdout too (gloo announces its connections there): of rank 0's stdout only the
    lines that are a JSON object are passed on, the rest goes to stderr with everything else."""
    import subprocess
    import threading

    def forward(pipe):
        for raw in iter(pipe.readline, b""):
            text = raw.decode("utf-8", "replace")
            target = sys.stdout if text.lstrip().startswith("{") else sys.stderr
            target.write(text)
            target.flush()
        pipe.close()

    port = _free_port()
    cmd = list(worker) if worker else [sys.executable, os.path.abspath(__file__)]
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if rank == 0 else sys.stderr))
    reader = threading.Thread(target=forward, args=(procs[0].stdout,), daemon=True)
    reader.start()
    code = 0
    pending = list(procs)
    while pending and code == 0:
        for proc in list(pending):
            rc = proc.poll()
            if rc is None:
                continue
            pending.remove(proc)
            if rc != 0:
                code = rc
                break
        else:
            time.sleep(poll_s)
    for proc in pending:  # a rank failed: the others would wait for it in a barrier forever
        proc.terminate()
    for proc in pending:
        try:
            proc.wait(timeout=10)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.wait()
    reader.join(timeout=10)
    return code


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: become one (nothing above or in here touches the GPU)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    # The HIP runtime maps streams onto hardware queues, four by default (the null stream holds one): a fourth
    # ticket stream would share a queue with another and serialise with it.  Small shards gain from a fourth trace
    # in flight (profiles/r3/streams.txt: 125k rays 24.7 -> 23.1 us per step), so the package asks for eight queues
    # when it is imported (pyrayt_amd/_runtime.py: in force as long as no HIP call has been made yet, whichever of
    # torch and pyrayt_amd comes first); a setting the user made stays.
    import numpy as np
    import torch
    from pyrayt_amd import engine

    import scenes
    from pyrayt_amd import distributed as pdist
    from pyrayt_amd import engine
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    n_devices = torch.cuda.device_count()
    # collectives run over RCCL ("nccl") on device tensors; PRT_DIST_BACKEND=gloo (CPU tensors) only
    # exists so that the multi-rank code path can be exercised on a single-GPU box
    backend = os.environ.get("PRT_DIST_BACKEND", "nccl")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if distributed and backend == "nccl" and n_devices < local_world:
        raise SystemExit(f"{local_world} ranks on this node but {n_devices} GPU(s) visible: RCCL needs one device per "
                         "rank (PRT_DIST_BACKEND=gloo lets several ranks share a GPU, for plumbing checks only)")
    local_device = local_rank % max(1, n_devices)
    torch.cuda.set_device(local_device)
    device = torch.device("cuda", local_device)
    comm_device = device if backend == "nccl" else torch.device("cpu")
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    # the workload: this rank's contiguous id range of the job
    n_job = args.rays
    strong = args.scaling == "strong"
    seed_shift = 0 if strong else 16 * rank
    base_seed = {"config2": 1234, "config3": 7, "config5": 11}.get(args.workload)  # (config 4 has no random rays)

    def build(k):
        """The scene and the k-th ray set of the workload (set 0 is the BASELINE one)."""
        CountedObject.reset_ids()
        if args.workload == "config4":
            return scenes.config4(scenes.product_api(), n_job // 8)
        return scenes.SCENES[args.workload](scenes.product_api(), n_job, seed=base_seed + seed_shift + k)

    n_sets = max(1, args.ray_sets) if base_seed is not None else 1
    parts, rays = build(0)
    n_job = rays.shape[1]
    lo, hi = pdist.shard_bounds(n_job, rank=rank, world=world) if strong else (0, n_job)

    def shard(block):
        if strong:  # one job, sharded by contiguous id range (ids are already global)
            return np.ascontiguousarray(block[:, lo:hi])
        block[12] += rank * n_job
        return block

    rays = shard(rays)
    n = rays.shape[1]
    snap = SceneSnapshot(parts)
    scene_options = {k: int(v or 1) for k, _, v in (item.partition("=") for item in args.options.split(",") if item)}
    scene = engine.DeviceScene(snap, options=scene_options)
    rays_dev = torch.from_numpy(rays).to(device)
    # the other ray sets of the rotation: same scene (the snapshot above), other seeds, resident in HBM
    ray_sets = [rays_dev] + [torch.from_numpy(shard(build(k)[1])).to(device) for k in range(1, n_sets)]

    limit = args.generation_limit
    pipelined = not args.no_pipeline
    # How the timed steps are issued.  "overlap": prt_trace_begin / prt_trace_end with `depth` traces in
    # flight, each ticket on its own HIP stream -- the host enqueues ahead AND the kernels of different
    # traces overlap on the device (a generation's workgroups leave the chip partly idle while they start
    # up and drain; another trace's kernels fill that).  "one_stream": the same with two tickets on one
    # stream (host hidden, kernels strictly one after the other).  "sync": prt_trace, one call at a time.
    # Measured on config 2 (profiles/r3/streams.txt): 1M rays 0.179 -> 0.151 ms per step with two streams
    # (three: 0.155), 125k rays 0.050 -> 0.040 -> 0.034 with two / three.
    # How many: the state a generation hands to the next (80 B per ray) is meant to stay in the 256 MB Infinity
    # Cache; traces in flight together share it.  As many traces as keep their next states within ~200 MB,
    # at most three (four for shards of 125k rays: see GPU_MAX_HW_QUEUES above): 1M rays -> 2, 500k and less -> 3,
    # 2M and more -> 1 (measured: config 2 at 1M rays 0.142 ms
    # with two against 0.146 with three; config 5 at 2M and config 4 at 8M lose 3-4 % with two).
    hw_queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)
    most = 4 if (hw_queues >= 8 and n <= 160_000) else 3  # (a fourth only where it was measured to pay, and has a queue)
    streams_wanted = args.streams if args.streams else max(1, min(most, int(200e6 // (80 * max(n, 1)))))
    mode = "sync" if not pipelined else ("overlap" if streams_wanted >= 2 else "one_stream")
    depth = streams_wanted if mode == "overlap" else 2  # traces in flight, the one being collected included
    # record blocks handed back to every step (what a design loop does once it has consumed the previous
    # frame); n * limit columns always suffice; every ticket in flight records into its own block
    # (... and one per ray set of the rotation, so that the LAST timed step of every ray set is still there to be
    # verified behind the timed region)
    blocks = [torch.empty((engine.RECORD_COLS, max(n, 1) * limit), dtype=torch.float64, device=device)
              for _ in range(max(depth, 2, n_sets))]
    block = blocks[0]

    def step(flags=args.flags, rays_in=None):
        return scene.trace(rays_dev if rays_in is None else rays_in, limit, flags=flags, out=block)

    def ray_set(k, rotate=True):
        return ray_sets[k % len(ray_sets)] if rotate else rays_dev

    streams = scene.ticket_streams(device, depth) if mode == "overlap" else None
    torch.cuda.synchronize(device)

    class Totals:
        def __init__(self):
            self.kernel_ms = self.launches = self.ray_generations = self.rows_recorded = self.rays_carried = 0.0
            self.rows_returned = 0  # rows of every step of the region, summed from what the traces returned
            self.busy = None        # overlapped regions issued by prt_trace_batch: the merged busy intervals

        def add(self, times=1):
            st = scene.trace_stats()
            self.kernel_ms += st["kernel_ms"] * times
            self.launches += st["kernel_launches"] * times
            self.ray_generations += st["ray_generations"] * times
            self.rows_recorded += st["rows"] * times
            self.rays_carried += st["rays_carried"] * times

        def add_set(self, k, times=1):
            """Statistics of ray set k's trace (taken once, below: an overlapped region cannot ask per trace)."""
            st = set_stats[k % len(set_stats)]
            self.launches += st["kernel_launches"] * times
            self.ray_generations += st["ray_generations"] * times
            self.rows_recorded += st["rows"] * times
            self.rays_carried += st["rays_carried"] * times

    batches = {}

    def prepared_batch(count, flags=args.flags, rotate=True):
        """The job table of `count` overlapped steps (built once per region size, outside the timed region)."""
        key = (count, flags, rotate)
        if key not in batches:
            # (every trace bracketed by its own pair of HIP events on its own stream, merged behind the region by the
            # library: PRT_TRACE_BUSY -> batch.busy())
            batches[key] = engine.TraceBatch(scene, [ray_set(k, rotate) for k in range(count)], limit, depth=depth,
                                             outs=blocks[:max(depth, len(ray_sets) if rotate else 1)],
                                             flags=flags | engine.TRACE_NO_TIMING | engine.TRACE_BUSY)
        return batches[key]

    def run_steps(count, totals=None, flags=args.flags, how=None, rotate=True):
        """`count` traces back to back, issued as `how` says (default: the bench's mode), step k on ray set
        k mod n_sets (rotate=False: all on set 0); returns (rows, counts) of the last one."""
        how = how or mode
        rows = counts = None
        if how == "sync":
            for k in range(count):
                rows, counts = step(flags, ray_set(k, rotate))
                if totals is not None:
                    totals.add()
                    totals.rows_returned += int(rows.shape[1])
            return rows, counts
        lanes = depth if how == "overlap" else 2
        if how == "overlap" and not args.python_loop:
            # the whole region as ONE library call (prt_trace_batch: the loop below, run by the library): what a
            # caller with its ray sets up front uses; no Python inside the timed region.  (Same step time as the
            # Python loop, --python-loop, at every shard size: profiles/r3/batch_issue.txt)
            batch = prepared_batch(count, flags, rotate)
            batch.run()
            if totals is not None:
                totals.busy = batch.busy()
                for k in range(min(count, len(ray_sets))):  # steps k, k + n_sets, ... trace set k
                    totals.add_set(k if rotate else 0, len(range(k, count, len(ray_sets))))
                totals.rows_returned += int(batch.totals.sum())
            torch.cuda.synchronize(device)
            return batch.result(-1)

        # overlapped traces are not bracketed with HIP events (an event pair would also see the other traces'
        # kernels; the generation kernel's own time comes from the one-stream region) and their statistics
        # are read once: every step of the region is the same trace
        if how == "overlap":
            flags = flags | engine.TRACE_NO_TIMING

        def begin(k):
            scene.trace_begin(k % lanes, ray_set(k, rotate), limit, blocks[k % lanes], flags=flags,
                              stream=streams[k % lanes] if how == "overlap" else None)

        for k in range(min(lanes - 1, count)):
            begin(k)
        for k in range(count):
            if k + lanes - 1 < count:
                begin(k + lanes - 1)
            rows, counts = scene.trace_end(k % lanes)
            if totals is not None:
                totals.rows_returned += int(rows.shape[1])
                if how == "overlap":
                    totals.add_set(k if rotate else 0)
                else:
                    totals.add()
        if how == "overlap":
            torch.cuda.synchronize(device)  # (the caller reads `rows` on the current stream)
        return rows, counts

    # what each ray set's trace does (launches, rays per generation, rows), asked once per set: an overlapped
    # region cannot ask trace by trace
    set_stats = []
    for k in range(len(ray_sets)):
        step(args.flags, ray_sets[k])
        step(args.flags, ray_sets[k])  # (the second trace of a set launches exactly its working generations)
        set_stats.append(scene.trace_stats())
    # ---- several GPUs: before anything is timed, the shards together against the reference's summaries --------------
    # Every rank traces its shard of each ray set; the frame is re-assembled on rank 0 through the path a user takes
    # (RCCL all-gathers + placement kernel inside the library when the group is RCCL-backed) and compared like a
    # single-GPU frame.  A mismatch ends the run, non-zero, on every rank: a scaling number for wrong rows is worth
    # nothing.  If the re-assembly itself fails (first contact with a new node), the shards are checked through their
    # additive statistics instead -- per generation x surface counts, checksum, column sums, Q5 ids -- and the line
    # says so.  `rccl_ranks` is what ncclCommCount reports for the library's communicator.
    pre_verification, rccl_ranks, bench_comm = None, None, None
    verifiable_job = (args.workload == "config2" and n_job == RAYS_PER_GPU and limit == GENERATION_LIMIT and strong)
    if distributed:
        if backend == "nccl":
            try:
                bench_comm = pdist.LibraryComm.from_group(dist.group.WORLD, device)
                rccl_ranks = bench_comm.info()["ranks"]
            except TimeoutError as exc:
                # fatal: the helper thread is still inside ncclCommInitRank and cannot be cancelled -- if the missing
                # rank turned up later it would complete a communicator nobody owns, and the ranks would disagree
                # about who has one.  Nothing is timed; the process leaves without running destructors behind it.
                print(json.dumps({"metric": "ray-surface intersections/sec, 1M-ray biconvex lens", "value": None,
                                  "n_gpus": world, "error": f"rank {rank}: {exc}"[:400]}), flush=True)
                os._exit(3)
            except Exception as exc:  # noqa: BLE001
                rccl_ranks = f"{type(exc).__name__}: {exc}"[:200]
                bench_comm = None
            # every rank uses the library's communicator or none does: a rank without one would answer the others'
            # ncclAllGather with a torch.distributed collective and both would wait for ever
            have = torch.tensor([1.0 if bench_comm is not None else 0.0], dtype=torch.float64, device=comm_device)
            dist.all_reduce(have, op=dist.ReduceOp.MIN)
            if float(have[0]) < 1.0 and bench_comm is not None:
                bench_comm.close()
                bench_comm = None
                rccl_ranks = "dropped: another rank could not create the library's communicator"
        if verifiable_job:
            per_seed, how, failed = {}, "frame re-assembled on rank 0 (pyrayt_amd.distributed.assemble_rows)", None
            for k in range(len(ray_sets)):
                seed = base_seed + k
                if not os.path.exists(summary_file(seed)):
                    continue
                want = np.load(summary_file(seed))
                got, got_counts = scene.trace(ray_sets[k], limit, flags=args.flags)
                checks = None
                if failed is None:
                    try:
                        full, _ = pdist.assemble_rows(got, got_counts, limit, dist.group.WORLD, "root", comm=bench_comm)
                        torch.cuda.synchronize(device)
                        if rank == 0:
                            checks = summary_checks(engine.to_host(full).T, want)
                        del full
                    except Exception as exc:  # noqa: BLE001
                        failed = f"{type(exc).__name__}: {exc}"[:200]
                    # (every rank takes the same path from here: one that failed tells the others)
                    flag = torch.tensor([0.0 if failed is None else 1.0], dtype=torch.float64, device=comm_device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                    if float(flag[0]) > 0:
                        failed = failed or "another rank's re-assembly failed"
                        how = f"additive statistics of the shards (the re-assembly failed: {failed})"
                        checks = None
                if failed is not None:
                    frame = engine.to_host(got).T
                    gens, surf = frame[:, 0].astype(np.int64), frame[:, 5].astype(np.int64)
                    pairs, pair_counts = np.unique(np.stack((gens, surf)), axis=1, return_counts=True)
                    mine = {"rows": frame.shape[0], "pairs": {tuple(p): int(c) for p, c in zip(pairs.T.tolist(), pair_counts)},
                            "checksum": int((surf * (gens + 1)).sum()), "sums": frame.sum(axis=0),
                            "q5": frame[(gens == 1) & (surf == int(want["gen_surface_pairs"][1].max())), 4].astype(np.int64)}
                    everyone = [None] * world
                    dist.all_gather_object(everyone, mine)
                    if rank == 0:
                        merged = {}
                        for part in everyone:
                            for key, c in part["pairs"].items():
                                merged[key] = merged.get(key, 0) + c
                        want_pairs = {tuple(p): int(c) for p, c in zip(want["gen_surface_pairs"].T.tolist(), want["gen_surface_counts"])}
                        checks = {"rows": sum(p["rows"] for p in everyone) == int(want["rows"]),
                                  "rows_per_generation_x_surface": merged == want_pairs,
                                  "q5_ids": bool(np.array_equal(np.sort(np.concatenate([p["q5"] for p in everyone])), want["q5_ids"])),
                                  "surface_checksum": sum(p["checksum"] for p in everyone) == int(want["surface_checksum"]),
                                  "column_sums": bool(np.allclose(sum(p["sums"] for p in everyone), want["column_sums"], rtol=1e-9, atol=1e-3))}
                if rank == 0:
                    per_seed[seed] = checks
            verdict = [None]
            if rank == 0 and per_seed:
                ok = all(all(c.values()) for c in per_seed.values())
                verdict[0] = (ok, {"against": "tests/golden/config2_1m_summary*.npz (written by the genuine reference)",
                                   "of": f"every ray set of the rotation, all {world} shards together, BEFORE the timed region: " + how,
                                   "seeds": sorted(per_seed), "checks": {str(k): v for k, v in sorted(per_seed.items())}})
            dist.broadcast_object_list(verdict, src=0)
            pre_verification = verdict[0]
            if pre_verification is not None and not pre_verification[0]:
                if rank == 0:
                    for seed_key, seed_checks in sorted((pre_verification[1].get("checks") or {}).items()):
                        failed_checks = [name for name, ok in (seed_checks or {}).items() if not ok]
                        if failed_checks:
                            print(f"bench.py: seed {seed_key}: failed {failed_checks}", file=sys.stderr, flush=True)
                    print(json.dumps({"metric": "ray-surface intersections/sec, 1M-ray biconvex lens", "value": None, "n_gpus": world,
                                      "verified": False, "verification": pre_verification[1]}), flush=True)
                dist.barrier()
                dist.destroy_process_group()
                raise SystemExit("bench.py: the shards' rows do NOT match the reference's summary: nothing was timed")
    spinup_steps = 0
    t_spin = time.perf_counter()
    while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
        run_steps(4)
        spinup_steps += 4
    rows, counts = run_steps(max(args.warmup, 1)) if args.warmup else step()
    if mode == "overlap" and not args.python_loop:
        prepared_batch(args.steps).run()  # (untimed: the region's job table and its pool of HIP events exist from here on)
    # ---- the timed region: --steps steps between barrier + synchronize, --reps times; the median repetition is the line
    repetitions = []
    for _ in range(max(1, args.reps)):
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        region = Totals()
        rows, counts = run_steps(args.steps, region)
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)
        region.elapsed = time.perf_counter() - t0
        repetitions.append(region)
    if distributed:  # a repetition takes as long as its slowest rank
        slowest = torch.tensor([r.elapsed for r in repetitions], dtype=torch.float64, device=comm_device)
        dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
        for r, took in zip(repetitions, slowest.tolist()):
            r.elapsed = took
    # the rows the LAST repetition's last steps left: one step per ray set of the rotation (verified further down,
    # before anything else records into those blocks)
    last_timed = {}
    if mode == "overlap" and not args.python_loop:
        done = prepared_batch(args.steps)
        for k in range(max(0, args.steps - len(ray_sets)), args.steps):
            last_timed[k % len(ray_sets)] = done.result(k)
    verifiable = verifiable_job
    if verifiable and world == 1:  # to the host now (one DMA each); looked at behind the side measurements
        last_timed = {k: (engine.to_host(r).T, c) for k, (r, c) in last_timed.items()}
    by_time = sorted(repetitions, key=lambda r: r.elapsed)
    timed = by_time[(len(by_time) - 1) // 2]  # the median (the faster of the middle two for an even count)
    elapsed = timed.elapsed
    ray_generations, rows_recorded, rays_carried = timed.ray_generations, timed.rows_recorded, timed.rays_carried
    launches = timed.launches
    # What the roofline prices is the timed region itself.  Overlapped traces (prt_trace_batch): every trace carries
    # its own pair of HIP events on its own stream and the library merges the intervals -- `kernel_ms` is the time the
    # device had at least one of the region's traces in flight.  One stream / synchronous: the HIP-event time of the
    # traces' launches (prt_trace_stats), which do not overlap.
    busy = timed.busy if (mode == "overlap" and timed.busy and timed.busy["traces"] > 0) else None
    if busy is not None:
        kernel_ms, kernel_launches = busy["union_ms"], launches
    else:
        kernel_ms, kernel_launches = timed.kernel_ms, launches
    # The generation kernel with the device to itself -- what `rocprofv3 --stats` averages when the same steps run on
    # one stream -- right behind the timed region (same process, same clocks): `roofline.one_stream`.
    def respin(how=None):
        """The GPU has been idle while the host looked at something (a copy to the host, a scene being built): untimed
        steps until the clocks are back up, as in front of the timed region."""
        r0 = time.perf_counter()
        while (time.perf_counter() - r0) * 1e3 < args.spinup_ms:
            run_steps(4, how=how)
        torch.cuda.synchronize(device)

    one_stream = None
    if mode == "overlap":
        respin("one_stream")
        run_steps(3, how="one_stream")
        torch.cuda.synchronize(device)
        alone = Totals()
        k0 = time.perf_counter()
        run_steps(args.steps, alone, how="one_stream")
        torch.cuda.synchronize(device)
        alone_s = time.perf_counter() - k0
        one_stream = {"ms_per_step": alone_s / args.steps * 1e3, "kernel_ms_per_step": alone.kernel_ms / args.steps,
                      "avg_launch_ms": alone.kernel_ms / alone.launches if alone.launches else 0.0,
                      "rows": alone.rows_returned,
                      "_bytes": 104.0 * alone.ray_generations + 120.0 * alone.rows_recorded + 104.0 * alone.rays_carried}
        if busy is None:  # (--python-loop: no merged intervals; the one-stream figures stand in, and the line says so)
            kernel_ms, kernel_launches = alone.kernel_ms, alone.launches
        run_steps(2)

    # side measurements (untimed region, every rank so that the ranks stay in step): the same step
    # (a) synchronous: one trace at a time through prt_trace, host and GPU strictly alternating;
    # (b) without the dense-mode hints of the previous trace (PRT_TRACE_NO_HINTS: what a first trace
    #     or a trace whose rays die differently runs on);
    # (c) with a ray count that changes from call to call (control words re-initialised every time; the hints
    #     of the previous trace still apply -- they are checked per tile): what a design loop that resizes its
    #     ray set every iteration sees.
    def side(label, fn, count):
        if count <= 0:
            return None
        respin()
        fn(3)
        torch.cuda.synchronize(device)
        tot = Totals()
        s0 = time.perf_counter()
        fn(count, tot)
        torch.cuda.synchronize(device)
        took = time.perf_counter() - s0
        return {"ms_per_step": took / count * 1e3, "kernel_ms_per_step": tot.kernel_ms / count,
                "avg_launch_ms": tot.kernel_ms / tot.launches if tot.launches else 0.0,
                "launches_per_step": tot.launches / count, "rows_per_s_this_gpu": tot.rows_returned / took,
                "_bytes": (104.0 * tot.ray_generations + 120.0 * tot.rows_recorded + 104.0 * tot.rays_carried),
                "_kernel_ms": tot.kernel_ms}

    def sync_steps(count, totals=None, flags=args.flags):
        run_steps(count, totals, flags=flags, how="sync")

    def resized_steps(count, totals=None):
        shorter = rays_dev[:, : max(1, n - 256)]
        for k in range(count):
            got, _ = step(args.flags, rays_dev if k % 2 == 0 else shorter)
            if totals is not None:
                totals.add()
                totals.rows_returned += int(got.shape[1])

    side_sync = side("synchronous", sync_steps, args.side_steps)
    side_no_hints = side("no hints", lambda c, t=None: run_steps(c, t, flags=args.flags | engine.TRACE_NO_HINTS,
                                                                  how="sync" if mode == "sync" else "one_stream"),
                         args.side_steps)
    side_resized = side("changing ray count", resized_steps, args.side_steps)
    # ... and (b) once more issued like the timed region, so that there is a figure to put next to `value`
    side_no_hints_overlap = side("no hints, overlapped", lambda c, t=None: run_steps(
        c, t, flags=args.flags | engine.TRACE_NO_HINTS), args.side_steps) if mode == "overlap" else None
    # (e) the round-3 headline form: the SAME ray set traced again and again, issued like the timed region -- dense
    #     hints of the previous trace active, the input possibly still in the Infinity Cache
    side_replay = side("replay", lambda c, t=None: run_steps(c, t, rotate=False), args.side_steps) \
        if len(ray_sets) > 1 else None
    # (f) a fresh scene's first trace: scene compilation, table upload, control-word initialisation, no hints
    cold = None
    if args.side_steps > 0:
        cold_times = []
        for _ in range(3):
            torch.cuda.synchronize(device)
            c0 = time.perf_counter()
            fresh = engine.DeviceScene(snap, options=scene_options)
            cold_rows, _ = fresh.trace(rays_dev, limit, flags=args.flags, out=block)
            torch.cuda.synchronize(device)
            cold_times.append(time.perf_counter() - c0)
            cold_count = int(cold_rows.shape[1])
            fresh.close()
        cold = {"ms": min(cold_times) * 1e3, "rows_per_s_this_gpu": cold_count / min(cold_times),
                "what": "DeviceScene(snapshot) + its first trace (no hints, control words initialised), best of 3; "
                        "kernels already loaded by this process"}
    # (g) record plans (round 6; include/prt.h prt_record_plan): the same steps when the caller wants only the rows of
    #     the detector -- `results.loc[results['surface'] == imager.get_id()]`, examples/lens_design.ipynb cells 11, 19,
    #     38 -- or no rows at all, only the sums its merit function is made of (cells 12, 15, 20): issued like the
    #     timed region (`overlapped`) and as blocking prt_trace calls (`synchronous`).  Their rows per second count the
    #     ray segments RESOLVED (what the metric counts), of which only the detector's are stored / summed.
    record_plans = None
    if args.side_steps > 0 and args.workload == "config2" and args.flags == 0:
        try:
            detector = int(snap.prims["surface_id"][-1])
            resolved_per_step = float(sum(set_stats[k % len(set_stats)]["rows"] for k in range(args.side_steps))) / args.side_steps
            record_plans = {}
            for label, plan in (("detector_rows", engine.RecordPlan(surfaces=(detector,), rows=True, generation_limit=limit)),
                                ("detector_sums", engine.RecordPlan(surfaces=(detector,), rows=False, stats=True,
                                                                    generation_limit=limit))):
                respin()
                for k in range(8):
                    got, got_counts = scene.trace(ray_set(k), limit, out=block, plan=plan)
                torch.cuda.synchronize(device)
                p0 = time.perf_counter()
                for k in range(args.side_steps):
                    got, got_counts = scene.trace(ray_set(k), limit, out=block)
                torch.cuda.synchronize(device)
                sync_s = (time.perf_counter() - p0) / args.side_steps
                stored = int(sum(got_counts))
                st = scene.trace_stats()
                for ticket in range(depth):
                    scene.set_plan(ticket, plan, device)
                batch = engine.TraceBatch(scene, [ray_set(k) for k in range(args.side_steps)], limit, depth=depth,
                                          outs=blocks[:max(depth, 2)], flags=engine.TRACE_NO_TIMING | engine.TRACE_BUSY)
                batch.run()
                torch.cuda.synchronize(device)
                p0 = time.perf_counter()
                batch.run()
                torch.cuda.synchronize(device)
                over_s = (time.perf_counter() - p0) / args.side_steps
                union = batch.busy()
                for ticket in range(depth):
                    scene.set_plan(ticket, None, device)
                # what such a step has to move: the caller's 13 rows in, lean state between the generations, 120 B per
                # stored row (SURVEY 8d's per-unit figures with the rows that are not asked for left out)
                carried = float(st["rays_carried"])
                bytes_step = 104.0 * n + STATE_BYTES_LEAN * (st["ray_generations"] - n) + STATE_BYTES_LEAN * carried + ROW_BYTES * stored
                record_plans[label] = {
                    "rows_stored_per_step": stored, "segments_resolved_per_step": resolved_per_step,
                    "synchronous_ms_per_step": sync_s * 1e3, "overlapped_ms_per_step": over_s * 1e3,
                    "value_synchronous": resolved_per_step * world / sync_s, "value_overlapped": resolved_per_step * world / over_s,
                    "launches_per_step": st["kernel_launches"],
                    "roofline": {"bound": "hbm", "bytes_per_step": bytes_step,
                                 "what": "13 state rows in + 56 B lean state each way between generations + 120 B per STORED row",
                                 "kernel_ms_per_step": union["union_ms"] / args.side_steps if union["traces"] else None,
                                 "achieved": bytes_step / (union["union_ms"] / args.side_steps * 1e-3) / 1e9 if union["traces"] else None,
                                 "frac": bytes_step / (union["union_ms"] / args.side_steps * 1e-3) / 1e9 / HBM_PEAK_GBS if union["traces"] else None,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s"}}
            scene.trace(rays_dev, limit, out=block, plan=None)
            # ... and the loop those plans are for, through the front end (RayTracer on the same system: sources on the
            # device, the detector's spot size read every iteration; every second figure with the detector moved first).
            # One GPU only: under torch.distributed a RayTracer shards its rays and gathers its frames -- collectives
            # that have no place in a side measurement of a multi-rank run.
            if distributed:
                raise StopIteration
            import pyrayt_amd as pyrayt

            CountedObject.reset_ids()
            loop_lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
            loop_src = pyrayt.components.ConeOfRays(cone_angle=6).move_x(-1.9)
            loop_det = pyrayt.components.baffle((1, 1)).move_x(1)
            tracer = pyrayt.RayTracer(loop_src, [loop_lens, loop_det], rays_per_source=n)

            def iteration(move, fused):
                if move:
                    loop_det.move_x(1e-4)
                if fused:
                    return tracer.trace_stats(surface=loop_det).values()["rms_radius"][0]
                return tracer.trace_device().group_stats(surface=loop_det.get_id())["rms_radius"].iloc[0]

            loop = {}
            for fused in (False, True):
                for move in (False, True):
                    for _ in range(5):
                        iteration(move, fused)
                    torch.cuda.synchronize(device)
                    p0 = time.perf_counter()
                    for _ in range(40):
                        iteration(move, fused)
                    torch.cuda.synchronize(device)
                    loop[("trace_stats" if fused else "trace_device+group_stats") + ("_moving_part" if move else "_unchanged")] = \
                        (time.perf_counter() - p0) / 40 * 1e3
            record_plans["design_loop_ms_per_iteration"] = dict(loop, what=(
                f"RayTracer on the bench's system, {n} rays from a device-side ConeOfRays, the detector's rms spot radius "
                "read on the host every iteration (tools/design_loop.py): the frame stored and reduced (prt_frame_stats) "
                "against the sums accumulated in the generation kernels (RayTracer.trace_stats)"))
        except StopIteration:
            pass
        except Exception as exc:  # noqa: BLE001
            record_plans = dict(record_plans or {}, error=f"{type(exc).__name__}: {exc}"[:300])
        respin()
    # --- in-run correctness tie: the rows of the timed workload against the reference's own summaries of it ---------
    # One GPU: the rows the last timed step of EACH ray set of the rotation left in its record block (brought to the
    # host right behind the timed region).  Several GPUs: checked before the timed region, see `pre_verification`.
    verified, verify_note = None, ("only BASELINE config 2 at 1M rays, generation_limit 10 (strong scaling) has "
                                   "reference summaries")
    if verifiable and world == 1:
        per_seed, what = {}, "the last timed step of each ray set (rows of the timed region's final repetition)"
        if not last_timed:  # (issue modes that do not keep a block per ray set: a synchronous trace of each set instead)
            what = "a synchronous trace of each ray set behind the timed region (this issue mode keeps no block per ray set)"
            for k in range(len(ray_sets)):
                got, got_counts = scene.trace(ray_sets[k], limit, flags=args.flags)
                last_timed[k] = (engine.to_host(got).T, got_counts)
        for k, (frame, _) in sorted(last_timed.items()):
            seed = base_seed + k
            if os.path.exists(summary_file(seed)):
                per_seed[seed] = summary_checks(frame, np.load(summary_file(seed)))
        if per_seed:
            verified = all(all(c.values()) for c in per_seed.values())
            verify_note = {"against": "tests/golden/config2_1m_summary*.npz (written by the genuine reference, "
                                      "tests/golden/generate_golden.py config2_summary)", "of": what,
                           "seeds": sorted(per_seed), "checks": {str(k): v for k, v in sorted(per_seed.items())}}
        last_timed.clear()
    elif pre_verification is not None:
        verified, verify_note = pre_verification
    rows, counts = run_steps(2)  # (leave the scene with the hints of the rotation for what follows)

    rows_per_step = int(rows.shape[1])
    rows_timed = float(timed.rows_returned)
    if distributed:
        agg = torch.tensor([elapsed, float(rows_per_step), rows_timed], dtype=torch.float64, device=comm_device)
        tmax = agg.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_rows_per_step = float(agg[1])
        rows_timed = float(agg[2])
    else:
        total_rows_per_step = float(rows_per_step)

    def sink_stats(rows_local, counts_local, group, comm):
        """ms of one DeviceFrame.group_stats call on the last step's rows (best of the last two of three)."""
        from pyrayt_amd.frame import DeviceFrame

        frame = DeviceFrame(rows_local, counts_local)
        detector = float(snap.prims["surface_id"][-1])
        per_group = max(1, n_job // 8)
        times = []
        for _ in range(3):
            torch.cuda.synchronize(device)
            if group is not None:
                dist.barrier()
            s0 = time.perf_counter()
            frame.group_stats(surface=detector, rays_per_source=per_group, n_groups=8, group=group, comm=comm)
            times.append(time.perf_counter() - s0)
        return min(times[1:]) * 1e3

    # frame re-assembly (not in the timed region): the library's RCCL all-gathers + placement kernel
    # when the group is RCCL-backed, torch.distributed + the placement kernel otherwise
    gather = None
    if distributed:
        try:  # an extra: its failure must not cost the benchmark line
            comm = bench_comm  # (the library's RCCL communicator made before the timed region; None on a gloo group)
            times = []
            for _ in range(3):  # the first pass also builds RCCL's channels
                torch.cuda.synchronize(device)
                dist.barrier()
                g0 = time.perf_counter()
                full, full_counts = pdist.assemble_rows(rows, counts, limit, dist.group.WORLD, "all", comm=comm, reuse=True)
                torch.cuda.synchronize(device)
                dist.barrier()
                times.append(time.perf_counter() - g0)
                assert full.shape[1] == int(total_rows_per_step), (full.shape, total_rows_per_step)
                del full
            agg = torch.tensor([min(times[1:])], dtype=torch.float64, device=comm_device)
            dist.all_reduce(agg, op=dist.ReduceOp.MAX)
            gather = {"ms": float(agg[0]) * 1e3, "first_call_ms": times[0] * 1e3, "rows": int(total_rows_per_step),
                      "GB_into_each_gpu": total_rows_per_step * 120 * (world - 1) / world / 1e9,
                      "transport": "RCCL inside libprt_hip (prt_allgather_rows)" if comm is not None
                                   else f"torch.distributed {backend} + prt_place_rows"}
            # ... and what most design loops want instead of the frame: the detector's spot / focus statistics per
            # source group, from the rows every rank kept (prt_frame_stats_sharded: per-rank sums + two small
            # all-reduces; nothing of the frame moves)
            stats_ms = sink_stats(rows, counts, dist.group.WORLD, comm)
            agg = torch.tensor([stats_ms], dtype=torch.float64, device=comm_device)
            dist.all_reduce(agg, op=dist.ReduceOp.MAX)
            gather["stats_ms"] = float(agg[0])
            gather["stats_transport"] = ("ncclAllReduce inside libprt_hip (prt_frame_stats_sharded)" if comm is not None
                                         else f"torch.distributed {backend} between prt_frame_reduce passes")
            # ... and the re-assembly PIPELINED behind the traces: the frame of trace k is gathered on a communication
            # stream while trace k + 1 runs (pyrayt_amd.distributed.trace_and_gather): the steady state of a loop that
            # wants every frame whole on every GPU
            if comm is not None:
                piped = []
                for _ in range(2):
                    torch.cuda.synchronize(device)
                    dist.barrier()
                    p0 = time.perf_counter()
                    frames = 0
                    for full, _ in pdist.trace_and_gather(scene, (ray_set(k) for k in range(args.steps)), limit, comm,
                                                          flags=args.flags):
                        frames += 1
                    torch.cuda.synchronize(device)
                    dist.barrier()
                    piped.append((time.perf_counter() - p0) / max(frames, 1))
                agg = torch.tensor([min(piped)], dtype=torch.float64, device=comm_device)
                dist.all_reduce(agg, op=dist.ReduceOp.MAX)
                gather["pipelined_ms_per_step"] = float(agg[0]) * 1e3
                comm.close()
        except Exception as exc:  # noqa: BLE001
            gather = dict(gather or {}, error=f"{type(exc).__name__}: {exc}"[:300])
    result_sink = None
    if not distributed:
        try:
            result_sink = {"stats_ms": sink_stats(rows, counts, None, None), "groups": 8,
                           "what": "DeviceFrame.group_stats of the detector's rows, 8 source groups (prt_frame_stats), incl. the D2H of the 8 x 8 result"}
        except Exception as exc:  # noqa: BLE001
            result_sink = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    # PCIe-inclusive end-to-end trace() on rank 0 (H2D rays, trace, D2H rows, DataFrame)
    end_to_end = None
    if rank == 0 and float(n) * limit * 120 > 16e9:
        end_to_end = {"skipped": "the frame of this job is tens of GB: not brought to the host three times for a side figure"}
    elif rank == 0:
        from pyrayt_amd.tracer import rows_to_frame

        try:
            times = []
            for _ in range(3):  # the first call also page-locks the host staging block
                torch.cuda.synchronize(device)
                e0 = time.perf_counter()
                up = torch.from_numpy(rays).to(device)
                r2, _ = scene.trace(up, limit, flags=args.flags)
                frame = rows_to_frame(r2)
                times.append(time.perf_counter() - e0)
                n_rows = frame.shape[0]
                del frame, r2, up
            end_to_end = {"ms": min(times[1:]) * 1e3, "rows_per_s": n_rows / min(times[1:]),
                          "first_call_ms": times[0] * 1e3}
            if args.workload == "config2" and args.flags == 0 and args.side_steps > 0:
                # ... and what the notebook's `results.loc[results['surface'] == imager.get_id()]` costs end to end when
                # the cut is made by the generation kernels (a record plan: the other rows are never written, never copied)
                detector_plan = engine.RecordPlan(surfaces=(int(snap.prims["surface_id"][-1]),), rows=True, generation_limit=limit)
                times = []
                for _ in range(3):
                    torch.cuda.synchronize(device)
                    e0 = time.perf_counter()
                    up = torch.from_numpy(rays).to(device)
                    r2, _ = scene.trace(up, limit, flags=args.flags, plan=detector_plan)
                    frame = rows_to_frame(r2)
                    times.append(time.perf_counter() - e0)
                    kept_rows = frame.shape[0]
                    del frame, r2, up
                end_to_end["detector_rows_only_ms"] = min(times[1:]) * 1e3
                end_to_end["detector_rows"] = kept_rows
                # ... and the spot diagram of cells 11 / 19: two columns (y1, z1) of those rows
                from pyrayt_amd.frame import DeviceFrame

                spot_plan = engine.RecordPlan(surfaces=detector_plan.surfaces, rows=True, columns=("y1", "z1"), generation_limit=limit)
                times = []
                for _ in range(3):
                    torch.cuda.synchronize(device)
                    e0 = time.perf_counter()
                    up = torch.from_numpy(rays).to(device)
                    r2, c2 = scene.trace(up, limit, flags=args.flags, plan=spot_plan)
                    frame = DeviceFrame(r2, c2, spot_plan.columns).to_pandas()
                    times.append(time.perf_counter() - e0)
                    del frame, r2, up
                scene.set_plan(0, None, device)
                end_to_end["detector_spot_columns_ms"] = min(times[1:]) * 1e3
        except Exception as exc:  # noqa: BLE001
            end_to_end = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    if rank != 0:
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    value = rows_timed / elapsed  # rows of every timed step, all ranks / the slowest rank's wall time
    # algorithmic bytes, SURVEY.md section 8d's per-unit figures (13-row state read 104 B, record row
    # 120 B, 13-row next state 104 B; = 328 B per ray-generation when every ray is recorded and goes on),
    # counted exactly: state read per ray alive at generation entry, row per recorded ray, next state
    # per ray that goes on.  The kernel moves less than that: between generations it carries 10 of the
    # 13 state rows (include/prt.h "compact state"), i.e. 80 B where the figure says 104 -- reported next
    # to it as `moved_bytes_per_launch`, which is what `traffic` (PMC) has to be compared with.
    algorithmic_bytes = 104.0 * ray_generations + 120.0 * rows_recorded + 104.0 * rays_carried
    full_rows = scene.telemetry()["full_rows_fallbacks"] > 0 or bool(args.flags & engine.TRACE_FULL_ROWS)
    # (the workloads of this file qualify for lean segments in every wave but the handful that straddle two sources;
    # `traffic` -- the PMC counters -- is the measurement, this is the expectation it is held against)
    state = STATE_BYTES_FULL if full_rows else STATE_BYTES_LEAN
    first_generation = float(n) * args.steps
    moved_bytes = (STATE_BYTES_FULL * first_generation + state * (ray_generations - first_generation) +
                   ROW_BYTES * rows_recorded + state * rays_carried)
    achieved = algorithmic_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # HBM bytes per launch from the PMC counters: measured by tools/traffic.sh in its own rocprofv3 passes (the
    # counters cannot be read from inside this process) and committed under profiles/; it counts only if it was
    # taken from THIS build of the library (tools/traffic.py stamps the file with the library's hash)
    import hashlib

    with open(engine.LIB_PATH, "rb") as fh:
        library_sha16 = hashlib.sha256(fh.read()).hexdigest()[:16]
    traffic, traffic_note = None, None
    traffic_file = next((f for f in (os.path.join(ROOT, "profiles", r, "traffic.json") for r in ("r6", "r5", "r4", "r3", "r2"))
                         if os.path.exists(f)), None)
    if traffic_file and args.flags == 0 and n == RAYS_PER_GPU and args.workload == "config2":
        with open(traffic_file) as fh:
            measured = json.load(fh)
        if measured.get("library_sha16") == library_sha16:
            # per launch like `achieved`: a repeated trace launches exactly its working generations
            # (the library sizes the first batch from the previous trace), a first trace one more
            blind = launches > len(counts) * args.steps
            traffic = measured.get("hbm_bytes_per_launch" if blind else "hbm_bytes_per_working_launch")
            traffic_note = (os.path.relpath(traffic_file, ROOT) + f" (library {library_sha16}; rocprofv3 PMC FETCH_SIZE + "
                            "WRITE_SIZE in separate passes, calibrated on tools/ubench/copy_f64; tools/traffic.sh)")
        else:
            traffic_note = (f"none: {os.path.relpath(traffic_file, ROOT)} was measured on library "
                            f"{measured.get('library_sha16', 'of an unstamped build')}, this run loaded {library_sha16} "
                            "(run tools/traffic.sh on this build)")
    line = {
        "metric": "ray-surface intersections/sec, 1M-ray biconvex lens",
        "value": value,
        "unit": "intersections/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": WORKLOADS[args.workload] +
                        (f": one {n_job}-ray job, contiguous id shards over {world} GPU(s)" if strong
                         else f": {n_job} rays per GPU (weak scaling)") +
                        f", generation_limit {limit}, rays resident in HBM; " +
                        (f"the timed steps rotate through {len(ray_sets)} distinct seeded ray sets (seeds {base_seed + seed_shift} ... "
                         f"{base_seed + seed_shift + len(ray_sets) - 1}): no step re-traces its predecessor's rays; the dense-mode hints "
                         "of the scene's previous trace apply (they are checked per tile)"
                         if len(ray_sets) > 1 else
                         "the timed step is a REPEATED IDENTICAL trace: dense-mode hints of the previous trace active, first batch sized by it") +
                        {"overlap": f", {depth} traces in flight on {depth} HIP streams (prt_trace_batch = prt_trace_begin / prt_trace_end per trace): the "
                                    "host enqueues ahead and the kernels of different traces overlap on the device",
                         "one_stream": ", one trace kept in flight on the same stream while the previous one's counts are "
                                       "collected (prt_trace_begin / prt_trace_end)",
                         "sync": ", synchronous (prt_trace)"}[mode] +
                        "; see value_replay / value_one_stream / value_synchronous / value_cold / value_no_hints / "
                        "value_first_trace for the other kinds of step",
            "ray_sets": len(ray_sets),
            "issue_mode": mode,
            "issued_by": ("prt_trace_batch (one library call per timed region)" if mode == "overlap" and not args.python_loop
                          else "a Python loop over prt_trace_begin / prt_trace_end" if mode != "sync" else "prt_trace"),
            "traces_in_flight": depth if mode != "sync" else 1,
            "streams": depth if mode == "overlap" else 1,
            "rays_job": n_job * (1 if strong else world),
            "rays_per_gpu": n,
            "devices_visible_per_rank_process": n_devices,
            "dist_backend": backend if distributed else None,
            "rccl_ranks": rccl_ranks,  # ncclCommCount of the library's communicator (None on one GPU / a gloo group)
            "rows_per_step_per_gpu": rows_per_step,
            "rows_per_generation": counts,
            # secondary metric of SURVEY.md section 8d: rays alive at generation entry x primitives
            "primitive_tests_per_s": ray_generations * len(snap.prims) * world / elapsed,
            "trace_flags": args.flags,
            "scene_options": scene_options,
            "spinup_steps_untimed": spinup_steps,
            # repeats the library made on its own since the scene was created (a dense hint that did not hold, the
            # look-back fallback): each is a trace run twice
            "telemetry": scene.telemetry(),
            "parallelism": f"ray data-parallel x{world} (contiguous id shards), no collective in the timed region",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            # (`frac` prices the timed region: with two traces in flight it is the throughput of two INTERLEAVED traces
            # over the time the device had at least one of them in flight -- not one kernel's duration.  The generation
            # kernel with the device to itself is `kernel_alone_frac` (= one_stream.frac, what rocprofv3 --stats of
            # `bench.py --streams 1` averages))
            "frac_means": ("algorithmic bytes of the region's launches / the union of the traces' busy intervals: two traces "
                           "interleaved" if busy is not None else "algorithmic bytes of a launch / its HIP-event duration"),
            "kernel_alone_frac": None,
            "frac_of_measured_copy": achieved / COPY_GBS,
            "traffic": traffic,
            "traffic_source": traffic_note,
            "library_sha16": library_sha16,
            "kernel": "k_generation" if not (args.flags & 2) else "k_hit + k_scan + k_shade + k_advance",
            "measured_on": ("the timed region (its median repetition): every trace bracketed by its own pair of HIP events on "
                            "its own stream, intervals merged by the library (PRT_TRACE_BUSY / prt_trace_batch_busy) -- "
                            "kernel_ms_per_step is the time per step during which at least one of the region's traces "
                            "had launches in flight, avg_launch_ms that time per generation launch; the kernel with the "
                            "device to itself is `one_stream`"
                            if busy is not None else
                            "one stream, the same steps right behind the timed region (HIP events per trace): this issue "
                            "mode has no merged intervals" if mode == "overlap" else
                            "the timed region (HIP events per trace on the launch stream; the traces do not overlap)"),
            "algorithmic_bytes_per_launch": algorithmic_bytes / launches if launches else 0,
            "avg_launch_ms": kernel_ms / kernel_launches if kernel_launches else 0,
            "launches_per_step": launches / args.steps,
            "kernel_ms_per_step": kernel_ms / args.steps,
            # what the device as a whole sustains in the timed region (all traces in flight together):
            # algorithmic bytes of a step over the step time
            "device_aggregate": {"achieved": algorithmic_bytes / elapsed / 1e9 if elapsed > 0 else 0.0,
                                 "frac": algorithmic_bytes / elapsed / 1e9 / HBM_PEAK_GBS if elapsed > 0 else 0.0,
                                 # on the bytes the kernels actually move (compact state): what to hold against a copy
                                 "moved": moved_bytes / elapsed / 1e9 if elapsed > 0 else 0.0,
                                 "moved_frac": moved_bytes / elapsed / 1e9 / HBM_PEAK_GBS if elapsed > 0 else 0.0,
                                 "moved_frac_of_measured_copy": moved_bytes / elapsed / 1e9 / COPY_GBS if elapsed > 0 else 0.0,
                                 "unit": "GB/s", "what": "bytes of the timed steps / their wall time (this GPU): `achieved` "
                                                         "on the algorithmic bytes of SURVEY 8d, `moved` on what the compact "
                                                         "state form transfers"},
            "bytes_per_ray_generation_if_all_survive": BYTES_PER_RAY_GENERATION,
            "moved_bytes_per_launch": moved_bytes / launches if launches else 0,
            # the traces' own intervals before merging: their sum per launch is what one launch takes on its stream while
            # another trace shares the device (about what `rocprofv3 --stats` averages for the overlapped command)
            "busy": None if busy is None else dict(busy, avg_launch_ms_on_its_stream=busy["sum_ms"] / launches if launches else 0.0),
            "one_stream": one_stream,
            "moved_frac": moved_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms > 0 else 0.0,
            "state_rows": "all 13" if full_rows else "7 of 13 between generations (w rows and generation row implied; intensity, "
                                                     "wavelength and id as three numbers per wave where a wave's rays stay in place: "
                                                     "lean segments; 10 rows for a ray set that does not qualify)",
        },
        "end_to_end_trace": end_to_end,
    }
    # every repetition of the timed region (`value` / `ms_per_step` are the median one's)
    line["repetitions"] = {"n": len(repetitions), "ms_per_step": [r.elapsed / args.steps * 1e3 for r in repetitions],
                           "value_min": rows_timed / max(r.elapsed for r in repetitions),
                           "value_max": rows_timed / min(r.elapsed for r in repetitions),
                           "kernel_ms_per_step": [r.busy["union_ms"] / args.steps if r.busy else None for r in repetitions]}
    if one_stream:
        bytes_alone = one_stream.pop("_bytes")
        gbs = bytes_alone / (one_stream["kernel_ms_per_step"] * args.steps * 1e-3) / 1e9 if one_stream["kernel_ms_per_step"] > 0 else 0.0
        one_stream["achieved"], one_stream["frac"] = gbs, gbs / HBM_PEAK_GBS
        line["roofline"]["kernel_alone_frac"] = gbs / HBM_PEAK_GBS
        one_stream["what"] = ("the same steps on ONE stream right behind the timed region: the generation kernel with the device to "
                              "itself (HIP events per trace; compare rocprofv3 --stats of `bench.py --streams 1`)")

    # the other kinds of step (this rank's GPU; untimed side runs of --side-steps traces each)
    def publish(kind):
        if kind is None:
            return None
        bytes_, ms_ = kind.pop("_bytes"), kind.pop("_kernel_ms")
        gbs = bytes_ / (ms_ * 1e-3) / 1e9 if ms_ > 0 else 0.0
        kind["frac"] = gbs / HBM_PEAK_GBS
        return kind

    side_sync, side_no_hints, side_resized = publish(side_sync), publish(side_no_hints), publish(side_resized)
    line["verified"] = verified
    line["verification"] = verify_note
    if side_replay:
        side_replay.pop("_bytes"), side_replay.pop("_kernel_ms")
        line["value_replay"] = side_replay["rows_per_s_this_gpu"] * world  # round 3's `value`: the same rays again and again
    if cold:
        line["value_cold"] = cold["rows_per_s_this_gpu"] * world
        line["cold"] = cold
    if one_stream:
        line["value_one_stream"] = one_stream["rows"] * world / (one_stream["ms_per_step"] * args.steps * 1e-3)
    if side_sync:
        line["value_synchronous"] = side_sync["rows_per_s_this_gpu"] * world
        line["synchronous"] = side_sync
    if side_no_hints_overlap:
        side_no_hints_overlap.pop("_bytes"), side_no_hints_overlap.pop("_kernel_ms")
        line["value_no_hints_overlapped"] = side_no_hints_overlap["rows_per_s_this_gpu"] * world  # issued like `value`
    if side_no_hints:
        line["value_no_hints"] = side_no_hints["rows_per_s_this_gpu"] * world  # one stream, like value_one_stream
        line["roofline"]["no_hints"] = {k: side_no_hints[k] for k in ("avg_launch_ms", "frac", "kernel_ms_per_step",
                                                                      "launches_per_step", "ms_per_step")}
    if side_resized:
        # a trace whose ray count differs from the previous one's (synchronous): the control words are
        # re-initialised, the hints still serve; a scene's very first trace has no hints (value_no_hints) and
        # its one-off costs (table upload, kernel load) are in end_to_end_trace.first_call_ms
        line["value_first_trace"] = side_resized["rows_per_s_this_gpu"] * world
        line["changing_ray_count"] = side_resized
    if result_sink:
        line["result_sink"] = result_sink
    line["scaling_model"] = scaling_model(n_job if strong else n_job * world, world, total_rows_per_step)
    if record_plans:
        line["record_plans"] = record_plans
        if "detector_rows" in record_plans and "value_overlapped" in record_plans.get("detector_rows", {}):
            line["value_filtered"] = record_plans["detector_rows"]["value_overlapped"]
        if "detector_sums" in record_plans and "value_overlapped" in record_plans.get("detector_sums", {}):
            line["value_sums_only"] = record_plans["detector_sums"]["value_overlapped"]
    if gather:
        line["gather"] = gather
        if "ms" in gather:  # a trace, then its re-assembly, one after the other
            line["value_with_gather"] = total_rows_per_step / (elapsed / args.steps + gather["ms"] * 1e-3)
        if "pipelined_ms_per_step" in gather:  # re-assembly of trace k behind trace k + 1: the steady state
            line["value_with_gather_pipelined"] = total_rows_per_step / (gather["pipelined_ms_per_step"] * 1e-3)

    if not args.no_cpu_baseline and world == 1:
        from oracle import prt_oracle
        import helpers

        m = min(args.cpu_rays, n)
        sample = np.ascontiguousarray(rays[:, :m])
        flat = helpers.flat_scene(snap)
        c0 = time.perf_counter()
        frame, _ = prt_oracle.trace(flat, sample, limit)
        cpu_s = time.perf_counter() - c0
        line["cpu_baseline"] = {
            "value": frame.shape[0] / cpu_s,
            "unit": "intersections/s",
            "cores": 1,
            "kind": "port",
            "sample": f"numpy oracle (oracle/prt_oracle.py) on the first {m} rays of the same "
                      f"workload, {frame.shape[0]} rows in {cpu_s:.1f} s, single process like "
                      f"the reference; host has {os.cpu_count()} logical CPUs",
        }
        # the bar next to it: the C restatement on every host core (one process per core over contiguous
        # id ranges of the same job; a child interpreter that never touches the GPU)
        try:
            import subprocess

            procs = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
            repeat = max(1, round(procs * 1.3e6 * 1.5 / (3.0 * n)))  # ~1.5 s of work per core at ~1.3e6 rows/s/core
            done = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", "--workload", args.workload, "--rays", str(n_job),
                                   "--procs", str(procs), "--repeat", str(repeat), "--limit", str(limit)],
                                  cwd=ROOT, capture_output=True, text=True, timeout=600)
            res = json.loads(done.stdout.strip().splitlines()[-1])
            model = ""
            try:
                with open("/proc/cpuinfo") as fh:
                    model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "")
            except OSError:
                pass
            line["cpu_baseline_all_cores"] = {
                "value": res["rows_per_s"], "unit": "intersections/s", "cores": procs, "kind": "port-c",
                "sample": f"C oracle (oracle/prt_oracle.c, gcc -O2) on the whole {n_job}-ray job, one process per "
                          f"logical CPU over contiguous id ranges, each slice traced {repeat}x: {res['rows']} rows in "
                          f"{res['seconds']:.2f} s; CPU: {model or 'unknown'} ({os.cpu_count()} logical CPUs)",
            }
        except Exception as exc:  # noqa: BLE001
            line["cpu_baseline_all_cores"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if verified is False:
        raise SystemExit("bench.py: the rows of the timed workload do NOT match the reference's summary: " + json.dumps(verify_note))


if __name__ == "__main__":
    main()
