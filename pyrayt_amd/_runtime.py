"""Process-wide settings that have to be made before the HIP runtime initialises (imported first by the package)."""
import os
import sys


def _ask_for_hardware_queues():
    """The HIP runtime maps streams onto hardware queues, four by default (the null stream holds one), and two
    streams that share a queue run their kernels one after the other: with the default, four traces in flight
    (``DeviceScene.trace_many(depth=4)``, small shards: 125k rays 24.7 -> 23.1 us per step) have the four queues to
    themselves only while the caller's own stream idles -- anything enqueued there shares a queue with a trace.

    The runtime reads GPU_MAX_HW_QUEUES when it INITIALISES -- on the first HIP call of the process, not when
    libamdhip64 is loaded (measured, tools/queue_probe.py -> profiles/r5/queue_probe.txt: set after ``import torch`` it
    is honoured, also after ``torch.cuda.device_count()``; after ``torch.cuda.is_available()``, which asks the runtime
    for its devices, it is not).  So the package asks for eight whenever the variable is unset, whichever of torch and
    pyrayt_amd is imported first; a setting the user made stays.  Returns how it went:
      "user"      the variable was set already
      "set"       set before torch was imported
      "set-late"  set with torch already imported: in force unless something initialised the runtime before (torch gives
                  no way to tell: ``torch.cuda.is_initialized()`` stays False after ``is_available()``); the first
                  request for more than three ticket streams checks that they really overlap (``queues_overlap``)
    """
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return "user"
    late = "torch" in sys.modules
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
    return "set-late" if late else "set"


HW_QUEUES = _ask_for_hardware_queues()

_overlap = {}


def queues_overlap(torch, streams, device):
    """Do kernels on these streams -- AND on the current stream, which the ticket streams share the hardware queues with
    (the caller's own work, the waits a trace's consumer enqueues) -- really run side by side?  One spin kernel of about
    a millisecond on one stream against the same on all of them (some 10 ms, once per process and stream count; the
    kernel has to dwarf the launch costs).  Streams that share a hardware queue serialise: with the runtime's default
    of four queues, four ticket streams alone still measure a ratio of 1.0 (profiles/r5/queue_probe.txt,
    "after_available") although a fifth stream in use at the same time has to share -- so the probe runs k + 1 streams,
    and any two of them on one queue double the time.  The answer is "yes" only below 1.5 x the single-stream time
    (2.0 is what one shared queue measures: a threshold there would be decided by noise).  Only asked when the queue
    setting was made late (see above) and more streams are wanted than the default covers."""
    key = (device.index or 0, len(streams))
    if key not in _overlap:
        import time

        cycles = 2_000_000
        probed = list(streams) + [torch.cuda.current_stream(device)]

        def run(k):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for s in probed[:k]:
                with torch.cuda.stream(s):
                    torch.cuda._sleep(cycles)
            torch.cuda.synchronize(device)
            return time.perf_counter() - t0

        run(len(probed))
        one = min(run(1) for _ in range(3))
        many = min(run(len(probed)) for _ in range(3))
        _overlap[key] = many < 1.5 * one
    return _overlap[key]
