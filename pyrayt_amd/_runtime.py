"""Process-wide settings that have to be made before the HIP runtime initialises (imported first by the package)."""
import os
import sys


def _ask_for_hardware_queues():
    """The HIP runtime maps streams onto hardware queues, four by default (the null stream holds one), and two
    streams that share a queue run their kernels one after the other: with the default a fourth trace in flight
    (``DeviceScene.trace_many(depth=4)``, small shards: 125k rays 24.7 -> 23.1 us per step) would not overlap.

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
    """Do kernels on these streams really run side by side?  One spin kernel of about a millisecond on one stream against
    the same on all of them (some 10 ms, once per process and stream count; the kernel has to dwarf the launch costs):
    streams that share a hardware queue take k times as long.  Only asked when the queue setting was made late (see
    above) and more streams are wanted than the default covers."""
    key = (device.index or 0, len(streams))
    if key not in _overlap:
        import time

        cycles = 2_000_000

        def run(k):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for s in streams[:k]:
                with torch.cuda.stream(s):
                    torch.cuda._sleep(cycles)
            torch.cuda.synchronize(device)
            return time.perf_counter() - t0

        run(len(streams))
        one = min(run(1) for _ in range(3))
        many = min(run(len(streams)) for _ in range(3))
        _overlap[key] = many < 0.5 * len(streams) * one
    return _overlap[key]
