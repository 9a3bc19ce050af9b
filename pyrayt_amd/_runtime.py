"""Process-wide settings that have to be made before the HIP runtime is loaded (imported first by the package)."""
import os
import sys


def _ask_for_hardware_queues():
    """The HIP runtime maps streams onto hardware queues, four by default (the null stream holds one), and two
    streams that share a queue run their kernels one after the other: with the default a fourth trace in flight
    (``DeviceScene.trace_many(depth=4)``, small shards: 125k rays 24.7 -> 23.1 us per step) would not overlap.
    The runtime reads GPU_MAX_HW_QUEUES when it is loaded, so the package asks for eight when it is imported
    BEFORE torch / the HIP runtime; a setting the user made stays.  Returns how it went ("user", "set", "late")."""
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return "user"
    if "torch" in sys.modules:  # the runtime is loaded already: its queues are what they are
        return "late"
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
    return "set"


HW_QUEUES = _ask_for_hardware_queues()
