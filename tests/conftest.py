import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in (ROOT, os.path.join(ROOT, "tests")):
    if path not in sys.path:
        sys.path.insert(0, path)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """tools/run_matrix.sh runs the GPU suites under every product-path switch: PRT_TEST_OPTIONS
    ("no_chain=1,hit_lanes=8") and PRT_TEST_FLAGS (PRT_TRACE_* bits) become the defaults of every
    DeviceScene the tests build.  The library itself reads nothing from the environment; this is the
    test harness configuring the product through its public attributes."""
    text, flags = os.environ.get("PRT_TEST_OPTIONS", ""), os.environ.get("PRT_TEST_FLAGS", "")
    if not text and not flags:
        return
    from pyrayt_amd import engine

    for item in filter(None, text.split(",")):
        key, _, value = item.partition("=")
        if key not in engine.OPTION_NAMES:
            raise SystemExit(f"PRT_TEST_OPTIONS: unknown scene option {key!r}")
        engine.DEFAULT_OPTIONS[key] = int(value or 1)
    if flags:
        engine.DEFAULT_TRACE_FLAGS = int(flags, 0)
