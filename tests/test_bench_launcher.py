"""`python bench.py --gpus N` without a launcher starts its own ranks (bench.spawn_ranks).  The launcher
never touches the GPU, so its logic is testable here with a stub worker in place of the benchmark."""
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent("""
    import json, os, sys, time
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
    assert int(os.environ["MASTER_PORT"]) > 0 and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    mode = sys.argv[1]
    if mode == "fail" and rank == 1:
        sys.exit(7)
    if mode == "fail":
        time.sleep(60)   # "waiting in a barrier" for the rank that died
    print(f"[Gloo] chatter from rank {rank}")   # libraries write to stdout too
    print(json.dumps({"rank": rank, "n_gpus": world, "argv": sys.argv[1:]}) if rank == 0 else f"noise from rank {rank}")
""")


def _launch(tmp_path, world, mode):
    stub = tmp_path / "stub_worker.py"
    stub.write_text(STUB)
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; "
            f"raise SystemExit(bench.spawn_ranks({world}, [{mode!r}, '--steps', '3'], worker=[sys.executable, {str(stub)!r}]))")
    t0 = time.perf_counter()
    done = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    return done, time.perf_counter() - t0


@pytest.mark.parametrize("world", [2, 4])
def test_launcher_passes_rank_zero_line_through(tmp_path, world):
    done, _ = _launch(tmp_path, world, "ok")
    assert done.returncode == 0, done.stderr
    lines = [ln for ln in done.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, done.stdout  # rank 0's JSON line and nothing else
    line = json.loads(lines[0])
    assert line == {"rank": 0, "n_gpus": world, "argv": ["ok", "--steps", "3"]}
    for rank in range(1, world):  # the other ranks' stdout ends up on stderr
        assert f"noise from rank {rank}" in done.stderr
    assert "[Gloo] chatter from rank 0" in done.stderr


def test_launcher_reports_a_failed_rank_and_stops_the_others(tmp_path):
    done, seconds = _launch(tmp_path, 3, "fail")
    assert done.returncode == 7
    assert seconds < 30  # did not sit out the survivors' 60 s
    assert done.stdout.strip() == ""


def test_bench_becomes_a_launcher_before_anything_touches_the_gpu():
    """The hand-over in main() comes before the first torch / library import."""
    source = open(os.path.join(ROOT, "bench.py")).read()
    main_body = source[source.index("def main():"):]
    assert main_body.index("spawn_ranks(") < main_body.index("import torch")
    launcher = source[source.index("def spawn_ranks("):source.index("def main():")]
    assert "import torch" not in launcher and "engine" not in launcher and "ctypes" not in launcher
