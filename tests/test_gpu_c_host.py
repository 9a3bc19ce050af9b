"""The C ABI driven by a host that is neither Python nor torch (examples/c_host/prt_trace_file.cpp: include/prt.h
+ the HIP runtime for device memory): golden fixtures traced by that program equal the reference's frames, the
Python binding's frames bit for bit, and its prt_trace_batch leg equals its own synchronous trace."""
import os
import subprocess

import numpy as np
import pytest

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "examples", "c_host")
HOST = os.path.join(HOST_DIR, "prt_trace_file")
MAGIC = 0x70727431


def build_host():
    """The C host programs (the library itself is only built when it is missing: rebuilding it in place would pull
    the file from under the processes that have it loaded -- the other tests of this session)."""
    from pyrayt_amd import engine

    if not os.path.exists(engine.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "pyrayt_amd", "csrc"), "libprt_hip.so"], check=True, capture_output=True)
    subprocess.run(["make", "-C", HOST_DIR], check=True, capture_output=True)
    return HOST


def write_input(path, snap, rays, limit, flags=0):
    """The scene snapshot's tables are numpy records with the layout of prt.h's structs: they go out as they are."""
    with open(path, "wb") as f:
        np.array([MAGIC, len(snap.prims), len(snap.nodes), len(snap.roots), len(snap.materials), rays.shape[1], limit, flags],
                 dtype="<i8").tofile(f)
        f.write(np.ascontiguousarray(snap.prims).tobytes())
        f.write(np.ascontiguousarray(snap.nodes).tobytes())
        roots = np.ascontiguousarray(snap.roots, dtype="<i4").tobytes()
        f.write(roots + b"\0" * (-len(roots) % 8))
        f.write(np.ascontiguousarray(snap.materials).tobytes())
        f.write(np.ascontiguousarray(rays, dtype="<f8").tobytes())


def read_output(path, limit):
    raw = open(path, "rb").read()
    total = int(np.frombuffer(raw, dtype="<i8", count=1)[0])
    counts = np.frombuffer(raw, dtype="<i8", count=limit, offset=8).tolist()
    rows = np.frombuffer(raw, dtype="<f8", offset=8 * (1 + limit)).reshape(15, total)
    while counts and counts[-1] == 0:
        counts.pop()
    return rows, counts


def test_the_c_host_builds_and_links_against_the_abi():
    """(CPU) hipcc compiles the program against include/prt.h and links it to libprt_hip.so."""
    host = build_host()
    assert os.access(host, os.X_OK)
    symbols = subprocess.run(["nm", "-D", "--undefined-only", host], capture_output=True, text=True, check=True).stdout
    for name in ("prt_scene_create", "prt_trace", "prt_trace_batch", "prt_trace_workspace_bytes", "prt_scene_destroy",
                 "prt_trace_set_plan"):
        assert name in symbols, name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["scene_config1.npz", "scene_config2.npz", "scene_config5.npz", "scene_mirrors_and_stops.npz", "scene_adv_bench_a.npz"])
@pytest.mark.parametrize("depth", [0, 2])
def test_fixture_traced_from_c_equals_the_golden_frame(tmp_path, name, depth):
    torch = pytest.importorskip("torch")
    from pyrayt_amd import engine

    if not os.path.exists(os.path.join(helpers.GOLDEN, name)):
        pytest.skip(f"no fixture {name}")
    fx = helpers.load(name)
    limit = int(fx["generation_limit"])
    snap = helpers.snapshot_of(fx)
    rays = np.ascontiguousarray(fx["rays0"])
    host = build_host()
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_input(src, snap, rays, limit)
    done = subprocess.run([host, src, dst, str(depth)], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-600:]
    rows, counts = read_output(dst, limit)
    helpers.assert_frames_match(rows.T, fx["frame"], what=f"{name} traced from C")
    # ... and what the Python binding returns for the same tables is the same bytes
    ds = engine.DeviceScene(snap)
    py_rows, py_counts = ds.trace(torch.from_numpy(rays).to("cuda:0"), limit)
    assert counts == py_counts
    assert np.array_equal(py_rows.cpu().numpy(), rows, equal_nan=True)
    ds.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["scene_config2.npz", "scene_config3.npz", "scene_mirrors_and_stops.npz"])
def test_record_plan_set_from_c(tmp_path, name):
    """prt_trace_set_plan from the torch-free host: the rows of one surface only, and their sums from the generation
    kernels -- against the reference's frame filtered and the frame oracle's sums of it."""
    pytest.importorskip("torch")
    from oracle import frame_oracle

    fx = helpers.load(name)
    limit = int(fx["generation_limit"])
    frame = fx["frame"]
    last = frame[frame[:, 0] == frame[:, 0].max()]
    ids, counts_of_ids = np.unique(last[:, 5], return_counts=True)
    surface = int(ids[np.argmax(counts_of_ids)])
    host = build_host()
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_input(src, helpers.snapshot_of(fx), np.ascontiguousarray(fx["rays0"]), limit)
    done = subprocess.run([host, src, dst, "0", str(surface)], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-600:]
    raw = open(dst, "rb").read()
    total = int(np.frombuffer(raw, dtype="<i8", count=1)[0])
    rows = np.frombuffer(raw, dtype="<f8", count=15 * total, offset=8 * (1 + limit)).reshape(15, total)
    sums = np.frombuffer(raw, dtype="<f8", offset=8 * (1 + limit) + 8 * 15 * total).reshape(limit, 12)
    want = frame[frame[:, 5] == surface]
    helpers.assert_frames_match(rows.T, want, what=f"{name}: rows of surface {surface} from C")
    for g in range(limit):
        ref = frame_oracle.reduce_sums(want.T, None, float(g), None, 1)[0]
        assert np.allclose(sums[g, :9], ref, rtol=1e-11, atol=1e-12), (name, g)
