"""C-ABI surface (CPU): libprt_hip.so loads without a GPU and exports exactly the entry points
include/prt.h declares; the product fails loudly -- it never falls back to a CPU path."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "prt.h")


@pytest.fixture(scope="module")
def lib_path():
    from pyrayt_amd import engine

    if not os.path.exists(engine.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "pyrayt_amd", "csrc")], check=True)
    return engine.LIB_PATH


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(prt_[a-z_]+)\s*\(", text)))


def test_header_declares_what_the_binding_binds():
    from pyrayt_amd import engine

    assert declared_functions() == sorted(engine.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in prt.h but not exported"
    lib.prt_version.restype = ctypes.c_int
    assert lib.prt_version() == 100
    # every header citation points at a reference file that SURVEY.md lists
    text = open(HEADER).read()
    for path in re.findall(r"(pyrayt/[\w/]+\.py|tinygfx/[\w/]+\.py)", text):
        assert path in ("pyrayt/_pyrayt.py", "pyrayt/materials.py", "tinygfx/g3d/world_objects.py",
                        "tinygfx/g3d/csg.py", "tinygfx/g3d/primitives.py", "pyrayt/components.py",
                        "tinygfx/g3d/renderers.py", "tinygfx/g3d/materials/gooch.py",
                        "tinygfx/g3d/operations.py"), path


def test_scene_create_validates_without_a_gpu(lib_path):
    """prt_scene_create is pure host code: malformed snapshots are rejected with a message."""
    from pyrayt_amd import components, engine
    from pyrayt_amd.scene import SceneSnapshot

    snap = SceneSnapshot([components.biconvex_lens(2, 2, 0.25, aperture=1), components.baffle((1, 1))])
    scene = engine.DeviceScene(snap)
    assert scene.component_rows(0) == 6 and scene.component_rows(1) == 2
    scene.close()
    bad = SceneSnapshot([components.baffle((1, 1))])
    bad.nodes["prim"][0] = 7  # out of range
    with pytest.raises(RuntimeError, match="malformed"):
        engine.DeviceScene(bad)
    bad = SceneSnapshot([components.baffle((1, 1))])
    bad.prims["type"][0] = 9
    with pytest.raises(RuntimeError, match="unknown type"):
        engine.DeviceScene(bad)


def test_no_cpu_fallback():
    """Without a GPU the product raises; it must not route through the oracle or numpy."""
    import torch

    import pyrayt_amd as pyrayt
    from pyrayt_amd import engine

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    tracer = pyrayt.RayTracer(pyrayt.components.LineOfRays(), pyrayt.components.baffle((1, 1)).move_x(1))
    with pytest.raises(engine.EngineUnavailable):
        tracer.trace()
    with pytest.raises(engine.EngineUnavailable):
        pyrayt.g3d.Sphere(1).intersect(pyrayt.g3d.bundle_of_rays(3))
    # and nothing under pyrayt_amd imports, loads or links the oracle
    pattern = re.compile(r"(import\s+.*oracle|from\s+.*oracle|libprt_oracle|prt_oracle)")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pyrayt_amd")):
        for name in files:
            if name.endswith((".py", ".hip", ".hpp", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, name)).read()
                assert not pattern.search(text), f"{name} references the oracle"


def test_generation_kernel_keeps_its_register_allocation(lib_path):
    """k_generation is tuned to 96 VGPRs without scratch (5 waves per SIMD).  A source change that looks
    harmless can cost it that (a waited-for atomic in the store path once did: 100 B of scratch per
    lane, 58 -> 73 us per launch), and nothing but the clock would notice: read it off the code object."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.READELF):
        pytest.skip("llvm-readelf not available")
    kernels = {name: res for name, res in mod.kernel_resources(lib_path).items() if "k_generation" in name}
    assert len(kernels) == 4, sorted(kernels)
    for name, res in kernels.items():
        assert res["private_segment_fixed_size"] == 0, (name, res)
        assert res["vgpr_spill_count"] == 0 and res["vgpr_count"] <= 96, (name, res)
