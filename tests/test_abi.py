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
    from pyrayt_amd import engine

    header_version = int(re.search(r"#define PRT_VERSION (\d+)", open(HEADER).read()).group(1))
    assert lib.prt_version() == header_version == engine.PRT_VERSION
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


def test_a_library_of_another_abi_version_is_refused(lib_path, monkeypatch):
    """include/prt.h changes incompatibly between versions (argument lists grow, output blocks widen): the
    binding compares prt_version() with the version it was written for and refuses a stale build."""
    from pyrayt_amd import engine

    monkeypatch.setattr(engine, "_lib", None)
    monkeypatch.setattr(engine, "PRT_VERSION", engine.PRT_VERSION + 1)
    with pytest.raises(engine.EngineUnavailable, match="ABI version"):
        engine.library()
    monkeypatch.setattr(engine, "PRT_VERSION", engine.PRT_VERSION - 1)
    assert engine.library() is not None


def test_user_defined_materials_are_classified_like_upstream_dispatches_them():
    """pyrayt/_pyrayt.py:408 calls surface.material.trace whatever the material is; which PRT_MAT_* kind serves
    an object is decided by where its trace / index_at are defined (pyrayt_amd.materials.device_kind)."""
    import scenes
    from pyrayt_amd import materials as m
    from pyrayt_amd.g3d.materials import gooch

    user = scenes.user_materials(scenes.product_api())

    class MyBK7(m.SellmeierRefractor):  # numbers only: keeps the closed form
        pass

    class Odd(m.BasicRefractor):  # redefines index_at: a table
        def index_at(self, wavelength):
            return 2.0

    class Duck:  # not even a TracableMaterial: upstream would call its trace() all the same
        def trace(self, surface, ray_set):
            return ray_set

    kinds = [(m.absorber, m.ABSORBER), (m.mirror, m.MIRROR), (m.glass["ideal"], m.CONST_INDEX),
             (m.glass["SF5"], m.SELLMEIER), (MyBK7(1, 2, 3), m.SELLMEIER), (Odd(1.5), m.TABLE),
             (user.CauchyGlass(1.5, 0.004), m.TABLE), (user.RetroReflector(), m.HOST),
             (user.LossyGlass(1.5, 0.004, 0.9), m.HOST), (user.ShiftingMirror(0.02), m.HOST), (Duck(), m.HOST),
             (gooch.BLACK, m.NONE), (object(), m.NONE)]
    for material, want in kinds:
        assert m.device_kind(material) == want, material
    # what super().trace() of a user's trace() reaches: the material's own shading arithmetic
    assert m.device_kind(user.LossyGlass(1.5, 0.004, 0.9), shading_only=True) == m.TABLE
    assert m.device_kind(user.RetroReflector(), shading_only=True) == m.NONE
    # constructors as upstream's (materials.py:12-24): base_material first, defaults per class
    assert user.CauchyGlass(1.5, 0.004)._base_material is gooch.BLUE
    assert user.RetroReflector()._base_material is gooch.BLACK
    assert m.TracableMaterial.__init__.__code__.co_varnames[:2] == ("self", "base_material")
    assert np.array_equal(m.table_indices(Odd(1.5), [0.4, 0.5]), [2.0, 2.0])


def test_snapshot_and_index_tables_without_a_gpu(lib_path):
    """The snapshot marks user-defined glasses (tables) and user-shaded surfaces; prt_scene_set_index_tables is
    host code until a device copy exists and validates what it is given."""
    import scenes
    from pyrayt_amd import engine
    from pyrayt_amd import materials as m
    from pyrayt_amd.g3d.objects import CountedObject
    from pyrayt_amd.scene import SceneSnapshot

    CountedObject.reset_ids()
    parts, _ = scenes.custom_mixed(scenes.product_api(), 16)
    snap = SceneSnapshot(parts)
    kinds = snap.materials["kind"].tolist()
    assert kinds == [m.HOST, m.HOST, m.TABLE, m.SELLMEIER, m.ABSORBER]
    assert [slot for slot, _ in snap.table_materials] == [2]
    assert [p for p, _ in snap.host_surfaces] == [0, 1, 2, 3]  # the lossy lens's three surfaces, the shifting mirror
    scene = engine.DeviceScene(snap)
    scene.ensure_tables([0.6, 0.5, 0.6, np.nan])
    lam, idx = scene._tables
    assert np.array_equal(lam, [0.5, 0.6]) and np.array_equal(idx[0], 1.52 + 0.0048 / lam ** 2)
    scene.ensure_tables([0.55])  # the union, re-evaluated
    assert np.array_equal(scene._tables[0], [0.5, 0.55, 0.6])
    lib = engine.library()
    ranges = np.zeros((5, 2), dtype=np.int64)
    ranges[2] = (0, 2)
    lam = np.array([0.6, 0.5])
    rc = lib.prt_scene_set_index_tables(scene.handle, ranges.ctypes.data, 5, lam.ctypes.data, lam.ctypes.data, 2)
    assert rc == -1 and b"ascending" in lib.prt_last_error()
    rc = lib.prt_scene_set_index_tables(scene.handle, ranges.ctypes.data, 4, lam.ctypes.data, lam.ctypes.data, 2)
    assert rc == -1
    ranges[2] = (1, 2)
    rc = lib.prt_scene_set_index_tables(scene.handle, ranges.ctypes.data, 5, lam.ctypes.data, lam.ctypes.data, 2)
    assert rc == -1 and b"bounds" in lib.prt_last_error()
    scene.close()


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
    assert len(kernels) == 8, sorted(kernels)  # CULL x COMPACT x PLAN (round 6: the record-plan instantiations, held to the same)
    for name, res in kernels.items():
        assert res["private_segment_fixed_size"] == 0, (name, res)
        assert res["vgpr_spill_count"] == 0 and res["vgpr_count"] <= 96, (name, res)


def test_package_asks_for_hardware_queues_whichever_import_comes_first():
    """GPU_MAX_HW_QUEUES is read when the HIP runtime INITIALISES (first HIP call), not when torch loads it
    (profiles/r5/queue_probe.txt): importing pyrayt_amd sets it before or after ``import torch`` (a fourth trace in
    flight then gets a queue of its own); a user's setting stays.  (Set late it is checked by running something the
    first time more than three ticket streams are asked for: tests/test_gpu_parity.py.)"""
    import sys

    def run(code, env_extra=None):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        env.update(env_extra or {})
        done = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
        assert done.returncode == 0, done.stderr[-1500:]
        return done.stdout.split()

    show = "import pyrayt_amd.engine as e, os; print(e.HW_QUEUES, os.environ.get('GPU_MAX_HW_QUEUES'))"
    assert run(show) == ["set", "8"]
    assert run("import torch; " + show) == ["set-late", "8"]
    assert run(show, {"GPU_MAX_HW_QUEUES": "2"}) == ["user", "2"]


def test_binding_constants_are_the_headers():
    """Every PRT_TRACE_* flag and PRT_ERR_* code the binding names has the header's value, and no two flags share a
    bit (the flags travel as one int through prt_trace / prt_trace_begin / prt_trace_batch)."""
    from pyrayt_amd import engine

    header = open(HEADER).read()
    flags = {name: int(value) for name, value in re.findall(r"#define PRT_TRACE_([A-Z_]+) (\d+)", header)}
    tickets = flags.pop("TICKETS")
    assert tickets == engine.TRACE_TICKETS
    assert len(flags) >= 11 and all(v & (v - 1) == 0 for v in flags.values()) and len(set(flags.values())) == len(flags)
    for name, value in flags.items():
        assert getattr(engine, "TRACE_" + name) == value, name
    errors = {name: int(value) for name, value in re.findall(r"#define PRT_ERR_([A-Z_]+) \((-\d+)\)", header)}
    for name in ("ROWS_CAP", "UNTRACABLE", "WAVELENGTH"):
        assert getattr(engine, "ERR_" + name) == errors[name], name


def test_binding_structs_have_the_layout_of_the_header(tmp_path):
    """The numpy record types the binding hands to the library (scene options, trace jobs, record plans, sources, cameras)
    against the C structs of include/prt.h: size and the offset of every field, read off a program gcc compiles from the
    header itself."""
    import shutil
    import subprocess

    from pyrayt_amd import engine

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    structs = {
        "prt_scene_options": (engine.OPTIONS_DTYPE, ["struct_size", "no_chain", "hit_lanes", "no_clearance", "reserved"]),
        "prt_trace_job": (engine.JOB_DTYPE, None),
        "prt_record_plan": (engine.PLAN_DTYPE, ["struct_size", "n_surfaces", "store_rows", "n_groups", "surfaces", "rays_per_source",
                                                "sums_out", "pivots", "ms_quantity", "ms_transform", "ms_about",
                                                "generation_limit", "columns"]),
        "prt_camera": (engine.CAMERA_DTYPE, ["world", "h_pixels", "v_pixels", "h_width", "v_width"]),
    }
    lines = ["#include <stdio.h>", "#include <stddef.h>", f'#include "{os.path.join(ROOT, "include", "prt.h")}"', "int main(void) {"]
    for name, (dtype, fields) in structs.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for field in fields or ():
            lines.append(f'  printf("{name}.{field} %zu\\n", offsetof({name}, {field}));')
    lines += ["  return 0;", "}"]
    source = tmp_path / "layout.c"
    source.write_text("\n".join(lines))
    binary = tmp_path / "layout"
    subprocess.run(["gcc", "-o", str(binary), str(source)], check=True, capture_output=True)
    got = dict(line.split() for line in subprocess.run([str(binary)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, (dtype, fields) in structs.items():
        assert int(got[name]) == dtype.itemsize, (name, got[name], dtype.itemsize)
        for field in fields or ():
            assert int(got[f"{name}.{field}"]) == dtype.fields[field][1], (name, field)


def test_scene_objects_count_their_changes():
    """g3d.objects.SceneEpoch moves on every attribute assignment of a scene object -- a transform, a material, a normal
    flip, a source's wavelength -- and stands still otherwise: what RayTracer keys its compiled scene and its generated
    rays on (the GPU side of that is tests/test_gpu_parity.py::test_raytracer_looks_at_its_system_again_...)."""
    import pyrayt_amd as pyrayt
    from pyrayt_amd.g3d.objects import SceneEpoch

    lens = pyrayt.components.biconvex_lens(2, 2, 0.25, aperture=1)
    source = pyrayt.components.ConeOfRays(cone_angle=6)
    plate = pyrayt.components.baffle((1, 1))
    seen = SceneEpoch.value
    lens.get_world_transform(); plate.bounding_box; source.wavelength; lens.surface_ids    # reads do not count
    assert SceneEpoch.value == seen
    glass = pyrayt.materials.BasicRefractor(1.5)
    ball = pyrayt.g3d.Sphere(1, material=glass)
    for change in (lambda: lens.move_x(0.1), lambda: plate.rotate_z(3), lambda: setattr(plate, "material", pyrayt.materials.mirror),
                   plate.invert_normals, lambda: setattr(source, "wavelength", 0.5), lambda: source.move_x(-1),
                   lambda: setattr(glass, "_refractive_index", 1.6),                   # a material's own numbers
                   lambda: setattr(ball.primitive, "params", (2.0,))):                  # a shape's parameters
        change()
        assert SceneEpoch.value > seen
        seen = SceneEpoch.value
