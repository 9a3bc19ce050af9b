#!/usr/bin/env python3
"""Register / scratch / LDS figures of the kernels in libprt_hip.so, read from the code object's metadata
(no GPU, no recompilation): the gfx950 ELF is cut out of the offload bundle inside the .so and its
AMDGPU metadata note is printed by llvm-readelf.

usage: tools/kernel_resources.py [libprt_hip.so] [name-substring ...]
Used by tests/test_abi.py to hold k_generation to its tuned allocation (96 VGPRs, no scratch, 5 waves/SIMD):
a change that looks harmless in the source can cost the kernel its registers (a waited-for atomic in the
store path once did: 100 B of scratch per lane, 58 -> 73 us per launch)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_object(lib_path, arch="gfx950"):
    """bytes of the device ELF for `arch` inside the shared library"""
    blob = open(lib_path, "rb").read()
    at = blob.find(MAGIC)
    if at < 0:
        raise RuntimeError(f"{lib_path}: no offload bundle")
    (count,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
    pos = at + len(MAGIC) + 8
    for _ in range(count):
        offset, size, triple_len = struct.unpack_from("<QQQ", blob, pos)
        triple = blob[pos + 24: pos + 24 + triple_len].decode()
        pos += 24 + triple_len
        if arch in triple and size:
            return blob[at + offset: at + offset + size]
    raise RuntimeError(f"{lib_path}: no {arch} code object in the bundle")


def kernel_resources(lib_path):
    """{kernel name: {vgpr_count, sgpr_count, private_segment_fixed_size, group_segment_fixed_size, ...}}"""
    with tempfile.NamedTemporaryFile(suffix=".elf") as tmp:
        tmp.write(code_object(lib_path))
        tmp.flush()
        text = subprocess.run([READELF, "--notes", tmp.name], capture_output=True, text=True, check=True).stdout
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        key, value = m.group(1), m.group(2).strip().strip("'")
        if key == "agpr_count":      # first key of a kernel's record (keys are sorted)
            cur = {}
        if cur is None:
            continue
        if key == "name":
            out[value] = cur
        elif re.fullmatch(r"-?\d+", value):
            cur[key] = int(value)
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(ROOT, "pyrayt_amd", "csrc", "libprt_hip.so")
    wanted = [a for a in sys.argv[1:] if not a.endswith(".so")]
    for name, res in sorted(kernel_resources(lib).items()):
        if wanted and not any(w in name for w in wanted):
            continue
        print(f"{name[:60]:60s} vgpr {res.get('vgpr_count'):4d} sgpr {res.get('sgpr_count'):4d} "
              f"scratch {res.get('private_segment_fixed_size'):5d} B/lane  static LDS {res.get('group_segment_fixed_size'):6d} B  "
              f"vgpr spills {res.get('vgpr_spill_count')}")
