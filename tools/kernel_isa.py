#!/usr/bin/env python3
"""Instruction streams of kernels in libprt_hip.so, for "did this source change touch the product kernel?" checks:
the gfx950 ELF is cut out of the offload bundle, llvm-objdump disassembles it, and per kernel whose (demangled) name
contains one of the given substrings the instruction text -- addresses and encodings stripped -- is hashed.

usage: tools/kernel_isa.py libA.so [libB.so] [--match k_generation] [--dump DIR]
With two libraries: prints which kernels of the match are identical and which differ (instruction counts beside them)."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import code_object  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def kernels(lib, match):
    with tempfile.NamedTemporaryFile(suffix=".elf") as tmp:
        tmp.write(code_object(lib))
        tmp.flush()
        text = subprocess.run([OBJDUMP, "-d", "-C", "--no-show-raw-insn", "--no-leading-addr", tmp.name],
                              capture_output=True, text=True, check=True).stdout
    out, name, body = {}, None, []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]* ?<(.*)>:$", line)
        if m:
            if name is not None:
                out[name] = body
            name, body = m.group(1), []
            continue
        if name is not None and line.strip():
            text_line = re.sub(r"<[^>]*>", "", re.sub(r"//.*$", "", line)).strip()  # (symbol references carry the kernel's name)
            if text_line and text_line != "...":
                body.append(text_line)
    if name is not None:
        out[name] = body
    return {k: v for k, v in out.items() if any(s in k for s in match)}


def main():
    args = sys.argv[1:]
    match, dump, libs = [], None, []
    while args:
        a = args.pop(0)
        if a == "--match":
            match.append(args.pop(0))
        elif a == "--dump":
            dump = args.pop(0)
        else:
            libs.append(a)
    match = match or ["k_generation"]
    tables = [kernels(lib, match) for lib in libs]
    for lib, table in zip(libs, tables):
        print(lib)
        for name, body in sorted(table.items()):
            digest = hashlib.sha256("\n".join(body).encode()).hexdigest()[:16]
            print(f"  {digest} {len(body):6d} instr  {name[:150]}")
            if dump:
                os.makedirs(dump, exist_ok=True)
                with open(os.path.join(dump, f"{os.path.basename(lib)}.{digest}.s"), "w") as f:
                    f.write(name + "\n" + "\n".join(body) + "\n")
    if len(tables) == 2:
        # kernels are paired by position among the matches of each library, sorted by name: a template parameter added
        # to a kernel changes its name, and the question is whether it changed its code
        import difflib
        a, b = sorted(tables[0].items()), sorted(tables[1].items())
        by_digest = {}
        for name, body in b:
            by_digest.setdefault(tuple(body), name)
        for name, body in a:
            if tuple(body) in by_digest:
                print(f"IDENTICAL  {name[:60]} == {by_digest[tuple(body)][:60]}")
                continue
            best = min(b, key=lambda item: abs(len(item[1]) - len(body)))
            delta = [d for d in difflib.unified_diff(body, best[1], lineterm="", n=0) if d[:1] in "+-" and d[:3] not in ("+++", "---")]
            print(f"DIFFERENT  {name[:60]} vs {best[0][:60]}: {len(delta)} differing lines")
            for d in delta[:8]:
                print("    " + d)


if __name__ == "__main__":
    main()
