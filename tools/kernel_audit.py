#!/usr/bin/env python
"""Kernel-family audit: what is in liballset_hip.so, what reaches it, what pins it.

    python tools/kernel_audit.py list                       # every __global__ instantiation in the built library, per family
    ALLSET_ABI_TRACE=gpurun_out/abi_trace.json rocprofv3 --kernel-trace --stats -d gpurun_out/audit -o audit -- \
        python -m pytest tests -q -m gpu                    # (GPU box) the whole suite: AUTO, strict and fp16x3 arithmetic
    python tools/kernel_audit.py report gpurun_out/abi_trace.json gpurun_out/audit > profiles/r06_kernel_audit.md

`report` joins three sources, none of them hand-written:
  * the library's own kernel table -- the `__device_stub__` symbols hipcc emits for every instantiated `__global__` template, read
    from the .so with `nm` and demangled;
  * the kernel names `rocprofv3 --kernel-trace --stats` saw while the whole `-m gpu` suite ran (every process of the run);
  * the per-entry-point call record the test session wrote (tests/conftest.py, ALLSET_ABI_TRACE): calls, arithmetic modes, first tests.
An instantiation the suite never launched is either unreachable from the dispatchers or untested; both are defects, and the report
lists them by name.  The entry-point half lists every exported symbol with its call count, the modes it ran under, and the tests
that pin it; a symbol no test calls is listed as such."""
from __future__ import annotations

import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "allset_amd", "liballset_hip.so")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return out


def canon(name: str) -> str:
    """A kernel name in a form both sources agree on: no return type, no parameter list, no `allset::`, no spaces, (bool)1 -> true."""
    name = name.strip().strip('"')
    name = re.sub(r"^void\s+", "", name)
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):                         # the parameter list starts at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    name = name[:cut]
    name = name.replace("allset::", "").replace("__device_stub__", "").replace(" ", "")
    name = name.replace("(bool)1", "true").replace("(bool)0", "false")
    name = re.sub(r"\((?:int|unsigned|long|unsignedint)\)(-?\d+)", r"\1", name)
    name = re.sub(r"\.kd$", "", name)
    return name


def library_kernels():
    syms = subprocess.run(["nm", "-C", "--defined-only", LIB], capture_output=True, text=True).stdout.splitlines()
    ks = set()
    for line in syms:
        if "__device_stub__" in line and "rocprim" not in line and "allset::" in line:
            ks.add(canon(line.split(" ", 2)[2]))
    return sorted(ks)


def family(k: str) -> str:
    return k.split("<")[0]


def exported_symbols():
    out = subprocess.run(["nm", "-D", "--defined-only", LIB], capture_output=True, text=True).stdout.splitlines()
    return sorted(s.split()[-1] for s in out if s.split()[-1].startswith("allset_"))


def launched_kernels(trace_dir):
    seen = collections.Counter()
    for path in glob.glob(os.path.join(trace_dir, "**", "*kernel_stats.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                nm = row.get("Name") or row.get("KernelName") or ""
                if "allset" in nm or "kernel" in nm:
                    seen[canon(nm)] += int(float(row.get("Calls", row.get("Count", 1)) or 1))
    return seen


def main():
    cmd = sys.argv[1] if len(sys.argv) > 1 else "list"
    ks = library_kernels()
    fams = collections.OrderedDict()
    for k in ks:
        fams.setdefault(family(k), []).append(k)
    if cmd == "list":
        print(f"{len(ks)} kernel instantiations in {os.path.relpath(LIB, ROOT)} ({os.path.getsize(LIB)} bytes, "
              f"{len(exported_symbols())} exported allset_* symbols)")
        for f, v in fams.items():
            print(f"{len(v):4d}  {f}")
        return
    trace = json.load(open(sys.argv[2]))
    seen = launched_kernels(sys.argv[3])
    print("# Kernel-family audit (generated: `python tools/kernel_audit.py report`; do not edit)\n")
    print(f"Library: `{os.path.relpath(LIB, ROOT)}`, {os.path.getsize(LIB)} bytes, {len(exported_symbols())} exported `allset_*` symbols, "
          f"{len(ks)} `__global__` instantiations in {len(fams)} families.  Launch record: `rocprofv3 --kernel-trace --stats` over the "
          "whole `pytest -m gpu` suite (which runs AUTO, strict and explicit-fp16x3 arithmetic).\n")
    print("## Kernel instantiations: built vs launched by the suite\n")
    print("| kernel family | built | launched by the suite | never launched (template arguments) |")
    print("|---|---|---|---|")
    dead_total = 0
    for f, v in fams.items():
        dead = [k for k in v if seen.get(k, 0) == 0]
        dead_total += len(dead)
        args = "; ".join("`" + (k[len(f):] or "-") + "`" for k in dead)
        print(f"| `{f}` | {len(v)} | {len(v) - len(dead)} | {args or '—'} |")
    print(f"\n{dead_total} of {len(ks)} instantiations were never launched.\n")
    unknown = sorted(k for k in seen if k not in set(ks) and "rocprim" not in k and not k.startswith(("at::", "void at::", "Cijk", "__amd")))
    print("## Exported entry points: calls in the suite, arithmetic modes, the tests that pin them\n")
    print("| entry point | calls | dense arithmetic at the call (auto / bf16x6 / fp16x3) | first tests |")
    print("|---|---|---|---|")
    for s in exported_symbols():
        r = trace.get(s)
        if r is None:
            print(f"| `{s}` | 0 | — | **none** |")
            continue
        m = r["modes"]
        tests = ", ".join("`" + t.replace("tests/", "") + "`" for t in r["tests"][:3])
        print(f"| `{s}` | {r['calls']} | {m.get('auto', 0)} / {m.get('bf16x6', 0)} / {m.get('fp16x3', 0)} | {tests} |")
    if unknown:
        print("\n(kernel names in the trace that are not this library's: " + ", ".join(f"`{u}`" for u in unknown[:12]) + ")")


if __name__ == "__main__":
    main()
