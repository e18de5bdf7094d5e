#!/usr/bin/env python3
"""Static instruction census of one kernel of integrator.hip, attributed to source functions and lines (no GPU needed).
Compiles the translation unit with -gline-tables-only into a scratch directory, disassembles the gfx950 code object with source
line markers and counts vector / LDS / memory instructions per inlined function and per line.  This is how the node step of
traverse.h was put on its diet in round 3 (profiles/r03_experiments.txt): the counts are static — a branch that is rarely taken
counts in full — so read them next to the PMC figures of profiles/*_pmc_*.txt.

usage: scripts/isa_attr.py '<mangled-name prefix or demangled substring>' [source file to list the top lines of]
  e.g. scripts/isa_attr.py _ZN2rt7k_shadeILb0ELb0ELb0E shading.h
       scripts/isa_attr.py 'k_trace_shadow_then_batch<false, false, false>' traverse.h
"""
import bisect
import collections
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hala-renderer_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(obj, out):
    data = open(obj, "rb").read()
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    cnt = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(cnt):
        eo, es, ts = struct.unpack_from("<QQQ", data, off); off += 24
        triple = data[off:off + ts].decode(); off += ts
        if "gfx" in triple:
            open(out, "wb").write(data[i + eo:i + eo + es])
            return
    raise SystemExit("no gfx code object in " + obj)


def functions(path):
    out = []
    for n, l in enumerate(open(path), 1):
        m = re.search(r"RT_DI\s+[\w:<>\*& ]+?\s+(\w+)\s*\(", l) or re.search(r"\)\s+(k_\w+)\(", l) or re.match(r"^__global__.*\s(k_\w+)\(", l)
        if m and not l.strip().startswith("//"):
            out.append((n, m.group(1)))
    return out


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    want = sys.argv[1]
    tmp = tempfile.mkdtemp(prefix="isa_attr_")
    obj = os.path.join(tmp, "integrator.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-mfma",
                           "-gline-tables-only", "-x", "hip", "-c", os.path.join(CSRC, "integrator.hip"), "-o", obj])
    co = os.path.join(tmp, "integrator.co")
    code_object(obj, co)
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "-l", "--mcpu=gfx950", co], capture_output=True, text=True, check=True).stdout.split("\n")
    start = end = None
    for i, l in enumerate(dis):
        m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
        if not m:
            continue
        name = m.group(1)
        pretty = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if start is None and (name.startswith(want) or want in pretty):
            start = i
            print("kernel:", pretty[:160])
        elif start is not None:
            end = i
            break
    if start is None:
        raise SystemExit("no kernel matches " + want)
    cur = None
    by_line = collections.Counter()
    kinds = collections.Counter()
    for l in dis[start:end]:
        m = re.match(r"^; (/[^:]+):(\d+)", l)
        if m:
            cur = (os.path.basename(m.group(1)), int(m.group(2)))
            continue
        t = l.strip().split()
        if t and "//" in l:
            k = "valu" if t[0].startswith("v_") else "lds" if t[0].startswith("ds_") else "vmem" if t[0].startswith(("global_", "buffer_", "flat_", "scratch_")) else "scalar"
            kinds[k] += 1
            if k != "scalar":
                by_line[cur] += 1
    print("instructions:", dict(kinds))
    src = {f: functions(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))}
    by_fn = collections.Counter()
    for k, c in by_line.items():
        if k is None:
            by_fn[("?", "?")] += c
            continue
        fl = src.get(k[0], [])
        idx = bisect.bisect_right([x[0] for x in fl], k[1]) - 1
        by_fn[(k[0], fl[idx][1] if idx >= 0 else "?")] += c
    print("vector + memory instructions by (inlined) function:")
    for k, v in by_fn.most_common(40):
        print(f"  {v:6d}  {k[0]}:{k[1]}")
    if len(sys.argv) > 2:
        print("top lines of", sys.argv[2])
        for k, v in sorted(by_line.items(), key=lambda x: -x[1]):
            if k and k[0] == sys.argv[2]:
                print(f"  {v:6d}  {k[0]}:{k[1]}")
                if v < 20:
                    break


if __name__ == "__main__":
    main()
