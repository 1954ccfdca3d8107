#!/usr/bin/env python3
"""Scans the gfx950 ISA inside libwetts_hip.so for SERIALISED LOAD ROUND TRIPS: runs of `global/buffer load ... s_waitcnt
vmcnt(0)` pairs, i.e. places where every load waits for the memory system before the next one is issued (no GPU needed).

Why: a source loop of the form `v = load(a); if (flag) v += load(b);` per element compiles -- with a uniform `flag` -- to
branch, two loads, `s_waitcnt vmcnt(0)`, add, for EVERY element.  In round 5 that shape cost the f32 headline 1.7 %
(conv_mfma_kernel's accumulator init: 64 serialised round trips per wave in front of its first MFMA, found through the SQ
counters, profiles/r05_sq_counters_mrf.txt) and a third of every LayerNorm launch.  `python tools/isa_scan.py [lib]` prints,
per kernel, the longest such run and the number of pairs; tests/test_cpu_kernel_resources.py pins the kernels that were fixed.
(Runs of 64 remain in conv_mfma_kernel's EDGE-tile paths -- per-element predicated loads of partial tiles -- by design.)"""
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(lib):
    """{mangled kernel name: [instruction lines]} of every gfx950 code object embedded in `lib`."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        base = os.path.join(td, os.path.basename(lib))
        with open(lib, "rb") as src, open(base, "wb") as dst:
            dst.write(src.read())
        subprocess.run([OBJDUMP, "--offloading", base], cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        for f in sorted(os.listdir(td)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", os.path.join(td, f)], capture_output=True, text=True, check=False).stdout
            name = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    name = m.group(1)
                    out[name] = []
                elif name and line.startswith("\t"):
                    out[name].append(line.strip())
    return out


def serial_runs(lines, window=6, gap=40):
    """(longest run, number) of `load ... s_waitcnt vmcnt(0)` pairs: a wait counts when a global / buffer load (not an
    LDS-DMA) sits within `window` instructions in front of it; pairs closer than `gap` instructions form a run."""
    ops = [l.split()[0] if l.split() else "" for l in lines]
    pos = []
    for i, l in enumerate(lines):
        if "s_waitcnt" in l and "vmcnt(0)" in l:
            back = ops[max(0, i - window):i]
            if any(("global_load" in o or "buffer_load" in o) and "lds" not in o for o in back):
                pos.append(i)
    best = cur = 1 if pos else 0
    for a, b in zip(pos, pos[1:]):
        cur = cur + 1 if b - a < gap else 1
        best = max(best, cur)
    return best, len(pos)


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=False)
    return dict(zip(names, p.stdout.splitlines())) if p.returncode == 0 else {n: n for n in names}


def scan(lib):
    isa = disassemble(lib)
    dm = demangle(list(isa))
    return {re.sub(r"\(.*", "", dm[n]).replace("void ", ""): serial_runs(lines) for n, lines in isa.items()}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                              "wetts_amd", "lib", "libwetts_hip.so")
    rows = sorted(scan(lib).items(), key=lambda kv: (-kv[1][0], kv[0]))
    print(f"{'longest run':>11s} {'pairs':>6s}  kernel")
    for k, (run, n) in rows:
        if run >= 4:
            print(f"{run:11d} {n:6d}  {k}")
