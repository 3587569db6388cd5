"""Find kernels whose global loads are issued one at a time ("load-pipelining pass").

Compiles every csrc/*.hip to gfx950 assembly (device side only, no GPU needed) and, per kernel, counts the global / buffer /
scratch loads whose NEXT memory-load instruction comes only after an `s_waitcnt vmcnt(0)`: each of those is a full round trip
that nothing else in the wave overlaps.  LDS-DMA (`global_load_lds_*`) is not counted.  A high count against the total is the
signature of (a) conversions or optional operands behind uniform branches, (b) a pointer fetched from the argument struct
inside a loop, (c) a store that may alias between two loads, (d) operands loaded where they are used instead of ahead of a wait
that exists anyway.

    python tools/isa_serial_loads.py [--min 3] [file.hip ...]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "motiondiffusion-moe_amd", "csrc")


def disassemble(src: str, out: str) -> None:
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-2000:]}")


def kernels(path: str):
    name, body = None, []
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        t = ln.strip()
        if t and not t.startswith(";") and not t.startswith("."):
            body.append(t)
        if "s_endpgm" in ln:
            yield name, body
            name = None


def serialized(body):
    loads = [i for i, t in enumerate(body) if re.match(r"(global_load|buffer_load|scratch_load)", t) and "lds" not in t]
    waits = [i for i, t in enumerate(body) if t.startswith("s_waitcnt") and "vmcnt(0)" in t]
    ser = 0
    for a, b in zip(loads, loads[1:]):
        if any(a < w < b for w in waits):
            ser += 1
    return len(loads), ser


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--min", type=int, default=3, help="report kernels with at least this many serialized loads")
    a = ap.parse_args()
    files = a.files or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    with tempfile.TemporaryDirectory() as tmp:
        outs = [os.path.join(tmp, os.path.basename(f) + ".s") for f in files]
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(lambda fo: disassemble(*fo), zip(files, outs)))
        for f, o in zip(files, outs):
            for name, body in kernels(o):
                n, ser = serialized(body)
                if ser >= a.min:
                    short = re.sub(r"^_ZN3mdm\d+_GLOBAL__N_1\d+", "", name)[:72]
                    print(f"{os.path.basename(f):16s} {short:72s} loads={n:4d} serialized={ser}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
