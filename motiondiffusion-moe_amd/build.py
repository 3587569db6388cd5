"""Build libmdm_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build(); also runnable:
    python motiondiffusion-moe_amd/build.py [--force] [--diag]
--diag builds the DIAGNOSTIC library libmdm_hip_diag.so (-DMDM_DIAG: the timing-only knock-out / stamped instantiations of the
fused expert MLP behind knobs 41..49, whose outputs are wrong by construction).  Only tools/mlp_ko.py and tools/mlp_stamps.py
load it (MDM_LIB=.../libmdm_hip_diag.so); the product library does not contain those kernels and refuses their knobs.
hipcc cross-compiles without a GPU.  Objects go to csrc/build/, the library next to this file so it
travels to the GPU box with the source snapshot."""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libmdm_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
         "-Wno-unused-result"]


# per-file flags: the fused MLP keeps its GELU on scalar fp32 instructions (packed fp32 is slow beside MFMAs)
EXTRA = {"mlp_stream.hip": ["-fno-slp-vectorize"]}


def _targets(diag: bool):
    if diag:
        return os.path.join(CSRC, "build_diag"), DIAG_LIB, FLAGS + ["-DMDM_DIAG"]
    return OBJ, LIB, FLAGS


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


DIAG_LIB = os.path.join(HERE, "libmdm_hip_diag.so")


def build(force: bool = False, verbose: bool = True, diag: bool = False) -> str:
    obj_dir, lib_path, flags = _targets(diag)
    os.makedirs(obj_dir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    jobs = []
    for s in srcs:
        o = os.path.join(obj_dir, os.path.basename(s)[:-4] + ".o")
        if force or _newer([s] + hdrs, o):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = ["hipcc"] + flags + EXTRA.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[mdm build] compiled {os.path.basename(s)}", flush=True)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(obj_dir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _newer(objs, lib_path):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print(f"[mdm build] linked {lib_path}", flush=True)
    return lib_path


if __name__ == "__main__":
    build(force="--force" in sys.argv, diag="--diag" in sys.argv)
