"""Ahead-of-time build of libe4s_hip.so for gfx950 (never at import).

    python -m e4s2024_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  The .so lands in-tree (e4s2024_amd/lib/) so that it travels to
the GPU box with the repo snapshot."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libe4s_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-fvisibility=hidden"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(obj, deps):
    return (not os.path.exists(obj)) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps)


def build(force: bool = False, verbose: bool = False, phase_prof: bool = False) -> str:
    """``phase_prof``: the tuning variant lib/libe4s_hip_prof.so (-DE4S_PHASE_PROF: per-workgroup phase timestamps, read by
    tools/phase_prof.py through E4S_HIP_LIB); never loaded by default."""
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build_prof" if phase_prof else "build")
    so = os.path.join(LIBDIR, "libe4s_hip_prof.so") if phase_prof else SO
    flags = FLAGS + (["-DE4S_PHASE_PROF"] if phase_prof else [])
    os.makedirs(objdir, exist_ok=True)
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(HERE), "include", "e4s_hip.h")]
    objs, procs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(so, objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", so] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return so


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, phase_prof="--phase-prof" in sys.argv))
