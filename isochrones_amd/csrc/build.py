"""Build libiso_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "iso_hip.hip")
OUT = os.path.join(HERE, "libiso_hip.so")
HEADER = os.path.join(HERE, "..", "..", "include", "isochrones_amd.h")


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def up_to_date() -> bool:
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(f) <= t for f in (SRC, HEADER, os.path.abspath(__file__)))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and up_to_date():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fno-fast-math", "-fgpu-rdc" if False else "-Wall", "-Wno-unused-function",
           "-o", OUT, SRC]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd, cwd=HERE)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
