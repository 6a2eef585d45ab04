"""Build libiso_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).
The translation units are compiled in parallel and linked into one shared library."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "libiso_hip.so")
OBJDIR = os.path.join(HERE, "build")
HEADERS = [os.path.join(HERE, "..", "..", "include", "isochrones_amd.h"),
           os.path.join(HERE, "iso_internal.h"), os.path.join(HERE, "iso_fast_kernel.h")] + \
    sorted(glob.glob(os.path.join(HERE, "kernels", "*.h"))) + sorted(glob.glob(os.path.join(HERE, "fast", "*.h")))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-Wall",
         "-Wno-unused-function"]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    me = os.path.abspath(__file__)
    if not force and not verbose and not _newer(OUT, sources() + HEADERS + [me]):
        return OUT                       # library newer than every source: nothing to do
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _newer(obj, [src, me] + HEADERS):
            cmd = [cc] + FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if verbose else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 2)) as ex:
            for rc in ex.map(lambda c: subprocess.run(c, cwd=HERE).returncode, jobs):
                if rc != 0:
                    raise RuntimeError("hipcc failed")
    if force or jobs or _newer(OUT, objs):
        subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs, cwd=HERE)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
