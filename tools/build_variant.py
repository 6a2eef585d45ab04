#!/usr/bin/env python
"""Build a second copy of libiso_hip.so with extra hipcc flags for A/B runs (select it with ISOCHRONES_AMD_LIB):

    python tools/build_variant.py nofmac -Xclang -target-feature -Xclang -fmacf64-inst
    ISOCHRONES_AMD_LIB=isochrones_amd/csrc/libiso_hip_nofmac.so python bench.py ...
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isochrones_amd.csrc import build as B          # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    objdir = os.path.join(B.OBJDIR, "variant_" + name)
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(B.HERE, "libiso_hip_%s.so" % name)
    cc = B.hipcc()
    objs, jobs = [], []
    for src in B.sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        jobs.append([cc] + B.FLAGS + extra + ["-c", src, "-o", obj])
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 2)) as ex:
        for rc in ex.map(lambda c: subprocess.run(c, cwd=B.HERE, stderr=subprocess.DEVNULL).returncode, jobs):
            if rc != 0:
                raise SystemExit("hipcc failed")
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, cwd=B.HERE)
    print(out)


if __name__ == "__main__":
    main()
