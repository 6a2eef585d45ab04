#!/usr/bin/env python
"""Build a second copy of libiso_hip.so with extra hipcc flags for A/B runs (select it with ISOCHRONES_AMD_LIB):

    python tools/build_variant.py nofmac -Xclang -target-feature -Xclang -fmacf64-inst
    ISOCHRONES_AMD_LIB=isochrones_amd/csrc/libiso_hip_nofmac.so python bench.py ...

    --src DIR    compile the *.hip of another source tree (e.g. an older commit extracted with git archive)
    --only a,b   compile only these translation units with the extra flags / from DIR; the others are the objects of the
                 regular build (isochrones_amd/csrc/build) or, with --base NAME, of the variant NAME
The library goes to variants/libs/libiso_hip_<name>.so when --src / --only is given (git-ignored, travels with gpurun),
next to it <name>.resources.txt (registers / scratch of every kernel compiled for it)."""
import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from isochrones_amd.csrc import build as B          # noqa: E402
from isochrones_amd.csrc import resources as R      # noqa: E402


def main():
    ap = argparse.ArgumentParser(allow_abbrev=False)
    ap.add_argument("name")
    ap.add_argument("--src", default=None)
    ap.add_argument("--only", default=None)
    ap.add_argument("--base", default=None)
    args, extra = ap.parse_known_args()
    here = os.path.abspath(args.src) if args.src else B.HERE
    special = bool(args.src or args.only)
    objdir = os.path.join(ROOT, "variants", "obj", args.name) if special else os.path.join(B.OBJDIR, "variant_" + args.name)
    os.makedirs(objdir, exist_ok=True)
    outdir = os.path.join(ROOT, "variants", "libs") if special else B.HERE
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, "libiso_hip_%s.so" % args.name)
    cc = B.hipcc()
    only = set(args.only.split(",")) if args.only else None
    basedir = os.path.join(ROOT, "variants", "obj", args.base) if args.base else B.OBJDIR
    objs, jobs = [], []
    import glob
    for src in sorted(glob.glob(os.path.join(here, "*.hip"))):
        tu = os.path.basename(src)[:-4]
        if only is not None and tu not in only:
            objs.append(os.path.join(basedir, tu + ".o"))
            continue
        obj = os.path.join(objdir, tu + ".o")
        objs.append(obj)
        jobs.append(([cc] + B.unit_flags(src) + ["-Rpass-analysis=kernel-resource-usage"] + extra + ["-c", src, "-o", obj], obj[:-2] + ".res"))

    def run(job):
        cmd, log = job
        p = subprocess.run(cmd, cwd=here, stderr=subprocess.PIPE, text=True, errors="replace")
        open(log, "w").write(p.stderr)
        if p.returncode != 0:
            sys.stderr.write(p.stderr[-4000:])
        return p.returncode

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 2)) as ex:
        for rc in ex.map(run, jobs):
            if rc != 0:
                raise SystemExit("hipcc failed")
    subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, cwd=here)
    table = {}
    for _, log in jobs:
        table.update(R.parse(open(log, errors="replace").read()))
    open(out[:-3] + ".resources.txt", "w").write(R.render(table))
    print(out)


if __name__ == "__main__":
    main()
