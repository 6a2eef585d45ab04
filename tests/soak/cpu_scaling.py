"""How does the CPU oracle (C port, OpenMP) scale on this host?  Prints evals/s per thread count."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
ic, mod = bench.build_model()
oic = bench.oracle_view(ic)[1]
desc = mod.model_desc()
pars = bench.make_samples(np.random.default_rng(1), 400_000, "prior_valid")
soa = np.ascontiguousarray(pars.T)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
for nt in (1, 2, 4, 8, 16, 32, 64, 128):
    oic.lnpost(desc, soa[:, :20000], nthreads=nt, parts=False)
    t = time.perf_counter()
    oic.lnpost(desc, soa, nthreads=nt, parts=False)
    dt = time.perf_counter() - t
    print(nt, "threads: %.3g evals/s" % (soa.shape[1] / dt), flush=True)
