import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench_configs
tmod, tpars = bench_configs.tree_model_and_samples(16)
tp = list(tpars[0])
for _ in range(2000): tmod.lnpost(tp)
best = 1e9
for rep in range(5):
    t = time.perf_counter()
    for _ in range(20000): tmod.lnpost(tp)
    best = min(best, (time.perf_counter() - t) / 20000)
print("%s tree lnpost(p) %.2f us" % (os.path.basename(os.environ.get("ISOCHRONES_AMD_LIB", "default")), best * 1e6))
