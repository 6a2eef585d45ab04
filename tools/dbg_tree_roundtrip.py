import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import isochrones_amd as ia
from isochrones_amd import priors
from tests.test_tree_cpu import build_notebook_tree
ic = ia.get_ichrone("mist", bands=["J", "H", "K"])
tree = ia.TreeStarModel(ic, obs=build_notebook_tree("nb"), parallax=(2.0, 0.05), Teff=(5834.0, 100), name="nb")
tree.set_prior(AV=priors.FlatPrior((0, 0.5)))
tree.fit(n_live_points=60, max_iter=150, seed=2)
tree.save("/tmp/tree.npz", overwrite=True)
tb = ia.TreeStarModel.load("/tmp/tree.npz", ic=ic)
q = tree.samples[list(tree.param_names)].values[:40]
print("rows", q.shape, len(tree.samples))
a, b = tb.lnpost(q), tree.lnpost(q)
print("lnpost", a, b, a - b)
print("prior", tb.lnprior(q) - tree.lnprior(q), "like", tb.lnlike(q) - tree.lnlike(q))
raw = lambda d: bytes(C.string_at(C.addressof(d), C.sizeof(d)))
d0, d1 = tree.tree_desc(), tb.tree_desc()
print("desc equal", raw(d0) == raw(d1))
if raw(d0) != raw(d1):
    for name, _ in d0._fields_:
        x, y = getattr(d0, name), getattr(d1, name)
        bx = bytes(C.string_at(C.addressof(x), C.sizeof(x))) if hasattr(x, "_fields_") or hasattr(x, "_length_") else x
        by = bytes(C.string_at(C.addressof(y), C.sizeof(y))) if hasattr(y, "_fields_") or hasattr(y, "_length_") else y
        if bx != by:
            print("field differs:", name)
            if name.startswith("prior_"):
                for f2, _ in x._fields_:
                    if getattr(x, f2) != getattr(y, f2):
                        print("   ", f2, repr(getattr(x, f2)), repr(getattr(y, f2)))
            elif name.startswith("bound"):
                print("   ", list(x), list(y))
for k in range(5):
    print("repeat", tree.lnpost(q) - b, tb.lnpost(q) - a)
