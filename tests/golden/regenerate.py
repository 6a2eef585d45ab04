"""Regenerate every fixture in this directory by RUNNING THE REFERENCE (timothydmorton/isochrones at
/root/reference) in the authoring container.  The work is done by oracle/make_golden.py (synthetic tables,
seeded samples) on top of oracle/ref_harness.py (numba identity shim + import-only stubs, so the
reference's interp.py / mags.py / likelihood.py / priors.py / starmodel.py / observation.py run as the pure
Python they are).  Nothing from the reference is stored here: the .npz files hold inputs and expected outputs.

    python tests/golden/regenerate.py            # everything
    python tests/golden/regenerate.py --only-priors | --only-tree | --only-eep | --only-isotrack

The reference tree does not exist on the GPU box; the tests only read the committed fixtures."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
runpy.run_path(os.path.join(ROOT, "oracle", "make_golden.py"), run_name="__main__")
