"""Create / use / drop every handle type repeatedly and watch the device's free memory."""
import gc, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isochrones_amd as ia
from isochrones_amd.catalog import CatalogPosterior, synthetic_catalog, initial_positions
from isochrones_amd.sampler import FusedEnsembleSampler


def once(seed):
    rng = np.random.default_rng(seed)
    fehs = np.array([-1.0, -0.5, 0.0, 0.5]); masses = ia.grids.mist_masses()[30:120:3]; eeps = np.arange(150.0, 650.0)
    ic = ia.synthetic_track(bands=("G", "BP", "RP"), fehs=fehs, masses=masses, eeps=eeps, eep_bounds=(150, 649),
                            limits=dict(mass=(masses[0], masses[-1]), feh=(-1.0, 0.5), age=(5, 10.13)))
    truth = np.array([1.05, 330.0, -0.1, 200.0, 0.15])
    T, g, f, mags = ic.interp_mag(truth, ["G", "BP", "RP"])
    mod = ia.SingleStarModel(ic, Teff=(T, 80), logg=(g, 0.1), G=(mags[0], 0.02), parallax=(5.0, 0.1), nu_max=(2000., 100.))
    n = 40000
    p = truth + 0.01 * rng.standard_normal((n, 5))
    mod.lnpost(p); mod.lnlike(p)
    pt = torch.as_tensor(np.ascontiguousarray(p.T), device="cuda")
    ic.interp_mag_device(pt, ["G", "RP"]); ic.interp_mag_device(pt, ["BP"])
    ic.model_grid.interp.interp_device([pt[2], pt[0], pt[1]], np.arange(6))
    ic.get_eep(1.0, 9.5, 0.0)
    mod2 = ia.SingleStarModel(ic, G=(mags[0], 0.02), BP=(mags[1], 0.02), parallax=(5.0, 0.1))
    mod2.fit_mcmc(nwalkers=32, nburn=5, niter=5, seed=1)
    mod2.fit_multinest(n_live_points=50, max_iter=200)
    cat, _ = synthetic_catalog(ic, 20, bands=["G", "BP", "RP"], seed=seed)
    post = CatalogPosterior.from_catalog(cat, ic, N=1, indices=np.arange(20))
    pos, lnp, failed = initial_positions(post, 16, rng_seed=1)
    fs = FusedEnsembleSampler(post, 16, seed=1); fs.run_mcmc(pos, 10, lnprob0=lnp); fs.close(); post.close()
    ic.release()


torch.cuda.init()
once(0); gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
base = torch.cuda.mem_get_info()[0]
for k in range(1, 13):
    once(k); gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    print(k, "free MiB delta vs first:", (torch.cuda.mem_get_info()[0] - base) / 2**20, flush=True)
