"""Catalog fit on `WORLD_SIZE` ranks (launched by torch.distributed.run): rank 0 builds the tables, every
rank receives them through broadcast_interpolator, fits its shard on its GPU and all-gathers the result
rows.  Backend: nccl (one GPU per rank) or, with ISO_WORLD_BACKEND=gloo, gloo with every rank on cuda:0
(single-GPU boxes).  Writes <out>/res<rank>.pkl and prints a JSON line on rank 0."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import isochrones_amd as ia
from isochrones_amd.catalog import broadcast_interpolator, fit_catalog, synthetic_catalog


def main():
    out = sys.argv[1]
    n_stars = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    backend = os.environ.get("ISO_WORLD_BACKEND", "nccl")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    ic0 = None
    if rank == 0:
        fehs = np.array([-1.0, -0.5, -0.25, 0.0, 0.25, 0.5]); masses = ia.grids.mist_masses()[25:140:2]
        ic0 = ia.synthetic_track(bands=("G", "BP", "RP"), fehs=fehs, masses=masses, eeps=np.arange(150.0, 700.0),
                                 eep_bounds=(150, 699), limits=dict(mass=(masses[0], masses[-1]), feh=(-1.0, 0.5), age=(5, 10.13)))
    ic = broadcast_interpolator(ic0, src=0)
    # every rank derives the same catalog from the shared tables
    cat, truth = synthetic_catalog(ic, n_stars, bands=["G", "BP", "RP"], seed=3, mag_unc=0.01)
    dist.barrier()
    t = time.perf_counter()
    res = fit_catalog(cat, ic, nwalkers=32, nburn=150, niter=100, seed=1)
    torch.cuda.synchronize()
    dist.barrier()
    wall = time.perf_counter() - t
    res.to_pickle(os.path.join(out, "res%d.pkl" % rank))
    truth.to_pickle(os.path.join(out, "truth.pkl"))
    if rank == 0:
        print(json.dumps({"n_stars": n_stars, "world": world, "backend": backend, "wall_s": wall,
                          "stars_per_s": n_stars / wall}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
