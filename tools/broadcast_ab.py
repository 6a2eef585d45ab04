"""Start-up cost of a rank that RECEIVES the tables of a catalog fit: `broadcast_interpolator` over a world-1 `nccl` (= RCCL)
group with `rebuild_on_src` (the one-GPU stand-in of a receiving rank), full-size MIST-shaped track tables, until the
interpolator's device tables and packs exist (`ic.handle()`):
    device   the tensors that arrived are handed to the library (iso_table_create_from_device) - round 6
    host     device -> numpy -> upload again (ISOCHRONES_AMD_BROADCAST=host) - rounds 1-5
One JSON line per route and repetition.  python tools/broadcast_ab.py [reps=3]"""
import json, os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import isochrones_amd as ia


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    ic0 = ia.synthetic_track(bands=("G", "BP", "RP"))
    p = [1.0, 355.0, 0.0, 100.0, 0.1]
    want = ic0.interp_mag(p, ["G", "BP", "RP"])
    for rep in range(reps):
        for route in ("device", "host"):
            os.environ["ISOCHRONES_AMD_BROADCAST"] = route
            tm = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ic = ia.broadcast_interpolator(ic0, src=0, rebuild_on_src=True, timings=tm)
            t1 = time.perf_counter()
            ic.handle(0)                                  # tables + hot pack + corner pack on the device
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            got = ic.interp_mag(p, ["G", "BP", "RP"])
            same = all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(got, want))
            print(json.dumps(dict(route=route, rep=rep, broadcast_and_rebuild_s=t1 - t0, device_tables_s=t2 - t1, total_s=t2 - t0,
                                  broadcast_s=tm.get("broadcast_s"), rebuild_s=tm.get("rebuild_s"), bytes=tm.get("broadcast_bytes"),
                                  same_numbers=bool(same))), flush=True)
            ic.release()
            del ic
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
