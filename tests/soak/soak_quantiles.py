"""Randomised check of iso_chain_quantiles (wave-per-pair selection with its hand-over, all sizes) against
numpy.quantile, bit for bit.  Usage on the GPU box: python tests/soak/soak_quantiles.py [seconds] [seed]."""
import os, sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from isochrones_amd import _cabi, device as dev

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib, ctx = _cabi.lib(), dev.context(0)
t0, rounds, pairs, bad = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    W = int(rng.choice([1, 2, 7, 16, 24, 32, 32, 32, 48, 64, 100]))
    T = int(rng.integers(1, max(2, min(8192 // W, 260)) + 1))
    if rng.random() < 0.4:
        # the compile-time-shaped wave kernel: W divides 64 and T W = 64 F (+ a tail < 64) with F in {12, 25, 50, 100}
        W = int(rng.choice([2, 4, 8, 16, 32, 32, 64]))
        F = int(rng.choice([12, 25, 50, 100]))
        T = (64 * F + int(rng.integers(0, 64)) + W - 1) // W
        if (T * W) // 64 != F:
            T = 64 * F // W
    S, D = int(rng.integers(1, 400)), int(rng.integers(1, 9))
    kind = rng.choice(["normal", "lognormal", "ties", "few", "const", "mixed", "inf"])
    x = rng.standard_normal((T, S * W, D))
    if kind == "lognormal":
        x = np.exp(3 * x)
    elif kind == "ties":
        x = np.round(x, int(rng.integers(0, 3)))
    elif kind == "few":
        x = rng.integers(0, int(rng.integers(2, 6)), size=x.shape).astype(float)
    elif kind == "const":
        x[:] = rng.standard_normal()
    elif kind == "mixed":
        x[:, :, ::2] = np.round(x[:, :, ::2], 1)
        x[rng.random(x.shape) < 0.3] = 0.25
    elif kind == "inf":
        x[rng.random(x.shape) < 0.01] = np.inf
        x[rng.random(x.shape) < 0.01] = -np.inf
    nq = int(rng.integers(1, 9))
    qs = np.ascontiguousarray(rng.choice([0.0, 1.0, 0.5, 0.16, 0.84, 0.025, 0.975, 1 / 3, 0.999, rng.random()], size=nq))
    # both chain layouts: row-major [step][row][param] and parameter-major [step][param][row]
    layout = int(rng.integers(0, 2))
    chain = torch.as_tensor(x if layout == _cabi.CHAIN_ROW_MAJOR else np.ascontiguousarray(x.transpose(0, 2, 1)), device="cuda")
    out = torch.zeros(S, D, nq, dtype=torch.float64, device="cuda")
    rc = lib.iso_chain_quantiles_layout(ctx, dev.ptr(chain), layout, T, S, W, D, qs.ctypes.data_as(C.POINTER(C.c_double)), nq,
                                        dev.ptr(out), None)
    assert rc == 0
    got = out.cpu().numpy()
    flat = x.reshape(T, S, W, D).transpose(1, 3, 0, 2).reshape(S, D, T * W)
    with np.errstate(invalid="ignore"):
        want = np.moveaxis(np.quantile(flat, qs, axis=2), 0, 2)
    same = (got == want) | (np.isnan(got) & np.isnan(want))
    if not same.all():
        bad += 1
        print("MISMATCH", kind, (T, S, W, D), qs, np.argwhere(~same)[:3], got[~same][:3], want[~same][:3], flush=True)
    rounds += 1
    pairs += S * D
print("quantile soak: %d rounds, %d (ensemble, parameter) pairs, %d mismatching rounds, %.0f s" % (rounds, pairs, bad, time.time() - t0))
sys.exit(1 if bad else 0)
