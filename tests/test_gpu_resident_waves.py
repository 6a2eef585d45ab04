"""The two kinds of resident waves - a model's mailbox wave for the per-point lnpost callback (iso_fast_mailbox.hip) and the
context's service wave of the scalar accessors (kernels/k_service.h) - next to the rest of the process: a batch evaluation
and a device-wide synchronise issued while they are resident complete within the waves' idle bound, the waves come back
afterwards with the same numbers, callers from several threads are served one at a time, and a table that is rebuilt
(add_column) is never answered from the old one.

Reference call forms: StarModel.lnpost(p) one point at a time (starmodel.py:797,952,966); DFInterpolator.__call__ /
ModelGridInterpolator.interp_value / interp_mag / get_eep with plain numbers (interp.py:631-660, models.py:390-445,501-542)."""
import threading
import time

import numpy as np
import pytest

import isochrones_amd as ia
from tests import _fixtures as fx

pytestmark = pytest.mark.gpu


def _model():
    import bench
    return bench.build_model()


def test_batches_and_device_synchronize_while_the_waves_are_resident():
    import torch
    import bench
    ic, mod = _model()
    p = [1.0, 355.0, 0.0, 100.0, 0.1]
    want_post = mod.lnpost(p)                                    # starts the model's mailbox wave
    want_val = ic.interp_value(p[:3], ["Teff", "logg", "age"])   # starts the context's service wave
    want_mag = ic.interp_mag(p, ["V"])
    big = torch.as_tensor(bench.make_samples(np.random.default_rng(3), 1_000_000, "prior_valid"), device="cuda")
    ref = mod.lnpost(big).cpu().numpy()
    worst_sync, worst_batch = 0.0, 0.0
    for it in range(50):
        assert mod.lnpost(p) == want_post                        # both waves resident (again) ...
        assert np.array_equal(ic.interp_value(p[:3], ["Teff", "logg", "age"]), want_val)
        t0 = time.perf_counter()
        torch.cuda.synchronize()                                 # ... a device-wide synchronise waits for them to leave: idle 1 ms each
        worst_sync = max(worst_sync, time.perf_counter() - t0)
        assert mod.lnpost(p) == want_post
        t0 = time.perf_counter()
        out = mod.lnpost(big)                                    # a 10^6-row batch on the null stream next to the resident waves
        torch.cuda.synchronize()
        worst_batch = max(worst_batch, time.perf_counter() - t0)
        if it % 10 == 0:
            assert np.array_equal(out.cpu().numpy(), ref, equal_nan=True)
    # (idle time-out 1 ms per wave + the batch's ~0.1 ms; generous bounds for a busy box: the failure mode is the 30 s lifetime)
    assert worst_sync < 0.05 and worst_batch < 0.05, (worst_sync, worst_batch)
    got = ic.interp_mag(p, ["V"])
    assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(got, want_mag))
    ic.release()


def test_threads_share_the_waves_one_caller_at_a_time():
    ic, mod = _model()
    rng = np.random.default_rng(5)
    rows = np.array([1.0, 355.0, 0.0, 100.0, 0.1]) * (1 + 1e-3 * rng.standard_normal((400, 5)))
    want_post = np.array([mod.lnpost(list(r)) for r in rows])
    want_val = np.array([ic.interp_value(list(r[:3]), ["Teff", "logg"]) for r in rows])
    errors = []

    def worker(k):
        try:
            for i in range(k, 400, 4):
                r = list(rows[i])
                if mod.lnpost(r) != want_post[i]:
                    errors.append(("lnpost", i))
                if not np.array_equal(ic.interp_value(r[:3], ["Teff", "logg"]), want_val[i]):
                    errors.append(("interp_value", i))
        except Exception as e:       # noqa: BLE001
            errors.append(repr(e))
    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]
    ic.release()


def test_a_rebuilt_table_is_never_answered_from_the_old_one():
    rng = np.random.default_rng(9)
    axes = [np.sort(rng.uniform(0, 10, n)) for n in (7, 9, 11)]
    grid = rng.standard_normal((7, 9, 11, 3))
    dfi = ia.DFInterpolator.from_arrays(grid, axes, ["a", "b", "c"])
    p = [float(a[2] + 0.3 * (a[3] - a[2])) for a in axes]
    before = dfi(p, ["a", "c"])
    extra = rng.standard_normal((7, 9, 11))
    dfi.add_column(extra, "d")                       # frees the device table; the next call uploads the new one
    after = dfi(p, ["a", "c", "d"])
    assert np.array_equal(after[:2], before)
    from oracle import oracle as orc
    want = orc.OracleTable(dfi.grid, dfi.index_columns).interp([np.array([x]) for x in p], [0, 2, 3])[0]
    fx.assert_close(after, want, 1e-12, atol=1e-13, what="after add_column")
    # many tables alive at once, calls alternating between them: the wave restages its axes for every change of target
    tabs = []
    for k in range(6):
        ax = [np.sort(rng.uniform(0, 10, n)) for n in (5 + k, 6, 4 + k)]
        g = rng.standard_normal((5 + k, 6, 4 + k, 2))
        tabs.append((ia.DFInterpolator.from_arrays(g, ax, ["u", "v"]), ax, g))
    for rep in range(5):
        for t, ax, g in tabs:
            q = [float(rng.uniform(a[0], a[-1])) for a in ax]
            want = orc.OracleTable(g, ax).interp([np.array([x]) for x in q], [0, 1])[0]
            fx.assert_close(t(q), want, 1e-12, atol=1e-13, what="alternating targets")
    for t, _, _ in tabs:
        t.release()
    dfi.release()
