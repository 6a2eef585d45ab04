"""Affine-invariant ensemble sampler (Goodman & Weare 2010 stretch move) whose walkers live in HBM.

The reference drives ``emcee.EnsembleSampler(nwalkers, npars, self.lnpost)`` one walker at a time
(isochrones/starmodel.py:886-972); emcee is not part of this build.  Here a half-ensemble is
proposed, evaluated by ONE batched ``lnpost`` launch and accepted/rejected on the device; nothing
crosses PCIe inside the loop.  The result attributes follow the emcee-v2 names the reference
reads (``chain`` [nwalkers, nsteps, ndim], ``lnprobability`` [nwalkers, nsteps], ``flatchain``,
``acceptance_fraction``, ``reset()``).

``lnpost_fn`` maps a CUDA float64 tensor [n, ndim] to a CUDA tensor [n] (e.g.
``BasicStarModel.lnpost``).  Also works with CPU tensors (used by the gloo/CPU tests with a toy
log-density; the product path always passes CUDA tensors).
"""
from __future__ import annotations

import math

import numpy as np


class EnsembleSampler:
    def __init__(self, nwalkers, ndim, lnpost_fn, a=2.0, seed=None, device=None):
        import torch
        if nwalkers % 2 or nwalkers < 2 * ndim:
            raise ValueError("need an even number of walkers, at least 2*ndim")
        self.nwalkers, self.ndim, self.lnpost_fn, self.a = int(nwalkers), int(ndim), lnpost_fn, float(a)
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed) if seed is not None else int(np.random.SeedSequence().entropy % (2 ** 63)))
        self.reset()

    def reset(self):
        self._chain = []
        self._lnprob = []
        self.naccepted = None
        self.iterations = 0

    # -- one full step = two half-ensemble updates ------------------------------------------
    def _half_step(self, pos, lnp, first, second):
        import torch
        ns = first.numel()
        idx = torch.randint(0, second.numel(), (ns,), generator=self.gen, device=self.device)
        u = torch.rand(ns, generator=self.gen, device=self.device, dtype=torch.float64)
        z = ((self.a - 1.0) * u + 1.0) ** 2 / self.a
        xk = pos[first]
        xj = pos[second[idx]]
        prop = xj + z[:, None] * (xk - xj)
        lnp_new = self.lnpost_fn(prop)
        lnq = (self.ndim - 1) * torch.log(z) + lnp_new - lnp[first]
        logu = torch.log(torch.rand(ns, generator=self.gen, device=self.device, dtype=torch.float64))
        accept = (logu < lnq) & torch.isfinite(lnp_new)
        pos[first] = torch.where(accept[:, None], prop, xk)
        lnp[first] = torch.where(accept, lnp_new, lnp[first])
        return accept

    def run_mcmc(self, p0, nsteps, lnprob0=None, store=True, thin=1):
        import torch
        pos = torch.as_tensor(p0, dtype=torch.float64, device=self.device).clone()
        if pos.shape != (self.nwalkers, self.ndim):
            raise ValueError("p0 must be [nwalkers, ndim]")
        lnp = self.lnpost_fn(pos).clone() if lnprob0 is None else torch.as_tensor(
            lnprob0, dtype=torch.float64, device=self.device).clone()
        if not bool(torch.isfinite(lnp).all()):
            raise ValueError("initial positions must have finite lnpost")
        half = self.nwalkers // 2
        allw = torch.arange(self.nwalkers, device=self.device)
        s0, s1 = allw[:half], allw[half:]
        if self.naccepted is None:
            self.naccepted = torch.zeros(self.nwalkers, dtype=torch.float64, device=self.device)
        for it in range(int(nsteps)):
            acc0 = self._half_step(pos, lnp, s0, s1)
            acc1 = self._half_step(pos, lnp, s1, s0)
            self.naccepted[:half] += acc0
            self.naccepted[half:] += acc1
            self.iterations += 1
            if store and (it % thin == 0):
                self._chain.append(pos.clone())
                self._lnprob.append(lnp.clone())
        return pos, lnp

    # -- emcee-v2 style views ---------------------------------------------------------------
    @property
    def chain(self):
        import torch
        if not self._chain:
            return torch.empty(self.nwalkers, 0, self.ndim, dtype=torch.float64, device=self.device)
        return torch.stack(self._chain, dim=1)

    @property
    def lnprobability(self):
        import torch
        if not self._lnprob:
            return torch.empty(self.nwalkers, 0, dtype=torch.float64, device=self.device)
        return torch.stack(self._lnprob, dim=1)

    @property
    def flatchain(self):
        return self.chain.reshape(-1, self.ndim)

    @property
    def flatlnprobability(self):
        return self.lnprobability.reshape(-1)

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(self.iterations, 1)


def summarize_chain(flatchain, flatlnprob):
    """Fixed-size per-star result row: [median, 16th, 84th] per parameter, then the MAP lnpost.
    Works on torch tensors or numpy arrays; returns a 1-D numpy array of length 3*ndim + 1."""
    x = flatchain.detach().cpu().numpy() if hasattr(flatchain, "detach") else np.asarray(flatchain)
    lp = flatlnprob.detach().cpu().numpy() if hasattr(flatlnprob, "detach") else np.asarray(flatlnprob)
    q = np.percentile(x, [50, 16, 84], axis=0)          # [3, ndim]
    return np.concatenate([q.T.ravel(), [lp.max() if lp.size else math.nan]])


class FusedEnsembleSampler:
    """The same stretch-move ensemble, but with proposal + fused lnpost + accept in ONE HIP kernel
    per half-ensemble (``iso_sampler_*``; Philox counter RNG in-kernel).  ``target`` is a
    :class:`BasicStarModel` (one ensemble; 13-32 bands run the band-tiled persistent kernel), a
    :class:`TreeStarModel` or :class:`IsoTrackModel` (one persistent launch per ``run_mcmc``, one workgroup per
    ensemble: ``iso_sampler_create_tree`` / ``iso_sampler_create_isotrack``) or a :class:`CatalogPosterior` (one
    ensemble per star, all advanced in lock-step; row = star * nwalkers + walker).  A 256-walker half-step is a
    single ~10 us launch instead of ~25 framework launches; whenever an ensemble fits a workgroup's
    LDS the library runs ALL iterations of a ``run_mcmc`` call in one persistent launch
    (workgroup-resident ensembles, positions in LDS; catalogs larger than the chip run in rounds),
    which produces bit-identical chains and is the faster form at every catalog size
    (``ISOCHRONES_AMD_SAMPLER=auto|persistent|stepwise`` selects the form)."""

    def __init__(self, target, nwalkers, a=2.0, seed=0, device=None, n_ensembles=1):
        import ctypes as C
        import torch
        from . import _cabi, device as dev
        from .catalog import CatalogPosterior
        self.target = target
        self.nwalkers = int(nwalkers)
        self.ndim = target.n_params
        self.is_catalog = isinstance(target, CatalogPosterior)
        host = getattr(target, "_host_terms", None)
        if host is not None and host():
            raise ValueError("the resident sampler evaluates priors in the kernel; this model has priors evaluated on the host "
                             "(use fit_mcmc, which takes the framework-op sampler for it)")
        # a model with n_ensembles > 1: that many independent ensembles of the one posterior in the same launches (shapes
        # then carry a leading ensemble axis, as a catalog's carry the star axis)
        self.multi_ensemble = (not self.is_catalog) and int(n_ensembles) > 1
        if self.is_catalog and int(n_ensembles) != 1:
            raise ValueError("n_ensembles is for a single model; a catalog has one ensemble per star")
        self.n_ensembles = target.n_models if self.is_catalog else int(n_ensembles)
        self.device_index = (target.device if self.is_catalog else
                             (dev.current_device() if device is None else device))
        self.device = torch.device("cuda", self.device_index)
        h = C.c_void_p()
        lib = _cabi.lib()
        from .starmodel import IsoTrackModel, TreeStarModel
        if self.is_catalog:
            _cabi.check(lib.iso_sampler_create_catalog(target._h, self.nwalkers, float(a), int(seed), C.byref(h)))
        elif isinstance(target, TreeStarModel):
            _cabi.check(lib.iso_sampler_create_tree(target.handle(self.device_index), self.n_ensembles, self.nwalkers,
                                                    float(a), int(seed), C.byref(h)))
        elif isinstance(target, IsoTrackModel):
            lo, hi, lnorm = target.age_prior_constants()
            _cabi.check(lib.iso_sampler_create_isotrack(target._iso_model.handle(self.device_index),
                                                        target._track_model.handle(self.device_index), lo, hi, lnorm,
                                                        self.n_ensembles, self.nwalkers, float(a), int(seed), C.byref(h)))
        elif self.multi_ensemble:
            _cabi.check(lib.iso_sampler_create_model_ensembles(target.handle(self.device_index), self.n_ensembles,
                                                               self.nwalkers, float(a), int(seed), C.byref(h)))
        else:
            _cabi.check(lib.iso_sampler_create_model(target.handle(self.device_index), self.nwalkers, float(a),
                                                     int(seed), C.byref(h)))
        self._h = h
        # the chain is stored parameter-major, [nsteps][ndim][rows]: coalesced stores in the sampler kernels, and the
        # per-(star, parameter) summaries read contiguous runs (every line of the chain is fetched once)
        _cabi.check(lib.iso_sampler_set_chain_layout(h, _cabi.CHAIN_PARAM_MAJOR))
        self.reset()

    def close(self):
        if getattr(self, "_h", None) is not None:
            from . import _cabi
            _cabi.lib().iso_sampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        import torch
        self._chain, self._lnprob = None, None
        self.accepted = torch.zeros(self.n_ensembles * self.nwalkers, dtype=torch.int32, device=self.device)
        self.iterations = 0

    def lnpost_rows(self, pos):
        """lnpost of [n_ens*W, ndim] rows through the regular batch entry points."""
        import torch
        if self.is_catalog:
            sid = torch.arange(self.n_ensembles, device=self.device, dtype=torch.int32).repeat_interleave(self.nwalkers)
            return self.target.lnpost(pos, sid)
        return self.target.lnpost(pos)          # a model's ensembles all evaluate the one posterior

    def run_mcmc(self, p0, nsteps, lnprob0=None, store=True, check=True, inplace=False):
        """p0: [W, ndim] (model) or [S, W, ndim] (catalog).  Returns (pos, lnprob) in that shape.
        ``check=False`` skips the test that every start point has a finite lnpost (a device-to-host round trip; a
        catalog fit has made sure on the device), ``inplace=True`` advances the caller's CUDA tensors themselves."""
        import torch
        from . import _cabi, device as dev
        rows = self.n_ensembles * self.nwalkers
        pos = torch.as_tensor(p0, dtype=torch.float64, device=self.device).reshape(rows, self.ndim).contiguous()
        lnp = (self.lnpost_rows(pos) if lnprob0 is None else
               torch.as_tensor(lnprob0, dtype=torch.float64, device=self.device).reshape(rows)).contiguous()
        if not inplace:
            pos, lnp = pos.clone(), (lnp.clone() if lnprob0 is not None else lnp)
        if check and not bool(torch.isfinite(lnp).all()):
            raise ValueError("initial positions must have finite lnpost")
        chain = torch.empty(nsteps, self.ndim, rows, dtype=torch.float64, device=self.device) if store else None
        clnp = torch.empty(nsteps, rows, dtype=torch.float64, device=self.device) if store else None
        _cabi.check(_cabi.lib().iso_sampler_run(self._h, dev.ptr(pos), dev.ptr(lnp), int(nsteps), dev.ptr(chain),
                                                dev.ptr(clnp), dev.ptr(self.accepted),
                                                dev.stream_ptr(self.device_index)))
        self.iterations += int(nsteps)
        if store:
            self._chain = chain if self._chain is None else torch.cat([self._chain, chain], dim=0)
            self._lnprob = clnp if self._lnprob is None else torch.cat([self._lnprob, clnp], dim=0)
        shape = (self.n_ensembles, self.nwalkers) if self._stacked else (self.nwalkers,)
        return pos.view(*shape, self.ndim), lnp.view(*shape)

    @property
    def chain_steps(self):
        """The stored chain as [nsteps, n_ens * W, ndim] (row = star * W + walker): a strided view of the
        parameter-major storage ``_chain`` [nsteps, ndim, n_ens * W]."""
        import torch
        if self._chain is None:
            return torch.empty(0, self.n_ensembles * self.nwalkers, self.ndim, dtype=torch.float64, device=self.device)
        return self._chain.permute(0, 2, 1)

    # emcee-v2 style views: [W, nsteps, ndim] for a model, [S, W, nsteps, ndim] for a catalog
    @property
    def chain(self):
        import torch
        if self._chain is None:
            return torch.empty(self.nwalkers, 0, self.ndim, dtype=torch.float64, device=self.device)
        c = self._chain.view(-1, self.ndim, self.n_ensembles, self.nwalkers).permute(2, 3, 0, 1)
        return c if self._stacked else c[0]

    @property
    def lnprobability(self):
        import torch
        if self._lnprob is None:
            return torch.empty(self.nwalkers, 0, dtype=torch.float64, device=self.device)
        c = self._lnprob.view(-1, self.n_ensembles, self.nwalkers).permute(1, 2, 0)
        return c if self._stacked else c[0]

    @property
    def flatchain(self):
        return self.chain.reshape(-1, self.ndim) if not self._stacked else self.chain.reshape(
            self.n_ensembles, -1, self.ndim)

    @property
    def flatlnprobability(self):
        return self.lnprobability.reshape(-1) if not self._stacked else self.lnprobability.reshape(
            self.n_ensembles, -1)

    def quantiles(self, q=(0.5, 0.16, 0.84)):
        """Per-ensemble quantiles of the stored chain, [S, ndim, len(q)] (model: [ndim, len(q)]) CUDA tensor,
        linear interpolation between order statistics as ``numpy.percentile``, bit for bit.  Selection on the device
        (``iso_chain_quantiles``: one wavefront per (ensemble, parameter) pair up to 6 656 values, a workgroup up to
        8 192, refinement passes streamed from the chain beyond - the reference's default 300 walkers x 100
        iterations); only more than 8 quantile levels at once fall back to a framework sort."""
        import ctypes as C
        import torch
        from . import _cabi, device as dev
        if self._chain is None:
            raise ValueError("no stored chain")
        nsteps = self._chain.shape[0]
        q = np.ascontiguousarray(q, dtype=np.float64)
        if q.size <= 8:
            out = torch.empty(self.n_ensembles, self.ndim, q.size, dtype=torch.float64, device=self.device)
            chain = self._chain.contiguous()
            _cabi.check(_cabi.lib().iso_chain_quantiles_layout(dev.context(self.device_index), dev.ptr(chain),
                                                               _cabi.CHAIN_PARAM_MAJOR, nsteps, self.n_ensembles,
                                                               self.nwalkers, self.ndim,
                                                               q.ctypes.data_as(C.POINTER(C.c_double)), q.size,
                                                               dev.ptr(out), dev.stream_ptr(self.device_index)))
        else:
            flat = self._chain.view(nsteps, self.ndim, self.n_ensembles, self.nwalkers).permute(2, 1, 0, 3)
            srt = torch.sort(flat.reshape(self.n_ensembles, self.ndim, -1), dim=2).values
            m = srt.shape[2]
            pick = torch.as_tensor(q, device=self.device) * (m - 1)
            i0 = pick.floor().long()
            i1 = torch.clamp(i0 + 1, max=m - 1)
            frac = pick - i0.to(torch.float64)
            out = srt[:, :, i0] * (1 - frac) + srt[:, :, i1] * frac
        return out if self._stacked else out[0]

    @property
    def acceptance_fraction(self):
        acc = self.accepted.to(dtype=__import__("torch").float64) / max(self.iterations, 1)
        return acc.view(self.n_ensembles, self.nwalkers) if self._stacked else acc

    @property
    def _stacked(self):
        """results carry a leading ensemble axis: catalogs (one ensemble per star) and multi-ensemble model samplers"""
        return self.is_catalog or self.multi_ensemble

    def gelman_rubin(self):
        """Potential scale reduction factor R-hat per parameter across the independent ensembles of a multi-ensemble
        model sampler (Gelman & Rubin 1992: between- vs within-chain variance of the ensemble means; each ensemble's
        stored samples are one "chain").  Values near 1 mean the ensembles agree."""
        import torch
        if not self.multi_ensemble or self._chain is None:
            raise ValueError("needs a stored chain of a sampler with n_ensembles > 1")
        x = self.flatchain                                             # [E, n, D]
        n = x.shape[1]
        means = x.mean(dim=1)
        Wv = x.var(dim=1, unbiased=True).mean(dim=0)
        B = means.var(dim=0, unbiased=True) * n
        return torch.sqrt(((n - 1) / n * Wv + B / n) / Wv)
