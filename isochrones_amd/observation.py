"""Observation trees ("next" row f4): arbitrary resolved / blended multi-star photometry, flattened
into a small per-model *program* that the device evaluates.

The reference builds a tree of Python objects (isochrones/observation.py:128-1306): observations
are ordered from coarsest to finest angular resolution; every source of an observation becomes a
node hanging below the closest already-placed node of a *different* observation whose resolution
still contains it; model stars ("leaves", label ``{system}_{tag}``) hang below the finest-level
nodes; the likelihood of a node compares its magnitude with the flux sum of all model stars below
it (or, for relative photometry, the difference to the observation's brightest source).

Here the same placement rules are applied once on the host and the result is stored as flat
arrays — for every observation node a (band, leaf-mask, reference-mask, value, sigma) record —
which is all the likelihood needs (reference: ``ObsNode.lnlike`` :464-491,
``ObservationTree.lnlike`` :1181-1234).  Names follow the reference (``Source``, ``Observation``,
``ObservationTree.from_df / add_observation / define_models / add_spectroscopy / add_limit /
add_parallax / add_AV``, ``param_description``, ``Nstars``, ``systems``, ``leaf_labels``,
``p2pardict``).
"""
from __future__ import annotations

import math

import numpy as np


def _distance(a, b):
    """Separation of two (separation, PA[deg]) positions (reference: isochrones/utils.py:78-93)."""
    r0, pa0 = a
    r1, pa1 = b
    ra0, dec0 = r0 * math.sin(pa0 * math.pi / 180), r0 * math.cos(pa0 * math.pi / 180)
    ra1, dec1 = r1 * math.sin(pa1 * math.pi / 180), r1 * math.cos(pa1 * math.pi / 180)
    return math.sqrt((ra1 - ra0) ** 2 + (dec1 - dec0) ** 2)


class Source:
    def __init__(self, mag, e_mag, separation=0.0, pa=0.0, relative=False, is_reference=False):
        self.mag = float(mag)
        self.e_mag = float(e_mag)
        self.separation = float(separation)
        self.pa = float(pa)
        self.relative = bool(relative)
        self.is_reference = bool(is_reference)

    def __repr__(self):
        return "({}, {}) @({}, {})".format(self.mag, self.e_mag, self.separation, self.pa)


class Observation:
    """One image: instrument name, band, approximate angular resolution, detected sources
    (kept sorted by separation)."""

    def __init__(self, name, band, resolution, sources=None, relative=False):
        self.name, self.band, self.resolution = name, band, resolution
        self.sources = []
        for s in sources or []:
            self.add_source(s)
        self.relative = relative
        self._set_reference()

    def add_source(self, source):
        if not isinstance(source, Source):
            raise TypeError("Can only add Source object.")
        k = 0
        for s in self.sources:
            if source.separation < s.separation:
                break
            k += 1
        self.sources.insert(k, source)

    @property
    def brightest(self):
        best = None
        for s in self.sources:
            if best is None or s.mag < best.mag:
                best = s
        return best

    def _set_reference(self):
        if self.sources:
            self.brightest.is_reference = True

    def __repr__(self):
        return "{}-{}".format(self.name, self.band)


class _Node:
    """Tree node: kind 'root' | 'obs' | 'dummy' | 'model'."""

    def __init__(self, kind, observation=None, source=None, reference=None, index=None, tag=None):
        self.kind = kind
        self.observation, self.source, self.reference = observation, source, reference
        self.index, self.tag = index, tag
        self.parent = None
        self.children = []

    def add_child(self, node):
        node.parent = self
        self.children.append(node)

    def walk(self):
        """children first, then the node itself (the reference's iteration order)."""
        for c in self.children:
            yield from c.walk()
        yield self

    def leaves(self):
        if not self.children and self.kind != "root":
            return [self]
        out = []
        for c in self.children:
            out += c.leaves()
        return out

    @property
    def label(self):
        if self.kind == "model":
            return "{}_{}".format(self.index, self.tag)
        if self.kind == "obs":
            return "{} {}".format(self.observation, self.source)
        return self.kind

    @property
    def reference_label(self):
        """The label the reference gives an observation node (observation.py:341-352):
        ``<instrument> <band>=(<mag>, <unc>) @(<separation>, <pa> [<resolution>])``."""
        if self.kind != "obs":
            return self.label
        o, s = self.observation, self.source
        return "{} {}=({:.2f}, {:.2f}) @({:.2f}, {:.0f} [{:.2f}])".format(o.name, o.band, s.mag, s.e_mag, s.separation,
                                                                          s.pa, o.resolution)

    def select_leaves(self, pattern):
        """All finest-level nodes below every node whose label matches ``pattern`` (a regular expression; this
        build's label or the reference's spelling of it), reference observation.py:234-249."""
        import re

        def bottom(node):                         # finest observation level: model stars do not count
            kids = [c for c in node.children if c.kind != "model"]
            if not kids:
                return [node]
            out = []
            for c in kids:
                out += bottom(c)
            return out

        hit = self.kind != "root" and (re.search(pattern, self.label) or re.search(pattern, self.reference_label))
        kids = [c for c in self.children if c.kind != "model"]
        if hit or not kids:
            return bottom(self) if hit else []
        out = []
        for c in kids:
            out += c.select_leaves(pattern)
        return out


class ObservationTree:
    spec_props = ["Teff", "logg", "feh", "density"]

    def __init__(self, observations=None, name=None):
        self.name = "root" if name is None else name
        self._observations = []
        self.spectroscopy, self.limits, self.parallax, self.AV = {}, {}, {}, {}
        self._model_spec = None
        self._build()
        for o in observations or []:
            self.add_observation(o)

    # -- construction -----------------------------------------------------------------------
    @classmethod
    def from_df(cls, df, **kwargs):
        """Columns: name, band, resolution, mag, e_mag, separation, pa, relative."""
        tree = cls(**kwargs)
        for (n, b), g in df.groupby(["name", "band"]):
            sources = [Source(s["mag"], s["e_mag"], s["separation"], s["pa"], s["relative"]) for _, s in g.iterrows()]
            tree.add_observation(Observation(n, b, g.resolution.mean(), sources=sources, relative=g.relative.any()))
        return tree

    def to_df(self):
        """The photometry as a DataFrame that :meth:`from_df` reads back (reference: observation.py:796-835)."""
        import pandas as pd
        rows = [dict(name=o.name, band=o.band, resolution=o.resolution, mag=s.mag, e_mag=s.e_mag,
                     separation=s.separation, pa=s.pa, relative=s.relative)
                for o in self._observations for s in o.sources]
        return pd.DataFrame(rows, columns=["name", "band", "resolution", "mag", "e_mag", "separation", "pa", "relative"])

    def print_ascii(self, fout=None, p=None):
        """The tree as indented text, one node per line (the reference draws the same hierarchy with asciitree,
        observation.py:167-172, 1175-1179); with a parameter vector ``p`` the model leaves show their parameters."""
        pardict = self.p2pardict(p) if p is not None else None
        lines = []

        def emit(node, prefix, last):
            text = self.name if node.kind == "root" and getattr(self, "name", None) else node.label
            if pardict is not None and node.kind == "model" and node.label in pardict:
                text += " = " + ", ".join("%.4g" % v for v in pardict[node.label])
            lines.append(prefix + ("" if node.kind == "root" else ("+-- " if last else "|-- ")) + str(text))
            kids = node.children
            for k, c in enumerate(kids):
                emit(c, prefix + ("" if node.kind == "root" else ("    " if last else "|   ")), k == len(kids) - 1)

        emit(self.root, "", True)
        out = "\n".join(lines) + "\n"
        if fout is None:
            print(out, end="")
        else:
            fout.write(out)

    def add_observation(self, obs):
        k = 0
        for o in self._observations:          # keep coarsest resolution first
            if obs.resolution > o.resolution:
                break
            k += 1
        self._observations.insert(k, obs)
        self._build()

    @property
    def observations(self):
        return self._observations

    def _closest(self, node):
        """Parent of a new node: the nearest node of another observation whose resolution still
        contains it (ties: first in children-first order); the root otherwise."""
        cands = [(math.inf, self.root)]
        for n in self.root.walk():
            if n is node or n.kind != "obs":
                continue
            if n.observation.name == node.observation.name and n.observation.band == node.observation.band:
                continue
            d = _distance((n.source.separation, n.source.pa), (node.source.separation, node.source.pa))
            cands.append((d, n))
        order = np.argsort([c[0] for c in cands], kind="stable")
        for k in order:
            d, n = cands[k]
            if n.kind == "obs" and (d < n.observation.resolution or n.observation.resolution == -1):
                return n
        return self.root

    def _build(self):
        self.root = _Node("root")
        for i, o in enumerate(self._observations):
            ref_node = _Node("obs", o, o.brightest)
            for s in o.sources:
                if s.relative and not s.is_reference:
                    node = _Node("obs", o, s, reference=ref_node)
                elif s.relative and s.is_reference:
                    node = ref_node
                else:
                    node = _Node("obs", o, s)
                parent = self.root if i == 0 else self._closest(node)
                parent.add_child(node)
        if not any(n.kind == "obs" for n in self.root.walk()):
            self.root.add_child(_Node("dummy"))
        if self._model_spec is not None:
            self.define_models(*self._model_spec)

    def define_models(self, ic=None, leaves=None, N=1, index=0):
        """Hang N model stars of physical system `index` below every finest-level node
        (scalars or one entry per node; an entry of `index` may itself be a list of length N)."""
        self._model_spec = None
        for n in list(self.root.walk()):
            if n.kind == "model":
                n.parent.children.remove(n)
        if leaves is None:
            hosts = self.root.leaves()
        elif isinstance(leaves, str):             # a pattern: the finest-level nodes below every matching node
            hosts = self.root.select_leaves(leaves)
            if not hosts:
                raise ValueError("no observation node matches {!r}".format(leaves))
        else:
            hosts = list(leaves)
        Ns = [int(N)] * len(hosts) if np.isscalar(N) else [int(x) for x in N]
        idx = [index] * len(hosts) if np.isscalar(index) else list(index)
        if len(Ns) != len(hosts) or len(idx) != len(hosts):
            raise ValueError("N / index need one entry per finest-level node (%d)" % len(hosts))
        for host, n, i in zip(hosts, Ns, idx):
            ilist = list(i) if isinstance(i, (list, tuple, np.ndarray)) else [int(i)] * n
            if len(ilist) != n:
                raise ValueError("If a list, index must be of length N.")
            for sysid in ilist:
                tag = len([l for l in self.root.leaves() if l.kind == "model" and l.index == int(sysid)])
                host.add_child(_Node("model", index=int(sysid), tag=tag))
        self._fix_labels()
        self._model_spec = (ic, leaves if isinstance(leaves, str) else None, N, index)

    def _fix_labels(self):
        """Within each system the star below the brightest source carries tag 0."""
        for s in self.systems:
            best, bmag = None, math.inf
            for n in self.get_system(s):
                if n.parent.kind != "obs":
                    continue
                if n.parent.source.mag < bmag:
                    bmag, best = n.parent.source.mag, n
            if best is not None and best.tag != 0:
                other = next(n for n in self.get_system(s) if n.tag == 0)
                other.tag, best.tag = best.tag, 0

    # -- queries ----------------------------------------------------------------------------
    # the reference's spellings (observation.py:234-300, 1088-1098)
    def select_leaves(self, pattern):
        return self.root.select_leaves(pattern)

    def select_observations(self, name):
        """Observation nodes of one instrument-band, named like the reference's ``obsname`` (``'2MASS-K'``)."""
        return [n for n in self.obs_nodes() if "{}-{}".format(n.observation.name, n.observation.band) == name]

    def get_obs_nodes(self):
        return self.obs_nodes()

    def get_model_nodes(self):
        return self.model_nodes()

    def get_leaf(self, label):
        for n in self.root.leaves():
            if n.label == label:
                return n
        return None

    def get_obs_leaves(self):
        """The finest-level observation nodes (the ones model stars hang below)."""
        seen, out = set(), []
        for n in self.root.leaves():
            host = n.parent if n.kind == "model" else n
            if host.kind == "obs" and id(host) not in seen:
                seen.add(id(host))
                out.append(host)
        return out

    obs_leaf_nodes = property(get_obs_leaves)

    @property
    def N_model_nodes(self):
        return len(self.model_nodes())

    def clear_models(self):
        self._model_spec = None
        for n in list(self.root.walk()):
            if n.kind == "model":
                n.parent.children.remove(n)

    def trim(self):
        """No-op, as in the reference (observation.py:1100-1107 returns before doing anything)."""
        return None

    def model_nodes(self):
        return [l for l in self.root.leaves() if l.kind == "model"]

    def get_system(self, ind):
        return [l for l in self.model_nodes() if l.index == ind]

    @property
    def leaf_labels(self):
        return [l.label for l in self.root.leaves()]

    @property
    def Nstars(self):
        N = {}
        for n in self.model_nodes():
            N[n.index] = N.get(n.index, 0) + 1
        return N

    @property
    def systems(self):
        return sorted(self.Nstars.keys())

    @property
    def param_description(self):
        pars = []
        N = self.Nstars
        for s in self.systems:
            pars += ["eep_{}_{}".format(s, j) for j in range(N[s])]
            pars += ["{}_{}".format(p, s) for p in ("age", "feh", "distance", "AV")]
        return pars

    def p2pardict(self, p):
        d, i, N = {}, 0, self.Nstars
        for s in self.systems:
            age, feh, dist, AV = p[i + N[s]: i + N[s] + 4]
            for j in range(N[s]):
                d["{}_{}".format(s, j)] = [p[i + j], age, feh, dist, AV]
            i += N[s] + 4
        return d

    def pardict2p(self, pardict):
        """The inverse of :meth:`p2pardict` (reference: observation.py:1132-1143)."""
        pars, N = [], self.Nstars
        for s in self.systems:
            pars += [pardict["{}_{}".format(s, j)][0] for j in range(N[s])]
            pars += list(pardict["{}_0".format(s)][1:])
        return pars

    def obs_nodes(self):
        return [n for n in self.root.walk() if n.kind == "obs"]

    @property
    def bands(self):
        return list({n.observation.band for n in self.obs_nodes()})

    # -- extra measurements -----------------------------------------------------------------
    def _pair_for(self, what, label, name, value, second):
        """Shared validation of the per-star constraints: a known model star, one of the spectroscopic
        properties, and a 2-sequence."""
        if label not in self.leaf_labels:
            raise ValueError("there is no model star {!r}; the tree's stars are {} (call define_models first)".format(
                label, self.leaf_labels))
        if name not in self.spec_props:
            raise ValueError("{} takes {}, not {!r}".format(what, " / ".join(self.spec_props), name))
        try:
            first, other = value
        except (TypeError, ValueError):
            raise ValueError("{}: {} needs a (value, {}) pair, got {!r}".format(what, name, second, value)) from None
        return first, other

    def add_spectroscopy(self, label="0_0", **props):
        """Teff / logg / feh measurements ``(value, uncertainty)`` of one model star (reference: observation.py:1029-1053)."""
        checked = {k: tuple(self._pair_for("add_spectroscopy", label, k, v, "uncertainty")) for k, v in props.items()}
        self.spectroscopy.setdefault(label, {}).update(checked)

    def add_limit(self, label="0_0", **props):
        """Hard ``(min, max)`` limits on Teff / logg / feh of one model star; ``None`` leaves a side open
        (reference: observation.py:1055-1078)."""
        for k, v in props.items():
            lo, hi = self._pair_for("add_limit", label, k, v, "maximum")
            self.limits.setdefault(label, {})[k] = (-np.inf if lo is None else lo, np.inf if hi is None else hi)

    def _system_pair(self, what, value, system):
        if system not in self.systems:
            raise ValueError("{}: system {!r} is not one of {}".format(what, system, self.systems))
        try:
            val, unc = value
        except (TypeError, ValueError):
            raise ValueError("{} needs a (value, uncertainty) pair, got {!r}".format(what, value)) from None
        return (val, unc)

    def add_parallax(self, plax, system=0):
        self.parallax[system] = self._system_pair("add_parallax", plax, system)

    def add_AV(self, AV, system=0):
        self.AV[system] = self._system_pair("add_AV", AV, system)

    # -- the flat program -------------------------------------------------------------------
    def program(self, bands=None):
        """Flatten to plain data.  Leaves are numbered in parameter order (system by system, tag
        order); every observation node becomes a term (band index, leaf bit-mask, reference
        bit-mask, relative flag, magnitude, sigma, reference magnitude), in the reference's
        children-first summation order."""
        systems = self.systems
        N = self.Nstars
        leaf_id, leaf_system, leaf_slot = {}, [], []
        for si, s in enumerate(systems):
            for j in range(N[s]):
                leaf_id["{}_{}".format(s, j)] = len(leaf_system)
                leaf_system.append(si)
                leaf_slot.append(j)
        bands = list(bands) if bands is not None else self.bands

        def mask(node):
            m = 0
            for l in node.leaves():
                if l.kind == "model":
                    m |= 1 << leaf_id[l.label]
            return m

        terms = []
        for n in self.obs_nodes():
            mag, dmag = n.source.mag, n.source.e_mag
            if np.isnan(dmag):
                continue                                   # contributes 0 (observation.py:473-474)
            if n.source.relative:
                if n.reference is None:
                    continue                               # the reference source itself: 0
                terms.append(dict(band=bands.index(n.observation.band), mask=mask(n), ref_mask=mask(n.reference),
                                  relative=1, mag=mag, unc=dmag, ref_mag=n.reference.source.mag))
            else:
                terms.append(dict(band=bands.index(n.observation.band), mask=mask(n), ref_mask=0, relative=0,
                                  mag=mag, unc=dmag, ref_mag=0.0))
        prop_id = {"Teff": 0, "logg": 1, "feh": 2}
        spec, limits = [], []
        for label, props in self.spectroscopy.items():
            for prop, (val, err) in props.items():
                if prop not in prop_id:
                    raise NotImplementedError("spectroscopic property %r is not evaluable (the reference's "
                                              "StarModel.lnlike only provides Teff, logg, feh)" % prop)
                spec.append(dict(leaf=leaf_id[label], prop=prop_id[prop], val=float(val), unc=float(err)))
        for label, props in self.limits.items():
            for prop, (lo, hi) in props.items():
                if prop not in prop_id:
                    raise NotImplementedError("limit on %r is not evaluable" % prop)
                limits.append(dict(leaf=leaf_id[label], prop=prop_id[prop], lo=float(lo), hi=float(hi)))
        return dict(systems=systems, n_stars=[N[s] for s in systems], leaf_system=leaf_system, leaf_slot=leaf_slot,
                    leaf_labels=list(leaf_id.keys()), bands=bands, terms=terms, spec=spec, limits=limits,
                    parallax={systems.index(s): v for s, v in self.parallax.items()},
                    AV={systems.index(s): v for s, v in self.AV.items()})
