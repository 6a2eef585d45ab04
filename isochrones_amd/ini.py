"""``star.ini`` files: the on-disk description of one star's measurements that the reference's
``StarModel.from_ini`` / ``BasicStarModel.write_ini`` read and write through configobj
(isochrones/starmodel.py:248-436, 1485-1499; examples under isochrones/tests/star1..star4).

configobj is not a dependency here: the files only use its simplest layer — ``key = value`` lines,
one level of ``[section]`` headers, ``#`` comments and comma-separated lists — which
:func:`read_ini` parses directly.  :func:`observation_rows` turns the sections into the
``name, band, resolution, relative, separation, pa, mag, e_mag`` rows the reference builds before
handing them to ``ObservationTree.from_df``.
"""
from __future__ import annotations

import re
from collections import OrderedDict

import numpy as np

# keywords that are never photometric bands (reference: StarModel._not_a_band, starmodel.py:95-113)
NOT_A_BAND = ("RA", "dec", "ra", "Dec", "maxAV", "parallax", "AV", "logg", "Teff", "feh", "density", "separation",
              "PA", "resolution", "relative", "N", "index", "id", "obsfile", "name", "max_distance", "nu_max",
              "delta_nu")


def _strip_comment(line):
    out, quote = [], None
    for ch in line:
        if quote:
            if ch == quote:
                quote = None
        elif ch in "'\"":
            quote = ch
        elif ch == "#":
            break
        out.append(ch)
    return "".join(out).strip()


def _unquote(s):
    s = s.strip()
    if len(s) >= 2 and s[0] == s[-1] and s[0] in "'\"":
        return s[1:-1]
    return s


def _value(text):
    """configobj's reading of a value: a string, or a list of strings when it holds commas."""
    text = text.strip()
    if "," in text and not (len(text) >= 2 and text[0] == text[-1] and text[0] in "'\""):
        items = [_unquote(x) for x in text.split(",")]
        if items and items[-1] == "":            # "a, b," -> ['a', 'b']
            items = items[:-1]
        return items
    return _unquote(text)


def read_ini(path):
    """-> (scalars, sections): two ordered dicts; ``sections[name]`` is itself an ordered dict."""
    scalars, sections = OrderedDict(), OrderedDict()
    where = scalars
    with open(path) as f:
        for lineno, raw in enumerate(f, 1):
            line = _strip_comment(raw)
            if not line:
                continue
            m = re.match(r"^\[\s*([^\[\]]+?)\s*\]$", line)
            if m:
                where = sections.setdefault(m.group(1), OrderedDict())
                continue
            if "=" not in line:
                raise ValueError("%s:%d: expected 'key = value' or '[section]', got %r" % (path, lineno, raw.rstrip()))
            k, v = line.split("=", 1)
            where[k.strip()] = _value(v)
    return scalars, sections


def parse_value(v):
    """A number, a list of numbers, or the string itself (reference: _parse_config_value, starmodel.py:51-60)."""
    try:
        return float(v)
    except (TypeError, ValueError):
        try:
            return [float(x) for x in v]
        except (TypeError, ValueError):
            return v


def parse_band(key):
    """The band an ini keyword names (``K_1`` -> ``K``), or None for the non-photometric keywords."""
    m = re.match(r"^([a-zA-Z0-9]+)(_\w+)?$", key.strip())
    if not m or m.group(1) in NOT_A_BAND:
        return None
    return m.group(1)


def get_bands(path):
    """Every photometric band an ini file mentions (reference: StarModel.get_bands, starmodel.py:229-246)."""
    scalars, sections = read_ini(path)
    found = []
    for k in list(scalars) + [k for sec in sections.values() for k in sec]:
        b = parse_band(k)
        if b is not None and b not in found:
            found.append(b)
    return found


def write_ini(path, scalars, sections=None):
    def fmt(v):
        if isinstance(v, (tuple, list, np.ndarray)):
            return ", ".join(repr(float(x)) if isinstance(x, (float, np.floating)) else str(x) for x in v)
        return repr(float(v)) if isinstance(v, (float, np.floating)) else str(v)
    with open(path, "w") as f:
        for k, v in scalars.items():
            f.write("%s = %s\n" % (k, fmt(v)))
        for name, sec in (sections or {}).items():
            f.write("\n[%s]\n" % name)
            for k, v in sec.items():
                f.write("%s = %s\n" % (k, fmt(v)))


def observation_rows(sections):
    """The photometry rows of the ``[instrument]`` sections, by the reference's rules
    (starmodel.py:336-424):

    * a section without ``resolution`` is an absolute, seeing-limited (4") measurement of everything;
    * a section with ``resolution`` lists companions relative to the primary unless ``relative`` says
      otherwise; companions carry ``separation_<tag>`` / ``PA_<tag>`` / ``<band>_<tag>``, and a relative
      section gets the reference star itself as a (0, 0.01) row per band;
    * rows whose magnitude or uncertainty is NaN are dropped.
    """
    rows = []
    for instrument, sec in sections.items():
        if "resolution" in sec:
            resolution, relative = float(sec["resolution"]), True
        else:
            resolution, relative = 4.0, False
        if "relative" in sec:
            relative = sec["relative"] == "True"
        tags, bands = [], []
        for label in sec:
            m = re.search(r"separation(_\w+)?", label)
            if m:
                if m.group(1) is not None and m.group(1) not in tags:
                    tags.append(m.group(1))
            elif re.search(r"PA", label) or re.search(r"id", label) or label in ("resolution", "relative"):
                continue
            else:
                b = re.search(r"([a-zA-Z0-9]+)(_\w+)?", label).group(1)
                if b not in bands:
                    bands.append(b)
        if bands and (not tags or bands[0] in sec):
            tags.append("")
        for b in bands:
            for tag in tags:
                key = b + tag
                if key not in sec:
                    continue
                try:
                    if isinstance(sec[key], str):
                        raise ValueError
                    mag, e_mag = (float(x) for x in sec[key])
                except (TypeError, ValueError):
                    raise ValueError("[%s] %s: expected 'magnitude, uncertainty', got %r" % (instrument, key, sec[key]))
                if "separation" + tag in sec:
                    sep, pa = float(sec["separation" + tag]), float(sec["PA" + tag])
                else:
                    sep, pa = 0.0, 0.0
                if not np.isnan(mag) and not np.isnan(e_mag):
                    rows.append(dict(name=instrument, band=b, resolution=resolution, relative=relative,
                                     separation=sep, pa=pa, mag=mag, e_mag=e_mag))
            if relative:
                rows.append(dict(name=instrument, band=b, resolution=resolution, relative=relative,
                                 separation=0.0, pa=0.0, mag=0.0, e_mag=0.01))
    return rows
