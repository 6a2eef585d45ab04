"""ctypes binding of the C ABI declared in include/isochrones_amd.h.

The shared library (isochrones_amd/csrc/libiso_hip.so) is hand-written HIP for gfx950; there is
no CPU fallback: if the library is missing or fails to load, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

ISO_MAX_BANDS = 32
ISO_MAX_STARS = 3
ISO_MAX_PARAMS = 7
ISO_MAX_COLS = 64

KIND_TRACK = 0
KIND_ISO = 1
ERR_INVALID, ERR_HIP, ERR_NOMEM = -1, -2, -3      # include/isochrones_amd.h
CHAIN_ROW_MAJOR = 0
CHAIN_PARAM_MAJOR = 1

PRIOR_FLAT = 1
PRIOR_FLATLOG = 2
PRIOR_POWERLAW = 3
PRIOR_GAUSS = 4
PRIOR_LOGNORMAL = 5
PRIOR_CHABRIER = 6
PRIOR_FEH = 7


class IsoPrior(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("bounded", C.c_int32),
        ("lo", C.c_double),
        ("hi", C.c_double),
        ("a", C.c_double), ("b", C.c_double), ("c", C.c_double), ("d", C.c_double),
        ("e", C.c_double), ("f", C.c_double), ("g", C.c_double), ("h", C.c_double),
    ]


class IsoModelDesc(C.Structure):
    _fields_ = [
        ("n_stars", C.c_int32),
        ("n_bands", C.c_int32),
        ("bc_cols", C.c_int32 * ISO_MAX_BANDS),
        ("mag_val", C.c_double * ISO_MAX_BANDS),
        ("mag_unc", C.c_double * ISO_MAX_BANDS),
        ("spec_val", C.c_double * 3),
        ("spec_unc", C.c_double * 3),
        ("has_parallax", C.c_int32),
        ("has_numax", C.c_int32),
        ("has_dnu", C.c_int32),
        ("reserved0", C.c_int32),
        ("plx_val", C.c_double), ("plx_unc", C.c_double),
        ("numax_val", C.c_double), ("numax_unc", C.c_double),
        ("dnu_val", C.c_double), ("dnu_unc", C.c_double),
        ("prior_mass", IsoPrior), ("prior_age", IsoPrior), ("prior_feh", IsoPrior),
        ("prior_distance", IsoPrior), ("prior_AV", IsoPrior),
        ("eep_lo", C.c_double), ("eep_hi", C.c_double),
        ("bound_lo", C.c_double * ISO_MAX_PARAMS), ("bound_hi", C.c_double * ISO_MAX_PARAMS),
    ]


TREE_MAX_SYSTEMS, TREE_MAX_LEAVES, TREE_MAX_BANDS, TREE_MAX_TERMS, TREE_MAX_SPEC = 4, 8, 16, 64, 24


class IsoTreeTerm(C.Structure):
    _fields_ = [("band", C.c_int32), ("relative", C.c_int32), ("mask", C.c_uint32), ("ref_mask", C.c_uint32),
                ("mag", C.c_double), ("unc", C.c_double), ("ref_mag", C.c_double)]


class IsoTreeProp(C.Structure):
    _fields_ = [("leaf", C.c_int32), ("prop", C.c_int32), ("a", C.c_double), ("b", C.c_double)]


class IsoTreeDesc(C.Structure):
    _fields_ = [
        ("n_systems", C.c_int32), ("n_leaves", C.c_int32), ("n_bands", C.c_int32), ("n_terms", C.c_int32),
        ("n_spec", C.c_int32), ("n_limits", C.c_int32),
        ("n_stars", C.c_int32 * TREE_MAX_SYSTEMS),
        ("leaf_system", C.c_int32 * TREE_MAX_LEAVES), ("leaf_slot", C.c_int32 * TREE_MAX_LEAVES),
        ("bc_cols", C.c_int32 * TREE_MAX_BANDS),
        ("terms", IsoTreeTerm * TREE_MAX_TERMS),
        ("spec", IsoTreeProp * TREE_MAX_SPEC), ("limits", IsoTreeProp * TREE_MAX_SPEC),
        ("has_plx", C.c_int32 * TREE_MAX_SYSTEMS), ("has_av", C.c_int32 * TREE_MAX_SYSTEMS),
        ("plx_val", C.c_double * TREE_MAX_SYSTEMS), ("plx_unc", C.c_double * TREE_MAX_SYSTEMS),
        ("av_val", C.c_double * TREE_MAX_SYSTEMS), ("av_unc", C.c_double * TREE_MAX_SYSTEMS),
        ("prior_mass", IsoPrior), ("prior_age", IsoPrior), ("prior_feh", IsoPrior),
        ("prior_distance", IsoPrior), ("prior_AV", IsoPrior),
        ("eep_lo", C.c_double), ("eep_hi", C.c_double),
        ("bound_lo", C.c_double * 4), ("bound_hi", C.c_double * 4),
    ]


#: every symbol include/isochrones_amd.h declares (tests check the .so exports all of them)
EXPORTED_SYMBOLS = (
    "iso_last_error", "iso_version", "iso_ctx_create", "iso_ctx_destroy",
    "iso_table_create", "iso_table_create_from_device", "iso_table_destroy", "iso_interp", "iso_interp_host",
    "iso_ic_create", "iso_ic_destroy", "iso_interp_mag", "iso_interp_mag_host",
    "iso_model_create", "iso_model_destroy", "iso_model_n_params", "iso_model_kernel_path",
    "iso_axis_bracket_host", "iso_debug_trace_kernels", "iso_debug_kernels", "iso_debug_sampler_plan",
    "iso_lnpost", "iso_lnpost_host", "iso_unit_cube", "iso_time_lnpost", "iso_time_lnpost_rotating",
    "iso_catalog_create", "iso_catalog_create_columns", "iso_catalog_destroy", "iso_catalog_lnpost", "iso_catalog_start_points", "iso_catalog_patch_failed",
    "iso_eep_table_create", "iso_eep_table_destroy", "iso_interp_eep", "iso_interp_eep_host",
    "iso_sampler_create_model", "iso_sampler_create_model_ensembles", "iso_sampler_create_catalog", "iso_sampler_destroy", "iso_sampler_run",
    "iso_sampler_create_tree", "iso_sampler_create_isotrack",
    "iso_sampler_set_chain_layout", "iso_chain_quantiles", "iso_chain_quantiles_layout",
    "iso_tree_model_create", "iso_tree_model_destroy", "iso_tree_lnpost", "iso_tree_lnpost_host",
)

_LIB = None


def library_path() -> str:
    env = os.environ.get("ISOCHRONES_AMD_LIB")
    if env:
        return env
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libiso_hip.so")


class IsoError(RuntimeError):
    """``rc`` is the C ABI's return code (ERR_INVALID: no kernel / bad argument; ERR_HIP, ERR_NOMEM: real failures)."""
    rc = None


def trace_kernels(on=True):
    """Test hook: start (and clear) / stop recording which kernel instantiations this thread's calls launch."""
    return lib().iso_debug_trace_kernels(1 if on else 0)


def traced_kernels():
    """Names of the distinct kernel instantiations launched by this thread since trace_kernels(True), as c++filt spells
    them (the keys of isochrones_amd.csrc.build.resource_table())."""
    L = lib()
    need = L.iso_debug_kernels(None, 0)
    buf = C.create_string_buffer(int(need))
    L.iso_debug_kernels(buf, need)
    return [k for k in buf.value.decode().split("\n") if k]


def last_sampler_plan():
    """What this thread's last ``iso_sampler_run`` decided (BasicStarModel / catalog kernels): a dict with the keys
    persistent, dense, threads, dense_stdp, group, per_cu, workgroups."""
    out = (C.c_int32 * 8)()
    check(lib().iso_debug_sampler_plan(out))
    return dict(zip(("persistent", "dense", "threads", "dense_stdp", "group", "per_cu", "workgroups"), list(out)[:7]))


def lib():
    """Load (once) and return the HIP library with argtypes set.  Raises if it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise IsoError(
            "isochrones_amd: HIP library not found at %s — build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)" % path)
    try:
        # torch bundles its own libamdhip64 (same SONAME); importing it first makes this library
        # bind to that runtime, so torch's streams / allocations and ours are one HIP context.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    L = C.CDLL(path)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pd = C.c_void_p  # device pointers travel as integers
    L.iso_last_error.restype = C.c_char_p
    L.iso_last_error.argtypes = []
    L.iso_version.restype = C.c_char_p
    L.iso_version.argtypes = []
    if hasattr(L, "iso_debug_trace_kernels"):        # (absent from libraries built from older source states: tools/build_variant.py --src)
        L.iso_debug_trace_kernels.argtypes = [C.c_int]
        L.iso_debug_kernels.argtypes = [C.c_char_p, C.c_int64]
        L.iso_debug_kernels.restype = C.c_int64
    if hasattr(L, "iso_debug_sampler_plan"):
        L.iso_debug_sampler_plan.argtypes = [C.POINTER(C.c_int32)]
    L.iso_ctx_create.argtypes = [C.POINTER(vp), C.c_int]
    L.iso_ctx_destroy.argtypes = [vp]
    L.iso_ctx_destroy.restype = None
    L.iso_table_create.argtypes = [vp, C.c_int, C.POINTER(i64), C.POINTER(dbl),
                                   C.POINTER(C.POINTER(dbl)), C.POINTER(vp)]
    if hasattr(L, "iso_table_create_from_device"):
        L.iso_table_create_from_device.argtypes = [vp, C.c_int, C.POINTER(i64), pd, C.POINTER(C.POINTER(dbl)), C.POINTER(vp)]
    L.iso_table_destroy.argtypes = [vp]
    L.iso_table_destroy.restype = None
    L.iso_interp.argtypes = [vp, C.POINTER(pd), i64, C.POINTER(i32), C.c_int, pd, vp]
    L.iso_ic_create.argtypes = [vp, vp, vp, C.c_int, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32),
                                C.POINTER(vp)]
    L.iso_ic_destroy.argtypes = [vp]
    L.iso_ic_destroy.restype = None
    L.iso_interp_mag.argtypes = [vp, pd, i64, i64, i64, C.POINTER(i32), C.c_int, pd, pd, pd, pd, vp]
    L.iso_model_create.argtypes = [vp, C.POINTER(IsoModelDesc), C.POINTER(vp)]
    L.iso_model_destroy.argtypes = [vp]
    L.iso_model_destroy.restype = None
    L.iso_model_n_params.argtypes = [vp]
    L.iso_model_kernel_path.argtypes = [vp]
    L.iso_axis_bracket_host.argtypes = [C.POINTER(pd), C.POINTER(i32), C.c_int, C.c_int, C.c_int, pd, i64,
                                        C.POINTER(i32), C.POINTER(i32)]
    L.iso_lnpost.argtypes = [vp, pd, i64, i64, i64, pd, pd, pd, vp]
    L.iso_lnpost_host.argtypes = [vp, vp, i64, vp, vp, vp]
    L.iso_unit_cube.argtypes = [vp, pd, i64, i64, i64, vp]
    L.iso_time_lnpost.argtypes = [vp, pd, i64, i64, i64, pd, C.c_int, vp, C.POINTER(dbl)]
    L.iso_time_lnpost_rotating.argtypes = [vp, C.POINTER(pd), C.POINTER(pd), C.c_int, i64, i64, i64, C.c_int, vp,
                                           C.POINTER(dbl)]
    L.iso_catalog_create.argtypes = [vp, C.POINTER(IsoModelDesc), i64, C.POINTER(vp)]
    hd = C.POINTER(dbl)
    L.iso_catalog_create_columns.argtypes = [vp, C.POINTER(IsoModelDesc), i64, hd, hd, hd, hd, C.POINTER(C.c_int32), hd, hd,
                                             hd, C.POINTER(vp)]
    L.iso_catalog_destroy.argtypes = [vp]
    L.iso_catalog_destroy.restype = None
    L.iso_catalog_lnpost.argtypes = [vp, pd, pd, i64, i64, i64, pd, vp]
    if hasattr(L, "iso_catalog_start_points"):
        L.iso_catalog_start_points.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_uint64, pd, pd, pd, vp]
    if hasattr(L, "iso_catalog_patch_failed"):
        L.iso_catalog_patch_failed.argtypes = [vp, C.c_int, pd, pd, pd, vp]
    L.iso_eep_table_create.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64), C.POINTER(dbl), i64, C.POINTER(dbl), i64,
                                       i64, dbl, C.POINTER(vp)]
    L.iso_eep_table_destroy.argtypes = [vp]
    L.iso_eep_table_destroy.restype = None
    L.iso_interp_eep.argtypes = [vp, pd, pd, pd, i64, pd, vp]
    hdp = C.POINTER(dbl)
    L.iso_interp_host.argtypes = [vp, vp, i64, vp, C.c_int, vp]           # host double* as void*: plain ints pass
    L.iso_interp_mag_host.argtypes = [vp, vp, i64, C.POINTER(C.c_int32), C.c_int, vp, vp, vp, vp]
    L.iso_interp_eep_host.argtypes = [vp, vp, vp, vp, i64, vp]
    L.iso_sampler_create_model.argtypes = [vp, C.c_int, dbl, C.c_uint64, C.POINTER(vp)]
    L.iso_sampler_create_catalog.argtypes = [vp, C.c_int, dbl, C.c_uint64, C.POINTER(vp)]
    L.iso_sampler_create_model_ensembles.argtypes = [vp, i64, C.c_int, dbl, C.c_uint64, C.POINTER(vp)]
    L.iso_sampler_create_tree.argtypes = [vp, i64, C.c_int, dbl, C.c_uint64, C.POINTER(vp)]
    L.iso_sampler_create_isotrack.argtypes = [vp, vp, dbl, dbl, dbl, i64, C.c_int, dbl, C.c_uint64, C.POINTER(vp)]
    L.iso_sampler_destroy.argtypes = [vp]
    L.iso_sampler_destroy.restype = None
    L.iso_sampler_run.argtypes = [vp, pd, pd, C.c_int, pd, pd, pd, vp]
    L.iso_chain_quantiles.argtypes = [vp, pd, i64, i64, C.c_int, C.c_int, C.POINTER(dbl), C.c_int, pd, vp]
    L.iso_chain_quantiles_layout.argtypes = [vp, pd, C.c_int, i64, i64, C.c_int, C.c_int, C.POINTER(dbl), C.c_int, pd, vp]
    L.iso_sampler_set_chain_layout.argtypes = [vp, C.c_int]
    L.iso_tree_model_create.argtypes = [vp, C.POINTER(IsoTreeDesc), C.POINTER(vp)]
    L.iso_tree_model_destroy.argtypes = [vp]
    L.iso_tree_model_destroy.restype = None
    L.iso_tree_lnpost.argtypes = [vp, pd, i64, i64, i64, pd, pd, pd, vp]
    L.iso_tree_lnpost_host.argtypes = [vp, vp, i64, vp, vp, vp]
    for name in EXPORTED_SYMBOLS:
        if (name.startswith("iso_debug_") or name in ("iso_catalog_start_points", "iso_catalog_patch_failed", "iso_table_create_from_device")) and not hasattr(L, name) and os.environ.get("ISOCHRONES_AMD_LIB"):
            continue                                   # a variant library built from an older source state
        fn = getattr(L, name)
        if fn.restype is C.c_int:
            fn.restype = C.c_int
    _LIB = L
    return L


def check(rc: int):
    if rc != 0:
        msg = lib().iso_last_error()
        e = IsoError("isochrones_amd C-ABI error %d: %s" % (rc, (msg or b"").decode()))
        e.rc = rc
        raise e
