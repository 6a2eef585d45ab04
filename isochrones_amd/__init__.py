"""isochrones_amd — MI355X-native implementation of the isochrones hot path
(DFInterpolator N-D interpolation -> interp_mag -> star_lnlike -> StarModel.lnpost).

Host side: Python mirroring the reference's interface for this path; compute: hand-written
HIP kernels for gfx950 behind the C ABI in include/isochrones_amd.h (no CPU fallback)."""
from .interp import DFInterpolator
from .models import (ModelGridInterpolator, EvolutionTrackInterpolator, IsochroneInterpolator,
                     synthetic_track, synthetic_isochrone, get_ichrone)
from .starmodel import (BasicStarModel, StarModel, TreeStarModel, SingleStarModel, BinaryStarModel,
                        TripleStarModel, IsoTrackModel)
from .observation import ObservationTree, Observation, Source
from ._cabi import IsoError
from .sampler import EnsembleSampler, FusedEnsembleSampler
from .catalog import (StarCatalog, CatalogPosterior, fit_catalog, synthetic_catalog, shard_of, shard_indices,
                      broadcast_interpolator)
from . import priors, grids, ingest, mist, nested, ini, persist, utils
from .starfit import starfit, batch_starfit

__version__ = "0.1.0"
