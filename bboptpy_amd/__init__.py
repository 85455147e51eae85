"""bboptpy_amd -- MI355X-native core of bboptpy's population-based optimizers.

Drop-in for the CMA-ES / DE / PSO classes of mike-gimelfarb/bboptpy
(`from bboptpy import ActiveCMAES` -> `from bboptpy_amd import ActiveCMAES`); the hot path
(sample -> evaluate -> rank -> update) runs as hand-written gfx950 kernels behind the C ABI
of include/bbopt_hip.h.  See DESIGN.md.
"""
from . import objectives
from .objectives import vectorized
from .multivariate import (MultivariateSolution, MultivariateSearch, BaseCMAES, CMAES,
                           ActiveCMAES, SepCMAES, IPopCMAES, BiPopCMAES, JADE, SHADE,
                           SANSDE, APSO, CSO, CCPSO)

__all__ = ["MultivariateSolution", "MultivariateSearch", "BaseCMAES", "CMAES", "ActiveCMAES",
           "SepCMAES", "IPopCMAES", "BiPopCMAES", "JADE", "SHADE", "SANSDE", "APSO", "CSO", "CCPSO", "objectives",
           "vectorized"]
