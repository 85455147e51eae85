"""Built-in objectives evaluated on the device (ids of bbo_objective_id, include/bbopt_hip.h).

The reference ships no objective functions; `rosenbrock` is its README objective
(/root/reference/README.md:111-112).  Pass one of these objects (or its name) as `f` to
optimize()/initialize() to keep the whole generation on the GPU.  Calling one with a NumPy
vector evaluates the same formula on the host -- a convenience for inspecting results; the only
optimizer code that uses it is CCPSO's optional local search (multivariate.py), whose objective
(the built-in composed with per-swarm weights) is a host callable by construction.
"""
import numpy as _np


class Builtin:
    def __init__(self, name, builtin_id, fn, box):
        self.name, self.builtin_id, self._fn, self.box = name, builtin_id, fn, box

    def __call__(self, x):
        return float(self._fn(_np.asarray(x, dtype=_np.float64)))

    def __repr__(self):
        return "<builtin objective %s (device id %d)>" % (self.name, self.builtin_id)


def _aux_t(n):
    return _np.arange(n) / max(n - 1, 1)


sphere = Builtin("sphere", 0, lambda x: _np.sum(x * x), (-10., 10.))
rosenbrock = Builtin("rosenbrock", 1, lambda x: _np.sum(
    100. * (x[1:] - x[:-1] ** 2) ** 2 + (1. - x[:-1]) ** 2), (-10., 10.))
rastrigin = Builtin("rastrigin", 2, lambda x: 10. * x.size + _np.sum(
    x * x - 10. * _np.cos(2. * _np.pi * x)), (-5.12, 5.12))
ellipsoid = Builtin("ellipsoid", 3, lambda x: _np.sum(
    10. ** (6. * _aux_t(x.size)) * x * x), (-10., 10.))
ackley = Builtin("ackley", 4, lambda x: -20. * _np.exp(-0.2 * _np.sqrt(_np.mean(x * x)))
                 - _np.exp(_np.mean(_np.cos(2. * _np.pi * x))) + 20. + _np.e, (-32., 32.))
griewank = Builtin("griewank", 5, lambda x: 1. + _np.sum(x * x) / 4000. - _np.prod(
    _np.cos(x / _np.sqrt(_np.arange(1, x.size + 1)))), (-600., 600.))
cigar = Builtin("cigar", 6, lambda x: x[0] ** 2 + 1e6 * _np.sum(x[1:] ** 2), (-10., 10.))
discus = Builtin("discus", 7, lambda x: 1e6 * x[0] ** 2 + _np.sum(x[1:] ** 2), (-10., 10.))
diffpow = Builtin("diffpow", 8, lambda x: _np.sum(
    _np.abs(x) ** (2. + 4. * _aux_t(x.size))), (-10., 10.))
schwefel12 = Builtin("schwefel12", 9, lambda x: _np.sum(_np.cumsum(x) ** 2), (-10., 10.))

ALL = (sphere, rosenbrock, rastrigin, ellipsoid, ackley, griewank, cigar, discus, diffpow,
       schwefel12)


def vectorized(f):
    """marks a Python objective as taking the whole population X[rows, n] and returning
    rows values (one host call per generation instead of one per candidate)"""
    f._bbo_vectorized = True
    return f
