"""Python class surface of the reference, over the HIP C ABI.

Mirrors the pybind11 module `_bboptpy` for the algorithms on the hot path
(/root/reference/py/multivariate_py.cpp): same class names, the same class hierarchy
(ActiveCMAES -> CMAES -> BaseCMAES -> MultivariateSearch, :99-115), the same positional
order, keyword names and defaults of every constructor (:103-171, :265-269), the four
methods optimize / initialize / iterate / solution (:376-420) and the MultivariateSolution
result object (:360-371).  Every constructor additionally accepts keyword-only extensions
that have no reference counterpart: `seed` (Philox key; default = fresh entropy, like the
reference's random_device seeding), `device`, `populations`, and `poll_every`.

All computation happens in libbbopt_hip.so on the GPU.  A Python callable objective is
supported through the host-callback path (X leaves HBM once per generation); the objects in
`bboptpy_amd.objectives` select the built-in on-device objectives instead.
"""
import ctypes as C
import os
import sys

import numpy as _np

from . import _ffi
from .objectives import Builtin


class MultivariateSolution:
    """result record, multivariate.h:81-115 / multivariate_py.cpp:360-371"""

    def __init__(self, x, n_evals, converged):
        self._x = _np.array(x, dtype=_np.float64)
        self._n_evals = int(n_evals)
        self._converged = bool(converged)

    @property
    def x(self):
        return self._x.copy()   # the reference returns a fresh array each time (:362-364)

    @property
    def converged(self):
        return self._converged

    @property
    def n_evals(self):
        return self._n_evals

    def __str__(self):
        # multivariate_solution::toString, multivariate.h:97-114 (std::to_string == "%f")
        sol = "".join("%f " % v for v in self._x)
        return ("x*: " + sol + "\n"
                + "objective calls: %d\n" % self._n_evals
                + "constraint calls: 0\n"
                + "B/B constraint calls: 0\n"
                + "converged: " + ("yes" if self._converged else "no/unknown"))

    def __repr__(self):
        return "<MultivariateSolution n_evals=%d converged=%s>" % (self._n_evals,
                                                                   self._converged)


def _as_vec(a, name):
    v = _np.ascontiguousarray(_np.asarray(a, dtype=_np.float64)).ravel()
    if v.size == 0:
        raise ValueError("%s must not be empty" % name)
    return v


class _ObjectiveBinding:
    """keeps the ctypes callback (and the user's callable) alive for the handle's life,
    like the std::function captured by value in multivariate_py.cpp:385-388"""

    def __init__(self, f, n):
        self.error = None
        self.struct = _ffi.Objective()
        self._keep = None
        if isinstance(f, Builtin):
            self.struct.kind = _ffi.OBJ_BUILTIN
            self.struct.builtin = f.builtin_id
            return
        if isinstance(f, str):
            if f not in _ffi.BUILTIN_IDS:
                raise ValueError("unknown built-in objective %r" % f)
            self.struct.kind = _ffi.OBJ_BUILTIN
            self.struct.builtin = _ffi.BUILTIN_IDS[f]
            return
        if not callable(f):
            raise TypeError("objective must be callable or a built-in objective")
        binding = self
        if getattr(f, "_bbo_vectorized", False):
            def batch(xptr, rows, ncol, ld, fout, user):
                try:
                    X = _np.ctypeslib.as_array(xptr, shape=(rows, ld))[:, :ncol].copy()
                    vals = _np.asarray(f(X), dtype=_np.float64).ravel()
                    if vals.size != rows:
                        raise ValueError("vectorized objective returned %d values for %d rows"
                                         % (vals.size, rows))
                    _np.ctypeslib.as_array(fout, shape=(rows,))[:] = vals
                    return 0
                except BaseException as e:   # propagate like pybind's error_already_set
                    binding.error = e
                    return 1
            self._keep = _ffi.BATCH_FN(batch)
            self.struct.kind = _ffi.OBJ_BATCH_CB
            self.struct.batch = self._keep
        else:
            def scalar(xptr, ncol, user, failed):
                try:
                    x = _np.ctypeslib.as_array(xptr, shape=(ncol,)).copy()
                    return float(f(x))
                except BaseException as e:
                    binding.error = e
                    failed[0] = 1
                    return 0.0
            self._keep = _ffi.SCALAR_FN(scalar)
            self.struct.kind = _ffi.OBJ_SCALAR_CB
            self.struct.scalar = self._keep


class MultivariateSearch:
    """MultivariateOptimizer (multivariate.h:132-146) as bound at multivariate_py.cpp:374-420"""

    _algo = None

    def __init__(self, *, seed=None, device=0, populations=1, poll_every=None):
        _ffi.lib()   # fail loudly at construction when the HIP library is missing
        self._params = _ffi.default_params(self._algo)
        if poll_every is not None:      # generations between host polls of the stop flags in run()
            self._params.poll_every = max(1, int(poll_every))
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        self._params.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._params.device = int(device)
        self._params.populations = int(populations)
        self._handle = None
        self._binding = None
        self._n = None

    # -- handle management --------------------------------------------------------
    def _create(self):
        h = C.c_void_p()
        _ffi.check(_ffi.lib().bbo_create(C.byref(self._params), C.byref(h)))
        return h

    def _ensure_handle(self):
        if self._handle is None:
            self._handle = self._create()
        return self._handle

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None:
                _ffi.lib().bbo_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def reseed(self, seed):
        """(extension) a new generator seed for the next initialize() / optimize(); the handle
        is re-created (the seed is part of bbo_params)"""
        self._params.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        if self._handle is not None:
            _ffi.lib().bbo_destroy(self._handle)
            self._handle = None

    def _check(self, status):
        if status < 0:
            err = self._binding.error if self._binding is not None else None
            if err is not None:
                self._binding.error = None
                raise err
            _ffi.check(status, self._handle)
        return status

    def _problem(self, f, lower, upper, guess):
        lower = _as_vec(lower, "lower")
        n = lower.size            # like the reference, n comes from `lower` (:380)
        upper = _as_vec(upper, "upper")
        guess = _as_vec(guess, "guess")
        pops = self._params.populations
        if upper.size < n or guess.size < n * pops:
            raise ValueError("upper/guess are shorter than lower (n = %d)" % n)
        self._binding = _ObjectiveBinding(f, n)
        self._n = n
        return n, lower, _np.ascontiguousarray(upper[:n]), _np.ascontiguousarray(
            guess[:n * pops])

    # -- the four reference methods --------------------------------------------------
    def optimize(self, f, lower, upper, guess):
        n, lower, upper, guess = self._problem(f, lower, upper, guess)
        h = self._ensure_handle()
        x = _np.zeros(n)
        fev, conv = C.c_int(), C.c_int()
        self._check(_ffi.lib().bbo_optimize(h, n, lower, upper, guess,
                                            C.byref(self._binding.struct), x,
                                            C.byref(fev), C.byref(conv)))
        return MultivariateSolution(x, fev.value, conv.value)

    def initialize(self, f, lower, upper, guess):
        n, lower, upper, guess = self._problem(f, lower, upper, guess)
        h = self._ensure_handle()
        self._check(_ffi.lib().bbo_init(h, n, lower, upper, guess,
                                        C.byref(self._binding.struct)))

    def iterate(self):
        if self._handle is None:
            raise RuntimeError("iterate() called before initialize()")
        self._check(_ffi.lib().bbo_iterate(self._handle))

    def solution(self, population=0):
        if self._handle is None:
            raise RuntimeError("solution() called before initialize()")
        x = _np.zeros(self._n)
        fev, conv = C.c_int(), C.c_int()
        self._check(_ffi.lib().bbo_solution_of(self._handle, int(population), x,
                                               C.byref(fev), C.byref(conv)))
        return MultivariateSolution(x, fev.value, conv.value)

    # -- extensions ---------------------------------------------------------------------
    def run(self, max_generations):
        """advance up to max_generations on the device without returning to Python;
        returns the number of generations launched"""
        if self._handle is None:
            raise RuntimeError("run() called before initialize()")
        done = C.c_int()
        self._check(_ffi.lib().bbo_run(self._handle, int(max_generations), C.byref(done)))
        return done.value

    def get_state(self, key, population=0):
        """named optimizer state as a float64 array (keys: DESIGN.md)"""
        if self._handle is None:
            raise RuntimeError("get_state() called before initialize()")
        L = _ffi.lib()
        cnt = self._check(L.bbo_get(self._handle, key.encode(), population, None, 0))
        out = _np.zeros(max(cnt, 1))
        self._check(L.bbo_get(self._handle, key.encode(), population,
                              out.ctypes.data_as(C.c_void_p), cnt))
        return out[:cnt]

    def set_state(self, key, value, population=0):
        v = _np.ascontiguousarray(_np.asarray(value, dtype=_np.float64)).ravel()
        self._check(_ffi.lib().bbo_set(self._handle, key.encode(), population, v, v.size))


class BaseCMAES(MultivariateSearch):
    """BaseCmaes, multivariate_py.cpp:99-101 (abstract in the reference)"""

    def phase(self, which):
        self._check(_ffi.lib().bbo_cma_phase_run(self._handle, int(which)))

    def set_params(self, np, sigma0, mfev):
        """BaseCmaes::setParams (base_cmaes.cpp:136-148; not bound to Python by the reference,
        used by its restart drivers): new lambda, sigma0 and budget for the next initialize() /
        optimize() of this object, which keeps B and C's off-diagonals (cmaes.cpp:53-59)"""
        p = self._params
        p.np, p.sigma0, p.mfev = int(np), float(sigma0), int(mfev)
        if self._handle is not None:
            self._check(_ffi.lib().bbo_cma_set_params(self._handle, p.np, p.sigma0, p.mfev))
        if p.bound:
            p.bound = 0     # the handle printed the reference's warning (or will not need to)

    def set_seed(self, seed):
        """(extension) a new Philox key WITHOUT re-creating the handle (reseed() does)"""
        self._params.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        if self._handle is not None:
            self._check(_ffi.lib().bbo_cma_set_seed(self._handle, self._params.seed))

    def evaluate(self, x):
        """the bound objective at one point (the restart drivers' extra evaluation,
        bipop_cmaes.cpp:86)"""
        if self._handle is None:
            raise RuntimeError("evaluate() called before initialize()")
        x = _np.ascontiguousarray(_np.asarray(x, dtype=_np.float64)).ravel()
        if x.size != self._n:
            raise ValueError("evaluate: x must have n = %d coordinates" % self._n)
        out = C.c_double()
        self._check(_ffi.lib().bbo_cma_evaluate(self._handle, x, C.byref(out)))
        return out.value

    def inject_normals(self, z):
        if z is None:
            self._check(_ffi.lib().bbo_cma_inject_normals(self._handle, None, 0))
            return
        z = _np.ascontiguousarray(_np.asarray(z, dtype=_np.float64)).ravel()
        self._check(_ffi.lib().bbo_cma_inject_normals(
            self._handle, z.ctypes.data_as(C.c_void_p), z.size))


class CMAES(BaseCMAES):
    """CMAES(mfev, tol, np, sigma0=2., bound=False, eigenrate=0.25) -- :103-108"""
    _algo = _ffi.ALGO_CMAES

    def __init__(self, mfev, tol, np, sigma0=2., bound=False, eigenrate=0.25, **ext):
        super().__init__(**ext)
        p = self._params
        p.mfev, p.tol, p.np = int(mfev), float(tol), int(np)
        p.sigma0, p.bound, p.eigenrate = float(sigma0), int(bool(bound)), float(eigenrate)


class ActiveCMAES(CMAES):
    """ActiveCMAES(mfev, tol, np, sigma0=2., bound=False, alphacov=2., eigenrate=0.25)
    -- :110-115"""
    _algo = _ffi.ALGO_ACTIVE_CMAES

    def __init__(self, mfev, tol, np, sigma0=2., bound=False, alphacov=2., eigenrate=0.25,
                 **ext):
        super().__init__(mfev, tol, np, sigma0, bound, eigenrate, **ext)
        self._params.alphacov = float(alphacov)


class SepCMAES(BaseCMAES):
    """SepCMAES(mfev, tol, np, sigma0=2., bound=False, adjustlr=True) -- :131-135
    (diagonal covariance, Ros & Hansen 2008; sep_cmaes.cpp)"""
    _algo = _ffi.ALGO_SEP_CMAES

    def __init__(self, mfev, tol, np, sigma0=2., bound=False, adjustlr=True, **ext):
        super().__init__(**ext)
        p = self._params
        p.mfev, p.tol, p.np = int(mfev), float(tol), int(np)
        p.sigma0, p.bound, p.adjustlr = float(sigma0), int(bool(bound)), int(bool(adjustlr))


class _RestartDriver(MultivariateSearch):
    def __init__(self, base, **ext):
        if not isinstance(base, BaseCMAES):
            raise TypeError("base must be a CMA-ES optimizer (CMAES / ActiveCMAES / SepCMAES)")
        if base._params.populations != 1:
            raise ValueError("restart drivers need a base optimizer with populations=1")
        super().__init__(**ext)
        self._base = base   # kept alive here; the reference only borrows the pointer
        self._base_handle = None

    def _create(self):
        h = C.c_void_p()
        base_h = self._base._ensure_handle()
        _ffi.check(_ffi.lib().bbo_create_restart(C.byref(self._params), base_h, C.byref(h)))
        self._base_handle = base_h
        return h

    def _ensure_handle(self):
        # the driver's handle borrows the base's engine: if the base re-created its handle
        # (reseed()), the driver's is stale and is rebuilt over the new one
        if self._handle is not None and self._base._handle is not self._base_handle:
            _ffi.lib().bbo_destroy(self._handle)
            self._handle = None
        return super()._ensure_handle()

    def _live(self):
        if self._handle is not None and self._base._handle is not self._base_handle:
            raise RuntimeError("the base optimizer was re-created (reseed) after this restart "
                               "driver was initialized: call initialize() / optimize() again")

    def iterate(self):
        self._live()
        super().iterate()

    def run(self, max_generations):
        self._live()
        return super().run(max_generations)


class IPopCMAES(_RestartDriver):
    """IPopCMAES(base, mfev, print=False, sigma0=2., nipop=True, ksigmadec=1.6,
    boundlambda=True) -- :137-142"""
    _algo = _ffi.ALGO_IPOP

    def __init__(self, base, mfev, print=False, sigma0=2., nipop=True, ksigmadec=1.6,
                 boundlambda=True, **ext):
        super().__init__(base, **ext)
        p = self._params
        p.mfev, p.print, p.sigma0 = int(mfev), int(bool(print)), float(sigma0)
        p.nipop, p.ksigmadec, p.boundlambda = int(bool(nipop)), float(ksigmadec), int(
            bool(boundlambda))


class BiPopCMAES(_RestartDriver):
    """BiPopCMAES(base, mfev, print=False, sigma0=2., maxlargeruns=9, nbipop=True,
    ksigmadec=1.6, kbudget=2.) -- :144-151"""
    _algo = _ffi.ALGO_BIPOP

    def __init__(self, base, mfev, print=False, sigma0=2., maxlargeruns=9, nbipop=True,
                 ksigmadec=1.6, kbudget=2., **ext):
        super().__init__(base, **ext)
        p = self._params
        p.mfev, p.print, p.sigma0 = int(mfev), int(bool(print)), float(sigma0)
        p.maxlargeruns, p.nipop = int(maxlargeruns), int(bool(nbipop))
        p.ksigmadec, p.kbudget = float(ksigmadec), float(kbudget)


class JADE(MultivariateSearch):
    """JADE(mfev, np, tol, archive=True, repaircr=True, pelite=0.05, cdamp=0.1, sigma=0.07)
    -- :159-164"""
    _algo = _ffi.ALGO_JADE

    def __init__(self, mfev, np, tol, archive=True, repaircr=True, pelite=0.05, cdamp=0.1,
                 sigma=0.07, **ext):
        super().__init__(**ext)
        p = self._params
        p.mfev, p.np, p.tol = int(mfev), int(np), float(tol)
        p.archive, p.repaircr = int(bool(archive)), int(bool(repaircr))
        p.pelite, p.cdamp, p.jade_sigma = float(pelite), float(cdamp), float(sigma)


class SHADE(MultivariateSearch):
    """SHADE(mfev, npinit, tol, archive=True, repaircr=True, h=100, npmin=4) -- :166-171"""
    _algo = _ffi.ALGO_SHADE

    def __init__(self, mfev, npinit, tol, archive=True, repaircr=True, h=100, npmin=4, **ext):
        super().__init__(**ext)
        p = self._params
        p.mfev, p.np, p.tol = int(mfev), int(npinit), float(tol)
        p.archive, p.repaircr, p.h, p.npmin = int(bool(archive)), int(bool(repaircr)), int(
            h), int(npmin)


class SANSDE(MultivariateSearch):
    """SANSDE(mfev, np, tol, repaircr=True, crref=5, pupdate=50, crupdate=25) -- :174-177
    (self-adaptive DE with neighbourhood search, Yang et al. 2008; sansde.cpp)"""
    _algo = _ffi.ALGO_SANSDE

    def __init__(self, mfev, np, tol, repaircr=True, crref=5, pupdate=50, crupdate=25, **ext):
        super().__init__(**ext)
        p = self._params
        p.mfev, p.np, p.tol = int(mfev), int(np), float(tol)
        p.repaircr = int(bool(repaircr))
        p.crref, p.pupdate, p.crupdate = int(crref), int(pupdate), int(crupdate)


class CSO(MultivariateSearch):
    """CSO(mfev, stol, np, pcompete=3, ring=False, correct=True, vmax=0.2) -- :272-275
    (competitive swarm optimizer, Cheng & Jin 2015; cso.cpp)"""
    _algo = _ffi.ALGO_CSO

    def __init__(self, mfev, stol, np, pcompete=3, ring=False, correct=True, vmax=0.2, **ext):
        super().__init__(**ext)
        p = self._params
        p.mfev, p.tol, p.np = int(mfev), float(stol), int(np)
        p.pcompete, p.ring = int(pcompete), int(bool(ring))
        p.correct, p.vmax = int(bool(correct)), float(vmax)


class CCPSO(MultivariateSearch):
    """CCPSO(mfev, sigmatol, np, pps, npps, correct=True, pcauchy=-1., local=None, localfreq=10)
    -- :291-295 (cooperatively coevolving PSO, Li & Yao 2012; ccpso.cpp).  `pps` lists the
    candidate swarm sizes (each must divide n).

    `local` (ccpso.cpp:116-118, 371-435): another optimizer.  Every `localfreq` generations one
    weight per swarm, scaling that swarm's coordinates of the context vector, is optimized by it
    inside the box that keeps the scaled vector in bounds, and the result replaces the context
    vector if it is better.  The generations run on the device; the search's objective evaluates
    on the host (the callable itself, or the built-in's formula), one population only.  Each
    search starts `local` afresh with the seed base + search index (the reference's CMA-ES
    objects start from the previous search's matrices, cmaes.cpp:53-54 -- not reproduced).
    A CMA-ES optimizer of THIS package is handed to the library (bbo_ccpso_set_local: the whole
    loop then runs behind bbo_iterate / bbo_optimize, as for a C caller); any other object with
    optimize(f, lower, upper, guess) -> solution carrying .x and .n_evals is driven from here
    between device generations through bbo_get / bbo_set."""
    _algo = _ffi.ALGO_CCPSO

    def __init__(self, mfev, sigmatol, np, pps, npps=None, correct=True, pcauchy=-1., local=None,
                 localfreq=10, **ext):
        super().__init__(**ext)
        pps = [int(v) for v in pps]
        npps = len(pps) if npps is None else int(npps)
        if not 1 <= npps <= min(16, len(pps)):
            raise ValueError("CCPSO: npps must be in [1, min(16, len(pps))]")
        p = self._params
        p.mfev, p.tol, p.np = int(mfev), float(sigmatol), int(np)
        p.npps = npps
        for k in range(npps):
            p.pps[k] = pps[k]
        p.correct, p.pcauchy = int(bool(correct)), float(pcauchy)
        if local is not None:
            if not callable(getattr(local, "optimize", None)):
                raise TypeError("CCPSO: local must provide optimize(f, lower, upper, guess)")
            if p.populations != 1:
                raise ValueError("CCPSO: the local optimizer works on one population")
        self._local, self._localfreq = local, int(localfreq)
        self._nlocal = 0
        self._local_seed0 = getattr(getattr(local, "_params", None), "seed", 0)
        # a CMA-ES object of this package: the search runs inside the library
        self._local_native = isinstance(local, BaseCMAES)
        self._local_handle = None

    def _attach_local(self):
        """(native local optimizer) hand its handle to the engine; again whenever either handle
        has been re-created"""
        lh = self._local._ensure_handle()
        if self._local_handle is not lh:
            self._check(_ffi.lib().bbo_ccpso_set_local(self._handle, lh, self._localfreq))
            self._local_handle = lh

    def _create(self):
        h = super()._create()
        self._local_handle = None
        return h

    def __del__(self):
        # detach first: the engine only borrows the local optimizer's handle
        try:
            if getattr(self, "_handle", None) is not None and getattr(self, "_local_native", False):
                _ffi.lib().bbo_ccpso_set_local(self._handle, None, 0)
        except Exception:
            pass
        super().__del__()

    # ---- swarm groups sharded over GPUs (bboptpy_amd.distributed.ShardedCCPSO) ---------------
    def set_shard(self, rank, world):
        """this object evaluates the swarms [nswarm rank / world, nswarm (rank + 1) / world)"""
        self._check(_ffi.lib().bbo_ccpso_set_shard(self._ensure_handle(), int(rank), int(world)))

    def phase(self, which):
        """0: regroup + evaluate this rank's swarms; 1: the rest of the generation"""
        self._check(_ffi.lib().bbo_ccpso_phase(self._handle, int(which)))

    def table_record(self):
        return self._check(_ffi.lib().bbo_ccpso_table_record(self._handle))

    def export_tables(self, out=None, device_ptr=None):
        """this rank's fitness record (fX | fY) into a host array, or to a device pointer"""
        if device_ptr is not None:
            self._check(_ffi.lib().bbo_ccpso_export_tables(self._handle, C.c_void_p(device_ptr), 1))
            return None
        if out is None:
            out = _np.zeros(self.table_record())
        self._check(_ffi.lib().bbo_ccpso_export_tables(self._handle,
                                                       out.ctypes.data_as(C.c_void_p), 0))
        return out

    def merge_tables(self, gathered=None, world=1, device_ptr=None):
        """every swarm's rows from the record of the rank that evaluated it"""
        if device_ptr is not None:
            self._check(_ffi.lib().bbo_ccpso_merge_tables(self._handle, C.c_void_p(device_ptr),
                                                          int(world), 1))
            return
        g = _np.ascontiguousarray(gathered, dtype=_np.float64)
        self._check(_ffi.lib().bbo_ccpso_merge_tables(self._handle, g.ctypes.data_as(C.c_void_p),
                                                      int(world), 0))

    # ---- the reference's loop with the local search between generations ------------------
    def initialize(self, f, lower, upper, guess):
        if self._local is not None and self._local_native:
            self._ensure_handle()
            self._attach_local()
            return super().initialize(f, lower, upper, guess)
        super().initialize(f, lower, upper, guess)
        if isinstance(f, str):
            from . import objectives as _objectives
            f = getattr(_objectives, f, None)
        self._f_host = f if callable(f) else None
        self._lower_h, self._upper_h = _as_vec(lower, "lower"), _as_vec(upper, "upper")[:self._n]
        self._nlocal = 0

    def iterate(self):
        if self._local is not None and self._local_native:
            self._attach_local()
            return super().iterate()
        gen = int(self.get_state("it")[0]) if self._local is not None else 0
        super().iterate()
        if self._local is not None and self._localfreq > 0 and gen % self._localfreq == 0:
            self._local_search()

    def run(self, max_generations):
        if self._local is not None and not self._local_native:
            raise RuntimeError("CCPSO.run() with a local optimizer driven from Python: use iterate()")
        if self._local is not None:
            self._attach_local()
        return super().run(max_generations)

    def optimize(self, f, lower, upper, guess):
        if self._local is None:
            return super().optimize(f, lower, upper, guess)
        if self._local_native:
            self._ensure_handle()
            self._attach_local()
            return super().optimize(f, lower, upper, guess)
        self.initialize(f, lower, upper, guess)
        mfev = self._params.mfev
        while True:                                   # ccpso.cpp:135-148
            self.iterate()
            if int(self.get_state("fev")[0]) >= mfev:
                converged = False
                break
            if int(self.get_state("conv")[0]):
                converged = True
                break
        return MultivariateSolution(self.get_state("yhat"), int(self.get_state("fev")[0]),
                                    converged)

    def _local_search(self):
        """CCPSOSearch::localSearch, ccpso.cpp:371-435"""
        n = self._n
        f = self._f_host
        if f is None:
            raise RuntimeError("CCPSO: the local optimizer needs a callable objective")
        yhat = self.get_state("yhat")
        k = self.get_state("k").astype(int)           # position -> coordinate, swarm-major
        cps, nsw = int(self.get_state("cpswarm")[0]), int(self.get_state("nswarm")[0])
        group = _np.empty(n, dtype=int)
        group[k] = _np.repeat(_np.arange(nsw), cps)
        scale = _np.where(_np.abs(yhat) < 1e-3, _np.where(yhat > 0., 1e-3, -1e-3), yhat)
        lb, ub = self._lower_h / scale, self._upper_h / scale
        lb, ub = _np.minimum(lb, ub), _np.maximum(lb, ub)
        wlb = _np.full(nsw, -_np.inf)
        wub = _np.full(nsw, _np.inf)
        _np.maximum.at(wlb, group, lb)
        _np.minimum.at(wub, group, ub)
        wguess = _np.maximum(wlb, _np.minimum(1., wub))

        def faux(w):
            return float(f(yhat * _np.asarray(w, dtype=_np.float64)[group]))

        if hasattr(self._local, "reseed"):
            self._local.reseed(self._local_seed0 + self._nlocal)
        self._nlocal += 1
        sol = self._local.optimize(faux, wlb, wub, wguess)
        w = _np.asarray(sol.x, dtype=_np.float64)
        fev = int(self.get_state("fev")[0]) + int(sol.n_evals)
        trial = yhat * w[group]
        inside = (not self._params.correct) or bool(
            _np.all((trial >= self._lower_h) & (trial <= self._upper_h)))
        if inside:
            fwy = float(f(trial))
            fev += 1
            if fwy < float(self.get_state("fyhat")[0]):
                self.set_state("yhat", trial)
                self.set_state("fyhat", [fwy])
                self.set_state("improved", [1.])
        self.set_state("fev", [float(fev)])


class APSO(MultivariateSearch):
    """APSO(mfev, tol, np, correct=True) -- :265-269"""
    _algo = _ffi.ALGO_APSO

    def __init__(self, mfev, tol, np, correct=True, **ext):
        super().__init__(**ext)
        p = self._params
        p.mfev, p.tol, p.np, p.correct = int(mfev), float(tol), int(np), int(bool(correct))
