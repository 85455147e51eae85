"""oracle/pyoracle.py -- TEST INFRASTRUCTURE (ctypes front-end of the CPU oracle).

Loads oracle/_build/libbbo_oracle.so (the CPU restatement, prefix ``orc_``) and,
when it exists, oracle/_ref/libbbo_ref.so (the real reference compiled in the
development container, prefix ``ref_``) behind one small Python interface so the
tests can drive either through the same calls.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under bboptpy_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# (BBO_ORACLE_SO: another build of the same restatement -- `make -C oracle asan` -- for the
# sanitizer run of tests/test_oracle_asan.py)
ORACLE_SO = os.environ.get("BBO_ORACLE_SO") or os.path.join(HERE, "_build", "libbbo_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libbbo_ref.so")

OBJ = {"sphere": 0, "rosenbrock": 1, "rastrigin": 2, "ellipsoid": 3, "ackley": 4,
       "griewank": 5, "cigar": 6, "discus": 7, "diffpow": 8, "schwefel12": 9}

RNG_MT, RNG_PHILOX, RNG_INJECT = 0, 1, 2

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C")
_ip = C.POINTER(C.c_int)


def build_oracle(force=False):
    """compile the oracle restatement (g++, a few seconds)"""
    if os.environ.get("BBO_ORACLE_SO"):
        return ORACLE_SO
    if force or not os.path.exists(ORACLE_SO) or any(
            os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(ORACLE_SO)
            for f in ("bbo_oracle.cpp", "bbo_oracle_pop.inc", "objectives.h", "philox.h", "zig_table.inc")):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    return ORACLE_SO


def build_ref():
    """compile the real reference where /root/reference exists (else no-op)"""
    subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
    return REF_SO if os.path.exists(REF_SO) else None


def have_ref():
    return os.path.exists(REF_SO)


def _vec(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class _Lib:
    """one of the two libraries; `p` is its symbol prefix"""

    def __init__(self, path, p):
        self.p = p
        self.lib = C.CDLL(path)
        L = self.lib
        f = self.f
        f("seed").argtypes = [C.c_uint32]
        f("seed").restype = None
        f("draw_uniform").argtypes = [C.c_double, C.c_double]
        f("draw_uniform").restype = C.c_double
        f("draw_int").argtypes = [C.c_int, C.c_int]
        f("draw_int").restype = C.c_int
        f("draw_normal").restype = C.c_double
        f("draw_raw").restype = C.c_uint32
        f("objective").argtypes = [C.c_int, C.c_int, _dp]
        f("objective").restype = C.c_double
        f("cma_create").restype = C.c_void_p
        f("cma_create").argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_double,
                                    C.c_int, C.c_double, C.c_double]
        for alg in ("cma", "shade", "jade", "sansde", "cso", "ccpso", "apso", "bipop", "ipop"):
            if not hasattr(L, p + alg + "_init"):
                continue
            f(alg + "_init").argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp]
            f(alg + "_init").restype = None
            f(alg + "_iterate").argtypes = [C.c_void_p]
            f(alg + "_iterate").restype = None
            f(alg + "_destroy").argtypes = [C.c_void_p]
            f(alg + "_destroy").restype = None
            f(alg + "_get").argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
            f(alg + "_get").restype = C.c_int
            f(alg + "_optimize").argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp,
                                             _dp, _ip, _ip]
            f(alg + "_optimize").restype = C.c_int
            if hasattr(L, p + alg + "_solution"):
                f(alg + "_solution").argtypes = [C.c_void_p, _dp, _ip, _ip]
                f(alg + "_solution").restype = None
        f("cma_converged").argtypes = [C.c_void_p]
        f("cma_converged").restype = C.c_int
        if hasattr(L, p + "shade_create"):
            f("shade_create").restype = C.c_void_p
            f("shade_create").argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                          C.c_int, C.c_int]
            f("jade_create").restype = C.c_void_p
            f("jade_create").argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                         C.c_double, C.c_double, C.c_double]
            f("apso_create").restype = C.c_void_p
            f("apso_create").argtypes = [C.c_int, C.c_double, C.c_int, C.c_int]
            if hasattr(L, p + "ccpso_create"):
                f("ccpso_create").restype = C.c_void_p
                f("ccpso_create").argtypes = [C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int),
                                              C.c_int, C.c_int, C.c_double]
            if hasattr(L, p + "ccpso_set_local"):
                f("ccpso_set_local").restype = None
                f("ccpso_set_local").argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64]
            if hasattr(L, p + "ccpso_phase"):
                f("ccpso_set_shard").restype = None
                f("ccpso_set_shard").argtypes = [C.c_void_p, C.c_int, C.c_int]
                f("ccpso_phase").restype = None
                f("ccpso_phase").argtypes = [C.c_void_p, C.c_int]
                f("ccpso_table_record").restype = C.c_int
                f("ccpso_table_record").argtypes = [C.c_void_p]
                f("ccpso_export_tables").restype = None
                f("ccpso_export_tables").argtypes = [C.c_void_p, _dp]
                f("ccpso_merge_tables").restype = None
                f("ccpso_merge_tables").argtypes = [C.c_void_p, _dp, C.c_int]
            if hasattr(L, p + "ccpso_local_fresh"):
                f("ccpso_local_fresh").restype = None
                f("ccpso_local_fresh").argtypes = [C.c_void_p, C.c_int]
            if hasattr(L, p + "cso_create"):
                f("cso_create").restype = C.c_void_p
                f("cso_create").argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_double]
            if hasattr(L, p + "sansde_create"):
                f("sansde_create").restype = C.c_void_p
                f("sansde_create").argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                               C.c_int, C.c_int]
            f("bipop_create").restype = C.c_void_p
            f("bipop_create").argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int,
                                          C.c_double, C.c_double]
            f("ipop_create").restype = C.c_void_p
            f("ipop_create").argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int,
                                         C.c_double, C.c_int]
            f("bipop_inner_get").argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
            f("bipop_inner_get").restype = C.c_int
        if p == "ref_":
            f("cma_peek_normals").argtypes = [C.c_void_p, _dp, C.c_int]
            f("cma_peek_normals").restype = None
        else:
            f("cma_set_rng").argtypes = [C.c_void_p, C.c_int, C.c_uint64]
            f("cma_set_rng").restype = None
            f("cma_inject_z").argtypes = [C.c_void_p, _dp, C.c_int]
            f("cma_inject_z").restype = None
            f("cma_set").argtypes = [C.c_void_p, C.c_char_p, _dp, C.c_int]
            f("cma_set").restype = C.c_int
            for s in ("cma_sample", "cma_evaluate_sort", "cma_update_distribution",
                      "cma_update_history"):
                f(s).argtypes = [C.c_void_p]
                f(s).restype = None
            f("cma_eigen").argtypes = [C.c_void_p, C.c_int]
            f("cma_eigen").restype = None
            f("cma_set_params").argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int]
            f("cma_set_params").restype = None
            f("set_hypot_mode").argtypes = [C.c_int]
            f("philox").argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.POINTER(C.c_uint32)]
            f("philox").restype = None
            f("philox_normals").argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, _dp]
            f("philox_normals").restype = None
            if hasattr(L, p + "log_unit"):
                f("log_unit").argtypes = [C.c_double]
                f("log_unit").restype = C.c_double
                f("sincos_turn").argtypes = [C.c_double, C.POINTER(C.c_double),
                                             C.POINTER(C.c_double)]
                f("sincos_turn").restype = None
            if hasattr(L, p + "exp_neg"):
                f("exp_neg").argtypes = [C.c_double]
                f("exp_neg").restype = C.c_double
                f("normal_quad").argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                             C.c_uint32, _dp]
                f("normal_quad").restype = None
                f("zig_strip").argtypes = [C.c_int, C.POINTER(C.c_double),
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
                f("zig_strip").restype = C.c_int
            if hasattr(L, p + "apso_set_chunk"):
                f("apso_set_chunk").argtypes = [C.c_void_p, C.c_int]
                f("apso_set_chunk").restype = None
            if hasattr(L, p + "pop_set_mode"):
                f("pop_set_mode").argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64]
                f("pop_set_mode").restype = None
                f("restart_set_rng").argtypes = [C.c_void_p, C.c_int, C.c_uint64]
                f("restart_set_rng").restype = None
                f("restart_set").argtypes = [C.c_void_p, C.c_char_p, C.c_double]
                f("restart_set").restype = C.c_int

    def f(self, name):
        return getattr(self.lib, self.p + name)

    def seed(self, s):
        self.f("seed")(int(s) & 0xFFFFFFFF)

    def objective(self, obj, x):
        x = _vec(x)
        return self.f("objective")(OBJ.get(obj, obj), x.size, x)


class Handle:
    """an optimizer object inside one of the libraries"""

    def __init__(self, lib, alg, ptr, keep=None):
        self.lib, self.alg, self.ptr = lib, alg, ptr
        self._keep = keep
        self.n = None

    def init(self, obj, lower, upper, guess):
        lower, upper, guess = _vec(lower), _vec(upper), _vec(guess)
        self.n = lower.size
        self.lib.f(self.alg + "_init")(self.ptr, OBJ.get(obj, obj), self.n, lower, upper,
                                        guess)

    def iterate(self):
        self.lib.f(self.alg + "_iterate")(self.ptr)

    def get(self, key, cap=None):
        getter = self.lib.f(self.alg + "_get")
        cnt = getter(self.ptr, key.encode(), None, 0)
        if cnt < 0:
            raise KeyError(key)
        out = np.zeros(max(cnt, 1), dtype=np.float64)
        getter(self.ptr, key.encode(), out.ctypes.data_as(C.c_void_p), cnt)
        return out[:cnt]

    def scalar(self, key):
        return float(self.get(key)[0])

    def optimize(self, obj, lower, upper, guess):
        lower, upper, guess = _vec(lower), _vec(upper), _vec(guess)
        self.n = lower.size
        x = np.zeros(self.n)
        fev, conv = C.c_int(), C.c_int()
        self.lib.f(self.alg + "_optimize")(self.ptr, OBJ.get(obj, obj), self.n, lower,
                                            upper, guess, x, C.byref(fev), C.byref(conv))
        return x, fev.value, bool(conv.value)

    def solution(self):
        x = np.zeros(self.n)
        fev, conv = C.c_int(), C.c_int()
        self.lib.f(self.alg + "_solution")(self.ptr, x, C.byref(fev), C.byref(conv))
        return x, fev.value, bool(conv.value)

    # ---- CMA-only helpers -------------------------------------------------
    def converged(self):
        return self.lib.f("cma_converged")(self.ptr)

    def peek_normals(self, count):
        out = np.zeros(count)
        self.lib.f("cma_peek_normals")(self.ptr, out, count)
        return out

    def set_rng(self, mode, seed=0):
        self.lib.f("cma_set_rng")(self.ptr, mode, seed)

    def inject_z(self, z):
        z = _vec(z).ravel()
        self.lib.f("cma_inject_z")(self.ptr, z, z.size)

    def set(self, key, value):
        v = _vec(value).ravel()
        r = self.lib.f("cma_set")(self.ptr, key.encode(), v, v.size)
        if r < 0:
            raise KeyError("%s (%d)" % (key, r))

    def step(self, name, *args):
        self.lib.f("cma_" + name)(self.ptr, *args)

    def set_mode(self, sync, rng_mode=RNG_MT, seed=0):
        """SHADE / JADE / APSO: async (reference) or sync (device) semantics + generator;
        restart drivers: generator of the driver and of its inner CMA"""
        if self.alg in ("bipop", "ipop"):
            self.lib.f("restart_set_rng")(self.ptr, rng_mode, seed)
        else:
            kind = {"apso": 1, "sansde": 2, "cso": 3, "ccpso": 4}.get(self.alg, 0)
            self.lib.f("pop_set_mode")(self.ptr, kind, 1 if sync else 0, rng_mode, seed)

    def set_chunk(self, chunk):
        """APSO, sync mode (oracle only): particles between two refreshes of the swarm's best inside
        a generation -- what the device's `chunk` state key reports (0: the whole swarm)"""
        self.lib.f("apso_set_chunk")(self.ptr, int(chunk))

    def rset(self, key, value):
        """restart drivers (oracle only): overwrite one bookkeeping field"""
        if self.lib.f("restart_set")(self.ptr, key.encode(), float(value)) < 0:
            raise KeyError(key)

    def destroy(self):
        if self.ptr:
            self.lib.f(self.alg + "_destroy")(self.ptr)
            self.ptr = None


def cma(lib, variant, mfev, tol, np_, sigma0=2., bound=False, alphacov=2., eigenrate=0.25,
        adjustlr=True):
    """variant: 'cmaes' | 'active' | 'sep' (SepCmaes: the alphacov slot carries adjustlr)"""
    v = {"cmaes": 0, "active": 1, "sep": 2}[variant]
    if v == 2:
        alphacov = 1. if adjustlr else 0.
    return Handle(lib, "cma", lib.f("cma_create")(v, mfev, tol, np_, sigma0, int(bound),
                                                   alphacov, eigenrate))


def shade(lib, mfev, npinit, tol, archive=True, repaircr=True, h=100, npmin=4):
    return Handle(lib, "shade", lib.f("shade_create")(mfev, npinit, tol, int(archive),
                                                       int(repaircr), h, npmin))


def jade(lib, mfev, np_, tol, archive=True, repaircr=True, pelite=0.05, cdamp=0.1,
         sigma=0.07):
    return Handle(lib, "jade", lib.f("jade_create")(mfev, np_, tol, int(archive),
                                                     int(repaircr), pelite, cdamp, sigma))


def sansde(lib, mfev, np_, tol, repaircr=True, crref=5, pupdate=50, crupdate=25):
    return Handle(lib, "sansde", lib.f("sansde_create")(mfev, np_, tol, int(repaircr), crref,
                                                         pupdate, crupdate))


def cso(lib, mfev, stol, np_, pcompete=3, ring=False, correct=True, vmax=0.2):
    return Handle(lib, "cso", lib.f("cso_create")(mfev, stol, np_, pcompete, int(ring),
                                                   int(correct), vmax))


def ccpso(lib, mfev, stol, np_, pps, correct=True, pcauchy=-1., local=None, localfreq=10,
          local_seed=0, local_fresh=False):
    """`local`: a cma Handle (ownership passes to the CCPSO object); local_fresh (oracle only):
    every local search starts from B = C = I like the device's, not from the reference's leftovers"""
    arr = (C.c_int * len(pps))(*[int(v) for v in pps])
    h = Handle(lib, "ccpso", lib.f("ccpso_create")(mfev, stol, np_, arr, len(pps),
                                                    int(correct), pcauchy))
    if local is not None:
        lib.f("ccpso_set_local")(h.ptr, local.ptr, int(localfreq), int(local_seed))
        local.ptr = None
        if local_fresh:
            lib.f("ccpso_local_fresh")(h.ptr, 1)
    return h


def apso(lib, mfev, tol, np_, correct=True):
    return Handle(lib, "apso", lib.f("apso_create")(mfev, tol, np_, int(correct)))


def bipop(lib, base, mfev, sigma0=2., maxlargeruns=9, nbipop=True, ksigmadec=1.6,
          kbudget=2.):
    """takes ownership of `base` (a cma Handle)"""
    ptr = lib.f("bipop_create")(base.ptr, mfev, sigma0, maxlargeruns, int(nbipop),
                                ksigmadec, kbudget)
    base.ptr = None
    return Handle(lib, "bipop", ptr)


def ipop(lib, base, mfev, sigma0=2., nipop=True, ksigmadec=1.6, boundlambda=True):
    ptr = lib.f("ipop_create")(base.ptr, mfev, sigma0, int(nipop), ksigmadec,
                               int(boundlambda))
    base.ptr = None
    return Handle(lib, "ipop", ptr)


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = _Lib(build_oracle(), "orc_")
    return _oracle


def reference():
    """the real reference, or None when oracle/_ref was never built here"""
    global _ref
    if _ref is None and have_ref():
        _ref = _Lib(REF_SO, "ref_")
    return _ref
