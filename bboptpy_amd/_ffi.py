"""ctypes binding of libbbopt_hip.so (include/bbopt_hip.h).

The library is the product: if it cannot be loaded, or no gfx950 device is visible,
everything here raises -- there is deliberately no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbbopt_hip.so")

# bbo_algo
ALGO_CMAES, ALGO_ACTIVE_CMAES, ALGO_SHADE, ALGO_JADE, ALGO_APSO, ALGO_IPOP, ALGO_BIPOP, \
    ALGO_SEP_CMAES, ALGO_SANSDE, ALGO_CSO, ALGO_CCPSO = range(11)
# bbo_objective_kind
OBJ_BUILTIN, OBJ_SCALAR_CB, OBJ_BATCH_CB = 0, 1, 2
# bbo_cma_phase
PHASE_SAMPLE_EVALUATE, PHASE_RANK, PHASE_UPDATE, PHASE_EIGEN, PHASE_HISTORY_STOP = range(5)

BUILTIN_IDS = {"sphere": 0, "rosenbrock": 1, "rastrigin": 2, "ellipsoid": 3, "ackley": 4,
               "griewank": 5, "cigar": 6, "discus": 7, "diffpow": 8, "schwefel12": 9}

SCALAR_FN = C.CFUNCTYPE(C.c_double, C.POINTER(C.c_double), C.c_int, C.c_void_p,
                        C.POINTER(C.c_int))
BATCH_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int,
                       C.POINTER(C.c_double), C.c_void_p)


class Params(C.Structure):
    _fields_ = [
        ("algo", C.c_int), ("mfev", C.c_int), ("tol", C.c_double), ("np", C.c_int),
        ("sigma0", C.c_double), ("bound", C.c_int), ("alphacov", C.c_double),
        ("eigenrate", C.c_double),
        ("archive", C.c_int), ("repaircr", C.c_int), ("pelite", C.c_double),
        ("cdamp", C.c_double), ("jade_sigma", C.c_double), ("h", C.c_int), ("npmin", C.c_int),
        ("correct", C.c_int),
        ("print", C.c_int), ("nipop", C.c_int), ("ksigmadec", C.c_double),
        ("boundlambda", C.c_int), ("maxlargeruns", C.c_int), ("kbudget", C.c_double),
        ("seed", C.c_uint64), ("device", C.c_int), ("populations", C.c_int),
        ("poll_every", C.c_int), ("adjustlr", C.c_int),
        ("crref", C.c_int), ("pupdate", C.c_int), ("crupdate", C.c_int),
        ("pcompete", C.c_int), ("ring", C.c_int), ("vmax", C.c_double),
        ("npps", C.c_int), ("pps", C.c_int * 16), ("pcauchy", C.c_double),
    ]


class Objective(C.Structure):
    _fields_ = [("kind", C.c_int), ("builtin", C.c_int), ("scalar", SCALAR_FN),
                ("batch", BATCH_FN), ("user", C.c_void_p)]


class BboError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("libbbopt_hip status %d: %s" % (status, message))
        self.status = status


_lib = None
_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = C.POINTER(C.c_int)


def lib():
    """load the HIP library once; raises if it has not been built"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "bboptpy_amd: %s is missing. Build it with `python __graft_entry__.py` (hipcc, "
            "gfx950). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.bbo_params_default.argtypes = [C.POINTER(Params), C.c_int]
    L.bbo_params_default.restype = None
    L.bbo_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_void_p)]
    L.bbo_create_restart.argtypes = [C.POINTER(Params), C.c_void_p, C.POINTER(C.c_void_p)]
    L.bbo_destroy.argtypes = [C.c_void_p]
    L.bbo_init.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, C.POINTER(Objective)]
    L.bbo_iterate.argtypes = [C.c_void_p]
    L.bbo_solution.argtypes = [C.c_void_p, _dp, _ip, _ip]
    L.bbo_solution_of.argtypes = [C.c_void_p, C.c_int, _dp, _ip, _ip]
    L.bbo_optimize.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, C.POINTER(Objective), _dp,
                               _ip, _ip]
    L.bbo_run.argtypes = [C.c_void_p, C.c_int, _ip]
    L.bbo_get.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    L.bbo_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int, _dp, C.c_int]
    L.bbo_cma_phase_run.argtypes = [C.c_void_p, C.c_int]
    L.bbo_cma_inject_normals.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.bbo_cma_set_params.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int]
    L.bbo_cma_set_seed.argtypes = [C.c_void_p, C.c_uint64]
    L.bbo_cma_evaluate.argtypes = [C.c_void_p, _dp, C.POINTER(C.c_double)]
    L.bbo_ccpso_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.bbo_ccpso_set_local.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.bbo_ccpso_phase.argtypes = [C.c_void_p, C.c_int]
    L.bbo_ccpso_table_record.argtypes = [C.c_void_p]
    L.bbo_ccpso_export_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.bbo_ccpso_merge_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.bbo_last_error.argtypes = [C.c_void_p]
    L.bbo_last_error.restype = C.c_char_p
    L.bbo_version.restype = C.c_char_p
    L.bbo_device_count.restype = C.c_int
    for name in ("bbo_create", "bbo_create_restart", "bbo_destroy", "bbo_init", "bbo_iterate",
                 "bbo_solution", "bbo_solution_of", "bbo_optimize", "bbo_run", "bbo_get",
                 "bbo_set", "bbo_cma_phase_run", "bbo_cma_inject_normals", "bbo_cma_set_params",
                 "bbo_cma_set_seed", "bbo_cma_evaluate", "bbo_ccpso_set_shard", "bbo_ccpso_set_local", "bbo_ccpso_phase",
                 "bbo_ccpso_table_record", "bbo_ccpso_export_tables", "bbo_ccpso_merge_tables"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


EXPORTED_SYMBOLS = (
    "bbo_params_default", "bbo_create", "bbo_create_restart", "bbo_destroy", "bbo_init",
    "bbo_iterate", "bbo_solution", "bbo_solution_of", "bbo_optimize", "bbo_run", "bbo_get",
    "bbo_set", "bbo_cma_phase_run", "bbo_cma_inject_normals", "bbo_cma_set_params",
    "bbo_cma_set_seed", "bbo_cma_evaluate", "bbo_ccpso_set_shard", "bbo_ccpso_set_local", "bbo_ccpso_phase",
    "bbo_ccpso_table_record", "bbo_ccpso_export_tables", "bbo_ccpso_merge_tables",
    "bbo_last_error", "bbo_version",
    "bbo_device_count",
)


def check(status, handle=None):
    if status < 0:
        msg = lib().bbo_last_error(handle)
        raise BboError(status, msg.decode() if msg else "")
    return status


def default_params(algo):
    p = Params()
    lib().bbo_params_default(C.byref(p), algo)
    return p
