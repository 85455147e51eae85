"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/bbopt_hip.h declares, the default parameters are the reference's keyword defaults,
the Python classes carry the reference's names / hierarchy / signatures, and the product
refuses to run without a GPU (no CPU fallback)."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bbopt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bbo_[a-z_]+)\s*\(", text)) - {"bbo_scalar_fn", "bbo_batch_fn"})


def test_library_exports_every_declared_symbol():
    from bboptpy_amd import _ffi
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), "libbbopt_hip.so does not export %s" % name
    assert set(names) == set(_ffi.EXPORTED_SYMBOLS)
    assert b"gfx950" in _ffi.lib().bbo_version()


def test_default_parameters_are_the_reference_defaults():
    from bboptpy_amd import _ffi
    p = _ffi.default_params(_ffi.ALGO_ACTIVE_CMAES)
    # py/multivariate_py.cpp:103-171,265-269
    assert (p.sigma0, p.bound, p.alphacov, p.eigenrate) == (2., 0, 2., 0.25)
    assert (p.archive, p.repaircr, p.pelite, p.cdamp, p.jade_sigma) == (1, 1, 0.05, 0.1, 0.07)
    assert (p.h, p.npmin, p.correct) == (100, 4, 1)
    assert (p.print, p.nipop, p.ksigmadec, p.boundlambda, p.maxlargeruns, p.kbudget) == \
        (0, 1, 1.6, 1, 9, 2.)
    assert p.populations == 1


def test_class_surface_matches_the_reference():
    import bboptpy_amd as bb

    def sig(cls):
        ps = inspect.signature(cls.__init__).parameters
        return [(k, v.default) for k, v in ps.items() if k not in ("self", "ext")]

    E = inspect.Parameter.empty
    assert sig(bb.CMAES) == [("mfev", E), ("tol", E), ("np", E), ("sigma0", 2.), ("bound", False),
                             ("eigenrate", 0.25)]
    assert sig(bb.ActiveCMAES) == [("mfev", E), ("tol", E), ("np", E), ("sigma0", 2.),
                                   ("bound", False), ("alphacov", 2.), ("eigenrate", 0.25)]
    assert sig(bb.IPopCMAES) == [("base", E), ("mfev", E), ("print", False), ("sigma0", 2.),
                                 ("nipop", True), ("ksigmadec", 1.6), ("boundlambda", True)]
    assert sig(bb.BiPopCMAES) == [("base", E), ("mfev", E), ("print", False), ("sigma0", 2.),
                                  ("maxlargeruns", 9), ("nbipop", True), ("ksigmadec", 1.6),
                                  ("kbudget", 2.)]
    assert sig(bb.JADE) == [("mfev", E), ("np", E), ("tol", E), ("archive", True),
                            ("repaircr", True), ("pelite", 0.05), ("cdamp", 0.1), ("sigma", 0.07)]
    assert sig(bb.SHADE) == [("mfev", E), ("npinit", E), ("tol", E), ("archive", True),
                             ("repaircr", True), ("h", 100), ("npmin", 4)]
    assert sig(bb.APSO) == [("mfev", E), ("tol", E), ("np", E), ("correct", True)]
    # ActiveCMAES -> CMAES -> BaseCMAES -> MultivariateSearch (multivariate_py.cpp:99-115)
    assert bb.ActiveCMAES.__mro__[:4] == (bb.ActiveCMAES, bb.CMAES, bb.BaseCMAES,
                                          bb.MultivariateSearch)
    for cls in (bb.CMAES, bb.JADE, bb.SHADE, bb.APSO, bb.IPopCMAES, bb.BiPopCMAES):
        for m in ("optimize", "initialize", "iterate", "solution"):
            assert callable(getattr(cls, m))


def test_solution_string_is_the_reference_format():
    from bboptpy_amd import MultivariateSolution
    s = MultivariateSolution(np.array([0.999989, 1.0000005, -2.5]), 6980, True)
    # multivariate.h:97-114: std::to_string per coordinate, four status lines
    assert str(s) == ("x*: 0.999989 1.000001 -2.500000 \nobjective calls: 6980\n"
                      "constraint calls: 0\nB/B constraint calls: 0\nconverged: yes")
    assert str(MultivariateSolution([0.], 1, False)).endswith("converged: no/unknown")
    x = s.x
    x[0] = 5.
    assert s.x[0] == 0.999989 and s.n_evals == 6980 and s.converged is True


def test_no_cpu_fallback():
    """without a GPU the product raises; it never routes through the oracle or NumPy"""
    import bboptpy_amd as bb
    from bboptpy_amd import _ffi
    if _ffi.lib().bbo_device_count() > 0:
        pytest.skip("a GPU is visible here")
    alg = bb.ActiveCMAES(mfev=100, tol=1e-4, np=8)
    with pytest.raises(_ffi.BboError) as ei:
        alg.optimize(bb.objectives.rosenbrock, -np.ones(4), np.ones(4), np.zeros(4))
    assert ei.value.status == -4
    for mod in ("pyoracle",):
        src = open(os.path.join(ROOT, "bboptpy_amd", "multivariate.py")).read()
        assert mod not in src


def test_builtin_objectives_agree_with_the_oracle(oracle_lib):
    """the host-side convenience formulas of bboptpy_amd.objectives (never used by the
    optimizers) describe the same functions as the device / oracle definitions"""
    import bboptpy_amd as bb
    rng = np.random.default_rng(0)
    for ob in bb.objectives.ALL:
        x = rng.uniform(-3, 3, 11)
        assert ob(x) == pytest.approx(oracle_lib.objective(ob.name, x), rel=1e-12, abs=1e-12)


def test_philox_known_answers(oracle_lib):
    """Random123 known-answer vectors for Philox4x32-10, and the host twin used by the
    concurrent BIPOP driver"""
    from bboptpy_amd.distributed import philox4x32_10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    out = (ctypes.c_uint32 * 4)()
    for ctr, key, want in kat:
        seed = key[0] | (key[1] << 32)
        oracle_lib.f("philox")(seed, *ctr, out)
        assert tuple(out) == want
        assert philox4x32_10(seed, *ctr) == want


def test_device_normal_arithmetic_is_accurate(oracle_lib):
    """The Box-Muller pieces shared bit-for-bit by oracle/philox.h and bbo_rng.hpp (hand-rolled
    ln on (0, 1] and sin/cos of 2 pi t) against long-double libm: a few ulp."""
    import ctypes
    rng = np.random.default_rng(7)
    b = rng.integers(0, 2 ** 53, 20000, dtype=np.uint64)
    us = np.concatenate([(b + 1) * 2.0 ** -53, [2.0 ** -53, 1.0, 1 - 2.0 ** -53, 0.5, 2.0 ** -30],
                         2.0 ** -rng.uniform(0, 53, 5000)])
    got = np.array([oracle_lib.f("log_unit")(float(u)) for u in us])
    ref = np.log(us.astype(np.longdouble))
    rel = np.abs((got - ref) / np.where(ref == 0, 1, ref)).astype(float)
    assert rel.max() <= 4 * 2.0 ** -53
    assert oracle_lib.f("log_unit")(1.0) == 0.0
    s, c = ctypes.c_double(), ctypes.c_double()
    ts = np.concatenate([b * 2.0 ** -53, np.arange(8) / 8.0, [1 - 2.0 ** -53]])
    S, Cc = [], []
    for t in ts:
        oracle_lib.f("sincos_turn")(float(t), ctypes.byref(s), ctypes.byref(c))
        S.append(s.value)
        Cc.append(c.value)
    ang = 8 * np.arctan(np.longdouble(1)) * ts.astype(np.longdouble)
    assert float(np.abs(np.array(S) - np.sin(ang)).max()) <= 4 * 2.0 ** -53
    assert float(np.abs(np.array(Cc) - np.cos(ang)).max()) <= 4 * 2.0 ** -53


def _zig_strips(oracle_lib):
    import ctypes
    w, k, f = ctypes.c_double(), ctypes.c_uint32(), ctypes.c_double()
    n = oracle_lib.f("zig_strip")(-1, ctypes.byref(w), ctypes.byref(k), ctypes.byref(f))
    W, K, F = np.zeros(n), np.zeros(n, dtype=np.int64), np.zeros(n + 1)
    for i in range(n + 1):
        oracle_lib.f("zig_strip")(i, ctypes.byref(w), ctypes.byref(k), ctypes.byref(f))
        if i < n:
            W[i], K[i] = w.value, k.value
        F[i] = f.value
    return n, W, K, F


def test_ziggurat_tables_are_a_ziggurat(oracle_lib):
    """The strips of the samplers' normal generator (normal_quad in bbo_rng.hpp, bbo_normal_quad
    in oracle/philox.h; tables from scripts/gen_ziggurat_table.py): decreasing right edges down
    to 0, equal areas, F = exp(-x^2 / 2) at the edges, K the largest integer position that is
    surely under the curve, and the base strip's area split between rectangle and tail."""
    from scipy.special import erfc
    n, W, K, F = _zig_strips(oracle_lib)
    assert n == 1024
    x = np.append(W * 2.0 ** 22, 0.)                 # x[0] virtual, x[1] = r, x[n] = 0
    assert (np.diff(x) < 0).all() and x[n] == 0.
    assert np.allclose(F[1:], np.exp(-0.5 * x[1:] ** 2), rtol=2e-14, atol=0)
    r = x[1]
    v = r * np.exp(-0.5 * r * r) + np.sqrt(np.pi / 2) * erfc(r / np.sqrt(2))
    assert abs(x[0] * F[1] / v - 1) < 1e-13          # x_0 = v / f(r)
    area = x[1:n] * (F[2:] - F[1:n])                 # strips 1 .. n-1
    assert np.abs(area / v - 1).max() < 1e-9
    ratio = 2.0 ** 22 * x[1:] / x[:n]
    assert (K <= ratio).all() and (K + 1 > ratio).all()
    assert K[n - 1] == 0 and (K[:n - 1] > 0).all()
    # what leaves the fast path (t odd in [1, 2^22): t >= K)
    slow = np.mean(1 - K / 2.0 ** 22)
    assert 0.004 < slow < 0.0045


def test_sampler_exp_is_accurate(oracle_lib):
    """exp(-s) of the wedge test (exp_neg / bbo_exp_neg) against long-double libm on the range
    the wedge test uses (s = x^2 / 2 <= r^2 / 2 = 8.2) and well beyond."""
    rng = np.random.default_rng(11)
    s = np.concatenate([rng.uniform(0, 8.2, 20000), rng.uniform(0, 700, 5000),
                        [0., 1e-300, 0.5 * np.log(2.), np.log(2.), 8.156, 700.]])
    got = np.array([oracle_lib.f("exp_neg")(float(v)) for v in s])
    ref = np.exp(-s.astype(np.longdouble))
    assert float(np.abs(got / ref - 1).max()) <= 4 * 2.0 ** -53
    assert oracle_lib.f("exp_neg")(0.) == 1.


def test_sampler_normals_are_standard_normal(oracle_lib):
    """2^22 normals of the samplers' generator: moments, Kolmogorov distance, tail counts
    (including the tail beyond the base strip's edge r = 4.04, which only the slow path makes) and
    a chi-square over 64 equiprobable cells."""
    from scipy import stats
    rows, n = 4096, 1024
    z = np.zeros(rows * n)
    oracle_lib.f("philox_normals")(20240611, 3, rows, n, z)
    N = z.size
    assert abs(z.mean()) < 4 / np.sqrt(N)
    assert abs(z.var() - 1) < 4 * np.sqrt(2. / N)
    assert abs(np.mean(z ** 3)) < 4 * np.sqrt(15. / N)
    assert abs(np.mean(z ** 4) - 3) < 4 * np.sqrt(96. / N)
    assert stats.kstest(z, "norm").statistic < 1.63 / np.sqrt(N)       # 1 % level
    for a in (1., 2., 3., 4., 4.0388498461095045, 4.5):
        pr = 2 * stats.norm.sf(a)
        cnt = int((np.abs(z) > a).sum())
        assert abs(cnt - N * pr) < 4.5 * np.sqrt(N * pr) + 1, (a, cnt, N * pr)
    assert (z > 4.0388498461095045).any() and (z < -4.0388498461095045).any()
    # chi-square over 64 equiprobable cells
    edges = stats.norm.ppf(np.linspace(0, 1, 65)[1:-1])
    cells = np.bincount(np.searchsorted(edges, z), minlength=64)
    chi2 = float(((cells - N / 64.) ** 2 / (N / 64.)).sum())
    assert chi2 < stats.chi2.ppf(0.999, 63), chi2


def test_philox_normals_are_standard_normal(oracle_lib):
    """Distribution check of the generator the device uses (mean, variance, KS distance) and of
    the CMA column layout (every column of a row is filled exactly once)."""
    from scipy import stats
    rows, n = 400, 128
    out = np.zeros(rows * n)
    oracle_lib.f("philox_normals")(2024, 3, rows, n, out)
    assert abs(out.mean()) < 4 / np.sqrt(out.size)
    assert abs(out.var() - 1) < 0.02
    assert stats.kstest(out, "norm").pvalue > 1e-3
    z = out.reshape(rows, n)
    assert abs(np.corrcoef(z[:, :-4].ravel(), z[:, 4:].ravel())[0, 1]) < 0.02
    # ragged n: the first n columns do not depend on how far the row extends
    out2 = np.zeros(rows * 37)
    oracle_lib.f("philox_normals")(2024, 3, rows, 37, out2)
    np.testing.assert_array_equal(out2.reshape(rows, 37), z[:, :37])
