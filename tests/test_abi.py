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


# the 11 optimizer classes of the path (SURVEY section 8 rows a, f) + the two abstract bases
PATH_CLASSES = ("CMAES", "ActiveCMAES", "SepCMAES", "IPopCMAES", "BiPopCMAES", "JADE", "SHADE",
                "SANSDE", "APSO", "CSO", "CCPSO")
# reference keyword -> bbo_params field where the names differ (include/bbopt_hip.h comments)
FIELD_OF = {("JADE", "sigma"): "jade_sigma", ("SHADE", "npinit"): "np", ("CSO", "stol"): "tol",
            ("CCPSO", "sigmatol"): "tol", ("BiPopCMAES", "nbipop"): "nipop"}
# keywords with no bbo_params field: objects / lists marshalled by the Python class itself
NOT_A_FIELD = {("IPopCMAES", "base"), ("BiPopCMAES", "base"), ("CCPSO", "pps"), ("CCPSO", "local"),
               ("CCPSO", "localfreq")}
# the ONE declared relaxation: the reference requires CCPSO's `npps`; here it may be omitted
# (len(pps)), every reference call still means the same
RELAXED_REQUIRED = {("CCPSO", "npps")}


def _surface():
    """tests/golden/class_surface.json: parsed from the text of py/multivariate_py.cpp by
    scripts/gen_class_surface.py (build container) -- names, order, defaults, bases"""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "class_surface.json")) as fh:
        return json.load(fh)["classes"]


def test_class_surface_fixture_is_current():
    """where the reference is present (the build container) the committed fixture is what the
    parser produces from it today"""
    src = "/root/reference/py/multivariate_py.cpp"
    if not os.path.exists(src):
        pytest.skip("the reference is not on this machine")
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "gen_class_surface", os.path.join(ROOT, "scripts", "gen_class_surface.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(src) as fh:
        assert mod.parse(fh.read()) == _surface()


def test_default_parameters_are_the_reference_defaults():
    """bbo_params_default(algo) holds, field by field, the default of every optional keyword of
    the reference's py::init for that class (py/multivariate_py.cpp:103-177,265-295)"""
    import bboptpy_amd as bb
    from bboptpy_amd import _ffi
    surf = _surface()
    checked = 0
    for name in PATH_CLASSES:
        p = _ffi.default_params(getattr(bb, name)._algo)
        assert p.algo == getattr(bb, name)._algo
        for kw in surf[name]["init"]["keywords"]:
            if kw["required"] or (name, kw["name"]) in NOT_A_FIELD:
                continue
            field = FIELD_OF.get((name, kw["name"]), kw["name"])
            got, want = getattr(p, field), kw["default"]
            assert got == (int(want) if isinstance(want, bool) else want), (name, kw["name"])
            checked += 1
    assert checked >= 40
    assert _ffi.default_params(_ffi.ALGO_ACTIVE_CMAES).populations == 1


def test_class_surface_matches_the_reference():
    """every class of the path: the reference's Python name, base class, keyword names in the
    reference's order, required / optional, and default values -- all from the fixture"""
    import bboptpy_amd as bb
    surf = _surface()
    E = inspect.Parameter.empty
    for name in PATH_CLASSES:
        cls, ref = getattr(bb, name), surf[name]
        ps = inspect.signature(cls.__init__).parameters
        mine = [(k, v.default) for k, v in ps.items() if k not in ("self", "ext")]
        want = [(kw["name"], E if kw["required"] else kw["default"]) for kw in ref["init"]["keywords"]]
        assert [k for k, _ in mine] == [k for k, _ in want], name
        for (k, got), (_, exp) in zip(mine, want):
            if (name, k) in RELAXED_REQUIRED:
                assert exp is E and got is None
                continue
            assert (got is E) == (exp is E), (name, k)
            if exp is not E:
                assert got == exp and type(got) is type(exp), (name, k, got, exp)
        # extensions are keyword-only (**ext): a positional reference call cannot hit them
        assert any(v.kind is inspect.Parameter.VAR_KEYWORD for v in ps.values()), name
        # the reference's base class is an ancestor here (helper classes may sit in between)
        assert getattr(bb, ref["base"]) in cls.__mro__[1:], name
    # the abstract bases: BaseCMAES(MultivariateSearch), no constructor in the reference
    assert surf["BaseCMAES"]["init"] is None and surf["BaseCMAES"]["base"] == "MultivariateSearch"
    assert bb.BaseCMAES.__mro__[1] is bb.MultivariateSearch
    assert bb.ActiveCMAES.__mro__[:4] == (bb.ActiveCMAES, bb.CMAES, bb.BaseCMAES,
                                          bb.MultivariateSearch)
    # MultivariateSearch: optimize / initialize (f, lower, upper, guess), iterate, solution
    for d in surf["MultivariateSearch"]["defs"]:
        fn = getattr(bb.MultivariateSearch, d["name"])
        names = [k for k in inspect.signature(fn).parameters if k != "self"]
        assert names[:len(d["keywords"])] == d["keywords"], d["name"]
        for cls in (getattr(bb, n) for n in PATH_CLASSES):
            assert callable(getattr(cls, d["name"]))
    # MultivariateSolution: __str__ and the read-only properties
    sol = surf["MultivariateSolution"]
    assert [d["name"] for d in sol["defs"]] == ["__str__"]
    for prop in sol["properties"]:
        assert isinstance(getattr(bb.MultivariateSolution, prop), property)
        assert getattr(bb.MultivariateSolution, prop).fset is None


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
