"""The CPU oracle against the golden vectors the REAL reference produced
(oracle/gen_golden.py, development container).  Bit-exact: the oracle replays the reference's
mt19937 draws and arithmetic order, so every float must be identical.  Runs everywhere (no GPU,
no /root/reference)."""
import numpy as np
import pytest

import pyoracle as po
from _golden import load, unhex


def test_mt19937_and_distributions(oracle_lib):
    g = load("rng_mt19937.json")
    O = oracle_lib
    O.seed(g["seed"])
    assert [int(O.f("draw_raw")()) for _ in g["raw"]] == g["raw"]
    O.seed(g["seed"])
    np.testing.assert_array_equal([O.f("draw_uniform")(-3., 7.) for _ in g["uniform_m3_7"]],
                                  unhex(g["uniform_m3_7"]))
    O.seed(g["seed"])
    assert [int(O.f("draw_int")(0, (k % 97) + 1)) for k in range(len(g["int_0_kmod97p1"]))] \
        == g["int_0_kmod97p1"]
    O.seed(g["seed"])
    O.f("reset_test_normal")()
    np.testing.assert_array_equal([O.f("draw_normal")() for _ in g["normal"]],
                                  unhex(g["normal"]))


def test_strategy_constants(oracle_lib):
    for rec in load("cma_constants.json"):
        n, lam = rec["n"], rec["lambda"]
        h = po.cma(oracle_lib, rec["variant"], 10 ** 6, 1e-4, lam)
        h.init("sphere", -np.ones(n), np.ones(n), np.zeros(n))
        for k, v in rec.items():
            if k in ("variant", "n", "lambda"):
                continue
            if k == "w_head":
                np.testing.assert_array_equal(h.get("weights")[:4], unhex(v))
            elif k == "w_tail":
                np.testing.assert_array_equal(h.get("weights")[-1:], unhex(v))
            else:
                assert h.scalar(k) == float.fromhex(v), (rec["variant"], n, lam, k)
        h.destroy()


@pytest.mark.parametrize("idx", range(5))
def test_cma_trajectory(oracle_lib, idx):
    rec = load("cma_runs.json")[idx]
    n, lam, box = rec["n"], rec["lambda"], rec["box"]
    oracle_lib.seed(rec["seed"])
    h = po.cma(oracle_lib, rec["variant"], rec["mfev"], rec["tol"], lam)
    h.init(rec["objective"], -box * np.ones(n), box * np.ones(n), unhex(rec["guess"]))
    states = {s["gen"]: s for s in rec["states"]}
    trace = rec.get("trace")
    head, tail = rec.get("trace_head"), rec.get("trace_tail")
    zs = unhex(rec["normals_first3"]).reshape(3, lam * n)
    gen, flag, rows = 0, 0, []
    while h.scalar("fev") < rec["mfev"]:
        h.iterate()
        if gen < 3:
            np.testing.assert_array_equal(h.get("zlast"), zs[gen])
        gen += 1
        if gen in states:
            for k, v in states[gen].items():
                if k != "gen":
                    np.testing.assert_array_equal(h.get(k), unhex(v), err_msg="gen %d %s" % (gen, k))
        D = h.get("D")
        rows.append([h.get("fit_val")[0], h.scalar("sigma"), np.linalg.norm(h.get("xmean")),
                     D[-1] / D[0]])
        flag = h.converged()
        if flag:
            break
    rows = np.array(rows)
    if trace is not None:
        np.testing.assert_array_equal(rows, np.array([unhex(r) for r in trace]))
    else:
        np.testing.assert_array_equal(rows[:60], np.array([unhex(r) for r in head]))
        np.testing.assert_array_equal(rows[-60:], np.array([unhex(r) for r in tail]))
    x, fev, conv = h.solution()
    res = rec["result"]
    assert (gen, flag, fev, conv) == (res["generations"], res["flag"], res["fev"],
                                      res["converged"])
    np.testing.assert_array_equal(x, unhex(res["x"]))


@pytest.mark.parametrize("idx", range(6))
def test_de_pso_generations(oracle_lib, idx):
    rec = load("pop_runs.json")[idx]
    n, box = rec["n"], rec["box"]
    oracle_lib.seed(rec["seed"])
    h = getattr(po, rec["algo"])(oracle_lib, **rec["params"])
    h.init(rec["objective"], -box * np.ones(n), box * np.ones(n), np.zeros(n))
    gen = 0
    for st in rec["states"]:
        while gen < st["gen"]:
            h.iterate()
            gen += 1
        for k, v in st.items():
            if k != "gen":
                np.testing.assert_array_equal(h.get(k), unhex(v),
                                              err_msg="%s gen %d %s" % (rec["algo"], gen, k))


@pytest.mark.parametrize("idx", range(6))
def test_restart_schedule(oracle_lib, idx):
    rec = load("restart_runs.json")[idx]
    n = rec["n"]
    oracle_lib.seed(rec["seed"])
    base = po.cma(oracle_lib, "active", 1, 1e-6, 4)
    h = getattr(po, rec["driver"])(oracle_lib, base, rec["mfev"])
    h.init(rec["objective"], -5. * np.ones(n), 5. * np.ones(n), unhex(rec["guess"]))
    for i, row in enumerate(rec["schedule"]):
        if i > 0:
            h.iterate()
        for k, v in row.items():
            np.testing.assert_array_equal(h.get(k), unhex(v), err_msg="restart %d %s" % (i, k))


@pytest.mark.parametrize("idx", range(4))
def test_sep_cma_trajectory(oracle_lib, idx):
    """SepCmaes (sep_cmaes.cpp:41-206): constants, first three generations' normals, states at
    generations 1, 2, 3, 10, 50, the whole (f_best, sigma, |m|, cond) trace and the result, bit
    for bit against the reference's run (tests/golden/sep_runs.json)."""
    rec = load("sep_runs.json")[idx]
    n, lam, box = rec["n"], rec["lambda"], rec["box"]
    oracle_lib.seed(rec["seed"])
    h = po.cma(oracle_lib, "sep", rec["mfev"], rec["tol"], lam, bound=rec["bound"],
               adjustlr=rec["adjustlr"])
    h.init(rec["objective"], -box * np.ones(n), box * np.ones(n), unhex(rec["guess"]))
    for k, v in rec["constants"].items():
        assert h.scalar(k) == float.fromhex(v), k
    states = {s["gen"]: s for s in rec["states"]}
    zs = unhex(rec["normals_first3"]).reshape(3, lam * n)
    gen, flag, rows = 0, 0, []
    while h.scalar("fev") < rec["mfev"]:
        h.iterate()
        if gen < 3:
            np.testing.assert_array_equal(h.get("zlast"), zs[gen])
        gen += 1
        if gen in states:
            for k, v in states[gen].items():
                if k != "gen":
                    np.testing.assert_array_equal(h.get(k), unhex(v), err_msg="gen %d %s" % (gen, k))
        D = h.get("D")
        rows.append([h.get("fit_val")[0], h.scalar("sigma"), np.linalg.norm(h.get("xmean")),
                     D.max() / D.min()])
        flag = h.converged()
        if flag:
            break
    rows = np.array(rows)
    if "trace" in rec:
        np.testing.assert_array_equal(rows, np.array([unhex(r) for r in rec["trace"]]))
    else:
        np.testing.assert_array_equal(rows[:60], np.array([unhex(r) for r in rec["trace_head"]]))
        np.testing.assert_array_equal(rows[-60:], np.array([unhex(r) for r in rec["trace_tail"]]))
    x, fev, conv = h.solution()
    res = rec["result"]
    assert (gen, flag, fev, conv) == (res["generations"], res["flag"], res["fev"],
                                      res["converged"])
    np.testing.assert_array_equal(x, unhex(res["x"]))


@pytest.mark.parametrize("idx", range(3))
def test_sansde_generations(oracle_lib, idx):
    """SaNSDESearch (sansde.cpp:58-300): swarm, per-individual CR, strategy / crossover /
    mutation adaptation state at generations 0, 1, 2, 5, 10, 25, 50, 60, bit for bit against
    the reference's run (tests/golden/sansde_runs.json)"""
    rec = load("sansde_runs.json")[idx]
    n, box = rec["n"], rec["box"]
    oracle_lib.seed(rec["seed"])
    h = po.sansde(oracle_lib, **rec["params"])
    h.init(rec["objective"], -box * np.ones(n), box * np.ones(n), np.zeros(n))
    gen = 0
    for st in rec["states"]:
        while gen < st["gen"]:
            h.iterate()
            gen += 1
        for k, v in st.items():
            if k != "gen":
                np.testing.assert_array_equal(h.get(k), unhex(v), err_msg="gen %d %s" % (gen, k))
    x, fev, conv = h.solution()
    np.testing.assert_array_equal(x, unhex(rec["result"]["x"]))
    assert (fev, conv) == (rec["result"]["fev"], rec["result"]["converged"])


@pytest.mark.parametrize("idx", range(4))
def test_cso_generations(oracle_lib, idx):
    """CSOSearch (cso.cpp:67-276), incl. the libstdc++ std::shuffle it calls and the
    birth-slot ring neighbourhood its stored pointers amount to: swarm, velocities, means and
    incumbent at generations 1, 2, 5, 20, 40, bit for bit (tests/golden/cso_runs.json)"""
    rec = load("cso_runs.json")[idx]
    n, box = rec["n"], rec["box"]
    oracle_lib.seed(rec["seed"])
    h = po.cso(oracle_lib, **rec["params"])
    h.init(rec["objective"], -box * np.ones(n), box * np.ones(n), np.zeros(n))
    np.testing.assert_array_equal(h.get("phil"), unhex(rec["phil"]))
    np.testing.assert_array_equal(h.get("phih"), unhex(rec["phih"]))
    gen = 0
    for st in rec["states"]:
        while gen < st["gen"]:
            h.iterate()
            gen += 1
        for k in rec["keys"]:
            np.testing.assert_array_equal(h.get(k), unhex(st[k]), err_msg="gen %d %s" % (gen, k))


@pytest.mark.parametrize("idx", range(3))
def test_ccpso_generations(oracle_lib, idx):
    """CCPSOSearch (ccpso.cpp:66-371, no local optimizer): random regrouping (std::shuffle),
    context-vector evaluations, personal / ring / global bests with the reference's stale-fY
    and last-particle-wins rules, Cauchy / normal resampling and the adaptive Cauchy rate, at
    generations 1, 2, 5, 20, 40, bit for bit (tests/golden/ccpso_runs.json)"""
    rec = load("ccpso_runs.json")[idx]
    n, box = rec["n"], rec["box"]
    oracle_lib.seed(rec["seed"])
    h = po.ccpso(oracle_lib, **rec["params"])
    h.init(rec["objective"], -box * np.ones(n), box * np.ones(n), np.zeros(n))
    gen = 0
    for st in rec["states"]:
        while gen < st["gen"]:
            h.iterate()
            gen += 1
        for k, v in st.items():
            if k != "gen":
                np.testing.assert_array_equal(h.get(k), unhex(v), err_msg="gen %d %s" % (gen, k))


@pytest.mark.parametrize("idx", range(3))
def test_ccpso_local_search_generations(oracle_lib, idx):
    """CCPSOSearch with its local optimizer (ccpso.cpp:116-118, 371-435): the weights of the
    swarms optimized by CMAES / ActiveCMAES every `localfreq` generations -- including the
    reference's habit of starting a search from the previous one's B and C (cmaes.cpp:53-54) --
    at generations 1, 2, 4, 10, 20, bit for bit (tests/golden/ccpso_local_runs.json)"""
    rec = load("ccpso_local_runs.json")[idx]
    n, box, lc = rec["n"], rec["box"], rec["local"]
    oracle_lib.seed(rec["seed"])
    base = po.cma(oracle_lib, lc["variant"], lc["mfev"], lc["tol"], lc["np"])
    h = po.ccpso(oracle_lib, local=base, localfreq=lc["localfreq"], **rec["params"])
    h.init(rec["objective"], -box * np.ones(n), box * np.ones(n), np.zeros(n))
    gen = 0
    for st in rec["states"]:
        while gen < st["gen"]:
            h.iterate()
            gen += 1
        for k, v in st.items():
            if k != "gen":
                np.testing.assert_array_equal(h.get(k), unhex(v), err_msg="gen %d %s" % (gen, k))
