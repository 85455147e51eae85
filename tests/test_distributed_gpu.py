"""GPU: the concurrent BIPOP driver (bboptpy_amd/distributed.py)
with its inner runs on the device.

* W = 1, small n: the driver IS the sequential one -- every run's (regime, lambda, sigma,
  evaluations used) equals the oracle Restart's (bipop_cmaes.cpp:61-267 restated and pinned to
  the reference) under the same Philox key, f* to 1e-7, and equals the single-GPU C++ driver
  (bbo_restart.hip) exactly.
* W = 2 at C5's n = 256 with the serial stand-in for the collective (both slots of a round run
  on this one GPU): the replicated bookkeeping equals the plan -- budgets are the sums of the
  evaluations the runs reported, every run stayed inside the cap it was planned with, regimes
  follow the NBIPOP rule given the budgets visible when the round was planned.
"""
import math

import numpy as np
import pytest

import pyoracle as po

pytestmark = pytest.mark.gpu


def test_world1_on_device_equals_oracle_restart_and_the_cxx_driver(hip, oracle_lib):
    from bboptpy_amd.distributed import ConcurrentBiPop
    n, seed, mfev = 6, 31, 30000
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(seed).uniform(-5, 5, n)
    d = ConcurrentBiPop(mfev=mfev, tol=1e-8, seed=seed, world_size=1, rank=0)
    sol = d.optimize("rastrigin", lo, up, guess)           # a built-in by NAME
    hist = d.state.history

    o = po.bipop(oracle_lib, po.cma(oracle_lib, "active", 1, 1e-8, 4), mfev)
    o.set_mode(False, po.RNG_PHILOX, seed)
    o.init("rastrigin", lo, up, guess)
    base = hip.ActiveCMAES(mfev=1, tol=1e-8, np=4)
    c = hip.BiPopCMAES(base, mfev=mfev, seed=seed)
    c.initialize(hip.objectives.rastrigin, lo, up, guess)
    rows_o = [(0, int(o.scalar("last_lambda")), o.scalar("last_sigma"),
               int(o.scalar("last_inner_fev")), o.scalar("fx"))]
    rows_c = [(0, int(c.get_state("last_lambda")[0]), c.get_state("last_sigma")[0],
               int(c.get_state("last_inner_fev")[0]), c.get_state("fx")[0])]
    while not (o.scalar("largerestarts") >= 9 or o.scalar("fev") >= mfev):
        o.iterate()
        c.iterate()
        rows_o.append((int(o.scalar("last_regime")), int(o.scalar("last_lambda")),
                       o.scalar("last_sigma"), int(o.scalar("last_inner_fev")), o.scalar("fx")))
        rows_c.append((int(c.get_state("last_regime")[0]), int(c.get_state("last_lambda")[0]),
                       c.get_state("last_sigma")[0], int(c.get_state("last_inner_fev")[0]),
                       c.get_state("fx")[0]))
    got = [(h["regime"], h["lam"], h["sigma"], h["used"], h["fx"]) for h in hist]
    assert got == rows_c                                   # the two device drivers: identical
    assert [r[:4] for r in got] == [r[:4] for r in rows_o]  # the oracle: same schedule
    np.testing.assert_allclose([r[4] for r in got], [r[4] for r in rows_o], rtol=1e-7, atol=1e-12)
    assert sol.n_evals == int(o.scalar("fev")) == int(c.get_state("fev")[0])
    assert not sol.converged


def test_world2_n256_bookkeeping_equals_the_plan(hip):
    from bboptpy_amd.distributed import ConcurrentBiPop
    n, seed, mfev = 256, 5, 400000
    lo, up = -5.12 * np.ones(n), 5.12 * np.ones(n)
    guess = np.random.default_rng(seed).uniform(-5, 5, n)
    # tol = 1e-2 and two large runs keep the n = 256 runs (3.4 ms per generation) to seconds
    d = ConcurrentBiPop(mfev=mfev, tol=1e-2, maxlargeruns=2, seed=seed, world_size=2, rank=0)
    sol = d.optimize(hip.objectives.sphere, lo, up, guess)
    st = d.state
    lamdef = 4 + int(3. * math.log(n))
    assert lamdef == 20
    assert st.history[0]["regime"] == 0 and st.history[0]["lam"] == lamdef
    large = sum(h["used"] for h in st.history if h["regime"] == 1)
    small = sum(h["used"] for h in st.history if h["regime"] == 2)
    assert (st.largebudget, st.smallbudget) == (large, small)
    assert st.largerestarts == sum(h["regime"] == 1 for h in st.history)
    assert st.smallrestarts == sum(h["regime"] == 2 for h in st.history)
    assert st.fev == sum(h["used"] + 1 for h in st.history) == sol.n_evals
    nl = 0
    for h in st.history:
        assert 0 < h["used"] <= h["maxfev"] + h["lam"]       # a generation may overshoot the cap
        assert h["used"] % h["lam"] == 0
        if h["regime"] == 1:
            nl += 1
            assert h["lam"] == lamdef * 2 ** nl
            assert h["sigma"] == max(2. * (1. / 1.6) ** nl, 0.02)
        elif h["regime"] == 2:
            assert lamdef <= h["lam"] <= max(lamdef, lamdef * 2 ** nl // 2)
    by_round = {}
    for h in st.history:
        by_round.setdefault(h["round"], []).append(h["slot"])
    assert max(len(v) for v in by_round.values()) == 2      # two concurrent runs per round
    assert st.fxbest == min(h["fx"] for h in st.history)
    assert hip.objectives.sphere(sol.x) == pytest.approx(st.fxbest, rel=1e-9)
    assert len(st.history) >= 4 and st.largerestarts == 2
