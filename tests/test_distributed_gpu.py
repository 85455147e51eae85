"""GPU: the concurrent BIPOP driver (bboptpy_amd/distributed.py)
with its inner runs on the device.

* W = 1, small n: the driver IS the sequential one -- its whole history equals the single-GPU
  C++ driver's (bbo_restart.hip) bit for bit, and its first run and first restart equal the
  oracle Restart's (bipop_cmaes.cpp:61-267 restated and pinned to the reference) under the same
  Philox key.  (The CPU test tests/test_distributed_cpu.py holds the whole W = 1 schedule
  against the oracle Restart with the oracle as the inner optimizer.)
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


def test_world1_on_device_equals_the_cxx_driver_and_the_oracle_plan(hip, oracle_lib):
    from bboptpy_amd.distributed import ConcurrentBiPop
    n, seed, mfev = 6, 31, 30000
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(seed).uniform(-5, 5, n)
    d = ConcurrentBiPop(mfev=mfev, tol=1e-8, seed=seed, world_size=1, rank=0)
    sol = d.optimize("rastrigin", lo, up, guess)           # a built-in by NAME
    hist = d.state.history

    base = hip.ActiveCMAES(mfev=1, tol=1e-8, np=4)
    c = hip.BiPopCMAES(base, mfev=mfev, seed=seed)
    c.initialize(hip.objectives.rastrigin, lo, up, guess)
    g = lambda k: c.get_state(k)[0]
    rows_c = [(0, int(g("last_lambda")), g("last_sigma"), int(g("last_maxfev")),
               int(g("last_inner_fev")), g("fx"))]
    while not (g("largerestarts") >= 9 or g("fev") >= mfev):
        c.iterate()
        rows_c.append((int(g("last_regime")), int(g("last_lambda")), g("last_sigma"),
                       int(g("last_maxfev")), int(g("last_inner_fev")), g("fx")))
    got = [(h["regime"], h["lam"], h["sigma"], h["maxfev"], h["used"], h["fx"]) for h in hist]
    # the Python rounds (set_params / set_seed / optimize / evaluate on one engine) and the C++
    # RestartDriver are the same computation on the same device: identical, bit for bit
    assert got == rows_c
    assert sol.n_evals == int(g("fev")) and not sol.converged
    np.testing.assert_array_equal(sol.x, c.get_state("xbest"))

    # and the oracle's Restart plans the same first restart from the same first run (the
    # horizon of inner-run equality is explained in tests/test_restart_gpu.py)
    o = po.bipop(oracle_lib, po.cma(oracle_lib, "active", 1, 1e-8, 4), mfev)
    o.set_mode(False, po.RNG_PHILOX, seed)
    o.init("rastrigin", lo, up, guess)
    assert (int(o.scalar("last_lambda")), o.scalar("last_sigma"), int(o.scalar("last_maxfev")),
            int(o.scalar("last_inner_fev"))) == got[0][1:5]
    o.iterate()
    assert (int(o.scalar("last_regime")), int(o.scalar("last_lambda")), o.scalar("last_sigma"),
            int(o.scalar("last_maxfev")), int(o.scalar("last_inner_fev"))) == got[1][:5]


@pytest.mark.parametrize("n,mfev,kw", [(6, 30000, {}), (3, 40000, {}),
                                       (4, 20000, dict(nipop=False, boundlambda=False))])
def test_ipop_world1_on_device_equals_the_cxx_driver(hip, oracle_lib, n, mfev, kw):
    """ConcurrentIPop with one slot per round on the device IS the sequential IPopCMAES
    (bbo_restart.hip, ipop_cmaes.cpp:112-162): the whole history -- lambda (incl. the cycling at
    10 n^2 for n = 3), sigma, cap, evaluations used, f* of every run -- the budget and the
    incumbent, bit for bit; and the oracle's IPOP restatement plans the same first restart from
    the same first run"""
    from bboptpy_amd.distributed import ConcurrentIPop
    seed = 40 + n
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(seed).uniform(-5, 5, n)
    d = ConcurrentIPop(mfev=mfev, tol=1e-8, seed=seed, world_size=1, rank=0, **kw)
    sol = d.optimize("rastrigin", lo, up, guess)
    got = [(h["lam"], h["sigma"], h["maxfev"], h["used"], h["fx"]) for h in d.state.history]

    base = hip.ActiveCMAES(mfev=1, tol=1e-8, np=4)
    c = hip.IPopCMAES(base, mfev=mfev, seed=seed, **kw)
    c.initialize(hip.objectives.rastrigin, lo, up, guess)
    g = lambda k: c.get_state(k)[0]
    row = lambda: (int(g("last_lambda")), g("last_sigma"), int(g("last_maxfev")),
                   int(g("last_inner_fev")), g("fx"))
    rows_c = [row()]
    while g("fev") < mfev:
        c.iterate()
        rows_c.append(row())
    assert got == rows_c
    assert sol.n_evals == int(g("fev")) and not sol.converged
    np.testing.assert_array_equal(sol.x, c.get_state("xbest"))
    assert len(got) >= 3

    o = po.ipop(oracle_lib, po.cma(oracle_lib, "active", 1, 1e-8, 4), mfev, **kw)
    o.set_mode(False, po.RNG_PHILOX, seed)
    o.init("rastrigin", lo, up, guess)
    orow = lambda: (int(o.scalar("last_lambda")), o.scalar("last_sigma"),
                    int(o.scalar("last_maxfev")), int(o.scalar("last_inner_fev")))
    assert orow() == got[0][:4]
    o.iterate()
    assert orow() == got[1][:4]


def test_ipop_world2_rounds_on_device(hip):
    """W = 2 with the serial stand-in for the collective: round k runs two consecutive doublings
    side by side, the replicated budget is the sum of what the runs reported, every run stayed
    inside its cap, and (2 ranks, 1 slot) equals (1 rank, 2 concurrent slots) bit for bit"""
    from bboptpy_amd.distributed import ConcurrentIPop
    n, seed, mfev = 8, 9, 60000
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(seed).uniform(-5, 5, n)
    a = ConcurrentIPop(mfev=mfev, tol=1e-8, seed=seed, world_size=2, rank=0)
    sa = a.optimize(hip.objectives.rastrigin, lo, up, guess)
    b = ConcurrentIPop(mfev=mfev, tol=1e-8, seed=seed, world_size=1, rank=0, slots_per_rank=2)
    sb = b.optimize(hip.objectives.rastrigin, lo, up, guess)
    assert a.state.history == b.state.history and sa.n_evals == sb.n_evals
    np.testing.assert_array_equal(sa.x, sb.x)
    st = a.state
    lamdef = 4 + int(3. * math.log(n))
    assert [h["lam"] for h in st.history[:4]] == [lamdef, 2 * lamdef, 4 * lamdef, 8 * lamdef]
    assert [(h["round"], h["slot"]) for h in st.history[:4]] == [(0, 0), (0, 1), (1, 0), (1, 1)]
    assert st.fev == sum(h["used"] + 1 for h in st.history) == sa.n_evals
    for h in st.history:
        assert 0 < h["used"] <= h["maxfev"] + h["lam"] and h["used"] % h["lam"] == 0
    assert st.fxbest == min(h["fx"] for h in st.history)
    assert hip.objectives.rastrigin(sa.x) == pytest.approx(st.fxbest, rel=1e-9)


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


# ---- CCPSO swarm groups sharded over ranks ------------------------------------------------------
@pytest.mark.parametrize("n,npp,pps,obj", [(24, 12, [2, 4, 6], "rosenbrock"),
                                           (1000, 30, [2, 5, 10, 50, 100, 250], "rosenbrock"),
                                           (300, 20, [3, 10, 50], "rastrigin")])
def test_sharded_ccpso_on_device_equals_unsharded(hip, n, npp, pps, obj):
    """W = 2 and W = 3 ranks as W engines on this one GPU (the serial stand-in for the
    all-gather: export_tables / merge_tables through host memory), every rank evaluating only
    its block of swarms (ccp_eval's candidate range): after every generation the state is
    bit-identical to the unsharded CCPSO's -- who evaluates a candidate does not change its f"""
    from bboptpy_amd.distributed import ShardedCCPSO
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    gens = 8
    ref = hip.CCPSO(mfev=10 ** 8, sigmatol=1e-12, np=npp, pps=pps, seed=5)
    ref.initialize(getattr(hip.objectives, obj), lo, up, np.zeros(n))
    want = []
    for _ in range(gens):
        ref.iterate()
        want.append((ref.get_state("fyhat")[0], int(ref.get_state("fev")[0]),
                     int(ref.get_state("nswarm")[0]), ref.get_state("yhat").copy(),
                     ref.get_state("x").copy(), ref.get_state("y").copy()))
    for world in (2, 3):
        d = ShardedCCPSO(10 ** 8, 1e-12, npp, pps, seed=5, world_size=world, rank=0)
        d.initialize(getattr(hip.objectives, obj), lo, up)
        for g in range(gens):
            d.iterate()
            for e in d._engines:                      # every rank: the same replicated state
                assert e.get_state("fyhat")[0] == want[g][0]
                assert int(e.get_state("fev")[0]) == want[g][1]
                assert int(e.get_state("nswarm")[0]) == want[g][2]
                np.testing.assert_array_equal(e.get_state("yhat"), want[g][3])
                np.testing.assert_array_equal(e.get_state("x"), want[g][4])
                np.testing.assert_array_equal(e.get_state("y"), want[g][5])


def test_sharded_handle_refuses_the_unsharded_entry_points(hip):
    """once the swarm groups are sharded (world > 1) only phase(0) / merge / phase(1) is a valid
    generation: iterate / run / optimize would update from stale rows, so they report
    BBO_ERR_STATE; and a rank's record holds its own block only (ceil(max swarms / W) np rows)"""
    from bboptpy_amd import _ffi
    n, npp, pps = 60, 10, [2, 3, 5]
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    e = hip.CCPSO(mfev=10 ** 6, sigmatol=1e-12, np=npp, pps=pps, seed=8)
    e.set_shard(1, 3)
    e.initialize(hip.objectives.rosenbrock, lo, up, np.zeros(n))
    assert e.table_record() == 2 * ((n // 2 + 2) // 3) * npp
    for call in (e.iterate, lambda: e.run(3)):
        with pytest.raises(_ffi.BboError) as ei:
            call()
        assert ei.value.status == -2 and "sharded" in str(ei.value)
    e.set_shard(0, 1)                      # back to one rank: the plain entry points work again
    e.initialize(hip.objectives.rosenbrock, lo, up, np.zeros(n))
    e.iterate()
    assert int(e.get_state("fev")[0]) > 0


def test_sharded_ccpso_host_objective_splits_the_calls(hip):
    """a Python objective: each of the W ranks calls it only for its own swarms' candidates
    (plus the replicated yhat re-evaluation), and the optimum found is the unsharded one's"""
    n, npp, pps = 12, 8, [2, 3]
    lo, up = -3. * np.ones(n), 3. * np.ones(n)
    calls = []

    def f(x):
        calls.append(1)
        return float(np.sum((x - 0.25) ** 2))

    from bboptpy_amd.distributed import ShardedCCPSO
    ref = hip.CCPSO(mfev=10 ** 8, sigmatol=1e-12, np=npp, pps=pps, seed=9)
    ref.initialize(f, lo, up, np.zeros(n))
    for _ in range(5):
        ref.iterate()
    single_calls = len(calls)
    calls.clear()
    d = ShardedCCPSO(10 ** 8, 1e-12, npp, pps, seed=9, world_size=2, rank=0)
    d.initialize(f, lo, up)
    for _ in range(5):
        d.iterate()
    np.testing.assert_array_equal(d.get_state("yhat"), ref.get_state("yhat"))
    assert int(d.get_state("fev")[0]) == int(ref.get_state("fev")[0])
    # both ranks together: the candidate evaluations ONCE (split over the ranks); only the
    # initial swarm (np calls) and the yhat re-evaluations (at most one per generation) twice
    assert single_calls < len(calls) <= single_calls + npp + 5


def test_concurrent_slots_on_one_gpu_equal_the_serial_plan(hip):
    """slots_per_rank = 4: four restart populations of a round run AT THE SAME TIME on this GPU
    (four engines, four HIP streams, four host threads inside the C library) -- and produce,
    bit for bit, the history of the same four slots run one after the other (W = 4 ranks in one
    process).  Inner runs are deterministic functions of (plan, seed): concurrency must not show."""
    from bboptpy_amd.distributed import ConcurrentBiPop
    n, seed, mfev = 24, 9, 120000
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(seed).uniform(-5, 5, n)
    hist = []
    for kw in (dict(world_size=4, rank=0), dict(world_size=1, rank=0, slots_per_rank=4),
               dict(world_size=2, rank=0, slots_per_rank=2)):
        d = ConcurrentBiPop(mfev=mfev, tol=1e-8, seed=seed, **kw)
        sol = d.optimize(hip.objectives.rastrigin, lo, up, guess)
        hist.append(([tuple(sorted(h.items())) for h in d.state.history], sol.n_evals,
                     sol.x.tolist()))
    assert hist[0] == hist[1] == hist[2]
    assert max(dict(h)["slot"] for h in hist[0][0]) >= 2      # rounds really had four runs


def test_sharded_ccpso_device_pointer_exchange(hip):
    """the RCCL form of the exchange: records leave and enter the engines through DEVICE pointers
    (torch CUDA tensors standing in for the all-gather's send / receive buffers), never through
    host memory -- same merged tables, same state as the unsharded optimizer.  Runs in a child
    process that imports torch BEFORE the HIP library is loaded, the order every torch.distributed
    program has (torch bundles its own HIP runtime; loaded second it finds no device)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "_ccpso_devptr_worker.py")],
                         capture_output=True, text=True, timeout=600, cwd=os.path.dirname(here))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "DEVPTR_OK" in out.stdout
