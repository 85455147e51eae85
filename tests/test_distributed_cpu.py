"""The N > 1 path on CPU: the concurrent BIPOP driver over a world_size-2 gloo group, with the
CPU oracle standing in for the device run (the product's collective and reduction logic is what
is under test here, not the inner optimizer)."""
import json
import math
import os
import socket
import subprocess
import sys

import numpy as np

from _dist_worker import drive

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_world1_degenerates_to_the_reference_schedule():
    """one slot per round: every cap is replaced by the real count before the next decision,
    so the regime / lambda / sigma sequence obeys bipop_cmaes.cpp:117-142,:207-214,:241-248"""
    res = drive(world=1, rank=0, mfev=40000)
    n, lamdef = 5, 4 + int(3. * math.log(5))
    large = small = nl = 0
    best, fbest = 1, math.inf
    for h in res["history"]:
        if h["regime"] == 0:
            assert h["lam"] == lamdef and h["sigma"] == 2.
        else:
            want = (1 if large <= small * 2. else 2) if best == 1 else (2 if small <= 2. * large else 1)
            assert h["regime"] == want
            if want == 1:
                assert h["lam"] == lamdef * 2 ** (nl + 1)
                assert h["sigma"] == max(2. * (1. / 1.6) ** (nl + 1), 0.02)
                large += h["used"]
                nl += 1
            else:
                assert h["maxfev"] <= large >> 1
                small += h["used"]
        if h["fx"] < fbest:
            fbest = h["fx"]
            if h["regime"]:
                best = h["regime"]
    assert res["large"] == [large, nl]
    assert float.fromhex(res["fxbest"]) == fbest
    assert res["fev"] == sum(h["used"] + 1 for h in res["history"])


def test_world1_equals_the_oracle_restart_driver_draw_for_draw():
    """W = 1 with one shared inner optimizer IS the sequential driver: same restart points, same
    (regime, lambda, sigma, evaluations, f*) per run and the same incumbent as the oracle's
    Restart (bipop_cmaes.cpp:61-267 restated, pinned to the reference) under the same Philox key"""
    import pyoracle as po
    n, seed, mfev = 5, 17, 40000
    res = drive(world=1, rank=0, mfev=mfev, n=n, seed=seed, shared=True)
    O = po.oracle()
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    o = po.bipop(O, po.cma(O, "active", 1, 1e-8, 4), mfev)
    o.set_mode(False, po.RNG_PHILOX, seed)
    o.init("rastrigin", lo, up, np.random.default_rng(seed).uniform(-5, 5, n))
    rows = [(0, int(o.scalar("last_lambda")), o.scalar("last_sigma"),
             int(o.scalar("last_inner_fev")), o.scalar("fx"))]
    while True:
        o.iterate()
        rows.append((int(o.scalar("last_regime")), int(o.scalar("last_lambda")),
                     o.scalar("last_sigma"), int(o.scalar("last_inner_fev")), o.scalar("fx")))
        if o.scalar("largerestarts") >= 9 or o.scalar("fev") >= mfev:
            break
    got = [(h["regime"], h["lam"], h["sigma"], h["used"], h["fx"]) for h in res["history"]]
    assert got == rows
    assert res["fev"] == int(o.scalar("fev"))
    assert float.fromhex(res["fxbest"]) == o.scalar("fxbest")
    np.testing.assert_array_equal([float.fromhex(v) for v in res["x"]], o.get("xbest"))


def test_gloo_world2_matches_the_serial_reduction(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "_dist_worker.py"), str(tmp_path)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=os.path.dirname(HERE))
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    assert r0 == r1                      # replicated state, no broadcast needed
    serial = drive(world=2, rank=0)      # same plan, slots run one after the other
    assert json.loads(json.dumps(serial)) == r0
    assert r0["rounds"] >= 2 and len(r0["history"]) > r0["rounds"]   # two runs per round


    # the same two ranks then ran the concurrent IPOP driver through the same group
    i0 = json.load(open(tmp_path / "ipop_rank0.json"))
    i1 = json.load(open(tmp_path / "ipop_rank1.json"))
    assert i0 == i1
    assert json.loads(json.dumps(drive(world=2, rank=0, kind="ipop", mfev=30000))) == i0
    assert i0["rounds"] >= 2 and [h["slot"] for h in i0["history"][:4]] == [0, 1, 0, 1]


def test_ipop_world1_equals_the_oracle_restart_driver_draw_for_draw():
    """ConcurrentIPop with one slot per round IS IPopCmaes (ipop_cmaes.cpp:65-189): same restart
    points, the same (lambda, sigma, evaluations, f*) per run, the same budget and incumbent as
    the oracle's restatement (pinned to the compiled reference) under the same Philox key --
    with the lambda cycling at 10 n^2 reached (n = 3: lambda_max = 90) and with it switched off"""
    import pyoracle as po
    for n, mfev, kw in ((5, 40000, {}), (3, 60000, {}), (3, 30000, dict(boundlambda=False)),
                        (4, 30000, dict(nipop=False, ksigmadec=2.5))):
        seed = 17 + n
        res = drive(world=1, rank=0, mfev=mfev, n=n, seed=seed, shared=True, kind="ipop", **kw)
        O = po.oracle()
        lo, up = -5. * np.ones(n), 5. * np.ones(n)
        o = po.ipop(O, po.cma(O, "active", 1, 1e-8, 4), mfev, **kw)
        o.set_mode(False, po.RNG_PHILOX, seed)
        o.init("rastrigin", lo, up, np.random.default_rng(seed).uniform(-5, 5, n))
        rows = [(int(o.scalar("last_lambda")), o.scalar("last_sigma"),
                 int(o.scalar("last_inner_fev")), o.scalar("fx"))]
        while o.scalar("fev") < mfev:
            o.iterate()
            rows.append((int(o.scalar("last_lambda")), o.scalar("last_sigma"),
                         int(o.scalar("last_inner_fev")), o.scalar("fx")))
        got = [(h["lam"], h["sigma"], h["used"], h["fx"]) for h in res["history"]]
        assert got == rows, (n, kw)
        assert res["fev"] == int(o.scalar("fev"))
        assert float.fromhex(res["fxbest"]) == o.scalar("fxbest")
        np.testing.assert_array_equal([float.fromhex(v) for v in res["x"]], o.get("xbest"))
        lams = [h["lam"] for h in res["history"]]
        if n == 3 and not kw:
            assert 90 in lams and lams.count(4 + int(3. * math.log(3))) >= 2     # cycled
        if kw.get("boundlambda") is False:
            assert lams == [lams[0] << k for k in range(len(lams))]


def test_ipop_rounds_run_consecutive_doublings():
    """round k of W slots runs lambda_def 2^(kW+1) ... 2^(kW+W) (run 0: lambda_def itself), the
    sigma sequence continues across rounds, and (W, S) / (W S, 1) give the same history"""
    a = drive(world=4, rank=0, kind="ipop", mfev=200000, n=6)
    b = drive(world=2, rank=0, slots=2, kind="ipop", mfev=200000, n=6)
    assert json.loads(json.dumps(a)) == json.loads(json.dumps(b))
    lamdef, lmax = 4 + int(3. * math.log(6)), 10 * 6 * 6
    lam = lamdef
    for i, h in enumerate(a["history"]):
        assert h["slot"] == i % 4 and h["round"] == i // 4
        if i > 0:                                            # ipop_cmaes.cpp:122-130
            lam <<= 1
            if lam > lmax:
                lam = lmax if lam - lmax < lmax - (lam >> 1) else lamdef
        assert h["lam"] == lam
        assert h["sigma"] == (2. if i == 0 else max(a["history"][i - 1]["sigma"] / 1.6, 0.02))
    assert [h["lam"] for h in a["history"][:5]] == [lamdef << k for k in range(5)]
    assert len(a["history"]) >= 5
    assert a["fev"] == sum(h["used"] + 1 for h in a["history"])


def test_slots_per_rank_is_the_same_plan_as_more_ranks():
    """(W ranks, S slots each) and (W S ranks, one slot each): the plan, the seeds and the
    reduction depend on the global slot only, so the histories are identical"""
    a = drive(world=4, rank=0)
    b = drive(world=2, rank=0, slots=2)
    c = drive(world=1, rank=0, slots=4)
    assert json.loads(json.dumps(a)) == json.loads(json.dumps(b)) == json.loads(json.dumps(c))
    assert max(h["slot"] for h in a["history"]) >= 2


# ---- CCPSO swarm groups sharded over ranks (bboptpy_amd.distributed.ShardedCCPSO) ---------------
def test_sharded_ccpso_serial_collective_equals_unsharded():
    """W = 1, 2, 3 ranks in one process (the serial stand-in for the all-gather), the oracle as
    the engine: state after every generation bit-identical to the unsharded optimizer's"""
    from _ccpso_worker import drive as cdrive, unsharded
    want = unsharded()
    for world in (1, 2, 3):
        assert cdrive(world=world, rank=0) == want, world


def test_sharded_ccpso_gloo_world2_equals_unsharded(tmp_path):
    from _ccpso_worker import unsharded
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "_ccpso_worker.py"), str(tmp_path)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=os.path.dirname(HERE))
    r0 = json.load(open(tmp_path / "ccpso_rank0.json"))
    r1 = json.load(open(tmp_path / "ccpso_rank1.json"))
    assert r0 == r1                      # replicated state: every rank holds the same swarm
    assert r0 == unsharded()             # and it is the unsharded optimizer's, bit for bit
