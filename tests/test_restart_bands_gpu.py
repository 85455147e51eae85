"""GPU: the IPOP / BIPOP restart drivers tied to the REFERENCE by outcome.

tests/test_restart_gpu.py holds every decision of the device drivers against the oracle's Restart
re-synchronised to the device's bookkeeping before each restart; what it cannot see is whether the
device's inner runs after the first restart are systematically better or worse than the
reference's.  This file closes that: the oracle in its reference mode (mt19937, the reference's
own draws -- pinned bit for bit to the compiled BiPopCmaes / IPopCmaes by
tests/test_oracle_vs_reference.py) runs P independently seeded schedules, the device runs P seeds
of the same problem, and the outcome distributions must agree (bipop_cmaes.cpp:109-267,
ipop_cmaes.cpp:112-162):

  * success rate (incumbent below the target before the budget / the ninth large run ends the
    schedule) within binomial noise of two samples of P = 32 (|difference| <= 9);
  * median evaluations-to-target of the successful schedules inside the reference's
    inter-quartile band widened by 1.25;
  * median number of restarts-to-target inside that band of the reference's, widened by 1.25 and
    one restart (the counts are small integers).
"""
import numpy as np
import pytest

import pyoracle as po

pytestmark = pytest.mark.gpu

P = 32


def _walk(init, step, get, mfev, is_bipop, target):
    """one schedule: (restarts-to-target, evaluations-to-target) or None; the reference's loop
    (bipop_cmaes.cpp:170-189): iterate until nine large runs or the budget"""
    init()
    k = 0
    while True:
        if get("fxbest") < target:
            return k, int(get("fev"))
        if get("fev") >= mfev or (is_bipop and get("largerestarts") >= 9) or k >= 60:
            return None
        step()
        k += 1


@pytest.mark.parametrize("driver,n,obj,target,mfev", [
    ("bipop", 6, "rastrigin", 1e-6, 150000),
    ("bipop", 10, "rosenbrock", 1e-6, 100000),
    ("ipop", 6, "rastrigin", 1e-6, 150000),
    ("ipop", 8, "rosenbrock", 1e-6, 100000),
])
def test_restart_driver_outcome_bands_match_reference(hip, oracle_lib, driver, n, obj, target, mfev):
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    is_bipop = driver == "bipop"
    dev, ref = [], []
    for p in range(P):
        guess = np.random.default_rng(900 + p).uniform(-5, 5, n)
        # device: Philox, seed p
        base = hip.ActiveCMAES(mfev=1, tol=1e-8, np=4)
        drv = (hip.BiPopCMAES if is_bipop else hip.IPopCMAES)(base, mfev=mfev, seed=100 + p)
        dev.append(_walk(lambda: drv.initialize(getattr(hip.objectives, obj), lo, up, guess),
                         drv.iterate, lambda k: drv.get_state(k)[0], mfev, is_bipop, target))
        # reference mode: mt19937 seeded per schedule
        oracle_lib.seed(7000 + p)
        ob = po.cma(oracle_lib, "active", 1, 1e-8, 4)
        o = (po.bipop if is_bipop else po.ipop)(oracle_lib, ob, mfev)
        ref.append(_walk(lambda: o.init(obj, lo, up, guess), o.iterate, o.scalar, mfev, is_bipop,
                         target))
        o.destroy()
    d_ok, r_ok = [v for v in dev if v], [v for v in ref if v]
    what = "%s n=%d %s" % (driver, n, obj)
    assert len(r_ok) >= 8 and len(d_ok) >= 8, (what, len(d_ok), len(r_ok))
    assert abs(len(d_ok) - len(r_ok)) <= 9, (what, len(d_ok), len(r_ok))
    e1, e3 = np.percentile([e for _, e in r_ok], [25, 75])
    med_e = np.median([e for _, e in d_ok])
    assert e1 / 1.25 <= med_e <= e3 * 1.25, (what, med_e, e1, e3)
    k1, k3 = np.percentile([k for k, _ in r_ok], [25, 75])
    med_k = np.median([k for k, _ in d_ok])
    assert k1 / 1.25 - 1 <= med_k <= k3 * 1.25 + 1, (what, med_k, k1, k3)
