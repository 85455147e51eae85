"""GPU: the generation-SYNCHRONOUS DE / PSO of the device against the compiled reference's
ASYNCHRONOUS algorithms AT THE SIZE OF BASELINE.json's CONFIGS, by convergence curve.

tests/test_bands_gpu.py ties the two at n = 10, np = 30-60.  At np = 4096 the in-generation
propagation the reference has (in-place replacement, shade.cpp:181-183, jade.cpp:175-176; gbest
refreshed inside the particle loop, apso.cpp:194-197) is another regime, so this file holds the
device against tests/golden/pop_bands.json: K = 8 runs of oracle/_ref (the reference compiled from
/root/reference; written by oracle/gen_pop_bands.py, which also says what became of the APSO seeds
whose reference process died of its out-of-bounds read) per case, best f at fixed evaluation
checkpoints:

  shade_lpsr   L-SHADE n=128 npinit=4096 Rastrigin, 600 np evaluations, population reduction on
  shade_fixed  the same with npmin = npinit (what bench.py times), 300 np evaluations
  jade         JADE n=128 np=4096 Rosenbrock, 600 np evaluations
  apso         APSO n=512 np=1024 Sphere, 40 iterations

The device runs the same problems with populations = 8 and is held to the reference
  (a) in VALUE: at every checkpoint its median log10(best f) within a quarter of a decade of the
      reference's median (f falls by 4 to 5 decades over these runs);
  (b) in SPEED: the evaluations its median curve needs to reach the level the reference's median
      holds at 25 / 50 / 75 / 100 % of the budget lie inside the reference's own inter-quartile
      band of evaluations-to-that-level, widened by 1.25 (the widening test_bands_gpu.py uses).
"""
import numpy as np
import pytest

from _golden import load

pytestmark = pytest.mark.gpu

K = 8


def _device_curves(hip, name, c):
    n, npop = c["n"], c["np"]
    lo, up = c["box"][0] * np.ones(n), c["box"][1] * np.ones(n)
    mfev = c["mfev"]
    if c["algo"] == "shade":
        g = hip.SHADE(mfev=mfev, npinit=npop, tol=0., npmin=c["npmin"], seed=4242, populations=K,
                      poll_every=1)
    elif c["algo"] == "jade":
        g = hip.JADE(mfev=mfev, np=npop, tol=0., seed=4243, populations=K, poll_every=1)
    else:
        g = hip.APSO(mfev=mfev, tol=0., np=npop, seed=4244, populations=K, poll_every=1)
    g.initialize(getattr(hip.objectives, c["objective"]), lo, up, np.zeros((K, n)))
    return g


def _best(g, algo, p):
    if algo == "apso":
        return float(g.get_state("fbest", p)[0])
    return float(np.min(g.get_state("f", p)))


def _curves(hip, name, case):
    c = case["config"]
    cps = case["checkpoints_fev"]
    g = _device_curves(hip, name, c)
    out = np.full((K, len(cps)), np.nan)
    k = 0
    guard = 0
    while k < len(cps):
        done = g.run(1)
        guard += 1
        assert guard < 200000
        fev = int(g.get_state("fev", 0)[0])
        if fev >= cps[k] or done == 0:
            f = [_best(g, c["algo"], p) for p in range(K)]
            while k < len(cps) and (fev >= cps[k] or done == 0):
                out[:, k] = f
                k += 1
    return np.log10(np.maximum(out, 1e-300))


def _evals_to_level(curve, cps, level):
    """first checkpoint at which a (monotone) curve is at or below `level`; inf if never"""
    hit = np.nonzero(curve <= level)[0]
    return cps[hit[0]] if hit.size else np.inf


@pytest.mark.parametrize("name", ["shade_lpsr", "shade_fixed", "jade", "apso"])
def test_config_scale_convergence_bands_match_the_compiled_reference(hip, name):
    doc = load("pop_bands.json")
    case = doc["cases"][name]
    assert len(case["seeds"]) == K and all(case["equals_oracle"])
    cps = np.array(case["checkpoints_fev"], dtype=float)
    ref = np.array(case["runs_log10_best_f"])
    dev = _curves(hip, name, case)
    assert np.isfinite(dev).all()
    ref_med, dev_med = np.median(ref, axis=0), np.median(dev, axis=0)
    # (a) value, everywhere
    worst = np.abs(dev_med - ref_med).max()
    assert worst <= 0.25, (name, "median log10 f off by %.3f decades" % worst,
                           np.round(dev_med - ref_med, 3).tolist())
    # (b) speed, at the levels the reference's median holds at the quarters of the budget
    report = []
    for frac in (0.25, 0.5, 0.75, 1.0):
        k = int(round(frac * len(cps))) - 1
        level = ref_med[k]
        e_ref = np.array([_evals_to_level(np.minimum.accumulate(r), cps, level) for r in ref])
        # (half of the reference's runs are at the level by construction; a run that never gets
        # there counts as the whole budget and a checkpoint more)
        e_ref = np.where(np.isfinite(e_ref), e_ref, cps[-1] + (cps[1] - cps[0]))
        q1, q3 = np.percentile(e_ref, [25, 75])
        e_dev = _evals_to_level(np.minimum.accumulate(dev_med), cps, level)
        if not np.isfinite(e_dev):      # (counted like a reference run that ends above the level)
            e_dev = cps[-1] + (cps[1] - cps[0])
        report.append((frac, level, e_dev, q1, q3))
        assert q1 / 1.25 <= e_dev <= q3 * 1.25, (name, report)
