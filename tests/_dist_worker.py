"""worker of tests/test_distributed_cpu.py: one rank of a gloo process group running the
concurrent BIPOP driver with the CPU oracle as the inner optimizer (test infrastructure)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def oracle_runner(lo, up, obj):
    import pyoracle as po
    O = po.oracle()

    def run(lam, sigma, maxfev, x0, seed):
        h = po.cma(O, "active", maxfev, 1e-8, lam, sigma0=sigma)
        h.set_rng(po.RNG_PHILOX, seed)
        x, fev, _ = h.optimize(obj, lo, up, x0)
        return x, fev, O.objective(obj, x)
    return run


def shared_oracle_runner(lo, up, obj, tol=1e-8):
    """ONE inner CMA-ES object re-parameterised before every run (setParams), as the
    reference's drivers and ConcurrentBiPop._device_run do: B and C carry over"""
    import pyoracle as po
    O = po.oracle()
    h = po.cma(O, "active", 1, tol, 4)

    def run(lam, sigma, maxfev, x0, seed):
        h.step("set_params", int(lam), float(sigma), int(maxfev))
        h.set_rng(po.RNG_PHILOX, seed)
        x, fev, _ = h.optimize(obj, lo, up, x0)
        return x, fev, O.objective(obj, x)
    return run


def drive(world=None, rank=None, mfev=60000, n=5, seed=17, shared=False, slots=1, kind="bipop",
          **kw):
    from bboptpy_amd.distributed import ConcurrentBiPop, ConcurrentIPop
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    make = shared_oracle_runner if shared else oracle_runner
    cls = ConcurrentBiPop if kind == "bipop" else ConcurrentIPop
    d = cls(mfev=mfev, seed=seed, runner=make(lo, up, "rastrigin"),
            world_size=world, rank=rank, slots_per_rank=slots, **kw)
    sol = d.optimize(None, lo, up, np.random.default_rng(seed).uniform(-5, 5, n))
    st = d.state
    return {"x": [float(v).hex() for v in sol.x], "fev": sol.n_evals, "fxbest": st.fxbest.hex(),
            "large": [st.largebudget, st.largerestarts], "small": [st.smallbudget, st.smallrestarts],
            "rounds": st.round, "history": st.history}


if __name__ == "__main__":
    import torch.distributed as dist
    dist.init_process_group("gloo")
    res = drive()
    with open(os.path.join(sys.argv[1], "rank%d.json" % dist.get_rank()), "w") as fh:
        json.dump(res, fh)
    res = drive(kind="ipop", mfev=30000)
    with open(os.path.join(sys.argv[1], "ipop_rank%d.json" % dist.get_rank()), "w") as fh:
        json.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()
