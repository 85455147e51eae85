"""worker of tests/test_rccl_gpu.py: ONE rank of a real RCCL group (backend "nccl", world size 1
-- what a one-GPU box allows) running the two multi-GPU drivers through the real collective on
device tensors, and the same drivers with no group at all; prints one JSON line.

torch is imported FIRST, before any HIP call of this process and before libbbopt_hip.so is
loaded (torch bundles its own HIP runtime), the order every torch.distributed program has."""
import json
import os
import socket
import sys

import torch                      # noqa: E402  (first on purpose)
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np               # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def bipop(n, mfev, tol, group_run, kind="bipop"):
    import bboptpy_amd as bb
    from bboptpy_amd.distributed import ConcurrentBiPop, ConcurrentIPop
    lo, up = -5.12 * np.ones(n), 5.12 * np.ones(n)
    guess = np.random.default_rng(n).uniform(-5, 5, n)
    kw = {} if group_run else dict(world_size=1, rank=0)      # explicit topology = no group
    # (no device= on the group run: the drivers take LOCAL_RANK / torch's current device there)
    if not group_run:
        kw["device"] = 0
    cls = ConcurrentBiPop if kind == "bipop" else ConcurrentIPop
    d = cls(mfev=mfev, tol=tol, seed=31, **kw)
    sol = d.optimize(bb.objectives.rastrigin, lo, up, guess)
    return {"history": [sorted((k, float(v).hex()) for k, v in h.items()) for h in d.state.history],
            "x": [float(v).hex() for v in sol.x], "fev": sol.n_evals,
            "collectives": getattr(d, "collectives", 0)}


def ccpso(n, npp, pps, group_run, gens=6):
    import bboptpy_amd as bb
    from bboptpy_amd.distributed import ShardedCCPSO
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    kw = dict(always_exchange=True) if group_run else dict(world_size=1, rank=0)
    d = ShardedCCPSO(10 ** 8, 1e-12, npp, pps, seed=5, device=0, **kw)
    d.initialize(bb.objectives.rosenbrock, lo, up)
    trace = []
    for _ in range(gens):
        d.iterate()
        trace.append([float(d.get_state("fyhat")[0]).hex(), int(d.get_state("fev")[0])])
    return {"trace": trace, "yhat": [float(v).hex() for v in d.get_state("yhat")],
            "x": [float(v).hex() for v in d.get_state("x")], "collectives": d.collectives}


def ipop(n, mfev, tol, group_run):
    return bipop(n, mfev, tol, group_run, kind="ipop")


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"
    out = {"backend": dist.get_backend()}
    # one real all_reduce first: the group works at all
    t = torch.tensor([3.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t)
    out["all_reduce"] = float(t.item())
    cases = {"bipop6": (bipop, (6, 30000, 1e-8)), "bipop256": (bipop, (256, 9000, 0.5)),
             "ipop6": (ipop, (6, 20000, 1e-8)),
             "ccpso24": (ccpso, (24, 12, [2, 4, 6])), "ccpso1000": (ccpso, (1000, 30, [2, 5, 10, 50]))}
    for name, (fn, a) in cases.items():
        out[name] = {"group": fn(*a, True), "nogroup": fn(*a, False)}
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
