"""worker of tests/test_distributed_cpu.py: one rank of a gloo group running ShardedCCPSO with the
CPU oracle as the engine (test infrastructure: the product's shard / gather / merge logic is
what is under test, not the optimizer arithmetic)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


class OracleCcpsoEngine:
    """the oracle's CCPSO behind the engine interface ShardedCCPSO drives"""

    def __init__(self, mfev, stol, np_, pps, seed):
        import pyoracle as po
        self.po, self.O = po, po.oracle()
        self.h = po.ccpso(self.O, mfev, stol, np_, pps)
        self.h.set_mode(True, po.RNG_PHILOX, seed)
        self._shard = (0, 1)

    def set_shard(self, rank, world):
        self._shard = (rank, world)
        self.O.f("ccpso_set_shard")(self.h.ptr, rank, world)

    def initialize(self, f, lower, upper, guess):
        self.h.init(f, lower, upper, guess)

    def phase(self, which):
        self.O.f("ccpso_phase")(self.h.ptr, which)

    def table_record(self):
        return self.O.f("ccpso_table_record")(self.h.ptr)

    def export_tables(self, out=None, device_ptr=None):
        if out is None:
            out = np.zeros(self.table_record())
        self.O.f("ccpso_export_tables")(self.h.ptr, out)
        return out

    def merge_tables(self, gathered=None, world=1, device_ptr=None):
        g = np.ascontiguousarray(gathered, dtype=np.float64).ravel()
        self.O.f("ccpso_merge_tables")(self.h.ptr, g, world)

    def get_state(self, key):
        if key == "conv":
            x = np.zeros(self.h.n)
            import ctypes as C
            fev, conv = C.c_int(), C.c_int()
            self.O.f("ccpso_solution")(self.h.ptr, x, C.byref(fev), C.byref(conv))
            return np.array([float(conv.value)])
        return self.h.get(key)


CFG = dict(n=24, np_=12, pps=[2, 4, 6], mfev=6000, stol=1e-9, seed=77, obj="rosenbrock")


def drive(world=None, rank=None, gens=12):
    from bboptpy_amd.distributed import ShardedCCPSO
    c = CFG
    lo, up = -5. * np.ones(c["n"]), 5. * np.ones(c["n"])
    d = ShardedCCPSO(c["mfev"], c["stol"], c["np_"], c["pps"], seed=c["seed"],
                     engine_factory=lambda: OracleCcpsoEngine(c["mfev"], c["stol"], c["np_"],
                                                              c["pps"], c["seed"]),
                     world_size=world, rank=rank)
    d.initialize(c["obj"], lo, up)
    trace = []
    for _ in range(gens):
        d.iterate()
        trace.append([float(d.get_state("fyhat")[0]).hex(), int(d.get_state("fev")[0]),
                      int(d.get_state("nswarm")[0])])
    return {"trace": trace, "yhat": [float(v).hex() for v in d.get_state("yhat")],
            "x": [float(v).hex() for v in d.get_state("x")]}


def unsharded(gens=12):
    import pyoracle as po
    c = CFG
    lo, up = -5. * np.ones(c["n"]), 5. * np.ones(c["n"])
    h = po.ccpso(po.oracle(), c["mfev"], c["stol"], c["np_"], c["pps"])
    h.set_mode(True, po.RNG_PHILOX, c["seed"])
    h.init(c["obj"], lo, up, np.zeros(c["n"]))
    trace = []
    for _ in range(gens):
        h.iterate()
        trace.append([float(h.scalar("fyhat")).hex(), int(h.scalar("fev")), int(h.scalar("nswarm"))])
    return {"trace": trace, "yhat": [float(v).hex() for v in h.get("yhat")],
            "x": [float(v).hex() for v in h.get("x")]}


if __name__ == "__main__":
    import torch.distributed as dist
    dist.init_process_group("gloo")
    res = drive()
    with open(os.path.join(sys.argv[1], "ccpso_rank%d.json" % dist.get_rank()), "w") as fh:
        json.dump(res, fh)
    dist.barrier()
    dist.destroy_process_group()
