"""child process of tests/test_distributed_gpu.py::test_sharded_ccpso_device_pointer_exchange"""
import os
import sys

import numpy as np
import torch          # first: torch's HIP runtime must be the one the process initialises

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.cuda.init()
import bboptpy_amd as hip   # noqa: E402

n, npp, pps, W = 60, 10, [2, 3, 5], 2
lo, up = -5. * np.ones(n), 5. * np.ones(n)
engs = []
for r in range(W):
    e = hip.CCPSO(mfev=10 ** 8, sigmatol=1e-12, np=npp, pps=pps, seed=8)
    e.set_shard(r, W)
    e.initialize(hip.objectives.rosenbrock, lo, up, np.zeros(n))
    engs.append(e)
ref = hip.CCPSO(mfev=10 ** 8, sigmatol=1e-12, np=npp, pps=pps, seed=8)
ref.initialize(hip.objectives.rosenbrock, lo, up, np.zeros(n))
reclen = engs[0].table_record()
allrec = torch.zeros(W * reclen, dtype=torch.float64, device="cuda")
for _ in range(6):
    ref.iterate()
    for e in engs:
        e.phase(0)
    for r, e in enumerate(engs):       # "all-gather": every rank's record into the big buffer
        e.export_tables(device_ptr=allrec[r * reclen:(r + 1) * reclen].data_ptr())
    torch.cuda.synchronize()
    for e in engs:
        e.merge_tables(world=W, device_ptr=allrec.data_ptr())
        e.phase(1)
    for e in engs:
        np.testing.assert_array_equal(e.get_state("yhat"), ref.get_state("yhat"))
        np.testing.assert_array_equal(e.get_state("x"), ref.get_state("x"))
        assert int(e.get_state("fev")[0]) == int(ref.get_state("fev")[0])
print("DEVPTR_OK")
