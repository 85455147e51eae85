"""The RCCL code path on the hardware there is: a real process group with backend "nccl" (world
size 1 on the one-GPU box) -- ConcurrentBiPop's all_gather of the round records and ShardedCCPSO's
all_gather_into_tensor of the fitness tables run on DEVICE tensors through RCCL, and must leave
every result bit-identical to the same driver run with no group (bipop_cmaes.cpp:109-164 sharded
by restart, ccpso.cpp:241-260 sharded by swarm group); then bench.py under RANK=0 WORLD_SIZE=1,
so its nccl branch (init_process_group, barrier, max_over_ranks) executes too."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _env():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_drivers_through_a_real_rccl_group_equal_the_no_group_run(hip):
    out = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_worker.py")],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=_env())
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RCCL_RESULT ")][-1]
    res = json.loads(line[len("RCCL_RESULT "):])
    assert res["backend"] == "nccl" and res["all_reduce"] == 3.5
    for name in ("bipop6", "bipop256", "ipop6"):
        g, s = res[name]["group"], res[name]["nogroup"]
        assert g["collectives"] >= 2 and s["collectives"] == 0      # one all_gather per round
        assert g["history"] == s["history"] and g["x"] == s["x"] and g["fev"] == s["fev"]
        assert len(g["history"]) >= 2                               # at least one restart happened
    for name in ("ccpso24", "ccpso1000"):
        g, s = res[name]["group"], res[name]["nogroup"]
        assert g["collectives"] == 6 and s["collectives"] == 0      # one all_gather per generation
        assert g["trace"] == s["trace"] and g["yhat"] == s["yhat"] and g["x"] == s["x"]


def test_bench_runs_under_a_world_size_1_nccl_group(hip):
    """bench.py as torchrun would start it (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the
    environment): init_process_group("nccl"), the barriers around the timed region and
    max_over_ranks all execute; one JSON line comes out"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_PORT=str(port))
    # the DEFAULT workload (what the driver launches), small: the bounded C5 leg then runs too --
    # ConcurrentBiPop under the nccl group, its all_gather on device tensors, max_over_ranks
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--populations", "8",
                          "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-single",
                          "--no-convergence"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["roofline"] is not None
    leg = rec["bipop_scaling"]
    assert leg["n_gpus"] == 1 and leg["rounds"] >= 3
    assert leg["large_restarts"] >= 1 and leg["small_restarts"] >= 1
    assert leg["packed_8_per_gpu"]["restarts"] >= 8
