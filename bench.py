#!/usr/bin/env python3
"""bench.py -- throughput of the per-generation hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
A *step* is one generation (sample -> evaluate -> rank -> update) of every population the
GPU holds.  Metric (BASELINE.json): candidate-evaluations per second.

Workloads (BASELINE.json configs; `--workload`):
  M   ActiveCMAES n=128 lambda=4096 Rosenbrock      <- the metric config (default)
  C3  ActiveCMAES n=128 lambda=1024 Rosenbrock
  C2  L-SHADE     n=128 np=4096    Rastrigin
  C4  APSO        n=512 np=65536   Sphere
  SEP SepCMAES    n=1024 lambda=4096 Ellipsoid       (the HBM-bound CMA variant)
  C5  BIPOP-CMA-ES n=256 Rastrigin: world_size concurrent restart populations, one per GPU,
      one RCCL all-gather of per-restart bests per round (bboptpy_amd.distributed); a "step" is
      one restart ROUND and --steps bounds the evaluation budget (steps * 25 000 per rank).
      Every default (M) line also carries a bounded C5 leg as `bipop_scaling`, so the driver's
      N = 1, 2, 4, 8 runs report the BIPOP multi-restart scaling north_star asks for -- with one
      restart population per GPU (the configuration as worded) and, as `packed_8_per_gpu`, with
      eight of them sharing every GPU (`--slots`).
`--populations P` independent populations of that exact shape are advanced in lockstep on
each GPU (population p uses Philox sub-stream p).  P = 1 is the strict single-run reading of
the config; the JSON line always carries BOTH the aggregate over P (`value`) and a
single-population measurement (`single_population`).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank runs the same
shape with different seeds, no data-path collective (weak scaling); the time is the MAX over
ranks between two barriers.  `python bench.py --gpus N` without a torchrun environment starts
the N ranks itself (a child `python -m torch.distributed.run`, before this process touches the
GPU) and passes their one JSON line through.

The timed region runs with the per-kernel HIP-event timers OFF and one host poll at its end
(poll_every = steps); the per-kernel shares and the roofline come from a second pass of the
same length with the timers on.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

FP64_PEAK_TFLOPS = 78.6     # MI355X fp64 vector = fp64 matrix peak (AMD datasheet; the guide's
                            # table stops at fp32: 157.3 TF, fp64 is half of it)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    "M": dict(algo="ActiveCMAES", n=128, np=4096, objective="rosenbrock", box=(-10., 10.), P=256),
    "C1": dict(algo="ActiveCMAES", n=10, np=20, objective="rosenbrock", box=(-10., 10.), P=4096),
    "C3": dict(algo="ActiveCMAES", n=128, np=1024, objective="rosenbrock", box=(-10., 10.), P=256),
    "C2": dict(algo="SHADE", n=128, np=4096, objective="rastrigin", box=(-5.12, 5.12), P=256),
    "JADE": dict(algo="JADE", n=128, np=4096, objective="rosenbrock", box=(-10., 10.), P=256),
    "SEP": dict(algo="SepCMAES", n=1024, np=4096, objective="ellipsoid", box=(-5., 5.), P=64),
    # the building block of C5 (what one restart population runs): profiling only
    "C5I": dict(algo="ActiveCMAES", n=256, np=20, objective="rastrigin", box=(-5.12, 5.12), P=1),
    "SANSDE": dict(algo="SANSDE", n=128, np=4096, objective="rosenbrock", box=(-10., 10.), P=256),
    "CSO": dict(algo="CSO", n=512, np=65536, objective="sphere", box=(-10., 10.), P=4),
    "CCPSO": dict(algo="CCPSO", n=1000, np=30, objective="rosenbrock", box=(-10., 10.), P=16,
                  pps=[2, 5, 10, 50, 100, 250]),
    "C4": dict(algo="APSO", n=512, np=65536, objective="sphere", box=(-10., 10.), P=1),
    "C4s": dict(algo="APSO", n=512, np=4096, objective="sphere", box=(-10., 10.), P=8),
}

CMA_KERNELS = ["cma_sample_eval", "cma_rank", "cma_whiten", "cma_gram", "cma_paths", "cma_cov",
               "cma_eigen", "cma_post", "cma_history_stop"]
DE_KERNELS = ["de_generation", "de_bookkeep", "de_archive_copy", "de_rank", "de_finish",
              "de_select"]
PSO_KERNELS = ["pso_center", "pso_ese", "pso_control", "pso_update", "pso_finish"]
CSO_KERNELS = ["cso_mean", "cso_shuffle", "cso_groups", "cso_compete", "cso_finish"]
CCPSO_KERNELS = ["ccp_regroup", "ccp_eval", "ccp_update", "ccp_position", "ccp_finish"]


def cso_kernel_costs(n, np_, P, pc=3):
    """CSO (bbo_cso_kernels.hpp): streaming passes; the learning step moves np (pc-1)/pc losers"""
    losers = np_ * (pc - 1) // pc
    # n <= 512 (the benchmark's CSO): the swarm mean comes from the column sums cso_compete leaves
    # per workgroup (4 groups each at 256 < n <= 512), not from another pass over the swarm
    ld = (n + 15) // 16 * 16
    lanes = 16 if ld <= 128 else 32 if ld <= 256 else 64
    nwg = -(-(np_ // pc) // (256 // lanes))
    return {
        "cso_mean": ("hbm", P * (nwg if ld <= 512 else np_) * 8 * n),
        "cso_shuffle": ("hbm", P * np_ * 24),
        "cso_groups": ("hbm", P * (np_ * 16 + (np_ // pc) * 8 * n)),
        "cso_compete": ("hbm", P * losers * (48 * n + 8)),
        "cso_finish": ("hbm", P * np_ * 20),
    }


def de_kernel_costs(n, np_, P):
    """40n+16 B per individual for the fused generation"""
    return {
        "de_generation": ("hbm", P * np_ * (40 * n + 16)),
        "de_bookkeep": ("hbm", P * np_ * 32),
        "de_archive_copy": ("hbm", None),        # 16n B per success: data dependent
        "de_rank": ("hbm", P * np_ * 16),
        "de_finish": ("hbm", P * np_ * 12),
        "de_select": ("hbm", P * np_ * 16 * n),
    }


def pso_chunk(np_):
    """particles between two refreshes of the swarm's best inside a generation (bbo_pso.hip: up to
    16 chunks of at least 64 particles, multiples of 16; 8 beyond 32768 particles) -- one pso_update LAUNCH moves one chunk"""
    nchunks = 8 if np_ > 32768 else min(16, (np_ + 63) // 64)
    return np_ if nchunks <= 1 else ((np_ + nchunks - 1) // nchunks + 15) // 16 * 16


def pso_kernel_costs(n, np_, P):
    return {
        "pso_center": ("hbm", P * np_ * 16 * n),
        "pso_ese": ("mfma", P * np_ * n * np_),          # d_ij = d_ji: half of the 2 n np^2 products
        "pso_control": ("hbm", P * np_ * 16),
        "pso_update": ("hbm", P * pso_chunk(np_) * (40 * n + 16)),     # per launch: one chunk
        "pso_finish": ("hbm", None),     # the refreshes of the best between chunks + the closing kernel: latency
    }


def cma_kernel_costs(n, lam, P):
    """algorithmic work of ONE launch (all P populations), SURVEY.md section 8d:
    flops for the MFMA-bound kernels, bytes for the bandwidth-bound ones"""
    mu = lam // 2
    # n <= 256, unbounded: the sampler hands down ||z||^2 and the whitening GEMM (mu 2n^2
    # flops) is not executed at all -- the kernel gathers mu norms through the ranking
    # (DESIGN.md "whitened norms"); beyond that it is the GEMM
    whiten = ("hbm", P * mu * 12) if n <= 256 else ("mfma", P * mu * 2 * n * n)
    return {
        "cma_sample_eval": ("mfma", P * lam * (2 * n * n + 8 * n)),
        "cma_whiten": whiten,
        "cma_gram": ("mfma", P * lam * n * (n + 1)),      # lower triangle only, like the reference
        "cma_eigen": ("mfma", P * 9 * n ** 3),
        # 16 < n <= 256 without a box (every benchmark shape but C1): C^-1/2 is not formed per
        # generation; n <= 128: the eigensolver packs B D itself, cma_post is not launched and
        # the timer brackets nothing; above, cma_post only packs
        "cma_post": ("mfma", None if 16 < n <= 256 else P * 2 * n ** 3),
        "cma_rank": ("hbm", P * lam * 16),
        "cma_paths": ("hbm", P * 8 * (n * n + 8 * n)),
        "cma_cov": ("hbm", P * 8 * 2 * (n * (n + 1) // 2)),
        "cma_history_stop": ("hbm", P * 8 * 4 * n),
    }


def measured_traffic(workload, P, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by scripts/collect_traffic.py from separate
    --pmc FETCH_SIZE / --pmc WRITE_SIZE runs; FETCH_SIZE doubled as the gfx950 note in
    MI355X_MICROARCH.md prescribes).  None when no pass matches this workload and P."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh)
    except (OSError, ValueError):
        return None
    e = t.get("%s:P%d" % (workload, P))
    if not e:
        return None
    # the timer slot is named after the phase; the launched kernel may be a shape-specific
    # variant of it (cma_sample_eval -> cma_sample_eval128, pso_ese -> pso_ese_sym + _finish)
    alias = {"SEP": {"cma_sample_eval": "sep_sample", "cma_gram": "sep_moments",
                     "cma_paths": "sep_paths"},
             "SANSDE": {"de_generation": "sansde_generation", "de_bookkeep": "sansde_bookkeep"},
             "CSO": {"cso_mean": "cso_colsum"}}.get(workload, {})
    kernel = alias.get(kernel, kernel)
    hits = [v.get("hbm_bytes_per_launch") for k, v in e.get("kernels", {}).items()
            if k == kernel or k.startswith(kernel)]
    hits = [h for h in hits if h is not None]
    return float(sum(hits)) if hits else None


def sep_kernel_costs(n, lam, P):
    """SepCMAES: every kernel is a streaming pass (bbo_sep_kernels.hpp)"""
    mu = lam // 2
    return {
        "cma_sample_eval": ("hbm", P * lam * (8 * n + 8)),     # X written once, f
        "cma_rank": ("hbm", P * lam * 16),
        "cma_gram": ("hbm", P * mu * 8 * n),                  # the selected rows, read once
        "cma_paths": ("hbm", P * 8 * 12 * n),
        "cma_history_stop": ("hbm", P * 8 * 4 * n),
    }


def make_optimizer(bb, wl, P, seed, device, poll=None):
    huge = 2 ** 31 - 1
    a = wl["algo"]
    ext = dict(seed=seed, device=device, populations=P, poll_every=poll)
    if a == "SepCMAES":
        return bb.SepCMAES(mfev=huge, tol=0., np=wl["np"], **ext)
    if a == "ActiveCMAES":
        # tol = 0: TolHistFun / TolX can never fire inside the timed region
        return bb.ActiveCMAES(mfev=huge, tol=0., np=wl["np"], **ext)
    if a == "SHADE":
        # npmin = npinit: population-size reduction off, steady-state throughput
        return bb.SHADE(mfev=huge, npinit=wl["np"], tol=0., npmin=wl["np"], **ext)
    if a == "CCPSO":
        return bb.CCPSO(mfev=huge, sigmatol=0., np=wl["np"], pps=wl["pps"], **ext)
    if a == "CSO":
        return bb.CSO(mfev=huge, stol=0., np=wl["np"], **ext)
    if a == "SANSDE":
        return bb.SANSDE(mfev=huge, np=wl["np"], tol=0., **ext)
    if a == "JADE":
        return bb.JADE(mfev=huge, np=wl["np"], tol=0., **ext)
    if a == "APSO":
        return bb.APSO(mfev=huge, tol=0., np=wl["np"], **ext)
    raise ValueError(a)


def measure(bb, wl, P, steps, warmup, seed, device, profile, barrier=None):
    """W untimed generations, then exactly `steps` timed ones (HIP-event timers off, one host
    poll at the end); with profile=True a second pass of `steps` generations with the per-kernel
    timers on follows and its report is returned"""
    n = wl["n"]
    lo = wl["box"][0] * np.ones(n)
    up = wl["box"][1] * np.ones(n)
    guess = np.random.default_rng(seed).uniform(wl["box"][0], wl["box"][1], (P, n))
    alg = make_optimizer(bb, wl, P, seed, device, poll=max(steps, 1))
    alg.initialize(getattr(bb.objectives, wl["objective"]), lo, up, guess)
    if warmup > 0:
        assert alg.run(warmup) == warmup
    count_all = wl["algo"] == "CCPSO"    # every population draws its own swarm size there
    pops = range(P) if count_all else (0,)
    fev0 = sum(alg.get_state("fev", p)[0] for p in pops)
    if barrier:
        barrier()
    t0 = time.perf_counter()
    done = alg.run(steps)            # returns after the stream has drained (hipStreamSynchronize)
    dt = time.perf_counter() - t0
    if barrier:
        barrier()
    assert done == steps, "a population stopped inside the timed region (%d of %d)" % (done,
                                                                                       steps)
    # objective evaluations of ONE population inside the timed region (np per generation for
    # CMA / DE; CSO evaluates its losers only, APSO adds its elitist-learning probes)
    fev = sum(alg.get_state("fev", p)[0] for p in pops) - fev0
    if count_all:
        fev /= P            # (the caller multiplies by P again)
    prof = None
    if profile:
        alg.set_state("profile", [1.0])
        assert alg.run(steps) == steps
        prof = alg.get_state("profile")
        alg.set_state("profile", [0.0])
    return dt, prof, fev, alg


# generations-to-tol of the reference itself (measured on its C++):
# the second half of the metric.  flag 5 = TolUpSigma (cmaes.cpp:193), flag 2 = TolHistFun.
REFERENCE_GENERATIONS_TO_TOL = {
    "M": {"generations": 2047, "flag": 5, "final_f": 84.4, "seed": 1},
    "C3": {"generations": 6228, "flag": 2, "final_f": 1.21e-4, "seed": 1},
}


def generations_to_tol(bb, wl, device, pops=8, cap=20000):
    """Run the SAME stopping rule as the reference (tol = 1e-4, all of cmaes.cpp:151-227) on
    `pops` independently seeded populations until each one stops on its own; report when."""
    n = wl["n"]
    lo, up = wl["box"][0] * np.ones(n), wl["box"][1] * np.ones(n)
    guess = np.random.default_rng(11).uniform(wl["box"][0], wl["box"][1], (pops, n))
    alg = bb.ActiveCMAES(mfev=2 ** 31 - 1, tol=1e-4, np=wl["np"], seed=11, device=device,
                         populations=pops)
    alg.initialize(getattr(bb.objectives, wl["objective"]), lo, up, guess)
    t0 = time.perf_counter()
    launched = alg.run(cap)
    dt = time.perf_counter() - t0
    its = [int(alg.get_state("it", p)[0]) for p in range(pops)]
    flags = [int(alg.get_state("flag", p)[0]) for p in range(pops)]
    fbest = [float(alg.get_state("fit_val", p)[0]) for p in range(pops)]
    return {"tol": 1e-4, "populations": pops, "generations": its, "stop_flag": flags,
            "best_f_at_stop": fbest, "generations_median": float(np.median(its)),
            "all_stopped": bool(launched < cap), "wall_s": dt}


def generations_to_ftarget(bb, wl, device, ftarget=1e-4, pops=8, cap=30000):
    """NON-REFERENCE metric (BASELINE.md section 2 asks for it next to the reference's own stop):
    generations until the best f of a generation is <= 1e-4, with the TolUpSigma test
    (cmaes.cpp:193) -- which ends the reference's lambda = 4096 run at f = 84 -- switched off
    (bbo_set "stop_off") and tol = 0 so TolHistFun / TolX stay quiet; all other tests as in the
    reference."""
    n = wl["n"]
    lo, up = wl["box"][0] * np.ones(n), wl["box"][1] * np.ones(n)
    guess = np.random.default_rng(11).uniform(wl["box"][0], wl["box"][1], (pops, n))
    alg = bb.ActiveCMAES(mfev=2 ** 31 - 1, tol=0., np=wl["np"], seed=11, device=device,
                         populations=pops, poll_every=64)
    alg.initialize(getattr(bb.objectives, wl["objective"]), lo, up, guess)
    alg.set_state("stop_off", [float(1 << 5)])
    alg.set_state("ftarget", [ftarget])
    t0 = time.perf_counter()
    launched = alg.run(cap)
    dt = time.perf_counter() - t0
    its = [int(alg.get_state("it", p)[0]) for p in range(pops)]
    flags = [int(alg.get_state("flag", p)[0]) for p in range(pops)]
    fbest = [float(alg.get_state("fit_val", p)[0]) for p in range(pops)]
    return {"label": "non-reference: TolUpSigma disabled, stop at best f <= ftarget (flag 10)",
            "ftarget": ftarget, "populations": pops, "generations": its, "stop_flag": flags,
            "best_f_at_stop": fbest, "generations_median": float(np.median(its)),
            "all_stopped": bool(launched < cap), "wall_s": dt}


def _cpu_handle(po, lib, wl, seed):
    """one optimizer of the compiled reference (or of the oracle) for workload `wl`; returns
    (handle, the np actually used, a note when it had to be reduced)"""
    n, lam = wl["n"], wl["np"]
    a = wl["algo"]
    note = ""
    if a == "APSO" and lam > 1024:
        # the reference's APSO generation is O(np^2 n): one generation at np = 65536 takes
        # about half an hour on one core; sample np = 1024 and say so
        note = " (np reduced from %d: the reference needs ~%.0f s per generation there)" % (
            lam, 0.45 * (lam / 1024.) ** 2)
        lam = 1024
    lo, up = wl["box"][0] * np.ones(n), wl["box"][1] * np.ones(n)
    guess = np.random.default_rng(seed).uniform(wl["box"][0], wl["box"][1], n)
    lib.seed(seed)
    if a == "ActiveCMAES":
        h = po.cma(lib, "active", 2 ** 31 - 1, 0., lam)
    elif a == "SepCMAES":
        h = po.cma(lib, "sep", 2 ** 31 - 1, 0., lam)
    elif a == "SHADE":
        h = po.shade(lib, 2 ** 31 - 1, lam, 0., npmin=lam)
    elif a == "JADE":
        h = po.jade(lib, 2 ** 31 - 1, lam, 0.)
    elif a == "SANSDE":
        h = po.sansde(lib, 2 ** 31 - 1, lam, 0.)
    elif a == "CSO":
        h = po.cso(lib, 2 ** 31 - 1, 0., lam)
    elif a == "CCPSO":
        h = po.ccpso(lib, 2 ** 31 - 1, 0., lam, wl["pps"])
    else:
        h = po.apso(lib, 2 ** 31 - 1, 0., lam)
    h.init(wl["objective"], lo, up, guess)
    return h, lam, note


def _cpu_run(wl, seed, budget_s, max_gens=200):
    """generations of the CPU implementation for `budget_s` seconds on THIS process's core"""
    import pyoracle as po
    lib = po.reference()
    kind = "reference" if lib is not None else "port"
    if lib is None:
        lib = po.oracle()
    h, lam, note = _cpu_handle(po, lib, wl, seed)
    fev0 = h.scalar("fev")
    gens = 0
    # the reference's CCPSO prints _fyhat to stdout every generation (ccpso.cpp:121): keep it
    # out of this script's one-line JSON
    saved_fd = None
    if wl["algo"] == "CCPSO":
        sys.stdout.flush()
        saved_fd = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        os.close(devnull)
    try:
        t0 = time.perf_counter()
        while True:
            h.iterate()
            gens += 1
            dt = time.perf_counter() - t0
            if dt >= budget_s or gens >= max_gens:
                break
    finally:
        if saved_fd is not None:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    return {"evals": h.scalar("fev") - fev0, "dt": dt, "gens": gens, "kind": kind, "np": lam,
            "note": note}


def host_core_count():
    """(cores of the host, cores THIS process may use): the scheduler affinity mask, cut down to
    the cgroup's CPU quota where one is set -- a GPU box hands each job a share of its host"""
    total = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = total
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return total, usable


def cpu_replica_main(spec):
    """child of cpu_baseline(): `bench.py --cpu-replica WORKLOAD:SEED:SECONDS` -- one independent
    replica in a process of its own (the reference's generator is one global, non-thread-safe
    object, /root/reference/src/random.hpp:168, so replicas cannot be threads); no GPU touched"""
    key, seed, budget = spec.split(":")
    r = _cpu_run(WORKLOADS[key], int(seed), float(budget), max_gens=10 ** 9)
    print(json.dumps(r))


def cpu_all_cores(key, replicas, budget_s=8.0):
    """N independent replicas of the CPU implementation, one PROCESS each, started together:
    aggregate evaluations / the slowest replica's wall time"""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-replica",
                               "%s:%d:%g" % (key, 100 + r, budget_s)], stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, env=env) for r in range(replicas)]
    outs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=budget_s * 4 + 60)
            lines = [ln for ln in o.decode().splitlines() if ln.startswith("{")]
            if pr.returncode == 0 and lines:
                outs.append(json.loads(lines[-1]))
        except subprocess.TimeoutExpired:
            pr.kill()
    if not outs:
        return None
    wall = max(o["dt"] for o in outs)
    return {"value": sum(o["evals"] for o in outs) / wall, "unit": "candidate-evals/s",
            "replicas": len(outs), "kind": outs[0]["kind"],
            "sample": "%d independent replicas, one process each (the reference's generator is a "
                      "process-wide global), %.1f s each" % (len(outs), wall)}


def cpu_baseline(wl, budget_s=12.0, key=None):
    """the same workload on the host cores of this box: the real reference when oracle/_ref
    travelled here, else the oracle restatement (a port).  `value` is ONE core -- the reference is
    single-threaded -- and `all_cores` the aggregate of one independent replica per usable core."""
    r = _cpu_run(wl, 1, budget_s)
    total, usable = host_core_count()
    out = {"value": r["evals"] / r["dt"], "unit": "candidate-evals/s", "cores": 1,
           "kind": r["kind"], "host_cores": total, "usable_cores": usable,
           "sample": "%d generations of %s n=%d np=%d %s, 1 thread, %.1f s%s" % (
               r["gens"], wl["algo"], wl["n"], r["np"], wl["objective"], r["dt"], r["note"])}
    if key is not None and usable > 1:
        out["all_cores"] = cpu_all_cores(key, min(usable, 64))
    return out


def max_over_ranks(dt):
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# The bounded C5 leg of every default line.  Sized so that ONE GPU shows the BIPOP schedule, not
# just its first run: inner tol 0.5 ends a run when sigma has about halved (TolX, cmaes.cpp:180-191)
# -- the first default run (lambda = 20) stops after ~1.2 s -- and 40000 evaluations per GPU then
# cover the first run, a large-regime restart and several small-regime ones (measured on one
# MI355X: 6 rounds, 1 large + 4 small restarts, 5.2 s).  bipop_cmaes.cpp:117-142,204-267.
C5_LEG_TOL = 0.5
C5_LEG_BUDGET = 40000


def bipop_leg(bb, world, rank, local_rank, use_dist, budget_per_rank, barrier=None, slots=1,
              tol=1e-8):
    """concurrent BIPOP-CMA-ES n = 256 Rastrigin over `world` ranks (C5): rounds of `slots`
    restarts per GPU (1 = the configuration as BASELINE.json words it), one RCCL all-gather of
    slots * (n + 7) doubles per round.  Returns on every rank (evaluations of all ranks, seconds =
    max over ranks, driver state)."""
    from bboptpy_amd.distributed import ConcurrentBiPop
    n = 256
    lo, up = -5.12 * np.ones(n), 5.12 * np.ones(n)
    guess = np.random.default_rng(7).uniform(-5.12, 5.12, n)
    budget = budget_per_rank * world * slots
    drv = ConcurrentBiPop(mfev=budget, tol=tol, sigma0=2., seed=2024, device=local_rank,
                          variant="active", slots_per_rank=slots)
    if barrier:
        barrier()
    t0 = time.perf_counter()
    drv.optimize(bb.objectives.rastrigin, lo, up, guess)
    dt = time.perf_counter() - t0
    if barrier:
        barrier()
    if use_dist:
        dt = max_over_ranks(dt)
    # the engines' own word on the multi-workgroup reduction (bbo_eig_mw.hpp): a wavefront that gave
    # up waiting for its partners sends its engine back to the one-workgroup kernel
    try:
        algs = list(getattr(drv, "_algs", {}).values())
        drv.state.spread_reduction = {
            "engines": len(algs),
            "gave_up": int(sum(int(a.get_state("eig_mw_fail")[0]) for a in algs)),
            "switched_off": int(sum(int(a.get_state("eig_mw_off")[0]) for a in algs))}
    except Exception:
        drv.state.spread_reduction = None
    return drv.state, dt, budget


def c5_inner_profile(bb, device, lam=20, gens=150):
    """per-kernel device time of the C5 building block (ActiveCMAES n = 256, lambda = lambda_def)"""
    n = 256
    wl = dict(algo="ActiveCMAES", n=n, np=lam, objective="rastrigin", box=(-5.12, 5.12))
    _, prof, _, _ = measure(bb, wl, 1, gens, 10, 5, device, profile=True)
    return wl, prof


def c5_cpu_baseline(budget_evals=3000, tol=1e-8):
    """the reference's BiPopCmaes (or the oracle's restatement of it) on one host core at n = 256
    Rastrigin, bounded to `budget_evals` evaluations (with tol = 1e-8 that is part of its first
    default-lambda run; with the C5 leg's tol the schedule gets as far as the budget lets it)"""
    import pyoracle as po
    lib = po.reference()
    kind = "reference" if lib is not None else "port"
    if lib is None:
        lib = po.oracle()
    n = 256
    lo, up = -5.12 * np.ones(n), 5.12 * np.ones(n)
    guess = np.random.default_rng(7).uniform(-5.12, 5.12, n)
    lib.seed(1)
    h = po.bipop(lib, po.cma(lib, "active", 1, tol, 4), budget_evals)
    t0 = time.perf_counter()
    h.init("rastrigin", lo, up, guess)
    restarts = 0
    while h.scalar("fev") < budget_evals and h.scalar("largerestarts") < 9:
        h.iterate()                                   # bipop_cmaes.cpp:170-189
        restarts += 1
    dt = time.perf_counter() - t0
    fev = h.scalar("fev")
    return {"value": fev / dt, "unit": "candidate-evals/s", "cores": 1, "kind": kind,
            "sample": "BiPopCmaes(ActiveCmaes, tol=%g) n=256 rastrigin, budget %d evaluations "
                      "(first run + %d restarts), 1 thread, %.1f s" % (tol, budget_evals, restarts,
                                                                         dt)}


def bench_bipop(args, world, rank, local_rank, use_dist, barrier):
    """C5: concurrent BIPOP over the ranks.  value = objective evaluations of ALL restarts of
    ALL ranks per second of wall time (max over ranks).  roofline: the dominant kernel of the
    inner runs (the n = 256 eigensolver at small lambda: serial latency, far from any roof)."""
    import bboptpy_amd as bb
    st, dt, budget = bipop_leg(bb, world, rank, local_rank, use_dist,
                               max(1, args.steps) * 25000, barrier, slots=args.slots)
    if rank != 0:
        return
    wl, prof = c5_inner_profile(bb, local_rank)
    kernels, roofline = kernel_report(CMA_KERNELS, cma_kernel_costs(wl["n"], wl["np"], 1), prof,
                                      "C5", 1)
    out = {
        "metric": "candidate-evals/sec", "value": st.fev / dt, "unit": "candidate-evals/s",
        "n_gpus": world, "steps": st.round, "warmup": 0,
        "ms_per_step": 1e3 * dt / max(st.round, 1), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BIPOP-CMA-ES (ActiveCMAES inner) n=256 rastrigin, %d concurrent "
                               "restart populations (%d per GPU), budget %d evaluations"
                               % (world * args.slots, args.slots, budget),
                   "n": 256, "objective": "rastrigin", "box": [-5.12, 5.12],
                   "rounds": st.round, "large_restarts": st.largerestarts,
                   "small_restarts": st.smallrestarts, "best_f": st.fxbest},
        "roofline": roofline, "kernels": kernels,
        "cpu_baseline": None if args.no_cpu_baseline else c5_cpu_baseline(),
    }
    print(json.dumps(out))


def kernel_report(names, costs, prof, workload, P):
    """per-kernel averages of the profiled pass priced with the algorithmic work per launch ->
    (kernels{}, roofline of the kernel with the largest time share)"""
    kernels = {}
    if prof is not None and names:
        for i, name in enumerate(names):
            ms, calls = prof[2 * i], prof[2 * i + 1]
            if calls <= 0 or name not in costs:
                continue
            bound, work = costs[name]
            avg_s = ms * 1e-3 / calls
            if work is None:
                peak, unit = (FP64_PEAK_TFLOPS, "TFLOP/s") if bound == "mfma" else (HBM_PEAK_GBS, "GB/s")
                kernels[name] = {"avg_us": avg_s * 1e6, "share": ms, "bound": bound,
                                 "achieved": None, "peak": peak, "unit": unit, "frac": None}
                continue
            if bound == "mfma":
                ach, peak, unit = work / avg_s / 1e12, FP64_PEAK_TFLOPS, "TFLOP/s"
            else:
                ach, peak, unit = work / avg_s / 1e9, HBM_PEAK_GBS, "GB/s"
            kernels[name] = {"avg_us": avg_s * 1e6, "share": ms, "bound": bound,
                             "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak}
        tot = sum(k["share"] for k in kernels.values())
        for k in kernels.values():
            k["share"] = k["share"] / tot
    roofline = None
    if kernels:
        dom = max(kernels, key=lambda k: kernels[k]["share"])
        kd = kernels[dom]
        roofline = {"kernel": dom, "bound": kd["bound"], "achieved": kd["achieved"],
                    "peak": kd["peak"], "unit": kd["unit"], "frac": kd["frac"],
                    "traffic": measured_traffic(workload, P, dom),
                    "avg_us": kd["avg_us"], "time_share": kd["share"]}
    return kernels, roofline


def costs_for(wl, P):
    """(timer-slot names, algorithmic work per launch) of workload `wl` at P populations"""
    a = wl["algo"]
    if a == "ActiveCMAES":
        return CMA_KERNELS, cma_kernel_costs(wl["n"], wl["np"], P)
    if a == "SepCMAES":
        return CMA_KERNELS, sep_kernel_costs(wl["n"], wl["np"], P)
    if a in ("SHADE", "JADE", "SANSDE"):
        costs = de_kernel_costs(wl["n"], wl["np"], P)
        if a == "SANSDE":   # x_i, best, three partners read, one row written
            costs["de_generation"] = ("hbm", P * wl["np"] * (48 * wl["n"] + 24))
        return DE_KERNELS, costs
    if a == "CSO":
        return CSO_KERNELS, cso_kernel_costs(wl["n"], wl["np"], P)
    if a == "CCPSO":
        # the candidate count changes with the subset size drawn: no fixed per-launch work;
        # ccp_eval is bound by the objective (2 (n/s) np full-dimension evaluations per
        # generation, no matrix instruction, hardly any HBM): the HBM figure -- X and Y read
        # once, the two fitness tables written -- only shows how far from a stream it is
        n_, np_ = wl["n"], wl["np"]
        costs = {k: ("hbm", None) for k in CCPSO_KERNELS}
        costs["ccp_eval"] = ("hbm", P * (2 * np_ * n_ * 8 + 2 * np_ * n_ * 8 // min(wl["pps"])))
        costs["ccp_position"] = ("hbm", P * 3 * np_ * n_ * 8)
        return CCPSO_KERNELS, costs
    return PSO_KERNELS, pso_kernel_costs(wl["n"], wl["np"], P)


# the other BASELINE.json configs, measured inside every default (M) line so that the driver's
# own run carries them (review of round 4, item 1): key -> generations timed.  C4 is ~35 ms a
# generation (its O(np^2 n) evolutionary-state estimate), the others 0.5 - 3 ms.
CONFIG_LEGS = {"C2": 20, "C3": 20, "C4": 8, "SEP": 20}


def config_leg(bb, key, device, cpu=True, cpu_budget_s=5.0):
    """one BASELINE config as its own bench line: warm-up, `steps` timed generations (timers
    off, one host poll), the same again with the per-kernel HIP-event timers on -> roofline of
    the kernel with the largest share; the reference on one host core beside it"""
    wl = WORKLOADS[key]
    P = wl["P"]
    steps = CONFIG_LEGS[key]
    warm = 3 if key == "C4" else 5
    dt, prof, fev_pop, alg = measure(bb, wl, P, steps, warm, 2000, device, profile=True)
    del alg
    names, costs = costs_for(wl, P)
    kernels, roofline = kernel_report(names, costs, prof, key, P)
    out = {"workload": "%s n=%d np=%d %s, %d independent population%s per GPU" % (
               wl["algo"], wl["n"], wl["np"], wl["objective"], P, "s" if P > 1 else ""),
           "value": P * fev_pop / dt, "unit": "candidate-evals/s", "steps": steps,
           "warmup": warm, "ms_per_step": 1e3 * dt / steps, "roofline": roofline,
           "kernels": {k: {"avg_us": v["avg_us"], "share": v["share"], "bound": v["bound"],
                           "frac": v["frac"]} for k, v in kernels.items()}}
    if cpu:
        r = _cpu_run(wl, 1, cpu_budget_s)
        out["cpu_baseline"] = {
            "value": r["evals"] / r["dt"], "unit": "candidate-evals/s", "cores": 1,
            "kind": r["kind"],
            "sample": "%d generations of %s n=%d np=%d %s, 1 thread, %.1f s%s" % (
                r["gens"], wl["algo"], wl["n"], r["np"], wl["objective"], r["dt"], r["note"])}
    return out


def spawn_ranks(n_gpus):
    """`bench.py --gpus N` outside torchrun: start the N ranks as a CHILD process tree before
    this process has touched the GPU (never exec after HIP init), pass their output through"""
    import subprocess
    # --standalone: torchrun picks a free rendezvous port itself (no bind/close/reuse race)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr",
           "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)    # >= 200 generations ...
    ap.add_argument("--warmup", type=int, default=20)   # ... after 20 warm-up generations
    ap.add_argument("--workload", default="M", choices=sorted(WORKLOADS) + ["C5"])
    ap.add_argument("--populations", type=int, default=None,
                    help="independent populations per GPU (default: per workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-convergence", action="store_true",
                    help="skip the generations-to-tol legs")
    ap.add_argument("--no-single", action="store_true",
                    help="skip the single-population legs (profiling runs: keeps rocprofv3's "
                         "per-kernel averages to the P-population launches)")
    ap.add_argument("--no-bipop", action="store_true", help="skip the bipop_scaling leg")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the C2 / C3 / C4 / SEP legs of the default line")
    ap.add_argument("--cpu-replica", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--slots", type=int, default=1,
                    help="C5: concurrent restart populations per GPU (default 1, as configured)")
    args = ap.parse_args()

    if args.cpu_replica:
        cpu_replica_main(args.cpu_replica)
        return
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    barrier = None
    dist = None
    use_dist = world > 1 or "RANK" in os.environ     # launched by torch.distributed.run
    if use_dist:
        import torch
        import torch.distributed as dist
        # BBO_BENCH_BACKEND=gloo: rehearsal of the N-rank path on a box with fewer GPUs than
        # ranks (RCCL refuses two ranks on one device): ranks share the GPUs round-robin and the
        # collectives run on host tensors.  Never used by the driver; its numbers mean nothing.
        backend = os.environ.get("BBO_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)

        def barrier():
            dist.barrier()
            torch.cuda.synchronize()
    import bboptpy_amd as bb

    if args.workload == "C5":
        bench_bipop(args, world, rank, local_rank, use_dist, barrier)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    wl = WORKLOADS[args.workload]
    P = args.populations if args.populations else wl["P"]
    dt, prof, fev_pop, _ = measure(bb, wl, P, args.steps, args.warmup, 1000 + rank, local_rank,
                                   profile=(rank == 0), barrier=barrier)
    if use_dist:
        dt = max_over_ranks(dt)
    total_evals = world * P * fev_pop
    value = total_evals / dt

    # BIPOP multi-restart scaling (north_star): a bounded C5 leg at this world size, all ranks
    bipop = None
    if args.workload == "M" and not args.no_bipop:
        st, bdt, budget = bipop_leg(bb, world, rank, local_rank, use_dist, C5_LEG_BUDGET, barrier,
                                    tol=C5_LEG_TOL)
        bipop = {"workload": "BIPOP-CMA-ES (ActiveCMAES inner, tol %g) n=256 rastrigin, one "
                             "concurrent restart population per GPU, %d evaluations per GPU"
                             % (C5_LEG_TOL, C5_LEG_BUDGET),
                 "n_gpus": world, "value": st.fev / bdt, "unit": "candidate-evals/s",
                 "wall_s": bdt, "rounds": st.round, "restarts": len(st.history),
                 "large_restarts": st.largerestarts, "small_restarts": st.smallrestarts,
                 "evaluations": st.fev, "best_f": st.fxbest, "scaling": "weak",
                 "spread_reduction": getattr(st, "spread_reduction", None)}
        # the same with 8 concurrent restart populations PACKED on every GPU (an inner run keeps a
        # dozen compute units busy at most: the n = 256 eigensolver's widest kernel is 8 workgroups)
        st8, bdt8, _ = bipop_leg(bb, world, rank, local_rank, use_dist, C5_LEG_BUDGET, barrier,
                                 slots=8, tol=C5_LEG_TOL)
        bipop["packed_8_per_gpu"] = {"value": st8.fev / bdt8, "unit": "candidate-evals/s",
                                     "wall_s": bdt8, "rounds": st8.round,
                                     "restarts": len(st8.history),
                                     "large_restarts": st8.largerestarts,
                                     "small_restarts": st8.smallrestarts,
                                     "evaluations": st8.fev, "best_f": st8.fxbest,
                                     "spread_reduction": getattr(st8, "spread_reduction", None)}
        if rank == 0 and world == 1:
            # what an inner run is made of (ActiveCMAES n = 256, lambda = lambda_def): per-kernel
            # device time and the roofline of its dominant kernel; and the reference's own
            # BiPopCmaes on one host core under the same tol, bounded to ~15 s
            wl5, prof5 = c5_inner_profile(bb, local_rank)
            k5, r5 = kernel_report(CMA_KERNELS, cma_kernel_costs(wl5["n"], wl5["np"], 1), prof5,
                                   "C5", 1)
            bipop["inner_run_roofline"] = r5
            bipop["inner_run_kernels"] = {k: {"avg_us": v["avg_us"], "share": v["share"]}
                                          for k, v in k5.items()}
            if not args.no_cpu_baseline:
                bipop["cpu_baseline"] = c5_cpu_baseline(10000, tol=C5_LEG_TOL)

    if rank == 0:
        # per-kernel device time (HIP events on the engine's stream) -> roofline
        names, costs = costs_for(wl, P)
        kernels, roofline = kernel_report(names, costs, prof, args.workload, P)
        single = None
        singles = None
        if world == 1 and not args.no_single:
            if P != 1:
                # (a generation of ONE population is ~0.4 ms: 200 of them, so that the one host
                # synchronisation at the end of the timed region is noise, not 1 % of it)
                s1 = max(200, args.steps)
                dt1, _, fev1, _ = measure(bb, wl, 1, s1, 10, 77, local_rank, profile=False)
                single = {"value": fev1 / dt1, "ms_per_step": 1e3 * dt1 / s1, "steps": s1,
                          "unit": "candidate-evals/s"}
            if args.workload == "M":
                # one optimisation run at a time (P = 1), the strict reading of each config
                singles = {"M": single}
                for key in ("C1", "C3"):
                    w1 = WORKLOADS[key]
                    sk = 400 if key == "C1" else max(200, args.steps)
                    d1, _, f1, _ = measure(bb, w1, 1, sk, 10, 78, local_rank, profile=False)
                    singles[key] = {"value": f1 / d1, "ms_per_step": 1e3 * d1 / sk, "steps": sk,
                                    "unit": "candidate-evals/s"}
        conv = ftar = None
        if (world == 1 and wl["algo"] == "ActiveCMAES" and not args.no_convergence
                and not args.no_single):
            conv = generations_to_tol(bb, wl, local_rank)
            conv["reference"] = REFERENCE_GENERATIONS_TO_TOL.get(args.workload)
            ftar = generations_to_ftarget(bb, wl, local_rank)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(wl, key=args.workload)
        configs = None
        if world == 1 and args.workload == "M" and not args.no_configs:
            configs = {key: config_leg(bb, key, local_rank, cpu=not args.no_cpu_baseline)
                       for key in CONFIG_LEGS}
        out = {
            "metric": "candidate-evals/sec", "value": value, "unit": "candidate-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s n=%d np=%d %s, %d independent populations per GPU "
                                   "in lockstep" % (wl["algo"], wl["n"], wl["np"],
                                                    wl["objective"], P),
                       "populations_per_gpu": P, "n": wl["n"], "np": wl["np"],
                       "objective": wl["objective"], "box": list(wl["box"])},
            "single_population": single,
            "single_population_configs": singles,
            "generations_to_tol": conv,
            "generations_to_ftarget": ftar,
            "bipop_scaling": bipop,
            "configs": configs,
            "roofline": roofline,
            "kernels": kernels,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
