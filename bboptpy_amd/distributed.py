"""The two multi-GPU paths of the hot path (one process per GPU): concurrent BIPOP-CMA-ES
restarts (ConcurrentBiPop) and CCPSO with its swarm groups sharded over the ranks (ShardedCCPSO,
at the end of this file).

Import order: with backend "nccl" the collectives run on torch CUDA tensors, and torch brings its
own HIP runtime, which has to be loaded before libbbopt_hip.so's: the two drivers import torch
themselves (`torch_first`) when they are given a process group or run under torchrun (RANK set),
and raise a clear error if the HIP library got there first.

Concurrent BIPOP-CMA-ES across the GPUs of one node.

The reference's BiPopCmaes (src/multivariate/cma/bipop_cmaes.cpp:109-267) is strictly
sequential: each regime decision depends on the budgets every earlier restart used.  This
module is the DOCUMENTED EXTENSION of SURVEY.md section 8e: restarts run in ROUNDS of W =
world_size concurrent runs, rank g owning slot g of every round (a complete CMA-ES run on its
own GPU: own mean, sigma, C, Philox key, lambda).  Inside a round nothing is exchanged.  After a
round ONE all_gather (RCCL over xGMI with backend "nccl"; gloo in the CPU tests) moves one
record per rank -- {ran, regime, lambda, sigma, budget, evaluations used, f, x[n]}, (n + 7)
doubles -- and every rank applies the same deterministic reduction in slot order, so the
replicated driver state (budgets, restart counters, incumbent) stays identical everywhere
without a broadcast.

Planning a round applies the reference's own rule (NBIPOP budget rule :117-142, large-regime
lambda/sigma :207-214, small-regime lambda/sigma/budget cap :241-248, per-run evaluation cap
:191-202 with the remaining budget split evenly over the slots still to plan) slot by slot,
charging each planned run its evaluation CAP until the real counts
arrive; with W = 1 every cap is replaced by the real count before the next decision, i.e. the
schedule degenerates to the reference's sequential one -- and to the single-GPU driver's
(bbo_restart.hip) draw for draw: run r (= round * W + slot; run 0 is the first default run)
takes its restart point, u and u' from the RESTART Philox stream at counter (r, k) and runs its
inner CMA-ES under the key seed + golden * r, on both.
"""
import math

import numpy as _np

from .multivariate import ActiveCMAES, CMAES, MultivariateSolution
from .objectives import Builtin



def torch_first(why):
    """A torch.distributed program uses torch's CUDA tensors for its collectives, and torch
    bundles its own HIP runtime, which must be the FIRST one the process loads -- imported after
    libbbopt_hip.so torch finds "no HIP GPUs" (measured).  Called explicitly by the two drivers
    below whenever a process group is in play; bboptpy_amd._ffi itself never imports torch."""
    import sys
    if "torch" in sys.modules:
        return
    from . import _ffi
    if _ffi._lib is not None and _ffi._lib.bbo_device_count() > 0:     # (no GPU: order is moot)
        raise RuntimeError(
            "%s: libbbopt_hip.so was loaded before torch.  `import torch` (and initialise the "
            "process group) before the first bboptpy_amd optimizer is created." % why)
    import torch  # noqa: F401


def _group_wanted(group, world_size):
    import os
    return world_size is None and (group is not None or os.environ.get("RANK") is not None)


def _default_device(device):
    """the GPU of this process when the caller named none: LOCAL_RANK under torchrun (one process
    per GPU -- with 0 for everybody every rank of an RCCL group would sit on cuda:0, which RCCL
    refuses as 'Duplicate GPU'), else torch's current device if torch has one, else 0"""
    import os
    import sys
    if device is not None:
        return int(device)
    if os.environ.get("LOCAL_RANK") is not None:
        return int(os.environ["LOCAL_RANK"])
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        return int(torch.cuda.current_device())
    return 0


def _device_of(dist, group, device):
    """where the collective's buffers live: the engine's own GPU under nccl (the same
    _default_device the engine was created on), host memory otherwise"""
    import torch
    if dist.get_backend(group) != "nccl":
        return torch.device("cpu")
    return torch.device("cuda", _default_device(device))


_GOLDEN = 0x9E3779B97F4A7C15
_M64 = (1 << 64) - 1
_STREAM_RESTART = 7


def philox4x32_10(seed, c0, c1, c2, c3):
    """host twin of bbo::philox4x32_10 (bboptpy_amd/csrc/bbo_rng.hpp) for the driver's few
    draws per round"""
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, \
            ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0 = (k0 + 0x9E3779B9) & 0xFFFFFFFF
        k1 = (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


def _u01(lo, hi):
    return float((((hi << 32) | lo) >> 11)) * 2.0 ** -53


class _State:
    """driver state replicated on every rank"""

    def __init__(self):
        self.fev = 0
        self.largebudget = self.smallbudget = 0
        self.largerestarts = self.smallrestarts = 0
        self.largelambda = 0
        self.bestregime = 1
        self.fxbest = math.inf
        self.xbest = None
        self.round = 0
        self.history = []


class _ConcurrentRestarts:
    """rounds of world_size * slots_per_rank concurrent restart runs: topology, the restart
    stream's draws, the inner runs, ONE all-gather of (n + 7) doubles per slot and round, the
    replicated reduction.  Subclasses state a schedule: plan_round / apply_round / finished."""

    _name = "ConcurrentRestarts"

    def _common(self, mfev, tol, sigma0, variant, seed, device, group, runner, world_size, rank,
                slots_per_rank):
        self.mfev, self.tol, self.sigma0 = int(mfev), float(tol), float(sigma0)
        self.variant, self.seed = variant, int(seed) & _M64
        self.device, self.group, self.runner = device, group, runner
        self._world, self._rank = world_size, rank
        # slots_per_rank concurrent restart populations PER GPU (each on its own engine and HIP
        # stream, driven from its own host thread): a round then has world_size * slots_per_rank
        # slots.  An inner run at n = 256 keeps ONE compute unit busy most of the time (the
        # eigensolver is one workgroup), so several of them share a GPU almost for free.
        self.slots = max(1, int(slots_per_rank))
        self._algs = {}
        if _group_wanted(group, world_size):
            torch_first(self._name)

    # -- topology ---------------------------------------------------------------------------
    def _topology(self):
        if self._world is not None:
            return self._world, (self._rank or 0), None
        import os
        import sys
        if "torch" not in sys.modules and "RANK" not in os.environ:
            return 1, 0, None       # nobody set up a process group: do not pay for `import torch`
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return dist.get_world_size(self.group), dist.get_rank(self.group), dist
        except ImportError:
            pass
        return 1, 0, None

    # -- the reference's rules --------------------------------------------------------------
    def _max_evals(self, lam, fev, slots_left=1):
        """bipop_cmaes.cpp:191-202, with the remaining budget SPLIT EVENLY among the slots of
        the round still to be planned (slots_left = 1, i.e. W = 1 or the last slot: the
        reference's `mfev - fev`).  Without the split the first run's cap -- at n = 256 it is
        1.5e7 evaluations, above any usual mfev -- would take the whole budget and leave the
        other GPUs of the round idle."""
        maxit = int(100. + 50. * (self.n + 3) * (self.n + 3) / math.sqrt(1. * lam))
        return min(maxit * lam, (self.mfev - fev) // max(1, slots_left))

    def _uniform(self, run, k, a, b):
        """draw k of run `run`: the sequential driver's counter layout (bbo_restart.hip
        `uniform`), so a plan does not depend on which rank executes the run"""
        w = philox4x32_10(self.seed, run & 0xFFFFFFFF, k, 0, _STREAM_RESTART << 24)
        return _u01(w[0], w[1]) * (b - a) + a

    # -- one inner run on this rank's GPU -----------------------------------------------------
    def _device_run(self, f, lam, sigma, maxfev, x0, seed, slot=0):
        """like the reference's drivers (bipop_cmaes.cpp:83-87): ONE inner optimizer per driver
        (here: per slot), re-parameterised through setParams before every run -- so B and C
        keep their off-diagonals from this slot's previous run (cmaes.cpp:53-59) -- and one
        extra evaluation of the point it returns"""
        if slot not in self._algs:
            cls = ActiveCMAES if self.variant == "active" else CMAES
            self._algs[slot] = cls(mfev=maxfev, tol=self.tol, np=lam, sigma0=sigma, seed=seed,
                                   device=_default_device(self.device))
        alg = self._algs[slot]
        alg.set_params(lam, sigma, maxfev)
        alg.set_seed(seed)
        sol = alg.optimize(f, self.lower, self.upper, x0)
        return sol.x, sol.n_evals, alg.evaluate(sol.x)

    def _run_slots(self, f, slots, st, total, first):
        """the records of the slots first .. first + len(slots) - 1 of this round.  Device runs
        of several slots go out concurrently (one host thread each: the C calls release the
        GIL, every engine has its own HIP stream); a `runner` is called slot after slot."""
        reclen = 7 + self.n

        def one(k):
            plan = slots[k]
            r = _np.zeros(reclen)
            if plan is None:
                return r
            s = first + k
            seed = (self.seed + _GOLDEN * (st.round * total + s)) & _M64
            if self.runner is not None:
                x, used, fx = self.runner(plan["lam"], plan["sigma"], plan["maxfev"], plan["x0"],
                                          seed)
            else:
                x, used, fx = self._device_run(f, plan["lam"], plan["sigma"], plan["maxfev"],
                                               plan["x0"], seed, slot=s)
            r[:7] = [1., plan["regime"], plan["lam"], plan["sigma"], plan["maxfev"], used, fx]
            r[7:] = x
            return r

        live = [k for k in range(len(slots)) if slots[k] is not None]
        if self.runner is not None or len(live) <= 1:
            return [one(k) for k in range(len(slots))]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=len(live)) as pool:
            return list(pool.map(one, range(len(slots))))

    def optimize(self, f, lower, upper, guess):
        self.lower = _np.ascontiguousarray(lower, dtype=_np.float64)
        self.upper = _np.ascontiguousarray(upper, dtype=_np.float64)
        self.guess = _np.ascontiguousarray(guess, dtype=_np.float64)
        self.n = self.lower.size
        self.lambdadef = 4 + int(3. * math.log(1. * self.n))
        world, rank, dist = self._topology()
        S = self.slots
        total = world * S                     # slots of a round
        st = _State()
        self.state = st
        reclen = 7 + self.n
        while True:
            slots = self.plan_round(st, total)
            if all(p is None for p in slots):
                break
            if dist is not None:
                import torch
                mine = _np.concatenate(self._run_slots(f, slots[rank * S:(rank + 1) * S], st, total,
                                                       rank * S))
                dev = _device_of(dist, self.group, self.device)
                mine_t = torch.from_numpy(mine).to(dev)
                out = [torch.empty_like(mine_t) for _ in range(world)]
                dist.all_gather(out, mine_t, group=self.group)
                records = [r for t in out for r in t.cpu().numpy().reshape(S, reclen)]
                self.collectives = getattr(self, "collectives", 0) + 1
            else:
                # no process group: every rank's slots run in this process, rank after rank (the
                # serial stand-in for the collective -- same plan, same reduction)
                records = []
                for g in range(world):
                    records += self._run_slots(f, slots[g * S:(g + 1) * S], st, total, g * S)
            self.apply_round(st, slots, records)
            if self.finished(st):
                break
        return MultivariateSolution(st.xbest, st.fev, False)


class ConcurrentBiPop(_ConcurrentRestarts):
    """BIPOP/NBIPOP-CMA-ES with world_size concurrent restart populations.

    Parameters mirror BiPopCMAES(base, mfev, print, sigma0, maxlargeruns, nbipop, ksigmadec,
    kbudget) (py/multivariate_py.cpp:144-151); instead of a `base` object the inner optimizer
    is described by `variant` ("active" | "cmaes") and `tol`.  `group` is a torch.distributed
    process group (None = the default group, or no collective at all when torch.distributed is
    not initialised: a single-process run).  `runner(lam, sigma, maxfev, x0, seed) -> (x,
    evaluations_used, f(x))` replaces the device run in the CPU tests.  `slots_per_rank` > 1 packs
    several concurrent restart populations onto each GPU (a rank's slots of a round run at the same
    time on separate engines / HIP streams); a round has world_size * slots_per_rank slots and
    the plan, the seeds and the reduction depend on the global slot only, so (W ranks, S slots)
    and (W S ranks, 1 slot) produce the same history.
    """

    _name = "ConcurrentBiPop"

    def __init__(self, mfev, tol=1e-8, sigma0=2., maxlargeruns=9, nbipop=True, ksigmadec=1.6,
                 kbudget=2., variant="active", seed=0, device=None, group=None, runner=None,
                 world_size=None, rank=None, slots_per_rank=1):
        self._common(mfev, tol, sigma0, variant, seed, device, group, runner, world_size, rank,
                     slots_per_rank)
        self.maxlargeruns, self.nbipop = int(maxlargeruns), bool(nbipop)
        self.ksigmadec, self.kbudget = float(ksigmadec), float(kbudget)

    def finished(self, st):
        return st.largerestarts >= self.maxlargeruns or st.fev >= self.mfev

    def plan_round(self, st, world):
        """the W runs of round st.round: a list of dicts (or None for an idle slot)"""
        slots = []
        lb, sb = st.largebudget, st.smallbudget
        nl, largelambda, fev = st.largerestarts, st.largelambda, st.fev
        for s in range(world):
            run = st.round * world + s          # global run index (0 = the first default run)
            if st.round == 0 and s == 0:
                # the reference's first default run from the user's guess (:76-87)
                maxfev = self._max_evals(self.lambdadef, fev, world - s)
                slots.append(dict(regime=0, lam=self.lambdadef, sigma=self.sigma0,
                                  maxfev=maxfev, x0=self.guess.copy()))
                fev += maxfev + 1
                continue
            if nl >= self.maxlargeruns or fev >= self.mfev:
                slots.append(None)
                continue
            if self.nbipop:
                if st.bestregime == 1:
                    regime = 1 if lb <= sb * self.kbudget else 2
                else:
                    regime = 2 if sb <= self.kbudget * lb else 1
            else:
                regime = 1 if lb <= sb else 2
            x0 = _np.array([self._uniform(run, j, self.lower[j], self.upper[j])
                            for j in range(self.n)])
            if regime == 1:
                lam = int(self.lambdadef * math.pow(2, nl + 1))
                if self.nbipop:
                    sigma = max(self.sigma0 * math.pow(1. / self.ksigmadec, nl + 1),
                                0.01 * self.sigma0)
                else:
                    sigma = self.sigma0
                maxfev = self._max_evals(lam, fev, world - s)
                if maxfev > 0:
                    nl += 1
                    largelambda = lam
                    lb += maxfev
            else:
                u = self._uniform(run, self.n, 0., 1.)
                u2 = self._uniform(run, self.n + 1, 0., 1.)
                lam = int(self.lambdadef * math.pow((0.5 * largelambda) / self.lambdadef, u * u))
                sigma = self.sigma0 * math.pow(10., -2. * u2)
                maxfev = min(self._max_evals(lam, fev, world - s), lb >> 1)
                if maxfev > 0:
                    sb += maxfev
            if maxfev <= 0 or lam < 4:
                slots.append(None)
                continue
            slots.append(dict(regime=regime, lam=lam, sigma=sigma, maxfev=maxfev, x0=x0))
            fev += maxfev + 1
        return slots

    def apply_round(self, st, slots, records):
        """the deterministic reduction every rank applies to the gathered records"""
        for s, (plan, rec) in enumerate(zip(slots, records)):
            if plan is None or rec[0] == 0.:
                continue
            used, fx, x = int(rec[5]), float(rec[6]), _np.array(rec[7:7 + self.n])
            st.fev += used + 1          # +1: the re-evaluation of the returned point (:86-87)
            if plan["regime"] == 1:
                st.largebudget += used
                st.largerestarts += 1
                st.largelambda = plan["lam"]
            elif plan["regime"] == 2:
                st.smallbudget += used
                st.smallrestarts += 1
            if st.xbest is None or fx < st.fxbest:
                st.fxbest, st.xbest = fx, x
                if plan["regime"] != 0:
                    st.bestregime = plan["regime"]
            st.history.append(dict(round=st.round, slot=s, regime=plan["regime"],
                                   lam=plan["lam"], sigma=plan["sigma"], maxfev=plan["maxfev"],
                                   used=used, fx=fx))
        st.round += 1


class ConcurrentIPop(_ConcurrentRestarts):
    """IPOP / NIPOP-aCMA-ES (IPopCmaes, src/multivariate/cma/ipop_cmaes.cpp:65-189) with
    world_size concurrent restart populations -- the IPOP variant of the all-gather of
    SURVEY.md section 8e, beside ConcurrentBiPop.

    The reference's schedule is a fixed sequence: run r >= 1 starts from a uniform point of the
    box with lambda doubled (:122-133; with `boundlambda` the doubling is cycled at lambda_max =
    10 n^2 -- to lambda_max if that is the nearer of the two, else back to lambda_def) and,
    with `nipop`, sigma divided by `ksigmadec` down to 0.01 sigma0 (:134-137); only the
    evaluation cap of a run depends on what the earlier runs used (:178-189).  Round k therefore
    runs the NEXT world_size * slots_per_rank entries of that sequence side by side -- lambda_def
    2^(kW+1) ... 2^(kW+W) while nothing cycles -- each charged its cap (the remaining budget
    split evenly over the slots still to plan) until the real counts arrive with the round's ONE
    all-gather of (n + 7) doubles per slot; every rank then applies the same reduction in slot
    order (budget, incumbent on strict improvement :150-153).  With one slot per round every cap
    is the reference's `mfev - fev` and the driver IS the sequential IPopCMAES / bbo_restart.hip,
    draw for draw and bit for bit: run r takes its restart point from the RESTART Philox stream
    at counter (r, k) and runs under the key seed + golden * r.

    Parameters mirror IPopCMAES(base, mfev, print, sigma0, nipop, ksigmadec, boundlambda)
    (py/multivariate_py.cpp:137-142); `variant` / `tol` describe the inner optimizer, the rest is
    ConcurrentBiPop's."""

    _name = "ConcurrentIPop"

    def __init__(self, mfev, tol=1e-8, sigma0=2., nipop=True, ksigmadec=1.6, boundlambda=True,
                 variant="active", seed=0, device=None, group=None, runner=None, world_size=None,
                 rank=None, slots_per_rank=1):
        self._common(mfev, tol, sigma0, variant, seed, device, group, runner, world_size, rank,
                     slots_per_rank)
        self.nipop, self.ksigmadec = bool(nipop), float(ksigmadec)
        self.boundlambda = bool(boundlambda)

    def finished(self, st):
        return st.fev >= self.mfev                      # ipop_cmaes.cpp:171-173

    def _next(self, lam, sigma):
        """(lambda, sigma) of the run after one with (lam, sigma): ipop_cmaes.cpp:122-137"""
        lam <<= 1
        if self.boundlambda:
            lmax = 10 * self.n * self.n
            if lam > lmax:
                lam = lmax if lam - lmax < lmax - (lam >> 1) else self.lambdadef
        if self.nipop:
            sigma = max(sigma / self.ksigmadec, 0.01 * self.sigma0)
        return lam, sigma

    def plan_round(self, st, world):
        slots = []
        fev = st.fev
        lam = getattr(st, "lam", self.lambdadef)
        sigma = getattr(st, "sigma", self.sigma0)
        for s in range(world):
            run = st.round * world + s
            if fev >= self.mfev:
                slots.append(None)
                continue
            if run == 0:
                # the first default run from the user's guess (:89-96)
                x0 = self.guess.copy()
                lam, sigma = self.lambdadef, self.sigma0
            else:
                x0 = _np.array([self._uniform(run, j, self.lower[j], self.upper[j])
                                for j in range(self.n)])
                lam, sigma = self._next(lam, sigma)
            maxfev = self._max_evals(lam, fev, world - s)
            if maxfev <= 0:
                slots.append(None)
                continue
            slots.append(dict(regime=0 if run == 0 else 1, lam=lam, sigma=sigma, maxfev=maxfev,
                              x0=x0))
            fev += maxfev + 1
        return slots

    def apply_round(self, st, slots, records):
        for s, (plan, rec) in enumerate(zip(slots, records)):
            if plan is None or rec[0] == 0.:
                continue
            used, fx, x = int(rec[5]), float(rec[6]), _np.array(rec[7:7 + self.n])
            st.fev += used + 1          # +1: the re-evaluation of the returned point (:93-94, :145-146)
            st.lam, st.sigma = plan["lam"], plan["sigma"]
            if plan["regime"] == 1:
                st.largerestarts += 1                  # (restarts so far: IPopCmaes::_it)
            if st.xbest is None or fx < st.fxbest:
                st.fxbest, st.xbest = fx, x
            st.history.append(dict(round=st.round, slot=s, regime=plan["regime"],
                                   lam=plan["lam"], sigma=plan["sigma"], maxfev=plan["maxfev"],
                                   used=used, fx=fx))
        st.round += 1


class ShardedCCPSO:
    """CCPSO2 (CCPSOSearch, src/multivariate/pso/ccpso.cpp) with the swarm groups of ONE swarm
    population sharded over the ranks -- the "multi-swarm variant" of SURVEY.md section 8e/8f-4.

    What is sharded: the 2 (n/s) np context-vector evaluations of a generation
    (updateSwarm, ccpso.cpp:241-260, each a full n-dimensional objective call through
    evaluate :152-171) -- they are independent while yhat is frozen, and they are where a
    generation's time goes.  Rank r evaluates the swarms [nswarm r / W, nswarm (r + 1) / W).
    What is exchanged: ONE all-gather per generation of the fitness records (fX | fY: 2 n np
    doubles per rank; RCCL over xGMI with backend "nccl", gloo in the CPU tests).  Everything
    else -- regrouping, personal / swarm / local bests, the move of yhat and its re-evaluation,
    the Cauchy rate, the position update, the stop test -- is replicated: every rank holds the
    whole state, draws the same Philox numbers (same seed everywhere) and applies the same
    update to the same merged tables, so no broadcast is needed and the result is bit-identical
    to the unsharded optimizer's for every world size.

    Constructor arguments are CCPSO's (py/multivariate_py.cpp:291-295; no `local` optimizer).
    `engine_factory()` replaces the device engine in the CPU tests (any object with
    set_shard / initialize / phase / table_record / export_tables / merge_tables / get_state).
    Without a process group, `world_size` > 1 runs all ranks in this process one after the other
    (the serial stand-in for the collective: same plan, same merge)."""

    def __init__(self, mfev, sigmatol, np, pps, npps=None, correct=True, pcauchy=-1., seed=0,
                 device=None, group=None, engine_factory=None, world_size=None, rank=None,
                 always_exchange=False):
        self.mfev = int(mfev)
        # always_exchange: run the all-gather + merge even in a group of ONE rank (where the
        # merge is the identity) -- how the RCCL path is exercised on a one-GPU machine
        self._always = bool(always_exchange)
        self.collectives = 0
        if _group_wanted(group, world_size) and engine_factory is None:
            torch_first("ShardedCCPSO")
        self._ctor = dict(mfev=mfev, sigmatol=sigmatol, np=np, pps=pps, npps=npps,
                          correct=correct, pcauchy=pcauchy)
        self.seed, self.device, self.group = int(seed) & _M64, device, group
        self._factory = engine_factory
        self._world, self._rank = world_size, rank
        self._engines = None

    def _make_engine(self):
        if self._factory is not None:
            return self._factory()
        from .multivariate import CCPSO
        return CCPSO(seed=self.seed, device=_default_device(self.device), **self._ctor)

    def _topology(self):
        import os
        import sys
        if self._world is not None:
            return self._world, (self._rank or 0), None
        if "torch" not in sys.modules and "RANK" not in os.environ:
            return 1, 0, None
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return dist.get_world_size(self.group), dist.get_rank(self.group), dist
        except ImportError:
            pass
        return 1, 0, None

    def initialize(self, f, lower, upper, guess=None):
        self.world, self.rank, self._dist = self._topology()
        lower = _np.ascontiguousarray(lower, dtype=_np.float64)
        upper = _np.ascontiguousarray(upper, dtype=_np.float64)
        guess = _np.zeros(lower.size) if guess is None else guess    # (the reference ignores it)
        serial = self._dist is None and self.world > 1
        ranks = range(self.world) if serial else (self.rank,)
        self._engines = []
        for r in ranks:
            e = self._make_engine()
            e.set_shard(r, self.world)
            e.initialize(f, lower, upper, guess)
            self._engines.append(e)
        self._buf = None

    def iterate(self):
        W = self.world
        for e in self._engines:
            e.phase(0)
        if W == 1 and not (self._always and self._dist is not None):
            pass                                  # one rank evaluated every swarm: nothing to merge
        elif self._dist is None:                  # serial stand-in: all ranks live in this process
            gathered = _np.stack([e.export_tables() for e in self._engines])
            for e in self._engines:
                e.merge_tables(gathered, W)
        else:
            import torch
            dist, e = self._dist, self._engines[0]
            reclen = e.table_record()
            dev = _device_of(dist, self.group, self.device)
            on_gpu = dev.type == "cuda"
            if self._buf is None:
                self._buf = (torch.zeros(reclen, dtype=torch.float64, device=dev),
                             torch.zeros(W * reclen, dtype=torch.float64, device=dev))
            mine, allrec = self._buf
            if on_gpu:                            # device to device: the record never visits the host
                e.export_tables(device_ptr=mine.data_ptr())
                with torch.cuda.device(dev):      # RCCL launches on the current device's stream
                    dist.all_gather_into_tensor(allrec, mine, group=self.group)
                    torch.cuda.current_stream(dev).synchronize()
                e.merge_tables(world=W, device_ptr=allrec.data_ptr())
            else:
                e.export_tables(out=mine.numpy())
                dist.all_gather(list(allrec.view(W, reclen).unbind(0)), mine, group=self.group)
                e.merge_tables(allrec.numpy(), W)
            self.collectives += 1
        for e in self._engines:
            e.phase(1)

    def get_state(self, key):
        return self._engines[0].get_state(key)

    def optimize(self, f, lower, upper, guess=None):
        self.initialize(f, lower, upper, guess)
        while True:                               # ccpso.cpp:135-148
            self.iterate()
            if int(self.get_state("fev")[0]) >= self.mfev:
                converged = False
                break
            if int(self.get_state("conv")[0]):
                converged = True
                break
        return MultivariateSolution(self.get_state("yhat"), int(self.get_state("fev")[0]),
                                    converged)
