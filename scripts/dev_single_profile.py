#!/usr/bin/env python3
"""dev tool (GPU box): per-kernel device time of ONE population (P = 1) next to the wall time per
generation -- the gap is launch / dependency latency."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench          # noqa: E402
import bboptpy_amd as bb   # noqa: E402

for key in sys.argv[1:] or ["C1", "C3", "M"]:
    wl = bench.WORKLOADS[key]
    steps = 400 if key == "C1" else 100
    dt, prof, fev, _ = bench.measure(bb, wl, 1, steps, 10, 5, 0, profile=True)
    names = bench.CMA_KERNELS
    parts = {names[i]: 1e3 * prof[2 * i] / max(prof[2 * i + 1], 1) for i in range(len(names))}
    print("%s P=1: %.1f us/generation wall, kernels sum %.1f us: %s" % (
        key, 1e6 * dt / steps, sum(parts.values()),
        ", ".join("%s %.1f" % (k.replace("cma_", ""), v) for k, v in parts.items())), flush=True)
