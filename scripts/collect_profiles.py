#!/usr/bin/env python3
"""Copy the summaries of a scripts/profile_all.sh run (gpurun_out/prof_<tag>_<WL>/) into profiles/
under the round's prefix and fold the PMC passes into profiles/traffic.json:
    python scripts/collect_profiles.py <tag> <prefix e.g. r05>"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]
P = {"M": 256, "C3": 256, "C2": 256, "C1": 4096, "C4": 1, "C5I": 1, "SEP": 64, "CSO": 4, "CCPSO": 16}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "prof_%s_*" % tag))):
    wl = d.rsplit("_", 1)[1]
    if wl not in P:
        continue
    base = os.path.join(ROOT, "profiles", "%s_%s_P%d" % (prefix, wl, P[wl]))
    stats = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], base + "_kernel_stats.csv")
    bj = os.path.join(d, "bench_trace.json")
    if os.path.exists(bj) and os.path.getsize(bj) > 0:
        shutil.copy(bj, base + "_bench_under_rocprof.json")
    f = glob.glob(os.path.join(d, "fetch", "**", "*counter_collection.csv"), recursive=True)
    w = glob.glob(os.path.join(d, "write", "**", "*counter_collection.csv"), recursive=True)
    if f and w:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "collect_traffic.py"),
                               "%s:P%d" % (wl, P[wl]), f[0], w[0]], stdout=subprocess.DEVNULL)
    print(wl, "stats" if stats else "-", "traffic" if f and w else "-")
