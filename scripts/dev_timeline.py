"""dev: per-generation timeline from a rocprofv3 --kernel-trace csv: for every kernel of the
steady-state generations its average duration and the average idle gap in front of it.
    python scripts/dev_timeline.py <kernel_trace.csv> [first_kernel_substring]"""
import csv
import sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "cma_sample_eval"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0].split("<")[0].replace("bbo::", "").replace("void ", ""),
       int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
starts = [i for i, e in enumerate(ev) if first in e[0]]
gens = [(starts[k], starts[k + 1]) for k in range(len(starts) - 1)]
gens = gens[len(gens) // 2:]          # steady state: the second half
acc = OrderedDict()
tot = 0.
for a, b in gens:
    seq = {}
    for i in range(a, b):
        name = ev[i][0]
        seq[name] = seq.get(name, 0) + 1
        key = "%s#%d" % (name, seq[name])
        dur = ev[i][2] - ev[i][1]
        gap = ev[i][1] - ev[i - 1][2] if i > 0 else 0
        s = acc.setdefault(key, [0., 0., 0])
        s[0] += dur
        s[1] += gap
        s[2] += 1
    tot += ev[b][1] - ev[a][1]
print("%d generations, %.1f us each" % (len(gens), tot / len(gens) / 1e3))
sd = sg = 0.
for k, (d, g, c) in acc.items():
    print("  %-34s %8.1f us   gap before %6.1f us   (x%d)" % (k, d / c / 1e3, g / c / 1e3, c))
    sd += d / len(gens) / 1e3
    sg += g / len(gens) / 1e3
print("  kernels %.1f us + gaps %.1f us" % (sd, sg))
