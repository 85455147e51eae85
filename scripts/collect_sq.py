#!/usr/bin/env python3
"""Fold the two SQ counter passes of scripts/profile_mfma.sh (rocprofv3 --pmc ..., csv output)
into profiles/<tag>_<WL>_P<P>_sq_counters.json: per kernel, the average of every counter per
launch plus three derived figures (MFMA utilisation at the nominal 2.4 GHz, the share of wavefront
cycles parked in s_waitcnt / barriers, vector instructions that are not MFMA).

usage: collect_sq.py OUT.json sq_counter_collection.csv sq2_counter_collection.csv
"""
import csv
import json
import re
import sys
from collections import defaultdict

N_SIMD = 1024          # 256 CUs x 4


def per_kernel(path):
    tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    dur, nd = defaultdict(float), defaultdict(int)
    seen = set()
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r"\(.*", "", row["Kernel_Name"])
            name = re.sub(r"<.*", "", name).split("::")[-1].replace("void ", "").strip()
            c = row["Counter_Name"]
            tot[name][c] += float(row["Counter_Value"])
            cnt[name][c] += 1
            key = (row.get("Dispatch_Id"), name)
            if key not in seen and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                seen.add(key)
                dur[name] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3
                nd[name] += 1
    out = {}
    for k in tot:
        out[k] = {c: tot[k][c] / cnt[k][c] for c in tot[k]}
        if nd[k]:
            out[k]["duration_us_under_pmc"] = dur[k] / nd[k]
    return out


def main():
    out_path, a, b = sys.argv[1:4]
    ka, kb = per_kernel(a), per_kernel(b)
    kernels = {}
    for k in sorted(set(ka) | set(kb)):
        if k.startswith("__amd"):
            continue
        v = dict(kb.get(k, {}))
        v.update(ka.get(k, {}))
        d_us = v.get("duration_us_under_pmc")
        if d_us and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            v["mfma_utilisation_at_2p4GHz"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * d_us * 2400.)
        if v.get("SQ_WAVE_CYCLES"):
            v["wait_any_share_of_wave_cycles"] = v.get("SQ_WAIT_ANY", 0.) / v["SQ_WAVE_CYCLES"]
        if "SQ_INSTS_VALU" in v:
            v["valu_non_mfma_instructions"] = v["SQ_INSTS_VALU"] - v.get("SQ_INSTS_MFMA", 0.)
        kernels[k] = v
    doc = {"kernels": kernels,
           "note": "averages per launch; separate rocprofv3 --pmc passes (scripts/profile_mfma.sh); "
                   "SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles "
                   "summed over the SIMDs"}
    with open(out_path, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    for k, v in kernels.items():
        print("%-24s %8.1f us  mfma %.3f  parked %.2f" % (
            k, v.get("duration_us_under_pmc", 0.), v.get("mfma_utilisation_at_2p4GHz", 0.),
            v.get("wait_any_share_of_wave_cycles", 0.)))


if __name__ == "__main__":
    main()
