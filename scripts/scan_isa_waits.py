#!/usr/bin/env python3
"""Compile every .hip of bboptpy_amd/csrc to gfx950 assembly and list, per kernel,
  * its scratch bytes (private-array spills or arrays the compiler could not keep in registers),
  * how often a global load sits behind global stores and is waited for with vmcnt(0) -- on
    gfx950 one in-order counter covers loads and stores, so that wait also waits for the stores'
    acknowledgements ("one pattern").
Build container only (hipcc cross-compiles; no GPU needed).  usage: scan_isa_waits.py [outdir]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bboptpy_amd", "csrc")
OUT = sys.argv[1] if len(sys.argv) > 1 else "/tmp/bbo_isa"
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only".split()


def main():
    os.makedirs(OUT, exist_ok=True)
    for f in sorted(os.listdir(SRC)):
        if not f.endswith(".hip"):
            continue
        asm = os.path.join(OUT, f[:-4] + ".s")
        subprocess.run(["hipcc"] + FLAGS + ["-o", asm, os.path.join(SRC, f)], check=True,
                       stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
        cur, hits, scratch = None, {}, {}
        for i, l in enumerate(lines):
            m = re.match(r"^(_Z\w+):", l)
            if m:
                cur = m.group(1)
            m = re.match(r"; ScratchSize: (\d+)", l)
            if m and cur and int(m.group(1)) > 0:
                scratch[cur] = int(m.group(1))
            if cur and "global_store" in l:
                seg = lines[i + 1:i + 14]
                li = [k for k, x in enumerate(seg) if "global_load" in x]
                if li and any("s_waitcnt vmcnt(0)" in x
                              for x in lines[i + 1 + li[0]:i + 13 + li[0]]):
                    hits[cur] = hits.get(cur, 0) + 1
        for k in sorted(set(hits) | set(scratch), key=lambda k: -hits.get(k, 0)):
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print("%-14s store->load->vmcnt(0): %3d   scratch %4d B   %s"
                  % (f, hits.get(k, 0), scratch.get(k, 0), name[:90]))


if __name__ == "__main__":
    main()
