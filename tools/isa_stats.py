#!/usr/bin/env python3
"""Register / scratch / LDS figures of the kernels in a hipcc -save-temps assembly dump (…gfx950.s):
   python tools/isa_stats.py <file.s> [substring filters...]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
filters = sys.argv[2:]
rows = []
for blk in re.findall(r"- \.agpr_count:.*?(?=- \.agpr_count:|\Z)", s, re.S):
    g = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", blk).group(1))
    rows.append((re.search(r"\.name:\s+(\S+)", blk).group(1), g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"),
                 g("group_segment_fixed_size")))
dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (n, v, sg, pr, lds), d in zip(rows, dem):
    d = d.replace("mgx::", "").split("(")[0]
    if not filters or any(f in d for f in filters) or pr > 0:
        print(f"{d[:80]:80s} vgpr={v} sgpr={sg} scratch={pr} lds={lds}")
