#!/usr/bin/env python3
"""Per-layer kernel durations from a rocprofv3 --kernel-trace database of tools/resnet_prof.py: the calls are grouped at
every conv_front_kernel / image_plus_share launch, and the last call of each run of equal groups is printed.
usage: layer_kernels.py results.db"""
import re
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
rows = list(db.execute(f"select name, start, end, grid_x, workgroup_x from {view[0]} order by start"))
groups, cur = [], []
for name, st, en, gx, wx in rows:
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("mmh::", "")
    if short.startswith("at::") or "rocclr" in short:
        continue
    if short.startswith("conv_front_kernel") and cur:
        groups.append(cur)
        cur = []
    cur.append((short[:40], (en - st) / 1e3, gx // max(wx, 1), st, en))
groups.append(cur)
prev = None
out = []
for g in groups:
    key = tuple((k[0], k[2]) for k in g)
    if key == prev:
        out[-1] = g
    else:
        out.append(g)
    prev = key
for g in out:
    span = (g[-1][4] - g[0][3]) / 1e3
    print(f"span {span:7.1f} us: " + "  ".join(f"{n.split('<')[0]}[{b}] {d:.1f}" for n, d, b, _, _ in g))
