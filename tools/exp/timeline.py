#!/usr/bin/env python3
"""Timeline of the last layer call in a rocprofv3 --kernel-trace database: per kernel start offset, duration and the
gap to its predecessor.  usage: timeline.py results.db n_kernels_per_call"""
import re
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
cols = [r[1] for r in db.execute(f"pragma table_info({view[0]})")]
rows = list(db.execute(f"select name, start, end from {view[0]} order by start"))
rows = rows[-n * 3:]
t0 = rows[0][1]
prev_end = None
for name, st, en in rows:
    short = re.sub(r"\(.*", "", name).replace("void ", "")[:60]
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(st - t0) / 1e3:9.1f} us  dur {(en - st) / 1e3:7.1f}  gap {gap:6.1f}  {short}")
    prev_end = en
