#!/usr/bin/env python3
"""Per-kernel summary (calls, total us, average us, share) of a rocprofv3 --kernel-trace run
stored as a rocpd SQLite database; prints CSV.  usage: prof_summary.py results.db [top_n]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("kernel,calls,total_us,avg_us,percent")
for name, calls, total, avg, pct in rows[:top]:
    short = name.replace("(anonymous namespace)::", "")
    short = re.sub(r"\(.*", "", short)           # drop the argument list
    short = re.sub(r"^void ", "", short)
    if len(short) > 90:
        short = short[:87] + "..."
    print(f'"{short}",{calls},{total:.1f},{avg:.2f},{pct:.2f}')
