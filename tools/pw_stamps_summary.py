#!/usr/bin/env python3
"""Summary of the product kernel's in-kernel clock stamps (MMH_PW_STAMPS=<file>, conv.hip: pw_mfma_kernel<true>): the
LAST table of the file (the launch behind the warm-up), workgroup 0, per wave and over all waves: mean shader cycles per
step spent (a) waiting for this step's digits and at the barrier, (b) issuing the next step's copies, (c) reading the
column fragments from LDS, (d) in the products (weight fragments read between them), (e) from the last product to the
top of the next step, and the step period.  usage: pw_stamps_summary.py stamps.txt"""
import sys
tables, cur = [], None
for line in open(sys.argv[1]):
    if line.startswith("#"):
        cur = {"head": line.strip(), "rows": []}
        tables.append(cur)
    elif cur is not None and line.strip():
        cur["rows"].append([int(x) for x in line.split()])
t = tables[-1]
print(t["head"])
print(f"# {len(tables)} launches in the file; this is the last one")
print("wave steps  wait+barrier  issue_copies  read_B_frags  products  to_next_top  period")
tot = [0.0] * 6
nw = 0
for w in range(8):
    rows = [r for r in t["rows"] if r[0] == w]
    if len(rows) < 3:
        continue
    rows = rows[1:]                       # the first step of the launch has nothing in flight yet
    seg = [0.0] * 6
    n = 0
    for a, b in zip(rows, rows[1:]):
        period = b[2] - a[2]
        if period <= 0 or period > 100000:  # an item boundary's epilogue sits between these two steps
            continue
        d1, d2, d3, d4 = a[3], a[4], a[5], a[6]
        vals = [d1, d2 - d1, d3 - d2, d4 - d3, period - d4, period]
        seg = [x + y for x, y in zip(seg, vals)]
        n += 1
    if not n:
        continue
    seg = [x / n for x in seg]
    tot = [x + y for x, y in zip(tot, seg)]
    nw += 1
    print(f"{w:4d} {n:5d}  {seg[0]:12.0f}  {seg[1]:12.0f}  {seg[2]:12.0f}  {seg[3]:8.0f}  {seg[4]:11.0f}  {seg[5]:6.0f}")
if nw:
    tot = [x / nw for x in tot]
    print(f" all        {tot[0]:12.0f}  {tot[1]:12.0f}  {tot[2]:12.0f}  {tot[3]:8.0f}  {tot[4]:11.0f}  {tot[5]:6.0f}")
    print("# shares of the step: " + "  ".join(f"{n} {v / tot[5]:.0%}" for n, v in zip(
        ["wait+barrier", "issue_copies", "read_B_frags", "products", "to_next_top"], tot[:5])))
