#!/usr/bin/env python3
"""Average per-launch counter values of one kernel from rocprofv3 --pmc CSV outputs
(*_counter_collection.csv under the given directories).  Prints a text summary and, with
--json FILE, writes {counter: average, "traffic_bytes": ...} applying the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts half of the streamed read bytes; both report KiB).
usage: pmc_summary.py KERNEL_SUBSTRING DIR [DIR...] [--json FILE]"""
import csv
import glob
import json
import sys

args = sys.argv[1:]
out_json = None
if "--json" in args:
    i = args.index("--json")
    out_json = args[i + 1]
    del args[i:i + 2]
kernel, dirs = args[0], args[1:]
acc = {}
for d in dirs:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
for k, v in avg.items():
    print(f"{k:<28}{v:>18.1f}  (n={len(acc[k])})")
if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
    fetch, write = 2 * avg["FETCH_SIZE"] * 1024, avg["WRITE_SIZE"] * 1024
    avg["traffic_bytes"] = fetch + write
    print(f"# HBM traffic per launch: 2 x FETCH_SIZE = {fetch / 1e6:.1f} MB read + WRITE_SIZE = {write / 1e6:.1f} MB "
          f"written = {(fetch + write) / 1e6:.1f} MB")
if out_json:
    json.dump(avg, open(out_json, "w"), indent=1)
