#!/usr/bin/env python3
"""Derived figures for a tools/prof_layer_pmc.sh summary (gpurun_out/lp_<tag>/pmc.txt): per kernel the HBM traffic
(2 x FETCH_SIZE + WRITE_SIZE, KiB; gfx950 correction of MI355X_MICROARCH.md) and its rate over the kernel's average
duration, matrix-core busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), VALU issue =
SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), waiting share = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.
usage: pmc_derive.py pmc.txt > profiles/r03_layer_pmc_<tag>.txt"""
import re
import sys
text = open(sys.argv[1]).read()
print(text.rstrip())
avg_us = {}
for line in text.splitlines():
    m = re.match(r'"mmh::([a-z_0-9]+)[^"]*",(\d+),([0-9.]+),([0-9.]+),', line)
    if m and m.group(1) not in avg_us:
        avg_us[m.group(1)] = float(m.group(4))
print("\n# ---- derived (tools/pmc_derive.py)")
for block in text.split("## ")[1:]:
    name, *rows = block.strip().splitlines()
    c = {}
    for r in rows:
        p = r.split()
        if len(p) >= 2 and p[0].isupper():
            c[p[0]] = float(p[1])
    if "GRBM_GUI_ACTIVE" not in c:
        continue
    chip_cycles = c["GRBM_GUI_ACTIVE"] / 8
    us = avg_us.get(name)
    out = [f"{name}: {us} us avg" if us else name]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c and us:
        traffic = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        out.append(f"HBM-side traffic {traffic / 1e6:.1f} MB = {traffic / us / 1e6:.2f} TB/s ({traffic / us / 1e6 / 8:.2f} of 8 TB/s)")
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        out.append(f"matrix-core busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * chip_cycles):.2f}")
    out.append(f"VALU issue {c['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * chip_cycles):.2f}")
    out.append(f"waiting {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f} of wave cycles")
    if c.get("SQ_INSTS_LDS"):
        out.append(f"LDS bank conflict cycles per LDS instruction {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_INSTS_LDS']:.2f}")
    print("# " + "; ".join(out))
