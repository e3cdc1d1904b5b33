#!/usr/bin/env python3
"""Summaries of tools/prof_r03.sh: top rows of every kernel_stats.csv, and per-kernel counter sums of every --pmc pass
(with the derived matrix-core utilisation where the MFMA counters are present).  Usage: prof_r03_summary.py <dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:110]


for f in sorted(glob.glob(root + "/*/**/*kernel_stats.csv", recursive=True)):
    print(f"## {os.path.relpath(f, root)}")
    for i, r in enumerate(csv.DictReader(open(f))):
        if i >= 8:
            break
        print(f"  {short(r['Name']):110s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e6:10.4f} ms  {float(r['Percentage']):6.2f} %")
for d in sorted(glob.glob(root + "/*")):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    tot, n = defaultdict(lambda: defaultdict(float)), defaultdict(int)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" or len(tot[k]) == 1:
                n[k] += 1
    print(f"## {os.path.relpath(d, root)} (sums over dispatches)")
    for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", tot[k].get("GRBM_GUI_ACTIVE", 0))):
        c = tot[k]
        if not any(x in k for x in ("k_pbs", "k_ks", "k_pfpks", "k_keyswitch")):
            continue
        line = "  " + k + ": " + ", ".join(f"{a} {v:.4g}" for a, v in sorted(c.items()))
        g = c.get("GRBM_GUI_ACTIVE")
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and g:
            # GRBM_GUI_ACTIVE sums the 8 XCDs: kernel cycles = / 8.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD
            # (MI355X_MICROARCH.md: "counts cycles"), 4 SIMDs x 256 CUs.  SQ_INSTS_VALU_MFMA_MOPS_I8 counts 512 int8
            # operations each (v_mfma_i32_16x16x64_i8 = 32,768 operations = 64 counts).
            cycles = g / 8.0
            line += f"  => matrix pipes busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cycles * 1024):.4f} of the chip's SIMD-cycles"
            if c.get("SQ_INSTS_VALU_MFMA_MOPS_I8"):
                ops = c["SQ_INSTS_VALU_MFMA_MOPS_I8"] * 512.0
                line += f", {ops / cycles:.0f} int8 op/cycle = {ops / cycles * 2.4e9 / 1e15:.3f} P int8-op/s at 2.4 GHz"
            if c.get("SQ_WAVE_CYCLES"):
                line += f"; {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_WAVE_CYCLES']):.3f} of the kernel's own wave-cycles"
        if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_WAVE_CYCLES"):
            line += f"  => VALU active {c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.3f} of wave-cycles"
        if c.get("SQ_LDS_IDX_ACTIVE") and g:
            line += f"  => LDS pipe active {c['SQ_LDS_IDX_ACTIVE'] / (g / 8.0 * 256):.3f} of CU-cycles"
        print(line[:1500])
