"""Summarise a rocprofv3 --pmc run: per-kernel sums of every counter (one line per counter).
Usage: pmc_summary.py <dir with *_counter_collection.csv> [kernel substring]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_pbs"
tot = defaultdict(float)
n = defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:28s} {tot[k]:18.0f}   dispatches {n[k]}")
