"""FETCH_SIZE / WRITE_SIZE of the lockstep k_pbs launches of a bench.py pass -> profiles/rNN/pmc_traffic.json.
Usage: pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [bootstraps per pass]
Counters come from separate rocprofv3 --pmc passes (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2:
MI355X_MICROARCH.md, rocprofv3 PMC slots).  Unit and gfx950 correction as that guide's HBM section prescribes:
both counters are in KiB-like units of 1,024 B; FETCH_SIZE tallies the 128-B requests of wide reads as 64 B, so it
is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Infinity-Cache hits are included (fabric-side requests)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def lockstep(name):
    # k_pbs<PbsCfg<F, LOGN, K, L, M, TW, PREFETCH, MINW, NB = 4>>
    return "k_pbs<" in name and name.rstrip(" >").split("(")[0].rstrip(" >").endswith(", 4")


def per_dispatch(d, counter):
    rows = defaultdict(float)
    grid = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or not lockstep(r["Kernel_Name"]):
                continue
            rows[r["Dispatch_Id"]] += float(r["Counter_Value"])
            grid[r["Dispatch_Id"]] = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
    return rows, grid


fd, wd, out = sys.argv[1:4]
fetch, grid = per_dispatch(fd, "FETCH_SIZE")
write, _ = per_dispatch(wd, "WRITE_SIZE")
if not fetch or not write:
    sys.exit("no lockstep k_pbs dispatches found in the counter files")
launches = len(fetch)
boots = sum(4 * g for g in grid.values())  # four bootstraps per workgroup
fetch_b = 2.0 * 1024.0 * sum(fetch.values())
write_b = 1024.0 * sum(write.values()) * launches / len(write)
res = {
    "kernel": "k_pbs<PbsCfg<..., NB = 4>> (lockstep build)",
    "launches": launches, "bootstraps_per_launch": boots / launches,
    "fetch_bytes_per_launch": fetch_b / launches, "write_bytes_per_launch": write_b / launches,
    "bytes_per_launch": (fetch_b + write_b) / launches,
    "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over `bench.py --steps 1 --warmup 0` "
              "(tools/pmc_traffic.sh); FETCH_SIZE x2 (gfx950: 128-B requests tallied as 64 B), fabric-side requests, "
              "Infinity-Cache hits included",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
