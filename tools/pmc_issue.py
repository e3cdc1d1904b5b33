"""SQ_INSTS_VALU / SQ_WAVES of the lockstep k_pbs -> profiles/rNN/pmc_issue.json (bench.py's roofline.valu_issue reads the
round-agnostic copy profiles/pmc_issue.json).
Usage: pmc_issue.py <dir of the SQ_INSTS_VALU pass> <dir of the SQ_WAVES pass> <out.json> <params> [source label]
Both passes are `rocprofv3 --kernel-trace --pmc ... -- python3 tools/prof_pbs.py <params> 1024 3` (whole lockstep rounds:
every wave of the dispatch belongs to the lockstep build; k + 1 waves per bootstrap)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def lockstep(name):
    return "k_pbs<" in name and name.rstrip(" >").split("(")[0].rstrip(" >").endswith(", 4")


def total(d, counter):
    s, n = 0.0, set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and lockstep(r["Kernel_Name"]):
                s += float(r["Counter_Value"])
                n.add(r["Dispatch_Id"])
    return s, len(n)


d1, d2, out, params = sys.argv[1:5]
insts, n1 = total(d1, "SQ_INSTS_VALU")
waves, n2 = total(d2, "SQ_WAVES")
busy, _ = total(d1, "SQ_ACTIVE_INST_VALU")
wcyc, _ = total(d1, "SQ_WAVE_CYCLES")
if not insts or not waves or n1 != n2:
    sys.exit(f"lockstep k_pbs dispatches: {n1} with SQ_INSTS_VALU, {n2} with SQ_WAVES")
k1 = {"boolean_default": 3, "helm_cuda": 2}.get(params)
res = {"kernel": "k_pbs<PbsCfg<..., NB = 4>> (lockstep build)", "params": params, "dispatches": n1,
       "valu_insts_per_wave": insts / waves, "waves_per_bootstrap": k1, "waves": waves,
       "valu_active_of_wave_cycles": (busy / wcyc) if wcyc else None,
       "source": sys.argv[5] if len(sys.argv) > 5 else
       "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU ... and --pmc SQ_WAVES ..., separate passes over `tools/prof_pbs.py <params> 1024 3`"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
