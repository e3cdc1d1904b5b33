# Round 6: k_pbs_tri10 (N = 1024, reference src/bin/helm.rs:141-146's set): the width table of the size dispatch (µ-bench), kernel-trace
# stats of launches of 768 and 512 bootstraps, and the issue-slot counters of the 768 launch.  Run on the GPU box from the
# repository root:   bash tools/prof_r06_tri10.sh   -> gpurun_out/r06/tri10/*
# Every rocprofv3 call has the program itself after `--`; --pmc passes carry --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/r06/tri10; mkdir -p $O
timeout -k 10 300 python3 tools/microbench_gates.py --sets helm_cuda --Bs 1,64,256,384,512,640,768,1024,4096 --out $O/microbench_helm_cuda.jsonl > $O/microbench.log 2>&1 || { tail -5 $O/microbench.log; exit 1; }
for B in 768 512; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$B -o s -- python3 tools/prof_pbs.py helm_cuda $B 5 > $O/stats_$B.log 2>&1 || { tail -5 $O/stats_$B.log; exit 1; }
  cp $(find $O/stats_$B -name "*kernel_stats.csv" | head -1) $O/helm_cuda_${B}_kernel_stats.csv
done
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1 -o a -- python3 tools/prof_pbs.py helm_cuda 768 3 > $O/sq1.log 2>&1 || { tail -5 $O/sq1.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2 -o b -- python3 tools/prof_pbs.py helm_cuda 768 3 > $O/sq2.log 2>&1 || { tail -5 $O/sq2.log; exit 1; }
python3 - <<'PY' > $O/pmc_tri10_768.json
import csv, glob, json
from collections import defaultdict
tot = defaultdict(float); disp = set()
for d in ("gpurun_out/r06/tri10/sq1", "gpurun_out/r06/tri10/sq2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_pbs_tri10" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add((d, r["Dispatch_Id"]))
n = 512
res = {"kernel": "k_pbs_tri10<Tri10Cfg<FpI, 3, 3>, true>", "params": "helm_cuda", "launch": "768 bootstraps = 256 workgroups of 12 waves", "dispatches_per_pass": len(disp) // 2,
       "valu_insts_per_wave": tot["SQ_INSTS_VALU"] / tot["SQ_WAVES"], "valu_insts_per_wave_step": tot["SQ_INSTS_VALU"] / tot["SQ_WAVES"] / n,
       "valu_active_of_wave_cycles": tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_WAVE_CYCLES"], "waves_per_simd": 3,
       "issue_slot_fraction": 3 * tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_WAVE_CYCLES"],
       "lds_insts_per_wave_step": tot["SQ_INSTS_LDS"] / tot["SQ_WAVES"] / n, "lds_bank_conflict_cycles_per_lds_active": tot["SQ_LDS_BANK_CONFLICT"] / max(1.0, tot["SQ_ACTIVE_INST_LDS"]),
       "source": "rocprofv3 --kernel-trace --pmc (two separate passes) over `tools/prof_pbs.py helm_cuda 768 3`: tools/prof_r06_tri10.sh"}
print(json.dumps(res, indent=1))
PY
cat $O/pmc_tri10_768.json; cat $O/microbench_helm_cuda.jsonl | cut -c1-220; head -5 $O/helm_cuda_768_kernel_stats.csv $O/helm_cuda_512_kernel_stats.csv
