# PMC passes for k_pbs (run on the GPU box): B bootstraps x 3 launches, build V (1 latency, 3 throughput)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export HELM_HIP_PBS_VARIANT=${V:-5}
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS -d gpurun_out/pmcA -o a -- python3 tools/prof_pbs.py boolean_default ${B:-1024} 3 > gpurun_out/pmcA.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d gpurun_out/pmcB -o b -- python3 tools/prof_pbs.py boolean_default ${B:-1024} 3 > gpurun_out/pmcB.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmcA k_pbs; python3 tools/pmc_summary.py gpurun_out/pmcB k_pbs
grep k_pbs gpurun_out/pmcA/*kernel_trace.csv | awk -F, '{print $NF, $(NF-1)}' | head -0
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/pmcA/*kernel_trace.csv'):
    d=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'k_pbs' in r['Kernel_Name']]
    print('k_pbs dispatches', len(d), 'avg ms', sum(d)/len(d)/1e6)
PY
