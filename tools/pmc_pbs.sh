cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
export HELM_HIP_PBS_VARIANT=${V:-3}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS -d gpurun_out/pmcA -o a -- python3 tools/prof_pbs.py boolean_default ${B:-1024} 3 > gpurun_out/pmcA.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d gpurun_out/pmcB -o b -- python3 tools/prof_pbs.py boolean_default ${B:-1024} 3 > gpurun_out/pmcB.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmcC -o c -- python3 tools/prof_pbs.py boolean_default ${B:-1024} 3 > gpurun_out/pmcC.log 2>&1
for x in A B C; do python3 tools/pmc_summary.py gpurun_out/pmc$x k_pbs; done
grep k_pbs gpurun_out/pmcA/*kernel_trace.csv | head -3 | cut -c1-300
