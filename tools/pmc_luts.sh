cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS -d gpurun_out/pmcL -o a -- python3 tools/prof_luts.py ${B:-256} 2 > gpurun_out/pmcL.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d gpurun_out/pmcL2 -o b -- python3 tools/prof_luts.py ${B:-256} 2 > gpurun_out/pmcL2.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmcL k_pbs64; python3 tools/pmc_summary.py gpurun_out/pmcL2 k_pbs64
grep k_pbs64 gpurun_out/pmcL/*kernel_trace.csv | head -2 | cut -c1-200
