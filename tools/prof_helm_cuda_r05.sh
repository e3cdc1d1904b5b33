# helm_cuda (reference src/bin/helm.rs:141-146) in the lazy field FpI and in the 51-bit field: kernel-trace stats and the
# issue-slot counters of the lockstep build.  Run on the GPU box from the repository root:
#   bash tools/prof_helm_cuda_r05.sh   -> gpurun_out/prof_hc/*
# Every rocprofv3 call has the program itself after `--`; --pmc passes carry --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/prof_hc; mkdir -p $O
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
for F in 50 51; do
  if [ $F = 51 ]; then export HELM_HIP_FIELD=51; else unset HELM_HIP_FIELD; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$F -o s -- python3 tools/prof_pbs.py helm_cuda 4096 3 > $O/stats_$F.log 2>&1 || { tail -5 $O/stats_$F.log; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_$F -o a -- python3 tools/prof_pbs.py helm_cuda 1024 3 > $O/sq1_$F.log 2>&1 || { tail -5 $O/sq1_$F.log; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2_$F -o b -- python3 tools/prof_pbs.py helm_cuda 1024 3 > $O/sq2_$F.log 2>&1 || { tail -5 $O/sq2_$F.log; exit 1; }
  python3 tools/pmc_issue.py $O/sq1_$F $O/sq2_$F $O/pmc_issue_helm_cuda_$F.json helm_cuda
  cp $(find $O/stats_$F -name "*kernel_stats.csv" | head -1) $O/helm_cuda_field${F}_kernel_stats.csv
done
unset HELM_HIP_FIELD
cat $O/pmc_issue_helm_cuda_50.json $O/pmc_issue_helm_cuda_51.json
head -4 $O/helm_cuda_field50_kernel_stats.csv $O/helm_cuda_field51_kernel_stats.csv
