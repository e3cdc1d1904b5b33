# Round 6: the counters bench.py quotes (profiles/pmc_issue.json, profiles/pmc_traffic.json), re-collected on the round's tree -
# the lockstep k_pbs lost its dead builds' template parameters, its instruction stream must be what it was.
# Separate --pmc passes, --kernel-trace only, the program itself after `--`.  -> gpurun_out/prof_r06/pmc_{issue,traffic}.json
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/prof_r06; mkdir -p $O
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_w1024 -o a -- python3 tools/prof_pbs.py boolean_default 1024 3 > $O/sq1_w1024.log 2>&1 || { tail -5 $O/sq1_w1024.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2_w1024 -o b -- python3 tools/prof_pbs.py boolean_default 1024 3 > $O/sq2_w1024.log 2>&1 || { tail -5 $O/sq2_w1024.log; exit 1; }
python3 tools/pmc_issue.py $O/sq1_w1024 $O/sq2_w1024 $O/pmc_issue.json boolean_default
B="--steps 1 --warmup 0 --no-cpu-baseline --no-other-modes --no-configs"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmcTF -o f -- python3 bench.py $B > $O/pmcTF.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmcTW -o w -- python3 bench.py $B > $O/pmcTW.log 2>&1 &&
python3 tools/pmc_traffic.py $O/pmcTF $O/pmcTW $O/pmc_traffic.json
cat $O/pmc_traffic.json
