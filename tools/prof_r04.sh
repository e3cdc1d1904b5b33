# Round-4 profiles of the build that is benchmarked.  Run on the GPU box from the repository root:
#   bash tools/prof_r04.sh            -> gpurun_out/prof_r04/*   (copy the summaries into profiles/r04/)
# Every rocprofv3 call has the program itself after `--`; --pmc passes carry --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/prof_r04; mkdir -p $O
BENCH="--steps 1 --warmup 1 --no-cpu-baseline --no-other-modes"
SMALL="--steps 1 --warmup 0 --blocks 4 --no-cpu-baseline --no-other-modes"
echo "== 0 the bench line itself, default arguments (same box as everything below)"
timeout -k 10 600 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -5 $O/bench_n1.err; exit 1; }
echo "== 1 kernel-trace stats, gates mode (bench.py $BENCH)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py $BENCH > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
echo "== 2 kernel-trace stats of one launch per width: 256 (wide), 512 (duo), 768 (trio), 1,024 (lockstep), 5 launches each"
for B in 256 512 768 1024; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/w$B -o w$B -- python3 tools/prof_pbs.py boolean_default $B 5 > $O/w$B.log 2>&1 || { tail -5 $O/w$B.log; exit 1; }
done
echo "== 3 issue-slot and LDS counters of k_pbs_duo (512), k_pbs_trio (768), the lockstep k_pbs (1,024) and k_pbs_wide (256)"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
for B in 256 512 768 1024; do
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_w$B -o a -- python3 tools/prof_pbs.py boolean_default $B 3 > $O/sq1_w$B.log 2>&1 || tail -5 $O/sq1_w$B.log
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2_w$B -o b -- python3 tools/prof_pbs.py boolean_default $B 3 > $O/sq2_w$B.log 2>&1 || tail -5 $O/sq2_w$B.log
done
echo "== 4 fabric traffic of the lockstep k_pbs launches at the bench's launch sizes"
B="--steps 1 --warmup 0 --no-cpu-baseline --no-other-modes"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmcTF -o f -- python3 bench.py $B > $O/pmcTF.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmcTW -o w -- python3 bench.py $B > $O/pmcTW.log 2>&1 &&
python3 tools/pmc_traffic.py $O/pmcTF $O/pmcTW $O/pmc_traffic.json
echo "== 5 the exchange machinery on one GPU: bench.py --force-comm (world-size-1 RCCL communicator inside the library)"
timeout -k 10 400 python3 bench.py --force-comm --no-cpu-baseline --no-other-modes > $O/bench_n1_forcecomm.json 2> $O/bench_fc.err || tail -5 $O/bench_fc.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fc -o fc -- python3 bench.py --force-comm --steps 1 --warmup 0 --blocks 8 --no-cpu-baseline --no-other-modes > $O/fc.log 2>&1 || tail -5 $O/fc.log
echo "== 6 the 8(d) micro-benchmark table"
rm -f $O/microbench.jsonl
timeout -k 10 900 python3 tools/microbench_gates.py --out $O/microbench.jsonl > $O/microbench.log 2>&1 || tail -5 $O/microbench.log
echo "== summaries"
python3 tools/prof_r03_summary.py $O > $O/summary.txt 2>&1; cat $O/summary.txt | cut -c1-260
