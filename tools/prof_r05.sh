# Round-5 profiles of the build that is benchmarked.  Run on the GPU box from the repository root:
#   bash tools/prof_r05.sh [steps...]   -> gpurun_out/prof_r05/*   (copy the summaries into profiles/r05/)
# Every rocprofv3 call has the program itself after `--`; --pmc passes carry --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}" || exit 1
O=gpurun_out/prof_r05; mkdir -p $O
STEPS="${*:-0 1 2 3 4 5 6 7}"
has() { case " $STEPS " in *" $1 "*) return 0;; esac; return 1; }
BENCH="--steps 1 --warmup 1 --no-cpu-baseline --no-other-modes --no-configs"
if has 0; then
echo "== 0 the bench line itself, default arguments (same box as everything below)"
timeout -k 10 600 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -5 $O/bench_n1.err; exit 1; }
fi
if has 1; then
echo "== 1 kernel-trace stats, gates mode (bench.py $BENCH)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py $BENCH > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
fi
if has 2; then
echo "== 2 kernel-trace stats of the 64-bit-torus kernels of this build: 1,024 three-input LUTs (m2c2, multi-bit), 2,048 two-input (m1c1), chi-squared u32"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lut_m2c2 -o lut_m2c2 -- python3 tools/prof_luts.py 1024 3 shortint_m2c2 > $O/lut_m2c2.log 2>&1 || tail -5 $O/lut_m2c2.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lut_mb -o lut_mb -- python3 tools/prof_luts.py 1024 3 shortint_m2c2_multibit3 > $O/lut_mb.log 2>&1 || tail -5 $O/lut_mb.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lut_m1c1 -o lut_m1c1 -- python3 tools/prof_luts.py 2048 3 shortint_m1c1 2 > $O/lut_m1c1.log 2>&1 || tail -5 $O/lut_m1c1.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/chi_mb -o chi_mb -- python3 tools/prof_chi.py shortint_m2c2_multibit3 > $O/chi_mb.log 2>&1 || tail -5 $O/chi_mb.log
fi
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
if has 3; then
echo "== 3 issue-slot and LDS counters: lockstep k_pbs (1,024; -> pmc_issue.json), k_pbs_wide (256), and the 64-bit kernels"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_w1024 -o a -- python3 tools/prof_pbs.py boolean_default 1024 3 > $O/sq1_w1024.log 2>&1 || tail -5 $O/sq1_w1024.log
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2_w1024 -o b -- python3 tools/prof_pbs.py boolean_default 1024 3 > $O/sq2_w1024.log 2>&1 || tail -5 $O/sq2_w1024.log
python3 tools/pmc_issue.py $O/sq1_w1024 $O/sq2_w1024 $O/pmc_issue.json boolean_default
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_w256 -o a -- python3 tools/prof_pbs.py boolean_default 256 3 > $O/sq1_w256.log 2>&1 || tail -5 $O/sq1_w256.log
for S in shortint_m2c2 shortint_m2c2_multibit3; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_$S -o a -- python3 tools/prof_luts.py 256 2 $S > $O/sq1_$S.log 2>&1 || tail -5 $O/sq1_$S.log
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2_$S -o b -- python3 tools/prof_luts.py 256 2 $S > $O/sq2_$S.log 2>&1 || tail -5 $O/sq2_$S.log
done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_m1c1 -o a -- python3 tools/prof_luts.py 512 2 shortint_m1c1 2 > $O/sq1_m1c1.log 2>&1 || tail -5 $O/sq1_m1c1.log
fi
if has 4; then
echo "== 4 fabric traffic of the lockstep k_pbs launches at the bench's launch sizes"
B="--steps 1 --warmup 0 --no-cpu-baseline --no-other-modes --no-configs"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmcTF -o f -- python3 bench.py $B > $O/pmcTF.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmcTW -o w -- python3 bench.py $B > $O/pmcTW.log 2>&1 &&
python3 tools/pmc_traffic.py $O/pmcTF $O/pmcTW $O/pmc_traffic.json
fi
if has 5; then
echo "== 5 the exchange machinery on one GPU: bench.py --force-comm (world-size-1 RCCL communicator inside the library), in order and overlapped"
timeout -k 10 400 python3 bench.py --force-comm --no-cpu-baseline --no-other-modes --no-configs > $O/bench_n1_forcecomm.json 2> $O/bench_fc.err || tail -5 $O/bench_fc.err
timeout -k 10 400 python3 bench.py --force-comm --overlap --no-cpu-baseline --no-other-modes --no-configs > $O/bench_n1_forcecomm_overlap.json 2> $O/bench_fco.err || tail -5 $O/bench_fco.err
fi
if has 6; then
echo "== 6 the 8(d) micro-benchmark table"
rm -f $O/microbench.jsonl
timeout -k 10 900 python3 tools/microbench_gates.py --out $O/microbench.jsonl > $O/microbench.log 2>&1 || tail -5 $O/microbench.log
fi
if has 7; then
echo "== 7 which memory-side counters gfx950 offers (is there one that separates Infinity-Cache hits from HBM reads?)"
rocprofv3 -L > $O/counters_available.txt 2>&1
grep -i -E "mall|dram|hbm|TCC_EA|TCC_HIT|TCC_MISS|TCC_REQ\b|TCC_READ\b" $O/counters_available.txt | cut -c1-200 > $O/counters_memory_side.txt
wc -l $O/counters_memory_side.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum -d $O/tcc_dram -o t -- python3 tools/prof_pbs.py boolean_default 4096 3 > $O/tcc_dram.log 2>&1 || tail -5 $O/tcc_dram.log
fi
echo "== summaries"
python3 tools/prof_r03_summary.py $O > $O/summary.txt 2>&1; cat $O/summary.txt | cut -c1-260 | head -150
