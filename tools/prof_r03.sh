# Round-3 profiles of the build that is benchmarked.  Run on the GPU box from the repository root:
#   bash tools/prof_r03.sh            -> gpurun_out/prof_r03/*   (copy the summaries into profiles/r03/)
# Every rocprofv3 call has the program itself after `--`; --pmc passes carry --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03; mkdir -p $O
BENCH="--steps 1 --warmup 1 --no-cpu-baseline --no-other-modes"
SMALL="--steps 1 --warmup 0 --blocks 4 --no-cpu-baseline --no-other-modes"
echo "== 1 kernel-trace stats, gates mode (bench.py $BENCH)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 bench.py $BENCH > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
echo "== 2 kernel-trace stats, LUT mode (1,024 three-input LUTs x 3; classical and multi-bit sets; 2,048 two-input LUTs x 3 on the binary's set)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lut -o lut -- python3 tools/prof_luts.py 1024 3 > $O/lut.log 2>&1 || { tail -5 $O/lut.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lutmb -o lutmb -- python3 tools/prof_luts.py 1024 3 shortint_m2c2_multibit3 > $O/lutmb.log 2>&1 || { tail -5 $O/lutmb.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lutm1 -o lutm1 -- python3 tools/prof_luts.py 2048 3 shortint_m1c1 2 > $O/lutm1.log 2>&1 || { tail -5 $O/lutm1.log; exit 1; }
echo "== 2b kernel-trace stats, arithmetic mode (chi-squared u32, default evaluation, classical set)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/chi -o chi -- python3 tools/prof_chi.py > $O/chi.log 2>&1 || { tail -5 $O/chi.log; exit 1; }
echo "== 3 kernel-trace stats, WoP-PBS wide gates (256 six-input gates, two bits per block, x 3)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wop -o wop -- python3 tools/wop_bench.py 256 6 2 > $O/wop.log 2>&1 || { tail -5 $O/wop.log; exit 1; }
echo "== 4 matrix-core counters: k_ks_mfma (gates mode), k_ks64_mfma (LUT mode and the WoP packing keyswitch)"
MF="SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc $MF -d $O/mfma_gates -o m -- python3 bench.py $SMALL > $O/mfma_gates.log 2>&1 || tail -5 $O/mfma_gates.log
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $MF -d $O/mfma_lut -o m -- python3 tools/prof_luts.py 1024 2 > $O/mfma_lut.log 2>&1 || tail -5 $O/mfma_lut.log
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc $MF -d $O/mfma_wop -o m -- python3 tools/wop_bench.py 256 6 2 > $O/mfma_wop.log 2>&1 || tail -5 $O/mfma_wop.log
echo "== 5 issue slots of k_pbs64s (classical and multi-bit) and of the lockstep k_pbs"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_lut -o a -- python3 tools/prof_luts.py 256 2 > $O/sq1_lut.log 2>&1 || tail -5 $O/sq1_lut.log
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2_lut -o b -- python3 tools/prof_luts.py 256 2 > $O/sq2_lut.log 2>&1 || tail -5 $O/sq2_lut.log
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_lutmb -o a -- python3 tools/prof_luts.py 256 2 shortint_m2c2_multibit3 > $O/sq1_lutmb.log 2>&1 || tail -5 $O/sq1_lutmb.log
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_lutm1 -o a -- python3 tools/prof_luts.py 512 2 shortint_m1c1 2 > $O/sq1_lutm1.log 2>&1 || tail -5 $O/sq1_lutm1.log
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1_gates -o a -- python3 bench.py $SMALL > $O/sq1_gates.log 2>&1 || tail -5 $O/sq1_gates.log
echo "== 6 fabric traffic of the lockstep k_pbs launches at the bench's launch sizes"
B="--steps 1 --warmup 0 --no-cpu-baseline --no-other-modes"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmcTF -o f -- python3 bench.py $B > $O/pmcTF.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmcTW -o w -- python3 bench.py $B > $O/pmcTW.log 2>&1 &&
python3 tools/pmc_traffic.py $O/pmcTF $O/pmcTW $O/pmc_traffic.json
echo "== summaries"
python3 tools/prof_r03_summary.py $O > $O/summary.txt 2>&1; cat $O/summary.txt | cut -c1-260
