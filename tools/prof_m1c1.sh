cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_m1c1; mkdir -p $O
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS"
SQ2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o st -- python3 tools/prof_luts.py 2048 3 shortint_m1c1 2 > $O/st.log 2>&1 || { tail -5 $O/st.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ1 -d $O/sq1 -o a -- python3 tools/prof_luts.py 512 2 shortint_m1c1 2 > $O/sq1.log 2>&1 || tail -5 $O/sq1.log
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $SQ2 -d $O/sq2 -o b -- python3 tools/prof_luts.py 512 2 shortint_m1c1 2 > $O/sq2.log 2>&1 || tail -5 $O/sq2.log
find $O -name "*.csv" | head -20
