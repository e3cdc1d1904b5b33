# kernel-trace statistics of the WoP-PBS wide-LUT path: 256 six-input gates, two bits per block (three passes, the first is the warm-up)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_wop -o wop -- python3 tools/wop_bench.py 256 6 2 > gpurun_out/prof_wop.log 2>&1 &&
head -16 gpurun_out/prof_wop/*kernel_stats.csv | cut -c1-260
