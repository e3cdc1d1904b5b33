# kernel-trace statistics of the bench command (1 step + 1 warm-up), round 2
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02 -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-modes > gpurun_out/prof_r02.log 2>&1 &&
head -12 gpurun_out/prof_r02/*kernel_stats.csv | cut -c1-300
