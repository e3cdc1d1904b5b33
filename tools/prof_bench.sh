cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r01.log 2>&1 && tail -1 gpurun_out/bench_r01.log | cut -c1-1500 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01b -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_r01b.log 2>&1
ls gpurun_out/prof_r01b | head; head -8 gpurun_out/prof_r01b/*kernel_stats.csv | cut -c1-260
