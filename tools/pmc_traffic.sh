# HBM-side traffic of the lockstep k_pbs launches at the bench's real launch sizes.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B="--steps 1 --warmup 0 --no-cpu-baseline --no-other-modes"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/pmcTF -o f -- python3 bench.py $B > gpurun_out/pmcTF.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d gpurun_out/pmcTW -o w -- python3 bench.py $B > gpurun_out/pmcTW.log 2>&1 &&
python3 tools/pmc_traffic.py gpurun_out/pmcTF gpurun_out/pmcTW gpurun_out/pmc_traffic.json
