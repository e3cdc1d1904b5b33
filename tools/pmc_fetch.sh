cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export HELM_HIP_PBS_VARIANT=${V:-5}
timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/pmcF -o f -- python3 tools/prof_pbs.py boolean_default 1024 2 > gpurun_out/pmcF.log 2>&1; echo "rc=$?"
python3 tools/pmc_summary.py gpurun_out/pmcF k_pbs
timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d gpurun_out/pmcW -o w -- python3 tools/prof_pbs.py boolean_default 1024 2 > gpurun_out/pmcW.log 2>&1; echo "rc=$?"
python3 tools/pmc_summary.py gpurun_out/pmcW k_pbs
