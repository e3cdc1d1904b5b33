# per-phase s_memtime stamps of the boolean blind-rotate kernels (k_pbs lockstep by default): builds a SEPARATE library
# libhelm_hip_stamps.so (-DHELM_WIDE_STAMPS on the main boolean unit, the other objects as the Makefile built them) on the box
# and runs B NAND gates through it (HELM_HIP_LIB).  Stamps add ~10 % to a step.  Output: cycles per step and phase for the
# waves of the first and last workgroup ("work | bar1 | sum | bar2 | inverse | publish").
# usage: bash tools/stamps_pbs.sh [set = boolean_default] [B = 1024] [variant = 5 (lockstep)]   (GPU box, repository root)
SET=${1:-boolean_default}; B=${2:-1024}; V=${3:-5}
cd "${GRAFT_REPO_ROOT:?}/helm_amd/csrc" || exit 1
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -mllvm -amdgpu-sched-strategy=max-ilp -DHELM_HIP_SPLIT_TU=1 -DHELM_WIDE_STAMPS -c -o /tmp/helm_hip_stamps.o helm_hip.hip &&
hipcc -O3 --offload-arch=gfx950 -fPIC -shared -o libhelm_hip_stamps.so /tmp/helm_hip_stamps.o helm_hip_wide.o helm_shortint.o helm_shortint_ilp.o helm_comm.o -ldl &&
cd "$GRAFT_REPO_ROOT" && HELM_HIP_LIB=libhelm_hip_stamps.so HELM_HIP_PBS_VARIANT=$V HELM_HIP_VERBOSE=1 timeout -k 10 300 python3 tools/prof_pbs.py $SET $B 2 2>&1 | grep -v "^\[helm_hip\] k_pbs M" | tail -40
rm -f "$GRAFT_REPO_ROOT/helm_amd/csrc/libhelm_hip_stamps.so"
