# per-phase s_memtime stamps of the 64-bit bootstrap kernels (k_pbs64s): rebuilds libhelm_hip.so with -DHELM_WIDE_STAMPS
# in the box's scratch copy of the repository, then runs one batch of B LUT bootstraps.  Usage: stamps64.sh [set] [B]
SET=${1:-shortint_m2c2}; B=${2:-256}
cd $GRAFT_REPO_ROOT/helm_amd/csrc &&
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -DHELM_WIDE_STAMPS -c -o /tmp/helm_shortint_stamps.o helm_shortint.hip &&
hipcc -O3 --offload-arch=gfx950 -fPIC -shared -o libhelm_hip.so helm_hip.o /tmp/helm_shortint_stamps.o &&
cd $GRAFT_REPO_ROOT && timeout -k 10 300 python3 - $SET $B <<'PY'
import sys, time, numpy as np
import helm_amd
name, B = sys.argv[1], int(sys.argv[2])
ck, sk = helm_amd.gen_keys_shortint(name, seed=1)
bits = np.random.default_rng(0).integers(0, 2, size=3 * B).astype(np.uint64)
w = sk.wires(4 * B)
w.upload(np.arange(3 * B), ck.encrypt(bits))
in_idx = np.arange(3 * B, dtype=np.int32).reshape(3, B).T.copy()
ar, tb, out = np.full(B, 3, np.int32), np.full(B, 0xE8, np.uint64), np.arange(3 * B, 4 * B, dtype=np.int32)
for _ in range(2):
    t0 = time.perf_counter(); w.eval_lut_level(ar, in_idx, tb, out); sk.sync(); print("ms", (time.perf_counter() - t0) * 1e3, flush=True)
PY
