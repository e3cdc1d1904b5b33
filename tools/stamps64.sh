# per-phase s_memtime stamps of the 64-bit bootstrap kernel k_pbs64s: builds a SEPARATE library libhelm_hip_stamps.so
# (-DHELM_WIDE_STAMPS, helm_shortint.hip as one translation unit, the other objects as the Makefile built them) on the box and
# runs one batch of B three-input LUT bootstraps through it (HELM_HIP_LIB).  Stamps add ~10 % to a step.
# usage: bash tools/stamps64.sh [set = shortint_m2c2] [B = 256]     (on the GPU box, from the repository root)
SET=${1:-shortint_m2c2}; B=${2:-256}
cd "${GRAFT_REPO_ROOT:?}/helm_amd/csrc" || exit 1
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC -DHELM_WIDE_STAMPS -Wno-unused-function -c -o /tmp/helm_shortint_stamps.o helm_shortint.hip &&
hipcc -O3 --offload-arch=gfx950 -fPIC -shared -o libhelm_hip_stamps.so helm_hip.o helm_hip_wide.o /tmp/helm_shortint_stamps.o helm_comm.o -ldl &&
cd "$GRAFT_REPO_ROOT" && HELM_HIP_LIB=libhelm_hip_stamps.so timeout -k 10 300 python3 - $SET $B <<'PY'
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
print("decrypt_ok", bool(np.array_equal(ck.decrypt(w.download(out)), (bits[:B] + bits[B:2 * B] + bits[2 * B:]) >= 2)))
PY
rm -f "$GRAFT_REPO_ROOT/helm_amd/csrc/libhelm_hip_stamps.so"
