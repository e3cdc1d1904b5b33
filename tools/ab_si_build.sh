# same-box A/B of a compile-time switch of helm_shortint.hip: builds the alternative in the box's scratch copy and runs
# the LUT micro-benchmark with both libraries (or, with a second argument, that command: e.g. "python3 tools/wop_bench.py
# 256 6 1" for the two-level build).  Usage: ab_si_build.sh "-DHELM_SI_PRIO=0" ["command"]
ALT="$1"; CMD="$2"
cd $GRAFT_REPO_ROOT/helm_amd/csrc && cp libhelm_hip.so /tmp/libhelm_hip_base.so &&
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -fPIC $ALT -c -o /tmp/helm_shortint_alt.o helm_shortint.hip &&
hipcc -O3 --offload-arch=gfx950 -fPIC -shared -o /tmp/libhelm_hip_alt.so helm_hip.o /tmp/helm_shortint_alt.o &&
cd $GRAFT_REPO_ROOT &&
for round in 1 2; do
  for v in base alt; do
    cp /tmp/libhelm_hip_$v.so helm_amd/csrc/libhelm_hip.so
    echo "== $v (round $round)"
    if [ -n "$CMD" ]; then timeout -k 10 300 $CMD 2>&1 | tail -1 | cut -c1-330; continue; fi
    timeout -k 10 200 python3 - <<'PY'
import time, numpy as np
import helm_amd
for name, B in (("shortint_m2c2", 1024), ("shortint_m2c2", 64), ("shortint_m2c2_multibit3", 1024)):
    ck, sk = helm_amd.gen_keys_shortint(name, seed=1)
    bits = np.random.default_rng(0).integers(0, 2, size=3 * B).astype(np.uint64)
    w = sk.wires(4 * B)
    w.upload(np.arange(3 * B), ck.encrypt(bits))
    in_idx = np.arange(3 * B, dtype=np.int32).reshape(3, B).T.copy()
    ar, tb, out = np.full(B, 3, np.int32), np.full(B, 0xE8, np.uint64), np.arange(3 * B, 4 * B, dtype=np.int32)
    w.eval_lut_level(ar, in_idx, tb, out); sk.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); w.eval_lut_level(ar, in_idx, tb, out); sk.sync(); ts.append(time.perf_counter() - t0)
    ok = bool(np.array_equal(ck.decrypt(w.download(out)), (bits[:B] + bits[B:2 * B] + bits[2 * B:]) >= 2))
    print(name, B, "ms", round(min(ts) * 1e3, 3), "luts/s", round(B / min(ts), 1), ok, flush=True)
    sk.close()
PY
  done
done
cp /tmp/libhelm_hip_base.so helm_amd/csrc/libhelm_hip.so
