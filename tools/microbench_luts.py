"""µ-bench: B independent 3-input LUT gates (gates::lut, KS + PBS each) on the shortint engine."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import helm_amd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "shortint_m2c2"
Bs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 64, 256, 512, 1024]
ck = helm_amd.SiClientKey.generate(name, seed=1)
sk = helm_amd.SiServerKey(ck)
rng = np.random.default_rng(0)
maxB = max(Bs)
bits = rng.integers(0, 2, size=3 * maxB).astype(np.uint64)
w = sk.wires(4 * maxB)
w.upload(np.arange(3 * maxB), ck.encrypt(bits))
sk.timing_enable(True)
for B in Bs:
    in_idx = np.stack([np.arange(B), maxB + np.arange(B), 2 * maxB + np.arange(B)], axis=1).astype(np.int32)
    out = np.arange(3 * maxB, 3 * maxB + B, dtype=np.int32)
    ar = np.full(B, 3, np.int32)
    tb = np.full(B, 0xE8, np.uint64)
    w.eval_lut_level(ar, in_idx, tb, out); sk.sync()
    sk.timing(reset=True)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        w.eval_lut_level(ar, in_idx, tb, out)
    sk.sync()
    dt = (time.perf_counter() - t0) / reps
    t = sk.timing(reset=True)
    dec = ck.decrypt(w.download(out))
    ok = np.array_equal(dec, (bits[:B] + bits[maxB:maxB + B] + bits[2 * maxB:2 * maxB + B]) >= 2)
    print(f"{name} B={B:6d} wall {dt*1e3:9.3f} ms  pbs {t.pbs_ms/reps:9.3f} ms  ks {t.ks_ms/reps:8.3f} ms  "
          f"lin {t.linear_ms/reps:7.3f} ms {B/dt:10.1f} LUTs/s  decrypt_ok={ok}", flush=True)
