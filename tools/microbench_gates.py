"""µ-bench: B independent NAND gates on fresh encryptions, gates/s vs B."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import helm_amd  # noqa: E402
import oracle  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "boolean_default"
Bs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 64, 256, 512, 768, 1024, 4096]
ck = helm_amd.ClientKey.generate(name, seed=1)
sk = helm_amd.ServerKey(ck)
p = ck.params
rng = np.random.default_rng(0)
maxB = max(Bs)
bits = rng.integers(0, 2, size=2 * maxB).astype(bool)
w = sk.wires(3 * maxB)
w.upload(np.arange(2 * maxB), ck.encrypt(bits))
sk.timing_enable(True)
for B in Bs:
    ops = np.full(B, oracle.NAND, dtype=np.int32)
    i0 = np.arange(B, dtype=np.int32)
    i1 = np.arange(maxB, maxB + B, dtype=np.int32)
    i2 = np.full(B, -1, dtype=np.int32)
    out = np.arange(2 * maxB, 2 * maxB + B, dtype=np.int32)
    prog = helm_amd.Program(sk, ops, i0, i1, i2, out, [0, B])
    prog.run(w); sk.sync()
    sk.timing(reset=True)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        prog.run(w)
    sk.sync()
    dt = (time.perf_counter() - t0) / reps
    t = sk.timing(reset=True)
    dec = ck.decrypt(w.download(out))
    ok = np.array_equal(dec, ~(bits[:B] & bits[maxB:maxB + B]))
    print(f"{name} B={B:6d} wall {dt*1e3:9.3f} ms  pbs {t.pbs_ms/reps:9.3f} ms  ks {t.ks_ms/reps:8.3f} ms  "
          f"{B/dt:10.1f} gates/s  decrypt_ok={ok}", flush=True)
