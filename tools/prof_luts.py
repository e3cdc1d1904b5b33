"""Profiling driver: one level of B 3-input LUT gates, repeated. Usage: prof_luts.py <B> <reps> [set = shortint_m2c2]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import helm_amd  # noqa
B, reps = int(sys.argv[1]), int(sys.argv[2])
ck = helm_amd.SiClientKey.generate(sys.argv[3] if len(sys.argv) > 3 else "shortint_m2c2", seed=1)
sk = helm_amd.SiServerKey(ck)
bits = np.random.default_rng(0).integers(0, 2, size=3 * B).astype(np.uint64)
w = sk.wires(4 * B)
w.upload(np.arange(3 * B), ck.encrypt(bits))
in_idx = np.arange(3 * B, dtype=np.int32).reshape(3, B).T.copy()
for _ in range(reps):
    w.eval_lut_level(np.full(B, 3, np.int32), in_idx, np.full(B, 0xE8, np.uint64), np.arange(3 * B, 4 * B))
sk.sync()
print("ok", np.array_equal(ck.decrypt(w.download(np.arange(3 * B, 4 * B))), (bits[:B] + bits[B:2 * B] + bits[2 * B:]) >= 2))
