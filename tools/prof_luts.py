"""Profiling driver: one level of B LUT gates, repeated. Usage: prof_luts.py <B> <reps> [set = shortint_m2c2] [arity = 3]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import helm_amd  # noqa
B, reps = int(sys.argv[1]), int(sys.argv[2])
A = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ck = helm_amd.SiClientKey.generate(sys.argv[3] if len(sys.argv) > 3 else "shortint_m2c2", seed=1)
sk = helm_amd.SiServerKey(ck)
bits = np.random.default_rng(0).integers(0, 2, size=A * B).astype(np.uint64)
w = sk.wires((A + 1) * B)
w.upload(np.arange(A * B), ck.encrypt(bits))
in_idx = np.arange(A * B, dtype=np.int32).reshape(A, B).T.copy()
table = 0xE8 if A == 3 else 0x8  # majority / AND
for _ in range(reps):
    w.eval_lut_level(np.full(B, A, np.int32), in_idx, np.full(B, table, np.uint64), np.arange(A * B, (A + 1) * B))
sk.sync()
want = bits.reshape(A, B).sum(axis=0) >= (2 if A == 3 else A)
print("ok", np.array_equal(ck.decrypt(w.download(np.arange(A * B, (A + 1) * B))), want))
