"""Profiling driver: one level of B NAND gates, repeated. Usage: prof_pbs.py <params> <B> <reps>"""
import sys
import numpy as np
sys.path.insert(0, ".")
import helm_amd  # noqa
NAND = 4  # HELM_GATE_NAND (include/helm_hip.h)
name, B, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
ck = helm_amd.ClientKey.generate(name, seed=1)
sk = helm_amd.ServerKey(ck)
rng = np.random.default_rng(0)
bits = rng.integers(0, 2, size=2 * B).astype(bool)
w = sk.wires(3 * B)
w.upload(np.arange(2 * B), ck.encrypt(bits))
prog = helm_amd.Program(sk, np.full(B, NAND, np.int32), np.arange(B), np.arange(B, 2 * B), np.full(B, -1),
                        np.arange(2 * B, 3 * B), [0, B])
for _ in range(reps):
    prog.run(w)
sk.sync()
print("ok", np.array_equal(ck.decrypt(w.download(np.arange(2 * B, 3 * B))), ~(bits[:B] & bits[B:])))
