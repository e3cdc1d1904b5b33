"""Repeat one level of B gates many times: every repetition must give bit-identical ciphertexts."""
import sys
import numpy as np
sys.path.insert(0, ".")
import helm_amd  # noqa: E402
import oracle  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "boolean_default"
Bs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 3, 64, 256, 700, 1024]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ck = helm_amd.ClientKey.generate(name, seed=1)
sk = helm_amd.ServerKey(ck)
rng = np.random.default_rng(0)
maxB = max(Bs)
bits = rng.integers(0, 2, size=3 * maxB).astype(bool)
w = sk.wires(4 * maxB)
w.upload(np.arange(3 * maxB), ck.encrypt(bits))
bad = 0
for B in Bs:
    ops = np.where(np.arange(B) % 3 == 0, oracle.MUX, np.where(np.arange(B) % 3 == 1, oracle.NAND, oracle.XOR)).astype(np.int32)
    i0 = np.arange(B, dtype=np.int32)
    i1 = np.arange(maxB, maxB + B, dtype=np.int32)
    i2 = np.where(ops == oracle.MUX, np.arange(2 * maxB, 2 * maxB + B), -1).astype(np.int32)
    out = np.arange(3 * maxB, 3 * maxB + B, dtype=np.int32)
    prog = helm_amd.Program(sk, ops, i0, i1, i2, out, [0, B])
    ref = None
    mism = 0
    for r in range(reps):
        prog.run(w)
        sk.sync()
        got = w.download(out)
        if ref is None:
            ref = got
        elif not np.array_equal(ref, got):
            mism += 1
            rows = np.nonzero((ref != got).any(axis=1))[0]
            print(f"  B={B} rep {r}: {len(rows)} rows differ, first rows {rows[:8]}", flush=True)
    bad += mism
    print(f"{name} B={B}: {mism}/{reps - 1} repetitions differ", flush=True)
print("DETERMINISTIC" if bad == 0 else "NONDETERMINISTIC")
