import sys, time, numpy as np
sys.path.insert(0, ".")
import helm_amd, oracle
ck = helm_amd.ClientKey.generate("boolean_default", seed=1)
orc = oracle.Oracle(ck.params.as_tuple7(), ck.bsk, ck.ksk)
B = 256
bits = np.random.default_rng(0).integers(0, 2, size=2 * B).astype(bool)
wires = np.zeros((3 * B, ck.params.n + 1), dtype=np.uint32)
wires[:2 * B] = ck.encrypt(bits)
for th in (16, 32, 64, 128, 256):
    t0 = time.perf_counter()
    orc.eval_level(wires, np.full(B, oracle.NAND, np.int32), np.arange(B), np.arange(B, 2 * B), np.full(B, -1), np.arange(2 * B, 3 * B), nthreads=th)
    dt = time.perf_counter() - t0
    print(th, "threads:", round(B / dt, 1), "gates/s", flush=True)
