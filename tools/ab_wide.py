"""Same-process A/B of k_pbs_wide wave placements: one context per HELM_HIP_WIDE_MAP value, alternating launches
of B bootstraps.  Usage: [PARAMS=boolean_default] [MAPS=0,1] ab_wide.py [B] [reps]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
import helm_amd, oracle  # noqa

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
MAPS = [int(x) for x in os.environ.get("MAPS", "0,1").split(",")]
PARAMS = os.environ.get("PARAMS", "boolean_default")
ck = helm_amd.ClientKey.generate(PARAMS, seed=1)
rng = np.random.default_rng(0)
bits = rng.integers(0, 2, size=2 * B).astype(bool)
cts = ck.encrypt(bits)
sks, progs, res = {}, {}, {m: [] for m in MAPS}
for m in MAPS:
    os.environ["HELM_HIP_WIDE_MAP"] = str(m)
    sks[m] = helm_amd.ServerKey(ck)
    w = sks[m].wires(3 * B)
    w.upload(np.arange(2 * B), cts)
    progs[m] = (helm_amd.Program(sks[m], np.full(B, oracle.NAND, np.int32), np.arange(B), np.arange(B, 2 * B),
                                 np.full(B, -1), np.arange(2 * B, 3 * B), [0, B]), w)
for rnd in range(6):
    for m in MAPS:
        prog, w = progs[m]
        prog.run(w)
        sks[m].sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            prog.run(w)
        sks[m].sync()
        res[m].append((time.perf_counter() - t0) / reps * 1e3)
tabs = [progs[m][1].download() for m in MAPS]
print("identical ciphertexts:", all(np.array_equal(tabs[0], t) for t in tabs[1:]))
for m in MAPS:
    print("map", m, "ms per launch:", " ".join(f"{x:.3f}" for x in res[m]), " median", f"{np.median(res[m]):.3f}")
