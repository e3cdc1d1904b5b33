"""Timing of the WoP-PBS wide-LUT path on one GPU: batches of wide LUT gates, per-stage breakdown (HIP events).
Usage: python tools/wop_bench.py [count] [n_inputs] [bits_per_block] [wop set] """
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))

import helm_amd
from helm_amd import wopbs
from helm_amd.shortint import si_named_params

count = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = int(sys.argv[2]) if len(sys.argv) > 2 else 6
b = int(sys.argv[3]) if len(sys.argv) > 3 else 2
wop_name = sys.argv[4] if len(sys.argv) > 4 else "wopbs_m1c1"
sp, sa, sb = si_named_params("shortint_m2c2")
wp, wa, wb = wopbs.wop_named_params(wop_name)
sp.message_modulus, sp.carry_modulus = wp.message_modulus, wp.carry_modulus
ck = helm_amd.SiClientKey(sp, sa, sb, seed=1)
wk = wopbs.WopClientKey(ck, wp, wa, wb, seed=2)
sk = helm_amd.SiServerKey(ck)
wsk = wopbs.WopServerKey(sk, wk)
rng = np.random.default_rng(0)
basis = wp.message_modulus
truth = rng.integers(0, 2, size=basis ** m, dtype=np.uint64)
xs = rng.integers(0, 1 << m, size=count)
bits_in = np.array([[(x >> (m - 1 - q)) & 1 for q in range(m)] for x in xs], dtype=np.uint64)
w = sk.wires(count * (m + 1))
w.upload(np.arange(count * m), ck.encrypt(bits_in.reshape(-1)))
in_idx = np.arange(count * m, dtype=np.int32).reshape(count, m)
out_idx = np.arange(count * m, count * (m + 1), dtype=np.int32)
for rep in range(3):
    wsk.timing(reset=True)
    t0 = time.perf_counter()
    wsk.eval_luts(w, in_idx, truth, out_idx, bits_per_block=b)
    sk.sync()
    dt = time.perf_counter() - t0
    t = wsk.timing()
    got = ck.decrypt_message_and_carry(w.download(out_idx))
    want = [int(truth[sum(int(v) * basis ** j for j, v in enumerate(r[::-1]))]) for r in bits_in]
    print(json.dumps({"gates": count, "inputs": m, "bits_per_block": b, "wop": wop_name, "wall_s": round(dt, 4),
                      "gates_per_s": round(count / dt, 1), "bootstraps_per_s": round(t["bootstraps"] / dt, 1),
                      "decrypt_ok": [int(v) for v in got] == want,
                      **{k: round(v, 2) if isinstance(v, float) else v for k, v in t.items()}}))
