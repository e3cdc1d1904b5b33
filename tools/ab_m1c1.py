#!/usr/bin/env python3
"""Same-box, same-process A/B of k_pbs64k's CRT pair under the set the reference binary installs for LUT mode (src/bin/helm.rs:301,
PARAM_MESSAGE_1_CARRY_1_KS_PBS: k = 3, N = 512): the 49-bit pair of rounds 4-5 (HELM_SI_FIELD=49) against the 46-bit pair the
loaded key selects since round 6 (both leading forward stages plain on the 17-bit digits, one recentring per transpose of
the inverse transform).  2,048 independent 2-input LUT gates per level - bench.py's lut_mode_m1c1 leg -, the levels of the
two contexts alternating; kernel time from the engine's HIP events.  One JSON line per setting; the ciphertexts must agree."""
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helm_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ck = helm_amd.SiClientKey.generate("shortint_m1c1", seed=1)
bits = np.random.default_rng(0).integers(0, 2, size=2 * B).astype(np.uint64)
enc = ck.encrypt(bits)
in_idx = np.arange(2 * B, dtype=np.int32).reshape(2, B).T.copy()
ar, tb, out = np.full(B, 2, np.int32), np.full(B, 0x6, np.uint64), np.arange(2 * B, 3 * B, dtype=np.int32)
ctxs = []
for name, env in (("49-bit pair (HELM_SI_FIELD=49)", {"HELM_SI_FIELD": "49"}), ("46-bit pair (the loaded key's choice)", {})):
    os.environ.pop("HELM_SI_FIELD", None)
    os.environ.update(env)
    sk = helm_amd.SiServerKey(ck)
    w = sk.wires(3 * B)
    w.upload(np.arange(2 * B), enc)
    w.eval_lut_level(ar, in_idx, tb, out)
    sk.sync()
    ctxs.append((name, sk, w, [], []))
os.environ.pop("HELM_SI_FIELD", None)
for rnd in range(7):
    for name, sk, w, wall, kern in ctxs:
        sk.timing_enable(True)
        sk.timing(reset=True)
        t0 = time.perf_counter()
        w.eval_lut_level(ar, in_idx, tb, out)
        sk.sync()
        wall.append((time.perf_counter() - t0) * 1e3)
        kern.append(sk.timing(reset=True).pbs_ms)
        sk.timing_enable(False)
for name, sk, w, wall, kern in ctxs:
    got = w.download(out)
    ok = bool(np.array_equal(ck.decrypt(got), bits[:B] ^ bits[B:]))
    print(json.dumps({"setting": name, "field_bits": sk.field_bits(), "luts": B, "k_pbs64k_ms_best": round(min(kern), 3),
                      "k_pbs64k_ms_median": round(sorted(kern)[len(kern) // 2], 3), "level_wall_ms_best": round(min(wall), 3),
                      "luts_per_s_best": round(B / min(wall) * 1e3, 1), "decrypt_ok": ok, "sha": hashlib.sha256(got.tobytes()).hexdigest()[:12]}))
