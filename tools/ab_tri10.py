#!/usr/bin/env python3
"""Same-box, same-process A/B of what serves launches of 513..768 bootstraps at N = 1024 (helm_cuda, reference
src/bin/helm.rs:141-146): a lockstep round with one SIMD in four empty (HELM_HIP_TRIO=0: rounds 1-5) against k_pbs_tri10
(round 6: three bootstraps per workgroup, twelve (polynomial, transform half) waves) with its issue priorities on / off, in the
lazy field FpI and in the 51-bit field; a full lockstep round (1,024) and k_pbs_duo (512) beside them for the width table of
helm_hip_launch_costs.  The engine reads its switches when a context is created: one context per setting, launches alternating.
usage: ab_tri10.py [B ...]   -> one JSON line per (setting, B): best and median of the rounds, digest of the ciphertexts"""
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helm_amd  # noqa: E402

KEYS = ("HELM_HIP_TRIO", "HELM_HIP_TRIO_FLAGS", "HELM_HIP_FIELD", "HELM_HIP_DUO1024")
SETTINGS = [("FpI: lockstep round (HELM_HIP_TRIO=0)", {"HELM_HIP_TRIO": "0"}),
            ("FpI: k_pbs_tri10, priorities on", {"HELM_HIP_TRIO": "1", "HELM_HIP_TRIO_FLAGS": "1"}),
            ("FpI: k_pbs_tri10, priorities off", {"HELM_HIP_TRIO": "1", "HELM_HIP_TRIO_FLAGS": "0"}),
            ("FpI: k_pbs_tri10 three AND two per workgroup (HELM_HIP_DUO1024=3)", {"HELM_HIP_TRIO": "1", "HELM_HIP_DUO1024": "3"}),
            ("FpH: lockstep round (HELM_HIP_TRIO=0)", {"HELM_HIP_TRIO": "0", "HELM_HIP_FIELD": "51"}),
            ("FpH: k_pbs_tri10, priorities on", {"HELM_HIP_TRIO": "1", "HELM_HIP_TRIO_FLAGS": "1", "HELM_HIP_FIELD": "51"})]
Bs = [int(x) for x in sys.argv[1:]] or [300, 400, 512, 600, 768, 1024]
ck = helm_amd.ClientKey.generate("helm_cuda", seed=1)
maxB = max(Bs)
bits = np.random.default_rng(0).integers(0, 2, size=2 * maxB).astype(bool)
enc = ck.encrypt(bits)
ctxs = []
for name, env in SETTINGS:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    sk = helm_amd.ServerKey(ck)
    w = sk.wires(3 * maxB)
    w.upload(np.arange(2 * maxB), enc)
    progs = {B: helm_amd.Program(sk, np.full(B, 4, np.int32), np.arange(B), np.arange(maxB, maxB + B), np.full(B, -1),
                                 np.arange(2 * maxB, 2 * maxB + B), [0, B]) for B in Bs}
    ctxs.append((name, sk, w, progs))
times = {(n, B): [] for n, *_ in ctxs for B in Bs}
for rnd in range(7):
    for name, sk, w, progs in ctxs:
        for B in Bs:
            progs[B].run(w)
            sk.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                progs[B].run(w)
            sk.sync()
            times[(name, B)].append((time.perf_counter() - t0) / 3 * 1e3)
for name, sk, w, progs in ctxs:
    for B in Bs:
        progs[B].run(w)
        sk.sync()
        out = w.download(np.arange(2 * maxB, 2 * maxB + B))
        ok = bool(np.array_equal(ck.decrypt(out), ~(bits[:B] & bits[maxB:maxB + B])))
        t = sorted(times[(name, B)])
        print(json.dumps({"setting": name, "B": B, "field": sk.field_bits(), "best_ms": round(t[0], 3), "median_ms": round(t[len(t) // 2], 3),
                          "decrypt_ok": ok, "sha": hashlib.sha256(out.tobytes()).hexdigest()[:12]}))
