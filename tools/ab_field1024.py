#!/usr/bin/env python3
"""Same-box, same-process A/B of the two fields an N = 1024 context can compute in (round 5): the lazy FpI (p = 5440^4 + 1,
chosen by helm_hip_load_bootstrap_key when the loaded key's own bound fits) against the 51-bit FpH (HELM_HIP_FIELD=51), on
helm_cuda (reference src/bin/helm.rs:141-146) at the width table's launch sizes.  One context per field, launches alternating;
the ciphertexts must be identical (both fields compute the same exact integers).
usage: ab_field1024.py [B ...]"""
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helm_amd  # noqa: E402

Bs = [int(x) for x in sys.argv[1:]] or [256, 512, 768, 1024, 4096]
ck = helm_amd.ClientKey.generate("helm_cuda", seed=1)
maxB = max(Bs)
bits = np.random.default_rng(0).integers(0, 2, size=2 * maxB).astype(bool)
enc = ck.encrypt(bits)
ctxs = []
for name, env in (("FpI (lazy, 5440^4 + 1)", None), ("FpH (51-bit, HELM_HIP_FIELD=51)", "51")):
    os.environ.pop("HELM_HIP_FIELD", None)
    if env:
        os.environ["HELM_HIP_FIELD"] = env
    sk = helm_amd.ServerKey(ck)
    w = sk.wires(3 * maxB)
    w.upload(np.arange(2 * maxB), enc)
    progs = {B: helm_amd.Program(sk, np.full(B, 4, np.int32), np.arange(B), np.arange(maxB, maxB + B), np.full(B, -1),
                                 np.arange(2 * maxB, 2 * maxB + B), [0, B]) for B in Bs}
    ctxs.append((name, sk, w, progs, sk.field_bits()))
os.environ.pop("HELM_HIP_FIELD", None)
times = {(n, B): [] for n, *_ in ctxs for B in Bs}
for rnd in range(7):
    for name, sk, w, progs, _ in ctxs:
        for B in Bs:
            progs[B].run(w)
            sk.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                progs[B].run(w)
            sk.sync()
            times[(name, B)].append((time.perf_counter() - t0) / 3 * 1e3)
for name, sk, w, progs, field in ctxs:
    for B in Bs:
        progs[B].run(w)
        sk.sync()
        out = w.download(np.arange(2 * maxB, 2 * maxB + B))
        ok = bool(np.array_equal(ck.decrypt(out), ~(bits[:B] & bits[maxB:maxB + B])))
        t = sorted(times[(name, B)])
        print(json.dumps({"field": name, "field_bits": field, "B": B, "best_ms": round(t[0], 3), "median_ms": round(t[len(t) // 2], 3),
                          "decrypt_ok": ok, "sha": hashlib.sha256(out.tobytes()).hexdigest()[:12]}))
