#!/usr/bin/env python3
"""Same-box, same-process A/B of the builds that serve launches of 257..512 bootstraps at N = 1024 (helm_cuda): the two-wave
build with all levels in flight (round 1), k_pbs_duo's compact layout in step / staggered under several priority settings.
The engine reads its switches when a context is created, so one context per setting, the launches alternating.
usage: ab_duo1024.py [B ...]   -> one line per (setting, B): best and median of the rounds, and a digest of the ciphertexts"""
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helm_amd  # noqa: E402

SETTINGS = [("two-wave (HELM_HIP_DUO1024=0)", {"HELM_HIP_DUO1024": "0"}),
            ("duo in step, flags 7", {"HELM_HIP_DUO1024": "1", "HELM_HIP_DUO1024_FLAGS": "7"}),
            ("duo in step, flags 1", {"HELM_HIP_DUO1024": "1", "HELM_HIP_DUO1024_FLAGS": "1"}),
            ("duo in step, flags 3", {"HELM_HIP_DUO1024": "1", "HELM_HIP_DUO1024_FLAGS": "3"}),
            ("duo in step, flags 0", {"HELM_HIP_DUO1024": "1", "HELM_HIP_DUO1024_FLAGS": "0"}),
            ("duo staggered, flags 7", {"HELM_HIP_DUO1024": "2", "HELM_HIP_DUO1024_FLAGS": "7"}),
            ("duo staggered, flags 1", {"HELM_HIP_DUO1024": "2", "HELM_HIP_DUO1024_FLAGS": "1"})]
Bs = [int(x) for x in sys.argv[1:]] or [300, 384, 512]
ck = helm_amd.ClientKey.generate("helm_cuda", seed=1)
maxB = max(Bs)
bits = np.random.default_rng(0).integers(0, 2, size=2 * maxB).astype(bool)
enc = ck.encrypt(bits)
ctxs = []
for name, env in SETTINGS:
    for k in ("HELM_HIP_DUO1024", "HELM_HIP_DUO1024_FLAGS"):
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
        print(json.dumps({"setting": name, "B": B, "best_ms": round(t[0], 3), "median_ms": round(t[len(t) // 2], 3), "decrypt_ok": ok,
                          "sha": hashlib.sha256(out.tobytes()).hexdigest()[:12]}))
