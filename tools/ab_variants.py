#!/usr/bin/env python3
"""Same-box A/B of builds of libhelm_hip.so: runs a micro-benchmark in a fresh process per build (HELM_HIP_LIB selects
the library, helm_amd/_native.py), alternating the builds over several rounds.
usage: ab_variants.py [--rounds R] [--bench lut|wop|m1c1|gates] variant [variant ...]
A variant is a file name under helm_amd/csrc/ (e.g. variants/libhelm_hip_base.so) or `default` (the Makefile's)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LUT = r'''
import json, time, numpy as np, sys
sys.path.insert(0, %r)
import helm_amd
res = {}
for name, B in (("shortint_m2c2", 1024), ("shortint_m2c2", 64), ("shortint_m2c2_multibit3", 1024)):
    ck, sk = helm_amd.gen_keys_shortint(name, seed=1)
    bits = np.random.default_rng(0).integers(0, 2, size=3 * B).astype(np.uint64)
    w = sk.wires(4 * B)
    w.upload(np.arange(3 * B), ck.encrypt(bits))
    in_idx = np.arange(3 * B, dtype=np.int32).reshape(3, B).T.copy()
    ar, tb, out = np.full(B, 3, np.int32), np.full(B, 0xE8, np.uint64), np.arange(3 * B, 4 * B, dtype=np.int32)
    w.eval_lut_level(ar, in_idx, tb, out); sk.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); w.eval_lut_level(ar, in_idx, tb, out); sk.sync(); ts.append(time.perf_counter() - t0)
    ok = bool(np.array_equal(ck.decrypt(w.download(out)), (bits[:B] + bits[B:2 * B] + bits[2 * B:]) >= 2))
    res[f"{name}:{B}"] = {"ms": round(min(ts) * 1e3, 3), "luts_per_s": round(B / min(ts), 1), "ok": ok,
                          "sha": __import__("hashlib").sha256(w.download(out).tobytes()).hexdigest()[:12]}
    sk.close()
print(json.dumps(res))
'''

M1C1 = r'''
import json, time, numpy as np, sys
sys.path.insert(0, %r)
import helm_amd
res = {}
for name, B in (("shortint_m1c1", 2048), ("shortint_m1c1", 512), ("si_toy_512_k2", 2048)):
    ck, sk = helm_amd.gen_keys_shortint(name, seed=1)
    bits = np.random.default_rng(0).integers(0, 2, size=2 * B).astype(np.uint64)
    w = sk.wires(3 * B)
    w.upload(np.arange(2 * B), ck.encrypt(bits))
    in_idx = np.arange(2 * B, dtype=np.int32).reshape(2, B).T.copy()
    ar, tb, out = np.full(B, 2, np.int32), np.full(B, 0x8, np.uint64), np.arange(2 * B, 3 * B, dtype=np.int32)
    w.eval_lut_level(ar, in_idx, tb, out); sk.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); w.eval_lut_level(ar, in_idx, tb, out); sk.sync(); ts.append(time.perf_counter() - t0)
    ok = bool(np.array_equal(ck.decrypt(w.download(out)), (bits[:B] + bits[B:]) >= 2))
    res[f"{name}:{B}"] = {"ms": round(min(ts) * 1e3, 3), "luts_per_s": round(B / min(ts), 1), "ok": ok,
                          "sha": __import__("hashlib").sha256(w.download(out).tobytes()).hexdigest()[:12]}
    sk.close()
print(json.dumps(res))
'''

GATES = r'''
import json, time, numpy as np, sys
sys.path.insert(0, %r)
import helm_amd
res = {}
for name, B in (("boolean_default", 4096), ("boolean_default", 1024), ("boolean_default", 256), ("helm_cuda", 4096)):
    ck = helm_amd.ClientKey.generate(name, seed=1)
    sk = helm_amd.ServerKey(ck)
    bits = np.random.default_rng(0).integers(0, 2, size=2 * B).astype(bool)
    w = sk.wires(3 * B)
    w.upload(np.arange(2 * B), ck.encrypt(bits))
    prog = helm_amd.Program(sk, np.full(B, 4, np.int32), np.arange(B), np.arange(B, 2 * B), np.full(B, -1), np.arange(2 * B, 3 * B), [0, B])
    for _ in range(2):
        prog.run(w)
    sk.sync()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); prog.run(w); prog.run(w); sk.sync(); ts.append((time.perf_counter() - t0) / 2)
    out = w.download(np.arange(2 * B, 3 * B))
    ok = bool(np.array_equal(ck.decrypt(out), ~(bits[:B] & bits[B:])))
    res[f"{name}:{B}"] = {"ms": round(min(ts) * 1e3, 3), "median_ms": round(sorted(ts)[len(ts) // 2] * 1e3, 3), "ok": ok,
                          "sha": __import__("hashlib").sha256(out.tobytes()).hexdigest()[:12]}
    sk.close()
print(json.dumps(res))
'''

WOP = r'''
import json, subprocess, sys
out = subprocess.run([sys.executable, %r + "/tools/wop_bench.py", "256", "6", "1"], capture_output=True, text=True).stdout
print(json.dumps({"wop_256x6_b1": out.strip().splitlines()[-1][:300]}))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--bench", default="lut")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    code = {"lut": LUT, "wop": WOP, "m1c1": M1C1, "gates": GATES}[a.bench] % ROOT
    for r in range(a.rounds):
        for v in a.variants:
            env = dict(os.environ)
            if v != "default":
                env["HELM_HIP_LIB"] = v
            else:
                env.pop("HELM_HIP_LIB", None)
            p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=400)
            line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "FAILED: " + p.stderr[-400:]
            print(f"round {r} {v}: {line}", flush=True)


main()
