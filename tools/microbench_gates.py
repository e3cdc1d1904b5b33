"""µ-bench of SURVEY.md §8(d): B independent gates on fresh encryptions, gates/s against B - one command for the
whole table of a round:

    python tools/microbench_gates.py [--sets boolean_default,helm_cuda,shortint_m2c2,shortint_m2c2_multibit3]
                                     [--Bs 1,64,256,512,768,1024,4096,16384] [--out profiles/rNN/microbench.jsonl]

Boolean sets: B NAND gates (one programmable bootstrap + keyswitch each) through helm_hip_program_run; 64-bit sets: B
3-input LUT gates (keyswitch + programmable bootstrap each) through helm_si_eval_lut_level.  Per B: wall time of one
launch (mean of `--reps` back-to-back launches after a warm-up), the bootstrap kernel's and the keyswitch's device time
(the engine's HIP events), the kernel builds the size dispatch chose, and a decrypt check.  One JSON object per line.
HELM_HIP_PBS_VARIANT / HELM_HIP_DUO in the environment force a build (see helm_hip.hip: launch_pbs_f)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helm_amd  # noqa: E402

NAND = 4  # helm_gate_op HELM_GATE_NAND (include/helm_hip.h)


def gates_curve(name, Bs, reps):
    ck = helm_amd.ClientKey.generate(name, seed=1)
    sk = helm_amd.ServerKey(ck)
    rng = np.random.default_rng(0)
    maxB = max(Bs)
    bits = rng.integers(0, 2, size=2 * maxB).astype(bool)
    w = sk.wires(3 * maxB)
    w.upload(np.arange(2 * maxB), ck.encrypt(bits))
    sk.timing_enable(True)
    for B in Bs:
        ops = np.full(B, NAND, dtype=np.int32)
        i0 = np.arange(B, dtype=np.int32)
        i1 = np.arange(maxB, maxB + B, dtype=np.int32)
        i2 = np.full(B, -1, dtype=np.int32)
        out = np.arange(2 * maxB, 2 * maxB + B, dtype=np.int32)
        prog = helm_amd.Program(sk, ops, i0, i1, i2, out, [0, B])
        prog.run(w)
        sk.sync()
        sk.timing(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            prog.run(w)
        sk.sync()
        dt = (time.perf_counter() - t0) / reps
        t = sk.timing(reset=True)
        ok = bool(np.array_equal(ck.decrypt(w.download(out)), ~(bits[:B] & bits[maxB:maxB + B])))
        yield {"set": name, "gate": "NAND", "B": B, "wall_ms": round(dt * 1e3, 4), "pbs_ms": round(t.pbs_ms / reps, 4),
               "pbs_lockstep_ms": round(t.pbs_main_ms / reps, 4), "lockstep_bootstraps": int(t.pbs_main_count // reps),
               "ks_ms": round(t.ks_ms / reps, 4), "gates_per_s": round(B / dt, 1), "clock_ghz": sk.kernel_clock_ghz(),
               "decrypt_ok": ok}
        prog.destroy()
    sk.close()


def luts_curve(name, Bs, reps):
    ck = helm_amd.SiClientKey.generate(name, seed=1)
    sk = helm_amd.SiServerKey(ck)
    rng = np.random.default_rng(0)
    maxB = max(Bs)
    bits = rng.integers(0, 2, size=3 * maxB).astype(np.uint64)
    w = sk.wires(4 * maxB)
    w.upload(np.arange(3 * maxB), ck.encrypt(bits))
    sk.timing_enable(True)
    for B in Bs:
        in_idx = np.stack([np.arange(B), maxB + np.arange(B), 2 * maxB + np.arange(B)], axis=1).astype(np.int32)
        out = np.arange(3 * maxB, 3 * maxB + B, dtype=np.int32)
        ar, tb = np.full(B, 3, np.int32), np.full(B, 0xE8, np.uint64)
        w.eval_lut_level(ar, in_idx, tb, out)
        sk.sync()
        sk.timing(reset=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            w.eval_lut_level(ar, in_idx, tb, out)
        sk.sync()
        dt = (time.perf_counter() - t0) / reps
        t = sk.timing(reset=True)
        ok = bool(np.array_equal(ck.decrypt(w.download(out)), (bits[:B] + bits[maxB:maxB + B] + bits[2 * maxB:2 * maxB + B]) >= 2))
        yield {"set": name, "gate": "3-input LUT (0xE8)", "B": B, "wall_ms": round(dt * 1e3, 4), "pbs_ms": round(t.pbs_ms / reps, 4),
               "ks_ms": round(t.ks_ms / reps, 4), "linear_ms": round(t.linear_ms / reps, 4), "gates_per_s": round(B / dt, 1),
               "decrypt_ok": ok}
    sk.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="boolean_default,helm_cuda,shortint_m2c2,shortint_m2c2_multibit3")
    ap.add_argument("--Bs", default="1,64,256,512,768,1024,4096,16384")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    Bs = [int(x) for x in a.Bs.split(",")]
    env = {k: v for k, v in os.environ.items() if k.startswith("HELM_HIP_")}
    out = open(a.out, "a") if a.out else None
    for name in a.sets.split(","):
        curve = luts_curve if name.startswith(("shortint", "si_")) else gates_curve
        for rec in curve(name, Bs, a.reps):
            if env:
                rec["env"] = env
            line = json.dumps(rec)
            print(line, flush=True)
            if out:
                out.write(line + "\n")
                out.flush()


if __name__ == "__main__":
    main()
