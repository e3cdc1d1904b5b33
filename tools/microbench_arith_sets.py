"""chi-squared u32 (arithmetic mode) and B independent LUTs under a named shortint set.
Usage: microbench_arith_sets.py <set> [B list]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import helm_amd  # noqa: E402
from helm_amd import ArithCircuit, Circuit, EvalCircuit, verilog_parser  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "shortint_m2c2_multibit3"
Bs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 256, 1024]
t0 = time.time()
ck, sk = helm_amd.gen_keys_shortint(name, seed=1)
print(f"{name}: keygen + upload {time.time() - t0:.1f} s, params {ck.params}", flush=True)
HERE = os.path.dirname(os.path.abspath(__file__))
NET = os.path.join(HERE, "..", "tests", "netlists")
gates_set, wire_set, input_wires, output_wires, dffs, _, _ = verilog_parser.read_verilog_file(f"{NET}/chi_squared_arith.v", True)
c = Circuit(gates_set, input_wires, output_wires, dffs)
c.sort_circuit()
c.compute_levels()
inputs = verilog_parser.read_input_wires(os.path.join(HERE, "..", "tests", "golden", "chi_squared_arith_1.inputs.csv"), "u32")
ac = ArithCircuit(ck, sk, c)
for rep in range(2):
    enc = EvalCircuit.encrypt_inputs(ac, wire_set, inputs)
    sk.sync()
    t0 = time.perf_counter()
    enc = EvalCircuit.evaluate_encrypted(ac, enc, 1 + rep, "u32")  # a new cycle each time (same-cycle memo)
    sk.sync()
    dt = time.perf_counter() - t0
out = {k: v.value for k, v in EvalCircuit.decrypt_outputs(ac, enc, True).items()}
print(f"chi-squared u32: {dt:.4f} s, {ac.pbs_per_cycle()} bootstraps, {ac.pbs_rounds_per_cycle()} rounds, "
      f"ok={out == {'alpha': 529, 'beta1': 242, 'beta2': 275, 'beta3': 1250}}", flush=True)
rng = np.random.default_rng(0)
maxB = max(Bs)
bits = rng.integers(0, 2, size=3 * maxB).astype(np.uint64)
w = sk.wires(4 * maxB)
w.upload(np.arange(3 * maxB), ck.encrypt(bits))
sk.timing_enable(True)
for B in Bs:
    in_idx = np.stack([np.arange(B), maxB + np.arange(B), 2 * maxB + np.arange(B)], axis=1).astype(np.int32)
    outr = np.arange(3 * maxB, 3 * maxB + B, dtype=np.int32)
    ar = np.full(B, 3, np.int32)
    tb = np.full(B, 0xE8, np.uint64)
    w.eval_lut_level(ar, in_idx, tb, outr); sk.sync()
    sk.timing(reset=True)
    t0 = time.perf_counter()
    w.eval_lut_level(ar, in_idx, tb, outr)
    sk.sync()
    dt = time.perf_counter() - t0
    t = sk.timing(reset=True)
    dec = ck.decrypt(w.download(outr))
    maj = (bits[:B] + bits[maxB:maxB + B] + bits[2 * maxB:2 * maxB + B]) >= 2
    print(f"{name} B={B:6d} wall {dt*1e3:9.3f} ms  pbs {t.pbs_ms:9.3f} ms  ks {t.ks_ms:8.3f} ms  {B/dt:10.1f} LUTs/s  "
          f"decrypt_ok={np.array_equal(dec.astype(bool), maj)}", flush=True)
