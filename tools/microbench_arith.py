"""Wall-clock of the arithmetic-mode chi-squared netlist (u32) and of the 8-bit LUT adder on the GPU."""
import os
import sys
import time

sys.path.insert(0, ".")
import helm_amd  # noqa: E402
from helm_amd import ArithCircuit, Circuit, LutCircuit, PtxtType, verilog_parser  # noqa: E402

NET = os.path.join("tests", "netlists")
ck, sk = helm_amd.gen_keys_shortint("shortint_m2c2", seed=1)


def circuit(path, arith):
    g, ws, i, o, d, _, _ = verilog_parser.read_verilog_file(path, arith)
    c = Circuit(g, i, o, d)
    c.sort_circuit()
    c.compute_levels()
    return c, ws


c, ws = circuit(f"{NET}/chi_squared_arith.v", True)
ac = ArithCircuit(ck, sk, c)
enc = ac.encrypt_inputs(ws, {"N0": PtxtType.U32(2), "N1": PtxtType.U32(7), "N2": PtxtType.U32(9)})
for rep in range(2):
    t0 = time.perf_counter()
    out = ac.evaluate_encrypted(enc, 1 + rep, "u32")  # a new cycle each time (same-cycle memo)
    dt = time.perf_counter() - t0
dec = {k: v.value for k, v in ac.decrypt_outputs(out, True).items()}
print(f"chi_squared u32: {dt:.3f} s, {ac.pbs_per_cycle()} bootstraps in {ac.pbs_rounds_per_cycle()} batched rounds "
      f"({ac.pbs_per_cycle() / dt:.0f} PBS/s), outputs {dec}")

c, ws = circuit(f"{NET}/8-bit-adder-lut-3-1.v", False)
lc = LutCircuit(ck, sk, c)
inp = {f"a[{i}]": PtxtType.Bool((0xB7 >> i) & 1) for i in range(8)}
inp.update({f"b[{i}]": PtxtType.Bool((0x6E >> i) & 1) for i in range(8)})
inp["cin"] = PtxtType.Bool(1)
enc = lc.encrypt_inputs(ws, inp)
for rep in range(2):
    t0 = time.perf_counter()
    out = lc.evaluate_encrypted(lc.encrypt_inputs(ws, inp), 1, "bool")  # a fresh map each time (same-cycle memo)
    dt = time.perf_counter() - t0
print(f"8-bit LUT adder: {dt:.3f} s, {lc.pbs_per_cycle()} bootstraps in 8 levels")
