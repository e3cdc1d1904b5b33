"""Profiling driver: chi_squared_arith.v u32, the default evaluation (lanes, carry-save), 1 warm-up + 1 traced pass.
Usage: prof_chi.py [set = shortint_m2c2]"""
import os, sys, time
sys.path.insert(0, ".")
import helm_amd
from helm_amd import ArithCircuit, Circuit, PtxtType, verilog_parser
name = sys.argv[1] if len(sys.argv) > 1 else "shortint_m2c2"
ck, sk = helm_amd.gen_keys_shortint(name, seed=1)
g, ws, i, o, d, _, _ = verilog_parser.read_verilog_file(os.path.join("tests", "netlists", "chi_squared_arith.v"), True)
c = Circuit(g, i, o, d); c.sort_circuit(); c.compute_levels()
ac = ArithCircuit(ck, sk, c)
if len(sys.argv) > 2: ac.set_lanes(int(sys.argv[2]))
enc = ac.encrypt_inputs(ws, {"N0": PtxtType.U32(2), "N1": PtxtType.U32(7), "N2": PtxtType.U32(9)})
ac.evaluate_encrypted(enc, 1, "u32")
t0 = time.perf_counter(); out = ac.evaluate_encrypted(enc, 2, "u32"); dt = time.perf_counter() - t0
print("chi-squared", name, round(dt, 4), "s", ac.pbs_per_cycle(), "bootstraps", ac.pbs_rounds_per_cycle(), "rounds",
      {k: v.value for k, v in ac.decrypt_outputs(out, True).items()})
