"""Stress of the radix layer's noise margin: random u32 add / sub / mult / squares, fresh keys per seed, both arithmetic
parameter sets; every decrypted block must be clean (value < 4) and every result exact.  Usage: stress_radix.py [seeds = 6] [cases = 8]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import helm_amd
from helm_amd import ArithCircuit, Circuit, PtxtType, verilog_parser

text = ("input A, B;\noutput S, D, P, Q, R;\nadd g0(A, B, S);\nsub g1(A, B, D);\nmult g2(A, B, P);\nmult g3(A, A, Q);\n"
        "mult g4(P, 3, t);\nadd g5(t, Q, R);\n")
gs, ws, i, o, d, _, _ = verilog_parser.read_verilog_text(text, True)
c = Circuit(gs, i, o, d)
c.sort_circuit()
c.compute_levels()
seeds, cases = (int(sys.argv[1]) if len(sys.argv) > 1 else 6), (int(sys.argv[2]) if len(sys.argv) > 2 else 8)
M = 1 << 32
bad = 0
for name in ("shortint_m2c2", "shortint_m2c2_multibit3"):
    for seed in range(1, seeds + 1):
        ck, sk = helm_amd.gen_keys_shortint(name, seed=seed)
        ac = ArithCircuit(ck, sk, c)
        rng = np.random.default_rng(seed)
        for k in range(cases):
            a, b = (int(x) for x in rng.integers(0, M, size=2))
            if k == 0:
                a, b = M - 1, M - 1
            out = ac.evaluate_encrypted(ac.encrypt_inputs(ws, {"A": PtxtType.U32(a), "B": PtxtType.U32(b)}), k + 1, "u32")
            for wname in ("S", "D", "P", "Q", "R"):
                vals = ck.decrypt_message_and_carry(out[wname])
                if any(int(v) >= 4 for v in vals):
                    bad += 1
                    print("DIRTY BLOCK", name, seed, k, wname, list(vals))
            got = {kk: int(v.value) for kk, v in ac.decrypt_outputs(out, True).items()}
            want = {"S": (a + b) % M, "D": (a - b) % M, "P": a * b % M, "Q": a * a % M, "R": (3 * a * b + a * a) % M}
            if got != want:
                bad += 1
                print("WRONG", name, seed, k, a, b, got, want)
        sk.close()
        print(name, "seed", seed, "ok" if not bad else f"{bad} failures so far", flush=True)
print("stress:", "PASS" if not bad else f"FAIL ({bad})")
sys.exit(1 if bad else 0)
