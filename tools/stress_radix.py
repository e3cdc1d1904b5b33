"""Stress of the radix layer's noise margin: random u32 add / sub / mult / squares, fresh keys per seed, both arithmetic
parameter sets; every decrypted block must be clean (value < 4) and every result exact.
Usage: stress_radix.py [seeds = 6] [cases = 8] [bits = 32] [div = 0: also A / B, A << (B % bits), A >> 3]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import helm_amd
from helm_amd import ArithCircuit, Circuit, PtxtType, verilog_parser

bits = int(sys.argv[3]) if len(sys.argv) > 3 else 32
with_div = len(sys.argv) > 4 and sys.argv[4] == "1"
text = ("input A, B;\noutput S, D, P, Q, R" + (", V, L, H" if with_div else "") + ";\nadd g0(A, B, S);\nsub g1(A, B, D);\nmult g2(A, B, P);\n"
        "mult g3(A, A, Q);\nmult g4(P, 3, t);\nadd g5(t, Q, R);\n" + ("div g6(A, B, V);\nshl g7(A, B, L);\nshr g8(A, 3, H);\n" if with_div else ""))
gs, ws, i, o, d, _, _ = verilog_parser.read_verilog_text(text, True)
c = Circuit(gs, i, o, d)
c.sort_circuit()
c.compute_levels()
seeds, cases = (int(sys.argv[1]) if len(sys.argv) > 1 else 6), (int(sys.argv[2]) if len(sys.argv) > 2 else 8)
M = 1 << bits
mk = {8: PtxtType.U8, 16: PtxtType.U16, 32: PtxtType.U32, 64: PtxtType.U64, 128: PtxtType.U128}[bits]
kind = f"u{bits}"
bad = 0
for name in ("shortint_m2c2", "shortint_m2c2_multibit3"):
    for seed in range(1, seeds + 1):
        ck, sk = helm_amd.gen_keys_shortint(name, seed=seed)
        ac = ArithCircuit(ck, sk, c)
        rng = np.random.default_rng(seed)
        for k in range(cases):
            a, b = (int.from_bytes(rng.bytes(bits // 8), "little") for _ in range(2))
            if with_div and k % 2:
                b >>= int(rng.integers(0, bits))  # divisors of every size
            if k == 0:
                a, b = M - 1, M - 1
            out = ac.evaluate_encrypted(ac.encrypt_inputs(ws, {"A": mk(a), "B": mk(b)}), k + 1, kind)
            for wname in ("S", "D", "P", "Q", "R") + (("V", "L", "H") if with_div else ()):
                vals = ck.decrypt_message_and_carry(out[wname])
                if any(int(v) >= 4 for v in vals):
                    bad += 1
                    print("DIRTY BLOCK", name, seed, k, wname, list(vals))
            got = {kk: int(v.value) for kk, v in ac.decrypt_outputs(out, True).items()}
            want = {"S": (a + b) % M, "D": (a - b) % M, "P": a * b % M, "Q": a * a % M, "R": (3 * a * b + a * a) % M}
            if with_div:
                want.update({"V": a // b if b else M - 1, "L": (a << (b % bits)) % M, "H": a >> 3})
            if got != want:
                bad += 1
                print("WRONG", name, seed, k, a, b, got, want)
        sk.close()
        print(name, "seed", seed, "ok" if not bad else f"{bad} failures so far", flush=True)
print("stress:", "PASS" if not bad else f"FAIL ({bad})")
sys.exit(1 if bad else 0)
