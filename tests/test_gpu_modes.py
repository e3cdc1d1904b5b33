"""LUT mode and arithmetic mode end to end on the GPU through the host front end, mirroring
reference tests/circuit_test.rs:266-311 (encrypted 8-bit LUT adder: every wire equals the
plaintext evaluation), :313-370 (chi-squared in arithmetic mode) and tests/gates_test.rs:127-310
(FheUint16 add / sub / mul known answers)."""
import os

import numpy as np
import pytest

import helm_amd
from helm_amd import ArithCircuit, Circuit, EvalCircuit, LutCircuit, PtxtType, verilog_parser
from helm_amd._host import Panic

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NET = os.path.join(HERE, "netlists")


@pytest.fixture(scope="module")
def keys():
    ck, sk = helm_amd.gen_keys_shortint("shortint_m2c2", seed=1)  # the 3-bit-capable class of circuit_test.rs:287
    yield ck, sk
    sk.close()


def _circuit(path_or_text, is_arith=False, is_text=False):
    rd = verilog_parser.read_verilog_text if is_text else verilog_parser.read_verilog_file
    gates_set, wire_set, input_wires, output_wires, dffs, _, _ = rd(path_or_text, is_arith)
    c = Circuit(gates_set, input_wires, output_wires, dffs)
    c.sort_circuit()
    c.compute_levels()
    return c, wire_set, input_wires, output_wires


def test_encrypted_8_bit_adder_lut(keys):  # circuit_test.rs:266-311
    client_key, server_key = keys
    circuit, wire_set, input_wires, output_wires = _circuit(f"{NET}/8-bit-adder-lut-3-1.v")
    a, b, cin = 0xB7, 0x6E, 1
    inputs = {f"a[{i}]": PtxtType.Bool((a >> i) & 1) for i in range(8)}
    inputs.update({f"b[{i}]": PtxtType.Bool((b >> i) & 1) for i in range(8)})
    inputs["cin"] = PtxtType.Bool(cin)
    ptxt = circuit.initialize_wire_map(wire_set, inputs, "bool")
    ptxt = circuit.evaluate(ptxt)
    lc = LutCircuit(client_key, server_key, circuit)
    enc = EvalCircuit.encrypt_inputs(lc, wire_set, inputs)
    enc = EvalCircuit.evaluate_encrypted(lc, enc, 1, "bool")
    assert lc.pbs_per_cycle() == 16
    for wire, want in ptxt.items():  # every wire, as the reference test does
        assert client_key.decrypt(enc[wire]) == int(bool(want)), wire
    out = EvalCircuit.decrypt_outputs(lc, enc, True)
    total = sum(out[f"sum[{i}]"].value << i for i in range(8)) + (out["cout"].value << 8)
    assert total == a + b + cin
    assert all(v.kind == "U64" for v in out.values())
    assert "Evaluated gates in level [1/" in lc.log()


def test_encrypted_8_bit_adder_lut_2_1_on_the_binarys_parameter_set():  # circuit_test.rs:266-311 + helm.rs:301
    """The reference's LUT test netlist (8-bit-adder-lut-2-1.v: 2-input LUTs, circuit_test.rs:272) under the set the
    reference BINARY installs for LUT mode, PARAM_MESSAGE_1_CARRY_1_KS_PBS (helm.rs:301; k = 3, N = 512): every wire
    equals the plaintext evaluation."""
    client_key, server_key = helm_amd.gen_keys_shortint("shortint_m1c1", seed=5)
    try:
        circuit, wire_set, input_wires, output_wires = _circuit(f"{NET}/8-bit-adder-lut-2-1.v")
        inputs = verilog_parser.read_input_wires(os.path.join(HERE, "golden", "8-bit-adder.inputs.csv"), "bool")
        ptxt = circuit.evaluate(circuit.initialize_wire_map(wire_set, inputs, "bool"))
        lc = LutCircuit(client_key, server_key, circuit)
        enc = EvalCircuit.evaluate_encrypted(lc, EvalCircuit.encrypt_inputs(lc, wire_set, inputs), 1, "bool")
        assert lc.pbs_per_cycle() == 40
        for wire, want in ptxt.items():
            assert client_key.decrypt(enc[wire]) == int(bool(want)), wire
        out = EvalCircuit.decrypt_outputs(lc, enc, True)
        a = sum(int(bool(inputs[f"a[{i}]"])) << i for i in range(8))
        b = sum(int(bool(inputs[f"b[{i}]"])) << i for i in range(8))
        assert sum(out[f"sum[{i}]"].value << i for i in range(8)) + (out["cout"].value << 8) == a + b + int(bool(inputs["cin"]))
    finally:
        server_key.close()


def test_lut_sequential_ready_latch(keys):
    """2-bit counter out of LUTs and DFFs, READY-latched outputs (circuit.rs:1002-1030)."""
    client_key, server_key = keys
    text = """input en;
output q0, READY;
dff g0(d0, q0);
dff g1(d1, q1);
lut g2(0x6, q0, en, d0);
lut g3(0x8, q0, en, c0);
lut g4(0x6, q1, c0, d1);
lut g5(0xAA, q1, q1, q1, READY);
"""
    circuit, wire_set, input_wires, output_wires = _circuit(text, is_text=True)
    lc = LutCircuit(client_key, server_key, circuit)
    enc = lc.encrypt_inputs(wire_set, {"en": PtxtType.Bool(True), "q0": PtxtType.Bool(False), "q1": PtxtType.Bool(False)})
    ready = lc.init_ready()
    state = []
    for _ in range(3):
        enc = lc.evaluate_encrypted(enc, 1, "bool")
        lc.evaluate_ready(enc, ready)
        state.append((client_key.decrypt(enc["q0"]), client_key.decrypt(enc["q1"])))
    assert state == [(1, 0), (0, 1), (1, 1)]
    assert lc.decrypt_outputs(ready, True)["READY"].value == 1


@pytest.mark.parametrize("x,y", [(10, 20), (20, 30), (30, 40)])
def test_fheuint16_known_answers(keys, x, y):  # gates_test.rs:127-310 (K-7)
    client_key, server_key = keys
    text = """input [15:0] A, B;
output [15:0] S, D, P, Q, R;
add g0(A, B, S);
sub g1(B, A, D);
mult g2(A, B, P);
add g3(A, 7, Q);
sub g4(B, 3, R);
"""
    circuit, wire_set, input_wires, output_wires = _circuit(text, is_arith=True, is_text=True)
    ac = ArithCircuit(client_key, server_key, circuit)
    enc = ac.encrypt_inputs(wire_set, {"A": PtxtType.U16(x), "B": PtxtType.U16(y)})
    enc = ac.evaluate_encrypted(enc, 1, "u16")
    out = ac.decrypt_outputs(enc, True)
    assert out["S"] == PtxtType.U16(x + y)
    assert out["D"] == PtxtType.U16(y - x)
    assert out["P"] == PtxtType.U16(x * y)
    assert out["Q"] == PtxtType.U16(x + 7)
    assert out["R"] == PtxtType.U16(y - 3)


def test_radix_wraparound_u8(keys):
    client_key, server_key = keys
    text = """input [7:0] A, B;
output [7:0] S, D, P, C, M;
add g0(A, B, S);
sub g1(A, B, D);
mult g2(A, B, P);
copy g3(A, C);
mult g4(A, 27, M);
"""
    circuit, wire_set, _, _ = _circuit(text, is_arith=True, is_text=True)
    ac = ArithCircuit(client_key, server_key, circuit)
    for cycle, (a, b) in enumerate(((200, 100), (3, 250), (255, 255), (0, 0)), start=1):  # a new cycle defeats the memo
        enc = ac.encrypt_inputs(wire_set, {"A": PtxtType.U8(a), "B": PtxtType.U8(b)})
        out = ac.decrypt_outputs(ac.evaluate_encrypted(enc, cycle, "u8"), True)
        assert out["S"].value == (a + b) % 256 and out["D"].value == (a - b) % 256
        assert out["P"].value == (a * b) % 256 and out["C"].value == a and out["M"].value == (a * 27) % 256


def test_chi_squared_u32(keys):  # circuit_test.rs:313-370, inputs K-5 (2, 7, 9)
    client_key, server_key = keys
    circuit, wire_set, input_wires, output_wires = _circuit(f"{NET}/chi_squared_arith.v", is_arith=True)
    inputs = verilog_parser.read_input_wires(os.path.join(HERE, "golden", "chi_squared_arith_1.inputs.csv"), "u32")
    ac = ArithCircuit(client_key, server_key, circuit)
    enc = EvalCircuit.encrypt_inputs(ac, wire_set, inputs)
    enc = EvalCircuit.evaluate_encrypted(ac, enc, 1, "u32")
    out = EvalCircuit.decrypt_outputs(ac, enc, True)
    n0, n1, n2 = 2, 7, 9
    M = 1 << 32
    want = {"alpha": ((4 * n0 * n2 - n1 * n1) ** 2) % M, "beta1": (2 * (2 * n0 + n1) ** 2) % M,
            "beta2": ((2 * n0 + n1) * (2 * n2 + n1)) % M, "beta3": (2 * (2 * n2 + n1) ** 2) % M}
    assert {k: v.value for k, v in out.items()} == want == {"alpha": 529, "beta1": 242, "beta2": 275, "beta3": 1250}
    assert all(v.kind == "U32" for v in out.values())
    assert ac.pbs_per_cycle() > 0 and ac.pbs_rounds_per_cycle() > 0


def test_chi_squared_on_lanes_is_bit_identical_and_shorter(keys):
    """Lanes (helm_si_ctx_fork): the two sub-circuits of chi-squared that share no wire (alpha's and the betas') run
    concurrently instead of meeting at every level boundary: the same ciphertexts on every wire, fewer rounds in a row
    (26 level-synchronous rounds -> the longer sub-circuit's 18).  The DEFAULT when the operator graph has two or more
    components: the sub-circuits as chains on one context whose look-up rounds are merged into launches of at most the
    device's capacity (RoundMerger); set_lanes(1) is the reference's level-by-level evaluation."""
    import time
    client_key, server_key = keys
    circuit, wire_set, _, _ = _circuit(f"{NET}/chi_squared_arith.v", is_arith=True)
    inputs = verilog_parser.read_input_wires(os.path.join(HERE, "golden", "chi_squared_arith_1.inputs.csv"), "u32")
    ac = ArithCircuit(client_key, server_key, circuit)
    enc_in = ac.encrypt_inputs(wire_set, inputs)
    ac.set_lanes(1)
    t0 = time.perf_counter()
    one = ac.evaluate_encrypted(enc_in, 1, "u32")
    t_one, rounds_one, pbs_one = time.perf_counter() - t0, ac.pbs_rounds_per_cycle(), ac.pbs_per_cycle()
    assert "Evaluated gates in level [1/4]" in ac.log()
    ac.set_lanes(2)
    ac.evaluate_encrypted(enc_in, 2, "u32")  # warm-up of the lane's scratch (a new cycle each time: the memo is per cycle)
    t0 = time.perf_counter()
    two = ac.evaluate_encrypted(enc_in, 3, "u32")
    t_two = time.perf_counter() - t0
    assert sorted(one.keys()) == sorted(two.keys())
    for wire in one.keys():
        assert np.array_equal(one[wire], two[wire]), wire
    assert {k: v.value for k, v in ac.decrypt_outputs(two, True).items()} == {"alpha": 529, "beta1": 242, "beta2": 275, "beta3": 1250}
    # a carry propagation over 16 blocks is 4 rounds (grouped by four; round 2: 6).  mult by 2 is one round (a shift): the
    # betas' chain is 1 + 4 + 9 + 1 = 15 rounds; alpha's products hand their terms to the subtraction in carry-save form:
    # 5 + 0 + 5 + 8 = 18 (the last product is a square: one reduction round fewer; round 2: 29); level by level
    # 5 + 4 + 9 + 8 = 26 (was 39)
    assert ac.pbs_per_cycle() == pbs_one and rounds_one == 26 and ac.pbs_rounds_per_cycle() == 18
    assert "2 independent sub-circuit(s)" in ac.log()
    print(f"chi-squared u32: {t_one:.3f} s level by level, {t_two:.3f} s on two lanes")
    assert t_two < t_one
    # the default: a fresh circuit runs its two components as chains and merges their rounds: 20 launches in a row, none
    # above the device's capacity while both chains run (18 would need both chains to fit next to each other every round)
    ac2 = ArithCircuit(client_key, server_key, circuit)
    dflt = ac2.evaluate_encrypted(enc_in, 1, "u32")
    log = ac2.log()
    assert 18 <= ac2.pbs_rounds_per_cycle() <= 22 and "2 independent sub-circuit(s)" in log and "rounds merged" in log
    for wire in one.keys():
        assert np.array_equal(one[wire], dflt[wire]), wire


def _decrypt_int(client_key, rows):
    """blocks of 2 message bits, least significant first -> the integer (every block must be clean: value < 4)"""
    vals = client_key.decrypt_message_and_carry(rows)
    assert all(int(v) < 4 for v in vals), list(vals)
    return sum(int(v) << (2 * i) for i, v in enumerate(vals))


def test_carry_save_products_feed_additions(keys):
    """Products whose consumers are additions / subtractions (possibly behind a multiplication by a power of four) hand
    over the two terms their reduction ends with instead of propagating carries; the consumer sums all terms and
    propagates once.  Every wire - the carry-save ones too - holds the same value as without the optimisation
    (reference semantics: src/gates.rs:331-385, 453-487: `*`, `+`, `-` on FheUintN), in fewer rounds in a row."""
    client_key, server_key = keys
    text = """input [15:0] A, B, C, D;
output [15:0] X, Y, Z, W;
mult g0(A, B, t0);
mult g1(C, D, t1);
sub g2(t0, t1, X);
mult g3(t0, 4, t2);
add g4(t2, t1, Y);
mult g5(A, C, t3);
add g6(t3, D, Z);
mult g7(t3, t3, W);
"""
    circuit, wire_set, _, _ = _circuit(text, is_arith=True, is_text=True)
    a, b, c, d = 1234, 567, 89, 4321
    m = 1 << 16
    want = {"t0": a * b % m, "t1": c * d % m, "X": (a * b - c * d) % m, "t2": a * b * 4 % m, "Y": (a * b * 4 + c * d) % m,
            "t3": a * c % m, "Z": (a * c + d) % m, "W": (a * c) ** 2 % m}
    results = {}
    for lazy in (True, False):
        ac = ArithCircuit(client_key, server_key, circuit)
        ac.set_lanes(1)
        ac.set_lazy_carries(lazy)
        enc = ac.encrypt_inputs(wire_set, {"A": PtxtType.U16(a), "B": PtxtType.U16(b), "C": PtxtType.U16(c), "D": PtxtType.U16(d)})
        out = ac.evaluate_encrypted(enc, 1, "u16")
        got = {w: _decrypt_int(client_key, out[w]) for w in want}
        assert got == want, (lazy, got)
        results[lazy] = (ac.pbs_rounds_per_cycle(), ac.pbs_per_cycle())
    # here level 1 also holds a product that must propagate (t3 feeds a product), so the level costs its 8 rounds either
    # way and the carry-save sums add one reduction round: 20 against 19 - the saving shows where the products are alone
    # (a carry propagation over 8 blocks is 4 rounds, round 2's Hillis-Steele form: 5; the square t3 * t3 needs one
    # reduction round fewer than a general product; 24 against 23 before both):
    assert results == {True: (20, results[True][1]), False: (19, results[False][1])}, results
    small, ws, _, _ = _circuit("input [15:0] A, B, C, D;\noutput [15:0] X;\nmult g0(A, B, t0);\nmult g1(C, D, t1);\nsub g2(t0, t1, X);\n",
                               is_arith=True, is_text=True)
    rounds = {}
    for lazy in (True, False):
        ac = ArithCircuit(client_key, server_key, small)
        ac.set_lazy_carries(lazy)
        enc = ac.encrypt_inputs(ws, {"A": PtxtType.U16(a), "B": PtxtType.U16(b), "C": PtxtType.U16(c), "D": PtxtType.U16(d)})
        out = ac.evaluate_encrypted(enc, 1, "u16")
        assert {w: _decrypt_int(client_key, out[w]) for w in ("t0", "t1", "X")} == {"t0": want["t0"], "t1": want["t1"], "X": want["X"]}
        rounds[lazy] = ac.pbs_rounds_per_cycle()
    # a * b - c * d on FheUint16: products 1 + 3 rounds, then 1 + 4 for the subtraction = 9; with every operator
    # propagating: 8 + 4 = 12
    assert rounds == {True: 9, False: 12}, rounds
    # on lanes (the default) the same ciphertexts as level by level
    ac = ArithCircuit(client_key, server_key, circuit)
    enc = ac.encrypt_inputs(wire_set, {"A": PtxtType.U16(a), "B": PtxtType.U16(b), "C": PtxtType.U16(c), "D": PtxtType.U16(d)})
    lanes = ac.evaluate_encrypted(enc, 1, "u16")
    ac.set_lanes(1)
    flat = ac.evaluate_encrypted(enc, 2, "u16")
    for w in sorted(flat.keys()):
        assert np.array_equal(lanes[w], flat[w]), w


def test_chains_of_additions_propagate_once(keys):
    """Sums and differences whose consumers are all additions / subtractions stay in carry-save form too (two terms, no
    propagation): a chain of additions propagates carries once, at its end.  Every wire of the returned map - the inner
    sums too - holds the reference's value (src/gates.rs:331-385, 453-487), in fewer rounds in a row."""
    client_key, server_key = keys
    text = """input [15:0] A, B, C, D, E;
output [15:0] S, T;
mult g0(A, B, p0);
mult g1(C, D, p1);
add g2(p0, p1, s0);
add g3(s0, E, s1);
sub g4(s1, A, s2);
add g5(s2, s0, S);
add g6(A, B, q0);
sub g7(q0, C, q1);
add g8(q1, D, q2);
add g9(q2, q2, T);
"""
    circuit, wire_set, _, _ = _circuit(text, is_arith=True, is_text=True)
    a, b, c, d, e = 51234, 60567, 65535, 4321, 65000
    m = 1 << 16
    want = {"p0": a * b % m, "p1": c * d % m}
    want["s0"] = (want["p0"] + want["p1"]) % m
    want["s1"] = (want["s0"] + e) % m
    want["s2"] = (want["s1"] - a) % m
    want["S"] = (want["s2"] + want["s0"]) % m
    want["q0"] = (a + b) % m
    want["q1"] = (want["q0"] - c) % m
    want["q2"] = (want["q1"] + d) % m
    want["T"] = (2 * want["q2"]) % m
    results, maps = {}, {}
    for lazy in (True, False):
        ac = ArithCircuit(client_key, server_key, circuit)
        ac.set_lanes(1)
        ac.set_lazy_carries(lazy)
        enc = ac.encrypt_inputs(wire_set, {"A": PtxtType.U16(a), "B": PtxtType.U16(b), "C": PtxtType.U16(c), "D": PtxtType.U16(d),
                                           "E": PtxtType.U16(e)})
        out = ac.evaluate_encrypted(enc, 1, "u16")
        got = {w: _decrypt_int(client_key, out[w]) for w in want}
        assert got == want, (lazy, got, want)
        results[lazy] = ac.pbs_rounds_per_cycle()
    print("chained additions, rounds in a row:", results)
    assert results[True] < results[False]
    # the default evaluation (two chains, merged rounds): the same values
    ac = ArithCircuit(client_key, server_key, circuit)
    enc = ac.encrypt_inputs(wire_set, {"A": PtxtType.U16(a), "B": PtxtType.U16(b), "C": PtxtType.U16(c), "D": PtxtType.U16(d),
                                       "E": PtxtType.U16(e)})
    out = ac.evaluate_encrypted(enc, 1, "u16")
    assert {w: _decrypt_int(client_key, out[w]) for w in want} == want


def test_many_independent_sub_circuits_merge_their_rounds(keys):
    """Six sub-circuits that share no wire (a product, a square behind a sum, a difference, a shift by a plaintext, a shift
    by a ciphertext, a division) run as chains whose look-up rounds are merged: the same ciphertext on every wire as level by
    level, in fewer launches than the level-by-level rounds."""
    client_key, server_key = keys
    text = """input [15:0] A, B, C, D, E, F, G, H, I, J, K;
output [15:0] X, Y, Z, W, V, U;
mult g0(A, B, X);
add g1(C, D, t0);
mult g2(t0, t0, Y);
sub g3(E, F, Z);
shl g4(G, 5, W);
shr g5(H, I, V);
div g6(J, K, U);
"""
    circuit, wire_set, _, _ = _circuit(text, is_arith=True, is_text=True)
    vals = dict(A=40000, B=51111, C=65535, D=12345, E=7, F=65000, G=0xBEEF, H=0xF00D, I=9, J=54321, K=123)
    m = 1 << 16
    want = {"X": vals["A"] * vals["B"] % m, "t0": (vals["C"] + vals["D"]) % m, "Z": (vals["E"] - vals["F"]) % m,
            "W": (vals["G"] << 5) % m, "V": vals["H"] >> 9, "U": vals["J"] // vals["K"]}
    want["Y"] = want["t0"] ** 2 % m
    enc_in = {k: PtxtType.U16(v) for k, v in vals.items()}
    ac = ArithCircuit(client_key, server_key, circuit)
    enc = ac.encrypt_inputs(wire_set, enc_in)
    merged = ac.evaluate_encrypted(enc, 1, "u16")
    log, launches = ac.log(), ac.pbs_rounds_per_cycle()
    assert "6 independent sub-circuit(s)" in log and "rounds merged" in log, log
    assert {w: _decrypt_int(client_key, merged[w]) for w in want} == want
    ac.set_lanes(1)
    flat = ac.evaluate_encrypted(enc, 2, "u16")
    assert {w: _decrypt_int(client_key, flat[w]) for w in want} == want
    for w in want:  # batching and launch boundaries do not change a single bit
        assert np.array_equal(merged[w], flat[w]), w
    print(f"six sub-circuits: {launches} merged launches, {ac.pbs_rounds_per_cycle()} rounds level by level")
    assert launches <= ac.pbs_rounds_per_cycle()


def test_evaluation_only_circuit_without_a_client_key(keys):
    """helm_host_si_circuit_new(client_key = NULL): the host encrypts and decrypts with its own keys (what the Rust shim
    does with tfhe's) and moves ciphertext words through the encrypted map; evaluate_encrypted is the whole-circuit
    evaluation (merged rounds, carry-save) - the same values as with the library's client key."""
    from helm_amd.circuit import SiEncWireMap
    client_key, server_key = keys
    text = "input [15:0] A, B, C;\noutput [15:0] X, Y;\nmult g0(A, B, t0);\nadd g1(t0, C, X);\nsub g2(A, C, Y);\n"
    circuit, wire_set, _, _ = _circuit(text, is_arith=True, is_text=True)
    vals = {"A": 40321, "B": 777, "C": 65000}
    ac = ArithCircuit(None, server_key, circuit)
    enc = SiEncWireMap(server_key, blocks=8)
    for name in sorted(wire_set):  # as the reference's encrypt_inputs: every wire is in the map (circuit.rs:970-1000),
        enc[name] = np.zeros((8, enc.row_words), dtype=np.uint64)  # non-inputs as trivial encryptions of zero
    for name, v in vals.items():
        digits = np.array([(v >> (2 * i)) & 3 for i in range(8)], dtype=np.uint64)
        enc[name] = client_key.encrypt(digits)
    out = ac.evaluate_encrypted(enc, 1, "u16")
    m = 1 << 16
    assert _decrypt_int(client_key, out["X"]) == (vals["A"] * vals["B"] + vals["C"]) % m
    assert _decrypt_int(client_key, out["Y"]) == (vals["A"] - vals["C"]) % m
    assert "rounds merged" in ac.log()
    with pytest.raises(Exception, match="evaluation-only"):
        ac.decrypt_outputs(out, False)


def test_shifts_and_division_u8(keys):  # gates.rs:386-452, 488-700 (div, shl, shr and their plain forms)
    client_key, server_key = keys
    text = """input [7:0] A, B;
output [7:0] Q, QS, L3, R3, L2, R1, LV, RV;
div g0(A, B, Q);
div g1(A, 7, QS);
shl g2(A, 3, L3);
shr g3(A, 3, R3);
shl g4(A, 2, L2);
shr g5(A, 1, R1);
shl g6(A, B, LV);
shr g7(A, B, RV);
"""
    circuit, wire_set, _, _ = _circuit(text, is_arith=True, is_text=True)
    ac = ArithCircuit(client_key, server_key, circuit)
    for cycle, (a, b) in enumerate(((201, 13), (77, 3), (5, 0)), start=1):
        enc = ac.encrypt_inputs(wire_set, {"A": PtxtType.U8(a), "B": PtxtType.U8(b)})
        out = {k: v.value for k, v in ac.decrypt_outputs(ac.evaluate_encrypted(enc, cycle, "u8"), True).items()}
        assert out["Q"] == (a // b if b else 255), (a, b, out)  # x / 0 = all ones, as tfhe's
        assert out["QS"] == a // 7
        assert out["L3"] == (a << 3) % 256 and out["R3"] == a >> 3 and out["L2"] == (a << 2) % 256 and out["R1"] == a >> 1
        assert out["LV"] == (a << (b % 8)) % 256 and out["RV"] == a >> (b % 8), (a, b, out)


def test_division_u16(keys):
    client_key, server_key = keys
    circuit, wire_set, _, _ = _circuit("input [15:0] A, B;\noutput [15:0] Q;\ndiv g0(A, B, Q);\n", is_arith=True, is_text=True)
    ac = ArithCircuit(client_key, server_key, circuit)
    enc = ac.encrypt_inputs(wire_set, {"A": PtxtType.U16(51234), "B": PtxtType.U16(321)})
    out = ac.decrypt_outputs(ac.evaluate_encrypted(enc, 1, "u16"), True)
    assert out["Q"] == PtxtType.U16(51234 // 321)


def test_radix_operators_random_u8(keys):
    """Every FheUint operator of arithmetic mode on random operands (one level, all gates batched)."""
    client_key, server_key = keys
    text = """input [7:0] A, B;
output [7:0] S, D, P, Q, L, R, SA, SS, SM, SQ, SL, SR;
add g0(A, B, S);
sub g1(A, B, D);
mult g2(A, B, P);
div g3(A, B, Q);
shl g4(A, B, L);
shr g5(A, B, R);
add g6(A, 77, SA);
sub g7(A, 77, SS);
mult g8(A, 77, SM);
div g9(A, 11, SQ);
shl g10(A, 5, SL);
shr g11(A, 5, SR);
"""
    circuit, wire_set, _, _ = _circuit(text, is_arith=True, is_text=True)
    ac = ArithCircuit(client_key, server_key, circuit)
    rng = np.random.default_rng(2024)
    for cycle in (1, 2, 3):
        a, b = int(rng.integers(0, 256)), int(rng.integers(1, 256))
        enc = ac.encrypt_inputs(wire_set, {"A": PtxtType.U8(a), "B": PtxtType.U8(b)})
        out = {k: v.value for k, v in ac.decrypt_outputs(ac.evaluate_encrypted(enc, cycle, "u8"), True).items()}
        want = {"S": (a + b) % 256, "D": (a - b) % 256, "P": (a * b) % 256, "Q": a // b, "L": (a << (b % 8)) % 256,
                "R": a >> (b % 8), "SA": (a + 77) % 256, "SS": (a - 77) % 256, "SM": (a * 77) % 256, "SQ": a // 11,
                "SL": (a << 5) % 256, "SR": a >> 5}
        assert out == want, (a, b)


def test_arithmetic_mode_under_the_multibit_set():
    """The reference installs PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS for arithmetic mode
    (src/bin/helm.rs:83): same circuits, multi-bit blind rotation (296 group steps).  chi-squared on u32
    (circuit_test.rs:313-370) and an FheUint16 known answer (gates_test.rs:127-310)."""
    client_key, server_key = helm_amd.gen_keys_shortint("shortint_m2c2_multibit3", seed=1)
    assert client_key.params.grouping_factor == 3 and client_key.params.n == 888
    try:
        circuit, wire_set, input_wires, output_wires = _circuit(f"{NET}/chi_squared_arith.v", is_arith=True)
        inputs = verilog_parser.read_input_wires(os.path.join(HERE, "golden", "chi_squared_arith_1.inputs.csv"), "u32")
        ac = ArithCircuit(client_key, server_key, circuit)
        enc = EvalCircuit.evaluate_encrypted(ac, EvalCircuit.encrypt_inputs(ac, wire_set, inputs), 1, "u32")
        out = {k: v.value for k, v in EvalCircuit.decrypt_outputs(ac, enc, True).items()}
        assert out == {"alpha": 529, "beta1": 242, "beta2": 275, "beta3": 1250}
        text = "input [15:0] A, B;\noutput [15:0] S, D, P;\nadd g0(A, B, S);\nsub g1(B, A, D);\nmult g2(A, B, P);\n"
        circuit, wire_set, input_wires, output_wires = _circuit(text, is_arith=True, is_text=True)
        ac = ArithCircuit(client_key, server_key, circuit)
        enc = ac.encrypt_inputs(wire_set, {"A": PtxtType.U16(30), "B": PtxtType.U16(40)})
        out = {k: v.value for k, v in ac.decrypt_outputs(ac.evaluate_encrypted(enc, 1, "u16"), True).items()}
        assert out == {"S": 70, "D": 10, "P": 1200}
    finally:
        server_key.close()


@pytest.mark.parametrize("width,kind", [(64, "u64"), (128, "u128")])
def test_wide_integers_add_sub_mul(keys, width, kind):
    """FheUint64 / FheUint128 (32 / 64 radix blocks): the width match of gates.rs:306-702 beyond u32."""
    client_key, server_key = keys
    text = "input A, B;\noutput S, D, P, M;\nadd g0(A, B, S);\nsub g1(A, B, D);\nmult g2(A, B, P);\nmult g3(A, 1000003, M);\n"
    circuit, wire_set, input_wires, output_wires = _circuit(text, is_arith=True, is_text=True)
    ac = ArithCircuit(client_key, server_key, circuit)
    rng = np.random.default_rng(width)
    a = int.from_bytes(rng.bytes(width // 8), "little")
    b = int.from_bytes(rng.bytes(width // 8), "little")
    mk = PtxtType.U64 if width == 64 else PtxtType.U128
    enc = ac.encrypt_inputs(wire_set, {"A": mk(a), "B": mk(b)})
    out = ac.decrypt_outputs(ac.evaluate_encrypted(enc, 1, kind), True)
    M = 1 << width
    assert out["S"].value == (a + b) % M and out["D"].value == (a - b) % M
    assert out["P"].value == (a * b) % M and out["M"].value == (a * 1000003) % M
    assert all(v.kind == kind.upper() for v in out.values())


@pytest.mark.parametrize("width,kind", [(8, "u8"), (16, "u16"), (32, "u32"), (64, "u64")])
def test_squares_share_their_cross_products(keys, width, kind):
    """mult(x, x): a_j a_k and a_k a_j are one look-up on 2 a_j a_k - the same value mod 2^bits as the general product
    (gates.rs:331-385: `*` on FheUintN), about half the bootstraps, never more rounds.  Extreme operands included: all
    blocks 3 makes every doubled cross product 18 (lo 2, hi 4)."""
    client_key, server_key = keys
    text = "input A, B;\noutput S, T, P;\nmult g0(A, A, S);\nmult g1(B, B, T);\n"
    sq, ws, _, _ = _circuit(text + "add g2(S, T, P);\n", is_arith=True, is_text=True)
    gen, wg, _, _ = _circuit("input A, B;\noutput S;\nmult g0(A, B, S);\n", is_arith=True, is_text=True)
    mk = {8: PtxtType.U8, 16: PtxtType.U16, 32: PtxtType.U32, 64: PtxtType.U64}[width]
    M = 1 << width
    rng = np.random.default_rng(width + 1)
    for a, b in ((M - 1, int.from_bytes(rng.bytes(width // 8), "little")), (int.from_bytes(rng.bytes(width // 8), "little"), 0x55 % M)):
        ac = ArithCircuit(client_key, server_key, sq)
        ac.set_lanes(1)
        out = ac.decrypt_outputs(ac.evaluate_encrypted(ac.encrypt_inputs(ws, {"A": mk(a), "B": mk(b)}), 1, kind), True)
        assert (out["S"].value, out["T"].value, out["P"].value) == (a * a % M, b * b % M, (a * a + b * b) % M)
        pbs_sq, rounds_sq = ac.pbs_per_cycle(), ac.pbs_rounds_per_cycle()
    ag = ArithCircuit(client_key, server_key, gen)
    out = ag.decrypt_outputs(ag.evaluate_encrypted(ag.encrypt_inputs(wg, {"A": mk(a), "B": mk(a)}), 1, kind), True)
    assert out["S"].value == a * a % M
    print(f"{kind}: two squares + their sum {pbs_sq} bootstraps in {rounds_sq} rounds; one general product {ag.pbs_per_cycle()} in {ag.pbs_rounds_per_cycle()}")
    assert pbs_sq < 2 * ag.pbs_per_cycle()


@pytest.mark.parametrize("width,kind", [(32, "u32"), (64, "u64")])
def test_division_and_shifts_wide(keys, width, kind):
    """div / shl / shr with encrypted and plain right-hand sides at FheUint32 and FheUint64 (the width match of
    gates.rs:386-452, 488-600; circuit.rs:1363-1435): one level, all six operators batched."""
    client_key, server_key = keys
    text = ("input A, B;\noutput Q, QS, L, R, LS, RS;\ndiv g0(A, B, Q);\ndiv g1(A, 1000003, QS);\nshl g2(A, B, L);\n"
            "shr g3(A, B, R);\nshl g4(A, 13, LS);\nshr g5(A, 21, RS);\n")
    circuit, wire_set, _, _ = _circuit(text, is_arith=True, is_text=True)
    ac = ArithCircuit(client_key, server_key, circuit)
    rng = np.random.default_rng(width)
    a = int.from_bytes(rng.bytes(width // 8), "little")
    b = int.from_bytes(rng.bytes(width // 16), "little") | 1  # half-width divisor: a multi-digit quotient
    mk = {32: PtxtType.U32, 64: PtxtType.U64}[width]
    enc = ac.encrypt_inputs(wire_set, {"A": mk(a), "B": mk(b)})
    out = {k: v.value for k, v in ac.decrypt_outputs(ac.evaluate_encrypted(enc, 1, kind), True).items()}
    M = 1 << width
    assert out == {"Q": a // b, "QS": a // 1000003, "L": (a << (b % width)) % M, "R": a >> (b % width),
                   "LS": (a << 13) % M, "RS": a >> 21}, (a, b)


def test_division_u128(keys):
    """FheUint128 / FheUint128 (64 radix blocks, 128 restoring steps), and x / 0 = all ones as tfhe's."""
    client_key, server_key = keys
    circuit, wire_set, _, _ = _circuit("input A, B, Z;\noutput Q, QZ;\ndiv g0(A, B, Q);\ndiv g1(A, Z, QZ);\n",
                                       is_arith=True, is_text=True)
    ac = ArithCircuit(client_key, server_key, circuit)
    a = 0xFEDCBA9876543210_0123456789ABCDEF
    b = 0x1_0000_0001_F00D
    enc = ac.encrypt_inputs(wire_set, {"A": PtxtType.U128(a), "B": PtxtType.U128(b), "Z": PtxtType.U128(0)})
    out = ac.decrypt_outputs(ac.evaluate_encrypted(enc, 1, "u128"), True)
    assert out["Q"].value == a // b and out["QZ"].value == (1 << 128) - 1


def test_preprocessed_behavioural_chi_squared_u32(keys):
    """README.md:116-120 end to end: behavioural arithmetic Verilog -> preprocessor --arithmetic -> arithmetic mode
    (BASELINE config 5), decrypting to the chi-squared known answer (2, 7, 9) -> (529, 242, 275, 1250)."""
    from helm_amd.preprocessor import preprocess
    client_key, server_key = keys
    raw = """module chi_squared(N0, N1, N2, alpha, beta1, beta2, beta3);
  input [31:0] N0, N1, N2;
  output [31:0] alpha, beta1, beta2, beta3;
  wire [31:0] t, u, v;
  assign t = 4 * N0 * N2 - N1 * N1;
  assign alpha = t * t;
  assign u = 2 * N0 + N1;
  assign v = 2 * N2 + N1;
  assign beta1 = 2 * (u * u);
  assign beta2 = u * v;
  assign beta3 = (v * v) << 1;
endmodule
"""
    circuit, wire_set, _, _ = _circuit(preprocess(raw, arithmetic=True), is_arith=True, is_text=True)
    ac = ArithCircuit(client_key, server_key, circuit)
    enc = ac.encrypt_inputs(wire_set, {"N0": PtxtType.U32(2), "N1": PtxtType.U32(7), "N2": PtxtType.U32(9)})
    out = {k: v.value for k, v in ac.decrypt_outputs(ac.evaluate_encrypted(enc, 1, "u32"), True).items()}
    assert out == {"alpha": 529, "beta1": 242, "beta2": 275, "beta3": 1250}


def test_merged_rounds_cut_at_capacity_one_are_bit_identical(keys):
    """RoundMerger with the capacity forced to ONE ciphertext per launch: every look-up round of every chain is cut into
    single-ciphertext launches, i.e. at every position a cut can fall.  The one-call guarantee (all keyswitches before any
    bootstrap writes) then only holds inside each part, so a batch that listed an in-place writer before a reader of the same
    row would return other ciphertexts here - the rounds' ordering contract (readers first), which the merger now checks for
    every round whatever the device's capacity.  Same ciphertext on every wire as the level-by-level evaluation."""
    client_key, server_key = keys
    text = """input [7:0] A, B, C, D, E, F, G;
output [7:0] X, Y, Z, W;
mult g0(A, B, p0);
mult g1(C, C, p1);
add g2(p0, p1, X);
sub g3(D, E, t0);
shl g4(t0, 3, Y);
add g5(F, 77, t1);
mult g6(t1, 6, Z);
shr g7(G, 1, W);
"""
    circuit, wire_set, _, _ = _circuit(text, is_arith=True, is_text=True)
    vals = dict(A=201, B=77, C=254, D=3, E=200, F=250, G=0xB7)
    m = 256
    want = {"p0": vals["A"] * vals["B"] % m, "p1": vals["C"] ** 2 % m, "t0": (vals["D"] - vals["E"]) % m,
            "t1": (vals["F"] + 77) % m, "W": vals["G"] >> 1}
    want.update(X=(want["p0"] + want["p1"]) % m, Y=(want["t0"] << 3) % m, Z=want["t1"] * 6 % m)
    ac = ArithCircuit(client_key, server_key, circuit)
    enc = ac.encrypt_inputs(wire_set, {k: PtxtType.U8(v) for k, v in vals.items()})
    ac.set_lanes(1)
    flat = ac.evaluate_encrypted(enc, 1, "u8")
    assert {w: _decrypt_int(client_key, flat[w]) for w in want} == want
    ac2 = ArithCircuit(client_key, server_key, circuit)
    default = ac2.evaluate_encrypted(enc, 1, "u8")
    assert "rounds merged" in ac2.log()
    launches_default = ac2.pbs_rounds_per_cycle()
    ac2.set_round_capacity(1)
    one = ac2.evaluate_encrypted(enc, 1, "u8")  # the capacity change reset the same-cycle memo: this is a real evaluation
    assert ac2.memo_hits() == 0
    assert ac2.pbs_rounds_per_cycle() == ac2.pbs_per_cycle() > launches_default  # one launch per look-up
    for w in flat.keys():
        assert np.array_equal(one[w], flat[w]), w
        assert np.array_equal(default[w], flat[w]), w


def test_arithmetic_memo_can_be_switched_off_and_reset(keys):
    """The arithmetic-mode same-cycle memo follows the reference (gates.rs:307-312: keyed on the cycle alone); the opt-out
    the other two evaluators do not need: set_memo(False) / reset_memo(), and a schedule switch resets it."""
    client_key, server_key = keys
    circuit, wire_set, _, _ = _circuit("input [7:0] A, B;\noutput [7:0] X;\nadd g0(A, B, X);\n", is_arith=True, is_text=True)
    ac = ArithCircuit(client_key, server_key, circuit)
    e1 = ac.encrypt_inputs(wire_set, {"A": PtxtType.U8(10), "B": PtxtType.U8(20)})
    e2 = ac.encrypt_inputs(wire_set, {"A": PtxtType.U8(100), "B": PtxtType.U8(7)})
    assert _decrypt_int(client_key, ac.evaluate_encrypted(e1, 1, "u8")["X"]) == 30
    assert _decrypt_int(client_key, ac.evaluate_encrypted(e2, 1, "u8")["X"]) == 30 and ac.memo_hits() == 1  # the reference's behaviour
    ac.reset_memo()
    assert _decrypt_int(client_key, ac.evaluate_encrypted(e2, 1, "u8")["X"]) == 107 and ac.memo_hits() == 1
    ac.set_memo(False)
    assert _decrypt_int(client_key, ac.evaluate_encrypted(e1, 1, "u8")["X"]) == 30 and ac.memo_hits() == 1
    ac.set_memo(True)
    assert _decrypt_int(client_key, ac.evaluate_encrypted(e2, 5, "u8")["X"]) == 107
    ac.set_lazy_carries(False)  # a schedule switch: the remembered cycle is gone
    assert _decrypt_int(client_key, ac.evaluate_encrypted(e1, 5, "u8")["X"]) == 30 and ac.memo_hits() == 1


def test_radix_level_scalar_power_of_two_beyond_the_width(keys):
    """helm_host_radix_level takes a raw 128-bit scalar: x * 2^s with s >= bits is x * 0 = 0 mod 2^bits, not a shift by
    s mod bits; x * (2^bits + 3) is x * 3 (gates.rs:560-600 multiplies mod 2^bits)."""
    import ctypes as C
    from helm_amd import _host as H
    client_key, server_key = keys
    nb = 4  # u8

    class Op(C.Structure):
        _fields_ = [("kind", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("out", C.c_int32),
                    ("scalar_lo", C.c_uint64), ("scalar_hi", C.c_uint64)]
    MUL_SCALAR = 9
    scalars = [1 << 8, 1 << 9, (1 << 8) + 3, 1 << 7, 1 << 100]
    ops = (Op * len(scalars))()
    for i, sc in enumerate(scalars):
        ops[i] = Op(MUL_SCALAR, 0, -1, nb * (1 + i), sc & (2**64 - 1), sc >> 64)
    x = 0xB5
    ints = 1 + len(scalars)
    scratch = int(H.host.helm_host_radix_scratch_rows(server_key._h, nb, C.cast(ops, C.c_void_p), len(scalars)))
    assert scratch >= 0
    w = server_key.wires(ints * nb + scratch)
    w.upload(np.arange(nb), client_key.encrypt(np.array([(x >> (2 * i)) & 3 for i in range(nb)], dtype=np.uint64)))
    pbs, rounds = C.c_int64(), C.c_int64()
    H.check(H.host.helm_host_radix_level(server_key._h, w._h, nb, C.cast(ops, C.c_void_p), len(scalars), ints * nb,
                                         C.byref(pbs), C.byref(rounds)))
    server_key.sync()
    got = [_decrypt_int(client_key, w.download(np.arange(nb * (1 + i), nb * (2 + i)))) for i in range(len(scalars))]
    assert got == [x * sc % 256 for sc in scalars] == [0, 0, x * 3 % 256, x * 128 % 256, 0]


def test_lut_mode_flip_flops_fed_by_flip_flops(keys):
    """LUT mode's DFF level (circuit.rs:1063-1069 copies) with a shift register and a swap of registers: rows that one
    flip-flop of the level writes and another reads are copied to scratch rows first (snapshot semantics, as the plaintext
    evaluator's; helm_si_eval_lut_level refuses a read-after-write inside a level) and every wire equals the plaintext
    evaluation on every cycle; the per-level timing lines can be switched off (no host synchronisation per level)."""
    client_key, server_key = keys
    text = """input x;
output q2, s0, s1;
dff ga(d, q0);
dff gb(q0, q1);
dff gc(q1, q2);
dff gd(s1, s0);
dff ge(s0, s1);
lut g9(0x6, x, q2, d);
"""
    circuit, wire_set, _, _ = _circuit(text, is_text=True)
    lc = LutCircuit(client_key, server_key, circuit)
    lc.set_timing_lines(False)
    state = {"x": 1, "q0": 0, "q1": 0, "q2": 0, "s0": 0, "s1": 0}
    enc = lc.encrypt_inputs(wire_set, {k: PtxtType.Bool(bool(v)) for k, v in state.items()})
    for w in ("q1", "s0"):
        enc.insert(w, client_key.encrypt(1))
        state[w] = 1
    ptxt = {w: PtxtType.None_() for w in wire_set}
    ptxt.update({k: PtxtType.Bool(bool(v)) for k, v in state.items()})
    for cycle in range(4):
        ptxt = circuit.evaluate(ptxt)
        enc = lc.evaluate_encrypted(enc, cycle + 1, "bool")
        for w in sorted(ptxt):
            assert int(client_key.decrypt(enc[w])) == int(bool(ptxt[w].value)), (cycle, w)
    assert "PBS time" not in lc.log()
    # the engine itself refuses a level with a read-after-write / write-after-write inside
    w = server_key.wires(8)
    w.upload(np.arange(3), client_key.encrypt(np.array([1, 0, 1], dtype=np.uint64)))
    with pytest.raises(helm_amd.HelmError, match="read-after-write"):
        w.eval_lut_level([2, 0], [[0, 1], [3, -1]], [0x6, 0], [3, 4])      # the copy reads row 3, which the LUT writes
    with pytest.raises(helm_amd.HelmError, match="write-after-write"):
        w.eval_lut_level([2, 2], [[0, 1], [1, 2]], [0x6, 0x8], [3, 3])
    w.eval_lut_level([2], [[0, 1]], [0x6], [0])                            # in place: row 0 <- XOR(row 0, row 1)
    server_key.sync()
    assert int(client_key.decrypt(w.download([0]))[0]) == 1
