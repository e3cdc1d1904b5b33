"""Same-cycle memo of the three evaluators (reference src/gates.rs:55-59: a Gate keeps `cycle` and its last encrypted
output; gates.rs:288-292 and 307-312 return it when the cycle repeats).

tests/gates_test.rs:110-311 (`caching_of_gate_evaluation`, K-7) is mirrored for arithmetic mode exactly as written:
FheUint16 {10, 20, 30, 40}; in the same cycle the gates are called again WITH OTHER OPERANDS and must hand back the
first result, faster; the next cycle computes anew.  Gates mode and LUT mode memoise only when nothing observable
changes (the reference's boolean probe is commented out, gates.rs:247-252, and its LUT probe never sees a stored
cycle, gates.rs:282-304): same cycle AND the very same unmodified input map -> identical ciphertexts, no new
bootstraps in get_timing."""
import os
import time

import numpy as np
import pytest

import helm_amd
from helm_amd import ArithCircuit, Circuit, GateCircuit, LutCircuit, PtxtType, verilog_parser

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NET = os.path.join(HERE, "netlists")


def _circuit(path_or_text, is_arith=False, is_text=False):
    rd = verilog_parser.read_verilog_text if is_text else verilog_parser.read_verilog_file
    gates_set, wire_set, input_wires, output_wires, dffs, _, _ = rd(path_or_text, is_arith)
    c = Circuit(gates_set, input_wires, output_wires, dffs)
    c.sort_circuit()
    c.compute_levels()
    return c, wire_set, input_wires, output_wires


@pytest.fixture(scope="module")
def si_keys():
    ck, sk = helm_amd.gen_keys_shortint("shortint_m2c2", seed=1)
    yield ck, sk
    sk.close()


def test_caching_of_gate_evaluation_arithmetic(si_keys):  # gates_test.rs:110-311
    client_key, server_key = si_keys
    text = "input [15:0] A, B;\noutput [15:0] S, D, P;\nadd g0(A, B, S);\nsub g1(B, A, D);\nmult g2(A, B, P);\n"
    circuit, wire_set, _, _ = _circuit(text, is_arith=True, is_text=True)
    ac = ArithCircuit(client_key, server_key, circuit)
    ptxt = [10, 20, 30, 40]
    enc = [ac.encrypt_inputs(wire_set, {"A": PtxtType.U16(ptxt[i]), "B": PtxtType.U16(ptxt[i + 1])}) for i in range(3)]
    server_key.timing_enable(True)
    server_key.timing(reset=True)

    def run(cycle, operands):
        t0 = time.perf_counter()
        out = ac.evaluate_encrypted(enc[operands], cycle, "u16")
        dt = time.perf_counter() - t0
        return out, {k: v.value for k, v in ac.decrypt_outputs(out, True).items()}, dt

    for cycle in (1, 2, 3):
        a, b = ptxt[cycle - 1], ptxt[cycle]
        want = {"S": a + b, "D": b - a, "P": a * b}
        hits = ac.memo_hits()
        out, dec, elapsed = run(cycle, cycle - 1)
        assert dec == want
        pbs = server_key.timing().pbs_count
        assert pbs > 0 and ac.memo_hits() == hits
        # "These should have been cached since the cycle is the same" (gates_test.rs:196-223): other operands, the
        # same cycle -> the first result, no bootstrap, faster
        other = (cycle + 1) % 3
        cached, dec_cached, elapsed_cached = run(cycle, other)
        assert dec_cached == want
        assert server_key.timing().pbs_count == pbs, "a memo hit must not launch bootstraps"
        assert ac.memo_hits() == hits + 1
        assert elapsed_cached < elapsed
        for wire in ("S", "D", "P"):
            assert np.array_equal(cached[wire], out[wire]), wire          # the very ciphertexts of the first call
        for wire in ("A", "B"):                                           # input wires are the caller's, not cached
            assert np.array_equal(cached[wire], enc[other][wire]), wire
        assert "already evaluated" in ac.log()
    server_key.timing_enable(False)


def test_lut_circuit_memo_and_pbs_time_lines(si_keys):
    client_key, server_key = si_keys
    circuit, wire_set, _, _ = _circuit(f"{NET}/8-bit-adder-lut-3-1.v")
    inputs = {f"a[{i}]": PtxtType.Bool((0xB7 >> i) & 1) for i in range(8)}
    inputs.update({f"b[{i}]": PtxtType.Bool((0x6E >> i) & 1) for i in range(8)})
    inputs["cin"] = PtxtType.Bool(1)
    lc = LutCircuit(client_key, server_key, circuit)
    enc = lc.encrypt_inputs(wire_set, inputs)
    server_key.timing_enable(True)
    server_key.timing(reset=True)
    first = lc.evaluate_encrypted(enc, 1, "bool")
    log = lc.log()
    # gates.rs:293-302: one "PBS time: {} us" line per LUT gate
    assert log.count("PBS time: ") == 16 and all(ln.endswith(" us") for ln in log.splitlines() if ln.startswith("PBS time"))
    pbs = server_key.timing().pbs_count
    assert pbs == 16
    again = lc.evaluate_encrypted(enc, 1, "bool")                        # same cycle, same unmodified map
    assert lc.memo_hits() == 1 and server_key.timing().pbs_count == pbs
    for wire in first.keys():
        assert np.array_equal(first[wire], again[wire]), wire
    again["cout"] = client_key.encrypt(0)                                 # the returned map is the caller's own copy
    third = lc.evaluate_encrypted(enc, 1, "bool")
    assert np.array_equal(third["cout"], first["cout"]) and lc.memo_hits() == 2
    lc.evaluate_encrypted(enc, 2, "bool")                                 # another cycle computes anew
    assert lc.memo_hits() == 2 and server_key.timing().pbs_count == 2 * pbs
    enc["cin"] = client_key.encrypt(0)                                    # a modified input map computes anew
    out = lc.evaluate_encrypted(enc, 2, "bool")
    assert lc.memo_hits() == 2 and server_key.timing().pbs_count == 3 * pbs
    dec = lc.decrypt_outputs(out, True)
    assert sum(dec[f"sum[{i}]"].value << i for i in range(8)) + (dec["cout"].value << 8) == 0xB7 + 0x6E
    server_key.timing_enable(False)


def test_gate_circuit_memo():
    client_key, server_key = helm_amd.gen_keys("toy_k2", seed=3)
    try:
        circuit, wire_set, input_wires, _ = _circuit(f"{NET}/2-bit-adder.v")
        gc = GateCircuit(client_key, server_key, circuit)
        vals = {"a[0]": True, "a[1]": False, "b[0]": True, "b[1]": True, "cin": True}
        enc = gc.encrypt_inputs(wire_set, {k: PtxtType.Bool(v) for k, v in vals.items()})
        server_key.timing_enable(True)
        server_key.timing(reset=True)
        first = gc.evaluate_encrypted(enc, 1, "bool")
        pbs = server_key.timing().pbs_count
        assert pbs == gc.pbs_per_cycle() > 0
        again = gc.evaluate_encrypted(enc, 1, "bool")
        assert gc.memo_hits() == 1 and server_key.timing().pbs_count == pbs
        for wire in first.keys():
            assert np.array_equal(first[wire], again[wire]), wire
        gc.evaluate_encrypted(enc, 2, "bool")                             # another cycle computes anew
        assert gc.memo_hits() == 1 and server_key.timing().pbs_count == 2 * pbs
        enc["cin"] = client_key.encrypt(False)                            # a modified input map computes anew
        out = gc.evaluate_encrypted(enc, 2, "bool")
        assert gc.memo_hits() == 1 and server_key.timing().pbs_count == 3 * pbs
        dec = gc.decrypt_outputs(out, True)
        assert dec["sum[0]"].value + 2 * dec["sum[1]"].value + 4 * dec["cout"].value == 1 + 3
        # a sequential loop feeds every cycle's output map back in with the driver's constant cycle = 1 (helm.rs:261):
        # a new map each time, never a hit
        hits = gc.memo_hits()
        nxt = gc.evaluate_encrypted(out, 2, "bool")
        gc.evaluate_encrypted(nxt, 2, "bool")
        assert gc.memo_hits() == hits
    finally:
        server_key.timing_enable(False)
        server_key.close()
