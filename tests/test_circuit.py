"""Mirror of the plaintext half of reference tests/circuit_test.rs plus the scheduler
semantics of src/circuit.rs:122-239."""
import os

import pytest

from helm_amd import Circuit, PtxtType, verilog_parser
from helm_amd._host import Panic

NET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "netlists")


def test_two_bit_adder():  # circuit_test.rs:17-45
    gates_set, wire_set, input_wires, _, _, _, _ = verilog_parser.read_verilog_file(f"{NET}/2-bit-adder.v", False)
    circuit = Circuit(gates_set, input_wires, [], [])
    circuit.sort_circuit()
    assert len(circuit.get_ordered_gates()) == 10
    circuit.compute_levels()
    wire_map = {w: PtxtType.Bool(True) for w in wire_set}
    wire_map.update({w: PtxtType.Bool(True) for w in input_wires})
    wire_map = circuit.evaluate(wire_map)
    assert len(wire_map) == 15
    assert len(input_wires) == 5
    assert wire_map["sum[0]"] == PtxtType.Bool(True)
    assert wire_map["sum[1]"] == PtxtType.Bool(True)
    assert wire_map["cout"] == PtxtType.Bool(True)
    assert wire_map["i0"] == PtxtType.Bool(False)
    assert wire_map["i1"] == PtxtType.Bool(False)


def test_two_bit_adder_all_inputs():
    gates_set, wire_set, input_wires, _, _, _, _ = verilog_parser.read_verilog_file(f"{NET}/2-bit-adder.v", False)
    circuit = Circuit(gates_set, input_wires, [], [])
    circuit.sort_circuit()
    circuit.compute_levels()
    for v in range(32):
        a, b, cin = v & 3, (v >> 2) & 3, v >> 4
        m = {w: PtxtType.None_() for w in wire_set}
        m.update({"a[0]": PtxtType.Bool(a & 1), "a[1]": PtxtType.Bool(a >> 1), "b[0]": PtxtType.Bool(b & 1),
                  "b[1]": PtxtType.Bool(b >> 1), "cin": PtxtType.Bool(cin)})
        out = circuit.evaluate(m)
        s = out["sum[0]"].value + 2 * out["sum[1]"].value + 4 * out["cout"].value
        assert s == a + b + cin


def test_levels_and_order():
    text = """input a, b;
output y;
xor gz(a, b, t1);
and ga(a, b, t0);
not gm(t0, t2);
or  gq(t1, t2, y);
"""
    gates, _, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(text, False)
    c = Circuit(gates, inputs, outputs, dffs)
    c.sort_circuit()
    # rounds of ready gates, each round sorted by gate name (circuit.rs:164)
    assert [g.gate_name for g in c.get_ordered_gates()] == ["ga", "gz", "gm", "gq"]
    c.compute_levels()
    assert c.get_ordered_gates() == []  # cleared (circuit.rs:238)
    lm = c.level_map()
    assert {k: sorted(g.gate_name for g in v) for k, v in lm.items()} == {1: ["ga", "gz"], 2: ["gm"], 3: ["gq"]}


def test_dff_goes_to_last_level_and_lut_msb_first():
    text = """input a, b, c;
output y;
lut g0(0xCA, a, b, c, y);
dff g1(y, q);
and g2(q, a, z);
"""
    gates, wire_set, inputs, outputs, dffs, has_luts, _ = verilog_parser.read_verilog_text(text, False)
    c = Circuit(gates, inputs, outputs, dffs)
    c.sort_circuit()
    assert c.get_ordered_gates()[-1].gate_type.name == "Dff"  # deferred to the end (circuit.rs:167)
    c.compute_levels()
    lm = c.level_map()
    assert max(lm) == len(lm) and [g.gate_name for g in lm[max(lm)]] == ["g1"]  # circuit.rs:226-234
    # LUT index: first input is the MSB (gates.rs:159-167): table 0xCA = a ? b : c
    for v in range(8):
        a, b, cc = (v >> 2) & 1, (v >> 1) & 1, v & 1
        m = {w: PtxtType.None_() for w in wire_set}
        m.update({"a": PtxtType.Bool(a), "b": PtxtType.Bool(b), "c": PtxtType.Bool(cc), "q": PtxtType.Bool(0)})
        out = c.evaluate(m)
        assert out["y"].value == (0xCA >> v) & 1
        assert out["q"].value == out["y"].value  # DFF latches in the last level


def test_nary_plaintext_gates_and_mux():  # gates.rs:154-157,189-232
    text = "input a, b, s;\nmux g0(a, b, s, m);\nxnor g1(a, b, x);\nnor g2(a, b, n);\n"
    gates, wire_set, inputs, _, _, _, _ = verilog_parser.read_verilog_text(text, False)
    c = Circuit(gates, inputs, [], [])
    c.sort_circuit()
    c.compute_levels()
    for v in range(8):
        a, b, s = v & 1, (v >> 1) & 1, v >> 2
        m = {w: PtxtType.None_() for w in wire_set}
        m.update({"a": PtxtType.Bool(a), "b": PtxtType.Bool(b), "s": PtxtType.Bool(s)})
        out = c.evaluate(m)
        assert out["m"].value == (a if s else b)
        assert out["x"].value == 1 - (a ^ b) and out["n"].value == 1 - (a | b)


def test_constants_are_level0_gates():
    """DEVIATION from circuit.rs:142-147 (constants dropped): see DESIGN.md."""
    gates, wire_set, inputs, _, _, _, _ = verilog_parser.read_verilog_text(
        "input a;\ncone g0(one);\nczero g1(zero);\nand g2(a, one, y);\nor g3(a, zero, z);\n", False)
    c = Circuit(gates, inputs, [], [])
    c.sort_circuit()
    assert [g.gate_name for g in c.get_ordered_gates()[:2]] == ["g0", "g1"]
    c.compute_levels()
    m = {w: PtxtType.None_() for w in wire_set}
    m["a"] = PtxtType.Bool(True)
    out = c.evaluate(m)
    assert out["one"].value == 1 and out["zero"].value == 0 and out["y"].value == 1 and out["z"].value == 1


def test_scheduler_assertions_and_errors():
    gates, _, inputs, _, _, _, _ = verilog_parser.read_verilog_text("input a;\nand g(a, ghost, y);\n", False)
    c = Circuit(gates, inputs, [], [])
    with pytest.raises(Panic, match="undriven|loop"):
        c.sort_circuit()
    gates, _, inputs, _, _, _, _ = verilog_parser.read_verilog_text("input a, b;\nand g(a, b, y);\n", False)
    c = Circuit(gates, inputs, [], [])
    with pytest.raises(Panic, match="assertion failed"):
        c.compute_levels()  # before sort_circuit (circuit.rs:176-177)
    c.sort_circuit()
    with pytest.raises(Panic, match="assertion failed"):
        c.sort_circuit()    # twice (circuit.rs:124)
    c.compute_levels()
    with pytest.raises(Panic):
        c.evaluate({"a": PtxtType.Bool(1)})  # missing wire -> index panic in the reference


def test_initialize_wire_map():  # circuit.rs:245-333
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(
        "input a;\noutput y;\nnot g0(a, y);\ndff g1(y, q);\n", False)
    c = Circuit(gates, inputs, outputs, dffs)
    m = c.initialize_wire_map(wire_set, {}, "bool")
    assert m["a"] == PtxtType.Bool(False) and m["q"] == PtxtType.Bool(False) and m["y"].kind == "None"
    m = c.initialize_wire_map(wire_set, {"a": PtxtType.Bool(True), "q": PtxtType.Bool(True)}, "bool")
    assert m["a"] == PtxtType.Bool(True) and m["q"] == PtxtType.Bool(False)  # DFF state starts at 0
    with pytest.raises(Panic, match='Input wire "q" not in input wires!'):
        c.initialize_wire_map(wire_set, {"a": PtxtType.Bool(True)}, "bool")
