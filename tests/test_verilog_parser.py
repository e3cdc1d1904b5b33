"""Mirror of reference tests/verilog_parser_test.rs.  The reference's fixtures live in
an absent submodule; the files used here are authored to the contents those tests imply
(tests/verilog_parser_test.rs:54-60, 72-76, 121-141)."""
import os

import pytest

from helm_amd import PtxtType
from helm_amd._host import Panic
from helm_amd.verilog_parser import (read_input_wires, read_verilog_file, read_verilog_text, write_output_wires,
                                     parse_input_wire, hex_to_bitstring)

HERE = os.path.dirname(os.path.abspath(__file__))
NET = os.path.join(HERE, "netlists")
GOLD = os.path.join(HERE, "golden")


def test_parse_two_bit_adder():  # verilog_parser_test.rs:5-12
    gates, wire_set, inputs, _, _, _, _ = read_verilog_file(f"{NET}/2-bit-adder.v", False)
    assert len(gates) == 10
    assert len(wire_set) == 10
    assert len(inputs) == 5


def test_input_wires_gates_parser():  # :14-26
    _, _, inputs, _, _, _, _ = read_verilog_file(f"{NET}/2-bit-adder.v", False)
    m = read_input_wires(f"{GOLD}/2-bit-adder.inputs.csv", "bool")
    assert len(m) == len(inputs)
    for w in inputs:
        assert w in m


def test_input_wires_arithmetic_parser():  # :28-44
    _, _, inputs, _, _, _, has_arith = read_verilog_file(f"{NET}/chi_squared_arith.v", True)
    m = read_input_wires(f"{GOLD}/chi_squared_arith_1.inputs.csv", "u32")
    assert has_arith
    assert len(m) == len(inputs)
    for w in inputs:
        assert w in m


def test_invalid_arithmetic_with_luts_parser():  # :46-52
    with pytest.raises(Panic, match="Can't mix LUTs with arithmetic operators!"):
        read_verilog_file(f"{NET}/invalid.v", True)


def test_bool_input_wires():  # :61-70
    m = read_input_wires(f"{GOLD}/2-bit-adder.inputs.csv", "bool")
    assert m["a[0]"] == PtxtType.Bool(True)
    assert m["a[1]"] == PtxtType.Bool(False)
    assert m["b[0]"] == PtxtType.Bool(False)
    assert m["b[1]"] == PtxtType.Bool(True)
    assert m["cin"] == PtxtType.Bool(False)


@pytest.mark.parametrize("t", ["u8", "u16", "u32", "u64", "u128"])
def test_integer_input_wires(t):  # :77-118
    m = read_input_wires(f"{GOLD}/chi_squared_arith_1.inputs.csv", t)
    mk = getattr(PtxtType, t.upper())
    assert m["N0"] == mk(2) and m["N1"] == mk(7) and m["N2"] == mk(9)


def test_bool_input_wires_array_as_int():  # :121-141
    m = read_input_wires(f"{GOLD}/bool_array.input.csv", "bool")
    assert [m[f"in1[{i}]"].value for i in range(4)] == [1, 0, 1, 1]
    assert [m[f"in2[{i}]"].value for i in range(4)] == [0, 1, 0, 1]
    assert [m[f"in3[{i}]"].value for i in range(6)] == [1, 1, 0, 1, 0, 0]


# ---- grammar details of verilog_parser.rs:31-276 not covered by the reference's tests ----
def test_gate_forms_and_lut_expansion():
    text = """module m(a, b, c, y);
// comment
input a, b, c;
output y;
wire w1, w2;
not g0(a, na);
buf g1(b,nb);
mux g2(a, b, c, m0);
lut g3(0x96, a, b, c, l0);
lut g4(150, a, b, c, l1);
dff g5(m0, q);
cone g6(one);
czero g7(zero);
nand g8(na,nb,y);
endmodule
"""
    gates, wire_set, inputs, outputs, dffs, has_luts, has_arith = read_verilog_text(text, False)
    by = {g.gate_name: g for g in gates}
    assert by["g0"].input_wires == ["a"] and by["g0"].output_wire == "na"
    assert by["g1"].input_wires == ["b"] and by["g1"].output_wire == "nb"
    assert by["g2"].input_wires == ["a", "b", "c"] and by["g2"].output_wire == "m0"
    assert by["g3"].lut_const == [0, 1, 1, 0, 1, 0, 0, 1] and by["g3"].input_wires == ["a", "b", "c"]
    assert by["g4"].lut_const == by["g3"].lut_const  # 150 == 0x96
    assert by["g6"].output_wire == "one" and by["g6"].input_wires == []
    assert by["g8"].input_wires == ["na", "nb"] and by["g8"].output_wire == "y"
    assert dffs == ["q"] and inputs == ["a", "b", "c", "q"]  # DFF outputs become inputs (:225-227)
    assert has_luts and not has_arith
    assert wire_set == {"na", "nb", "m0", "l0", "l1", "q", "one", "zero", "y"}
    assert outputs == ["y"]


def test_ranged_declarations():
    text = "module m(x, y);\ninput [3:0] x;\noutput [0:1] y;\nand g(x[0], x[1], y[0]);\nendmodule\n"
    _, _, inputs, outputs, _, _, _ = read_verilog_text(text, False)
    assert inputs == ["x[0]", "x[1]", "x[2]", "x[3]"] and outputs == ["y[0]", "y[1]"]
    # arithmetic keeps bare bus names for every listed identifier (:180-185)
    text = "module m(A, B, Y);\ninput [31:0] A, B;\noutput [31:0] Y;\nadd g(A, B, Y);\nendmodule\n"
    _, _, inputs, outputs, _, _, has_arith = read_verilog_text(text, True)
    assert inputs == ["A", "B"] and outputs == ["Y"] and has_arith


def test_parser_panics():
    with pytest.raises(Panic, match="no gates detected"):
        read_verilog_text("module m(a);\ninput a;\nendmodule\n", False)
    with pytest.raises(Panic, match='Invalid gate type "frob"'):
        read_verilog_text("frob g(a, b, c);\n", False)
    with pytest.raises(Panic, match="Failed to parse hex"):
        read_verilog_text("lut g(0xZZ, a, b, y);\n", False)
    with pytest.raises(Panic):
        read_verilog_file("/nonexistent/file.v", False)


def test_duplicate_gate_name_first_wins():  # HashSet<Gate> keyed by name (gates.rs:64-87)
    gates, _, _, _, _, _, _ = read_verilog_text("and g(a, b, x);\nor g(a, b, y);\n", False)
    assert len(gates) == 1 and next(iter(gates)).gate_type.name == "And"


def test_parse_input_wire_and_hex():  # lib.rs:90-106, 181-194
    assert parse_input_wire("1", "bool") == PtxtType.Bool(True)
    assert parse_input_wire("true", "bool") == PtxtType.Bool(True)
    assert parse_input_wire("yes", "bool") == PtxtType.Bool(False)
    assert parse_input_wire("255", "u8") == PtxtType.U8(255)
    with pytest.raises(Panic):
        parse_input_wire("256", "u8")
    assert parse_input_wire(str(2**128 - 1), "u128") == PtxtType.U128(2**128 - 1)
    assert hex_to_bitstring("D") == "1101" and hex_to_bitstring("a5") == "10100101"


def test_csv_errors_and_write(tmp_path):
    bad = tmp_path / "bad.csv"
    bad.write_text("wire,a,b,c\nx,1,2,3\n")
    with pytest.raises(Panic, match="either two or three columns"):
        read_input_wires(str(bad), "bool")
    out = tmp_path / "out.csv"
    write_output_wires(str(out), {"cout": PtxtType.Bool(True), "s": PtxtType.U32(7)})
    assert out.read_text().splitlines() == ["cout, true", "s, 7"]
