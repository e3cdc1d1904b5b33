"""The shortint oracle (oracle/shortint_oracle.c) against what the reference pins for LUT mode:
decrypted truth tables of gates::lut() for every arity it distinguishes (reference
src/gates.rs:746-785, index convention src/gates.rs:159-167), generate_lookup_table's
semantics on every plaintext value, and the 8-bit LUT adder on every wire
(reference tests/circuit_test.rs:266-311).  CPU only."""
import itertools
import os

import numpy as np
import pytest

import helm_amd
import oracle
from helm_amd import Circuit, PtxtType, verilog_parser

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def toy():
    ck = helm_amd.SiClientKey.generate("si_toy_512", seed=2)
    return ck, oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk)


def test_decompose_and_modswitch_definitions():
    L = oracle.lib64()
    import ctypes as C
    rng = np.random.default_rng(1)
    for logB, l in ((23, 1), (15, 2), (3, 5), (4, 4)):
        for x in [0, 1, 2**63, 2**64 - 1] + [int(v) for v in rng.integers(0, 2**63, 40)]:
            d = (C.c_int64 * l)()
            L.orc64_decompose(x, logB, l, d)
            assert all(-(1 << (logB - 1)) <= v <= (1 << (logB - 1)) for v in d)
            rec = sum(int(v) << (64 - logB * (j + 1)) for j, v in enumerate(d)) % 2**64
            err = (rec - x + 2**63) % 2**64 - 2**63
            assert abs(err) <= 1 << (64 - logB * l - 1)  # closest representable
    assert L.orc64_modswitch(2**63, 11) == 1024 and L.orc64_modswitch(2**64 - 1, 11) == 0
    assert L.orc64_modswitch((1 << 52) - 1, 11) == 0 and L.orc64_modswitch(1 << 52, 11) == 1  # round half up


def test_apply_lookup_table_every_value(toy):
    ck, orc = toy
    f = lambda v: (v * v + 1) % ck.t
    lut = orc.make_lut(f)
    for v in range(ck.t):
        out = orc.apply_lut(ck.encrypt(v), lut)
        assert ck.decrypt_message_and_carry(out) == f(v)
        assert orc.decrypt(ck.glwe_secret, out) == f(v)  # the oracle's own decrypt agrees with the client's


@pytest.mark.parametrize("arity", [2, 3, 4])
def test_lut_gate_truth_tables(toy, arity):
    ck, orc = toy
    rng = np.random.default_rng(arity)
    table = int(rng.integers(1, 2 ** (2 ** arity)))
    combos = list(itertools.product((0, 1), repeat=arity))
    n_in = arity * len(combos)
    wires = np.zeros((n_in + len(combos), ck.dim + 1), dtype=np.uint64)
    wires[:n_in] = ck.encrypt(np.array(combos, dtype=np.uint64).reshape(-1))
    in_idx = np.arange(n_in, dtype=np.int32).reshape(len(combos), arity)
    out_idx = np.arange(n_in, n_in + len(combos), dtype=np.int32)
    orc.eval_lut_level(wires, np.full(len(combos), arity, np.int32), in_idx, np.full(len(combos), table, np.uint64), out_idx)
    got = ck.decrypt_message_and_carry(wires[n_in:])
    for g, bits in enumerate(combos):
        idx = sum(b << (arity - 1 - q) for q, b in enumerate(bits))  # first input = MSB
        assert int(got[g]) == (table >> idx) & 1, bits


def test_one_input_lut_is_copy_or_negation(toy):  # gates.rs:765-770
    ck, orc = toy
    wires = np.zeros((4, ck.dim + 1), dtype=np.uint64)
    wires[:2] = ck.encrypt([1, 1])
    orc.eval_lut_level(wires, [1, 1], [[0], [1]], [0x0, 0x2], [2, 3])
    assert list(ck.decrypt_message_and_carry(wires[2:])) == [1, ck.t - 1]


def test_8_bit_lut_adder_every_wire(toy):  # circuit_test.rs:266-311
    ck, orc = toy
    gates, wire_set, inputs, outputs, dffs, has_luts, _ = verilog_parser.read_verilog_file(
        os.path.join(HERE, "netlists", "8-bit-adder-lut-3-1.v"), False)
    assert has_luts and len(gates) == 16
    c = Circuit(gates, inputs, outputs, dffs)
    c.sort_circuit()
    c.compute_levels()
    a, b, cin = 0x5C, 0xE9, 0
    inp = {f"a[{i}]": PtxtType.Bool((a >> i) & 1) for i in range(8)}
    inp.update({f"b[{i}]": PtxtType.Bool((b >> i) & 1) for i in range(8)})
    inp["cin"] = PtxtType.Bool(cin)
    ptxt = c.evaluate(c.initialize_wire_map(wire_set, inp, "bool"))
    names = list(inputs) + sorted(wire_set)
    row = {w: i for i, w in enumerate(names)}
    wires = np.zeros((len(names), ck.dim + 1), dtype=np.uint64)
    wires[:len(inputs)] = ck.encrypt([int(bool(inp[w])) for w in inputs])
    for level, gs in sorted(c.level_map().items()):
        ar = [len(g.input_wires) for g in gs]
        ii = [[row[w] for w in g.input_wires] for g in gs]
        tb = [sum((int(v) & 1) << i for i, v in enumerate(g.lut_const)) for g in gs]
        orc.eval_lut_level(wires, ar, ii, tb, [row[g.output_wire] for g in gs])
    for w, want in ptxt.items():
        assert int(ck.decrypt(wires[row[w]])) == int(bool(want)), w
