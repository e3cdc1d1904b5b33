"""Encrypted netlists on the GPU through the host front end, mirroring the encrypted
half of reference tests/circuit_test.rs, plus bit-exactness against the oracle,
the committed golden vectors, sharded evaluation and the AES-128 known answer."""
import os

import numpy as np
import pytest

import helm_amd
import oracle
from helm_amd import Circuit, EncWireMap, EvalCircuit, GateCircuit, PtxtType, verilog_parser
from helm_amd._host import Panic
from helm_amd.distributed import level_arrays
from helm_amd.netlists import aes128, aes128_reference_encrypt

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NET = os.path.join(HERE, "netlists")


@pytest.fixture(scope="module")
def keys():
    client_key, server_key = helm_amd.gen_keys()  # tfhe::boolean::gen_keys(), helm.rs:241
    yield client_key, server_key
    server_key.close()


def _circuit(path_or_text, is_text=False):
    rd = verilog_parser.read_verilog_text if is_text else verilog_parser.read_verilog_file
    gates_set, wire_set, input_wires, output_wires, dffs, _, _ = rd(path_or_text, False)
    c = Circuit(gates_set, input_wires, output_wires, dffs)
    c.sort_circuit()
    c.compute_levels()
    return c, wire_set, input_wires, output_wires


def test_encrypted_two_bit_adder(keys):  # circuit_test.rs:47-94
    datatype = "bool"
    client_key, server_key = keys
    circuit, wire_set, input_wires, _ = _circuit(f"{NET}/2-bit-adder.v")
    ptxt_wire_map = {w: PtxtType.Bool(True) for w in wire_set}
    ptxt_wire_map.update({w: PtxtType.Bool(True) for w in input_wires})
    ptxt_wire_map = circuit.evaluate(ptxt_wire_map)
    enc_wire_map = EncWireMap(server_key)
    for wire in wire_set:
        enc_wire_map.insert(wire, client_key.encrypt(False))
    for input_wire in input_wires:
        enc_wire_map.insert(input_wire, client_key.encrypt(True))
    gc = GateCircuit(client_key, server_key, circuit)
    enc_wire_map = EvalCircuit.evaluate_encrypted(gc, enc_wire_map, 1, datatype)
    dec_wire_map = {w: client_key.decrypt(enc_wire_map[w]) for w in sorted(enc_wire_map.keys())}
    for key in ptxt_wire_map:
        assert ptxt_wire_map[key] == PtxtType.Bool(dec_wire_map[key])
    assert "Evaluated gates in level [1/" in gc.log()


def test_two_bit_adder_csv_inputs_and_decrypt_outputs(keys):  # K-2
    client_key, server_key = keys
    circuit, wire_set, input_wires, output_wires = _circuit(f"{NET}/2-bit-adder.v")
    input_map = verilog_parser.read_input_wires(os.path.join(HERE, "golden", "2-bit-adder.inputs.csv"), "bool")
    gc = GateCircuit(client_key, server_key, circuit)
    enc = gc.encrypt_inputs(wire_set, input_map)
    assert len(enc) == 15
    enc = gc.evaluate_encrypted(enc, 1, "bool")
    out = gc.decrypt_outputs(enc, False)
    assert out == {"sum[0]": PtxtType.Bool(True), "sum[1]": PtxtType.Bool(True), "cout": PtxtType.Bool(False)}
    assert " cout: false" in gc.log()
    assert gc.pbs_per_cycle() == 10
    # no inputs given -> every input false (lib.rs:166-178 "dummy", circuit.rs:462)
    enc = gc.evaluate_encrypted(gc.encrypt_inputs(wire_set, {"dummy": PtxtType.Bool(False)}), 1, "bool")
    assert all(not v.value for v in gc.decrypt_outputs(enc).values())
    with pytest.raises(Panic, match='Input wire "a\\[1\\]" not in input wires!'):
        gc.encrypt_inputs(wire_set, {"a[0]": PtxtType.Bool(True)})


def test_two_bit_adder_every_wire_bit_exact_vs_oracle(keys):
    client_key, server_key = keys
    circuit, wire_set, input_wires, _ = _circuit(f"{NET}/2-bit-adder.v")
    orc = oracle.Oracle(client_key.params.as_tuple7(), client_key.bsk, client_key.ksk)
    names = sorted(wire_set) + list(input_wires)
    index = {w: i for i, w in enumerate(names)}
    ops, i0, i1, i2, out, off = level_arrays(circuit, index)
    wires = np.zeros((len(names), client_key.params.n + 1), dtype=np.uint32)
    bits = {"a[0]": 1, "a[1]": 1, "b[0]": 1, "b[1]": 0, "cin": 1}
    for w, v in bits.items():
        wires[index[w]] = client_key.encrypt(bool(v))
    dev = server_key.wires(len(names))
    dev.upload(np.arange(len(names)), wires)
    prog = helm_amd.Program(server_key, ops, i0, i1, i2, out, off)
    prog.run(dev)
    for l in range(len(off) - 1):
        s = slice(off[l], off[l + 1])
        orc.eval_level(wires, ops[s], i0[s], i1[s], i2[s], out[s])
    got = dev.download()
    for w in names:
        assert np.array_equal(got[index[w]], wires[index[w]]), w
    s = sum(int(client_key.decrypt(got[index[w]])) << i for i, w in enumerate(["sum[0]", "sum[1]", "cout"]))
    assert s == 3 + 1 + 1


def test_c880_class_every_wire(keys):  # config 2 (stand-in netlist)
    client_key, server_key = keys
    circuit, wire_set, input_wires, output_wires = _circuit(f"{NET}/alu-c880-class.v")
    rng = np.random.default_rng(0x48454C4D)
    inputs = {w: PtxtType.Bool(int(rng.integers(0, 2))) for w in input_wires}
    ptxt = {w: PtxtType.None_() for w in wire_set}
    ptxt.update(inputs)
    ptxt = circuit.evaluate(ptxt)
    gc = GateCircuit(client_key, server_key, circuit)
    enc = gc.evaluate_encrypted(gc.encrypt_inputs(wire_set, inputs), 1, "bool")
    for w in sorted(ptxt):
        assert client_key.decrypt(enc[w]) == bool(ptxt[w].value), w
    assert gc.pbs_per_cycle() > 330
    # ... and bit for bit: the oracle's SIMD route from the evaluator's own input ciphertexts, level by level, every wire
    orc = oracle.Oracle(client_key.params.as_tuple7(), client_key.bsk, client_key.ksk, use_ntt=False, use_fp=True)
    names = list(input_wires) + sorted(wire_set)
    index = {w: i for i, w in enumerate(names)}
    ops, i0, i1, i2, out, off = level_arrays(circuit, index)
    host = np.zeros((len(names), client_key.params.n + 1), dtype=np.uint32)
    for w in input_wires:
        host[index[w]] = enc[w]
    for l in range(len(off) - 1):
        s = slice(off[l], off[l + 1])
        orc.eval_level_fp(host, ops[s], i0[s], i1[s], i2[s], out[s])
    for w in sorted(wire_set):
        assert np.array_equal(enc[w], host[index[w]]), f"{w}: the evaluator's ciphertext differs from the oracle's"


def test_golden_vectors_on_gpu():
    g = np.load(os.path.join(HERE, "golden", "gates_toy.npz"))
    n, k, N, l, logB, ksl, kslogB = [int(x) for x in g["params"]]
    p = helm_amd.Params(32, n, k, N, l, logB, ksl, kslogB, 0, 1)
    sk = helm_amd.ServerKey(params=p, bsk=g["bsk"], ksk=g["ksk"])
    w = sk.wires(2 + len(g["ops"]))
    w.upload([0, 1], g["inputs"])
    w.eval_gate_level(g["ops"], g["in0"], g["in1"], g["in2"], np.arange(2, 2 + len(g["ops"]), dtype=np.int32))
    assert np.array_equal(w.download()[2:], g["expected"])
    sk.close()


def test_sharded_levels_equal_unsharded(keys):
    """helm_hip_program_run_level_shard + scatter_level for every rank of world 3,
    executed back to back on one GPU, must reproduce program_run bit for bit."""
    import ctypes
    client_key, server_key = keys
    circuit, wire_set, input_wires, _ = _circuit(f"{NET}/8-bit-adder.v")
    names = sorted(wire_set) + list(input_wires)
    index = {w: i for i, w in enumerate(names)}
    ops, i0, i1, i2, out, off = level_arrays(circuit, index)
    rng = np.random.default_rng(3)
    wires = np.zeros((len(names), client_key.params.n + 1), dtype=np.uint32)
    for w in input_wires:
        wires[index[w]] = client_key.encrypt(bool(rng.integers(0, 2)))
    prog = helm_amd.Program(server_key, ops, i0, i1, i2, out, off)
    ref = server_key.wires(len(names))
    ref.upload(np.arange(len(names)), wires)
    prog.run(ref)
    world = 3
    shard = server_key.wires(len(names))
    shard.upload(np.arange(len(names)), wires)
    rows_max = max(prog.chunk_rows(l, world) for l in range(prog.n_levels))
    gathered = server_key.wires(rows_max * world)  # device scratch: reuse a wire table as the gather buffer
    base = gathered.device_ptr()
    row_bytes = (client_key.params.n + 1) * 4
    for l in range(prog.n_levels):
        rows = prog.chunk_rows(l, world)
        for r in range(world):
            prog.run_level_shard(shard, l, r, world, base + r * rows * row_bytes)
        prog.scatter_level(shard, l, world, base)
    server_key.sync()
    assert np.array_equal(shard.download(), ref.download())


def test_sequential_circuit_with_ready_latch(keys):  # helm.rs:249-274, circuit.rs:482-504
    client_key, server_key = keys
    text = """input en;
output q0, READY;
dff g0(d0, q0);
dff g1(d1, q1);
xor g2(q0, en, d0);
and g3(q0, en, c0);
xor g4(q1, c0, d1);
buf g5(q1, READY);
"""
    circuit, wire_set, input_wires, output_wires = _circuit(text, is_text=True)
    gc = GateCircuit(client_key, server_key, circuit)
    enc = gc.encrypt_inputs(wire_set, {"en": PtxtType.Bool(True), "q0": PtxtType.Bool(False), "q1": PtxtType.Bool(False)})
    ready_map = gc.init_ready()
    state = []
    for cycle in range(3):
        enc = gc.evaluate_encrypted(enc, 1, "bool")
        assert enc.contains_key("READY")
        gc.evaluate_ready(enc, ready_map)
        state.append((client_key.decrypt(enc["q0"]), client_key.decrypt(enc["q1"])))
    # 2-bit counter: q after 1,2,3 cycles
    assert state == [(True, False), (False, True), (True, True)]
    out = gc.decrypt_outputs(ready_map, True)
    # READY = q1 seen before the latch of each cycle: cycle 3 has READY = 1 -> outputs latched then
    assert out["READY"] == PtxtType.Bool(True)


def test_boolean_mode_rejects_lut_and_arith_gates(keys):  # gates.rs:257-264
    client_key, server_key = keys
    circuit, wire_set, input_wires, _ = _circuit("input a, b, c;\nlut g(0x96, a, b, c, y);\n", is_text=True)
    gc = GateCircuit(client_key, server_key, circuit)
    enc = gc.encrypt_inputs(wire_set, {})
    with pytest.raises(Panic, match="can't be mixed with Boolean"):
        gc.evaluate_encrypted(enc, 1, "bool")
    w = server_key.wires(4)
    with pytest.raises(helm_amd.HelmError, match="out of range"):
        w.eval_gate_level([oracle.AND], [0], [9], [-1], [1])
    with pytest.raises(helm_amd.HelmError, match="missing operand"):
        w.eval_gate_level([oracle.AND], [0], [-1], [-1], [1])


def test_full_size_properties(keys):
    """Size-independent properties on a 2,048-gate level of the full parameter set:
    determinism (same inputs -> identical ciphertext bits), NOT is an involution on bits,
    x NAND x == NOT x after decryption, and XOR(x, x) decrypts to false."""
    client_key, server_key = keys
    B = 1024
    rng = np.random.default_rng(11)
    bits = rng.integers(0, 2, size=B).astype(bool)
    w = server_key.wires(6 * B)
    w.upload(np.arange(B), client_key.encrypt(bits))
    ar = np.arange(B, dtype=np.int32)
    neg1 = np.full(B, -1, np.int32)
    w.eval_gate_level(np.full(B, oracle.NOT), ar, neg1, neg1, ar + B)
    w.eval_gate_level(np.full(B, oracle.NOT), ar + B, neg1, neg1, ar + 2 * B)
    w.eval_gate_level(np.concatenate([np.full(B, oracle.NAND), np.full(B, oracle.XOR)]), np.concatenate([ar, ar]),
                      np.concatenate([ar, ar]), np.concatenate([neg1, neg1]), np.concatenate([ar + 3 * B, ar + 4 * B]))
    w.eval_gate_level(np.full(B, oracle.NAND), ar, ar, neg1, ar + 5 * B)
    t = w.download()
    assert np.array_equal(t[2 * B:3 * B], t[:B])                 # NOT(NOT(x)) == x, bit for bit
    assert np.array_equal(t[5 * B:6 * B], t[3 * B:4 * B])        # deterministic
    assert np.array_equal(client_key.decrypt(t[3 * B:4 * B]), ~bits)
    assert not client_key.decrypt(t[4 * B:5 * B]).any()


def test_aes128_fips197_encrypted(keys):  # K-8 / config 4 on one GPU, one block
    client_key, server_key = keys
    circuit, wire_set, input_wires, output_wires = _circuit(aes128(), is_text=True)
    key, pt = bytes(range(16)), bytes.fromhex("00112233445566778899aabbccddeeff")
    kv, pv = int.from_bytes(key, "big"), int.from_bytes(pt, "big")
    inputs = {f"key[{i}]": PtxtType.Bool((kv >> i) & 1) for i in range(128)}
    inputs.update({f"pt[{i}]": PtxtType.Bool((pv >> i) & 1) for i in range(128)})
    gc = GateCircuit(client_key, server_key, circuit)
    enc = gc.evaluate_encrypted(gc.encrypt_inputs(wire_set, inputs), 1, "bool")
    out = gc.decrypt_outputs(enc, False)
    ct = sum(out[f"ct[{i}]"].value << i for i in range(128)).to_bytes(16, "big")
    assert ct == bytes.fromhex("69c4e0d86a7b0430d8cdb78070b4c55a") == aes128_reference_encrypt(key, pt)


def test_packed_launches_equal_level_schedule_bit_exact():
    """Launch packing (helm_host_pack_levels): 6 AES blocks under a toy set, quantum forced small so that the
    packing really re-times the batch - every wire of every block must hold the SAME ciphertext bits as under the
    level schedule of circuit.rs:524-543, and the outputs decrypt to a software AES."""
    from helm_amd.distributed import pack_levels
    ck = helm_amd.ClientKey.generate("toy_k2", seed=21)
    sk = helm_amd.ServerKey(ck)
    circuit, wire_set, input_wires, output_wires = _circuit(aes128(), is_text=True)
    names = list(input_wires) + sorted(wire_set)
    index = {w: i for i, w in enumerate(names)}
    ops, i0, i1, i2, out, off = level_arrays(circuit, index)
    blocks, nw, nl = 6, len(names), len(off) - 1
    t = lambda a: np.concatenate([np.concatenate([np.where(a[off[l]:off[l + 1]] >= 0, a[off[l]:off[l + 1]] + b * nw, -1)
                                                  for b in range(blocks)]) for l in range(nl)]).astype(np.int32)
    opsT = np.concatenate([np.tile(ops[off[l]:off[l + 1]], blocks) for l in range(nl)]).astype(np.int32)
    arrs = (opsT, t(i0), t(i1), t(i2), t(out), (off * blocks).astype(np.int64))
    packed = pack_levels(*arrs, 128)
    assert packed[6] and len(packed[5]) != len(arrs[5])
    rng = np.random.default_rng(4)
    keys_pt = [(bytes(rng.integers(0, 256, 16, dtype=np.uint8)), bytes(rng.integers(0, 256, 16, dtype=np.uint8)))
               for _ in range(blocks)]
    rows, bits = [], []
    for b, (key, pt) in enumerate(keys_pt):
        kv, pv = int.from_bytes(key, "big"), int.from_bytes(pt, "big")
        for i in range(128):
            rows += [b * nw + index[f"key[{i}]"], b * nw + index[f"pt[{i}]"]]
            bits += [(kv >> i) & 1, (pv >> i) & 1]
    cts = ck.encrypt(np.array(bits, dtype=bool))
    tables = []
    for a in (arrs, packed[:6]):
        prog = helm_amd.Program(sk, *a)
        w = sk.wires(nw * blocks)
        w.upload(np.array(rows, np.int32), cts)
        prog.run(w)
        tables.append(w.download())
        prog.destroy()
        w.free()
    assert np.array_equal(tables[0], tables[1])
    for b, (key, pt) in enumerate(keys_pt):
        dec = ck.decrypt(tables[1][[b * nw + index[f"ct[{i}]"] for i in range(128)]])
        assert sum(int(dec[i]) << i for i in range(128)).to_bytes(16, "big") == aes128_reference_encrypt(key, pt)
    sk.close()


def test_gate_circuit_packs_wide_levels():
    """GateCircuit::evaluate_encrypted on a netlist whose levels exceed one lockstep round: the host front end
    packs the launches itself (gate_circuit.cpp) and every wire still decrypts to the plaintext evaluation."""
    import torch
    cu = torch.cuda.get_device_properties(0).multi_processor_count
    width = 4 * cu + 40  # one round and a bit per level
    lines = ["input [%d:0] a;" % (width - 1), "input [%d:0] b;" % (width - 1), "output [%d:0] y;" % (width - 1)]
    for i in range(width):
        lines.append(f"xor g0_{i}(a[{i}], b[{i}], t{i});")
        lines.append(f"and g1_{i}(t{i}, a[{(i + 1) % width}], u{i});")
        lines.append(f"or g2_{i}(u{i}, b[{(i + 3) % width}], y[{i}]);")
    client_key = helm_amd.ClientKey.generate("toy_k2", seed=9)
    server_key = helm_amd.ServerKey(client_key)
    circuit, wire_set, input_wires, output_wires = _circuit("\n".join(lines) + "\n", is_text=True)
    rng = np.random.default_rng(6)
    inputs = {w: PtxtType.Bool(bool(rng.integers(0, 2))) for w in input_wires}
    ptxt = circuit.evaluate(circuit.initialize_wire_map(wire_set, inputs, "bool"))
    gc = GateCircuit(client_key, server_key, circuit)
    enc = gc.evaluate_encrypted(gc.encrypt_inputs(wire_set, inputs), 1, "bool")
    assert "packed launches" in gc.log()
    assert gc.pbs_per_cycle() == 3 * width
    for wire, want in ptxt.items():
        assert client_key.decrypt(enc[wire]) == bool(want), wire
    server_key.close()


@pytest.mark.parametrize("order", ["forward", "backward"])
def test_flip_flops_fed_by_flip_flops_follow_the_plaintext_evaluator(keys, order):
    """Every DFF sits in ONE last level (circuit.rs:174-239), so a flip-flop fed by another reads a wire written in its own
    level.  The reference evaluates such a level as a race (par_iter, circuit.rs:531 and :348-381); this repository's
    plaintext evaluator gives a level snapshot semantics (every gate reads the values from before the level: flip-flops on
    one clock edge) and the encrypted evaluation follows it by copying the rows in question to scratch rows first (the
    engine refuses a level with a read-after-write inside): on every wire, every cycle, encrypted == plaintext - a shift
    register in either gate order, and the swap of two registers."""
    client_key, server_key = keys
    chain = ["dff ga(d, q0);", "dff gb(q0, q1);", "dff gc(q1, q2);", "dff gd(s1, s0);", "dff ge(s0, s1);"]
    if order == "backward":
        chain = [c.replace("ga", "gz").replace("gb", "gy").replace("gc", "gx") for c in chain]  # sort_circuit orders by name
    text = "input x;\noutput q2, s0, s1;\n" + "\n".join(chain) + "\nxor g9(x, q2, d);\n"
    circuit, wire_set, input_wires, output_wires = _circuit(text, is_text=True)
    names = [g.gate_name for g in circuit.level_map()[max(circuit.level_map())]]
    assert len(names) == 5
    gc = GateCircuit(client_key, server_key, circuit)
    state = {"x": True, "q0": False, "q1": True, "q2": False, "s0": True, "s1": False}
    enc = gc.encrypt_inputs(wire_set, {k: PtxtType.Bool(v) for k, v in state.items()})
    # DFF outputs start at encrypt(false) (circuit.rs:474-476): put the test's initial state in
    for w in ("q1", "s0"):
        enc.insert(w, client_key.encrypt(True))
    ptxt = {w: PtxtType.None_() for w in wire_set}
    ptxt.update({k: PtxtType.Bool(v) for k, v in state.items()})
    for cycle in range(4):
        ptxt = circuit.evaluate(ptxt)
        enc = gc.evaluate_encrypted(enc, cycle + 1, "bool")
        for w in sorted(ptxt):
            assert bool(client_key.decrypt(enc[w])) == bool(ptxt[w].value), (order, cycle, w, names)
