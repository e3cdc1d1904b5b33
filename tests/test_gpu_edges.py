"""Edge cases of the C ABI on the GPU: empty batches, empty and ragged levels, levels of free gates only, a shard
chunk smaller than the world, circuits without a bootstrap - the cases a host that walks arbitrary netlists runs into
(reference src/circuit.rs:531-560 iterates whatever the level map holds, including levels of NOT / BUF gates and
circuits whose outputs are inputs)."""
import numpy as np
import pytest

import helm_amd
import oracle
from helm_amd import Circuit, verilog_parser

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def toy():
    ck = helm_amd.ClientKey.generate("toy", seed=21)
    sk = helm_amd.ServerKey(ck)
    yield ck, sk
    sk.close()


def test_empty_batches_are_no_ops(toy):
    ck, sk = toy
    p = ck.params
    assert sk.pbs_batch(np.zeros((0, p.n + 1), np.uint32), np.zeros((1, p.N), np.uint32)).shape == (0, p.k * p.N + 1)
    assert sk.keyswitch_batch(np.zeros((0, p.k * p.N + 1), np.uint32)).shape == (0, p.n + 1)
    w = sk.wires(4)
    e = np.zeros(0, np.int32)
    w.eval_gate_level(e, e, e, e, e)  # a level without gates
    w.upload(e, np.zeros((0, p.n + 1), np.uint32))
    assert w.download(e).shape == (0, p.n + 1)
    sk.sync()


def test_program_with_empty_free_and_ragged_levels(toy):
    """Levels: [AND, XOR] | [] | [NOT, BUF] (no bootstrap) | [OR x 5] - through the packed program and level by level."""
    ck, sk = toy
    orc = oracle.Oracle(ck.params.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True)
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2, size=4).astype(bool)
    A, X, N, B = oracle.AND, oracle.XOR, oracle.NOT, oracle.BUF
    ops = np.array([A, X, N, B] + [oracle.OR] * 5, np.int32)
    in0 = np.array([0, 2, 4, 5, 4, 5, 6, 7, 0], np.int32)
    in1 = np.array([1, 3, -1, -1, 5, 6, 7, 4, 3], np.int32)
    in2 = np.full(9, -1, np.int32)
    out = np.arange(4, 13, dtype=np.int32)
    off = np.array([0, 2, 2, 4, 9], np.int64)
    w = sk.wires(13)
    w.upload(np.arange(4, dtype=np.int32), ck.encrypt(bits))
    prog = helm_amd.Program(sk, ops, in0, in1, in2, out, off)
    assert [prog.level_pbs(l) for l in range(4)] == [2, 0, 0, 5]
    prog.run(w)
    sk.sync()
    got = w.download()
    # the oracle, level by level on the same input ciphertexts
    exp = np.ascontiguousarray(got.copy())
    exp[4:] = 0
    for l in range(4):
        a, b = int(off[l]), int(off[l + 1])
        if b > a:
            orc.eval_level(exp, ops[a:b], in0[a:b], in1[a:b], in2[a:b], out[a:b])
    assert np.array_equal(got, exp)
    v = ck.decrypt(got)
    a, b, c, d = bits
    want = [a & b, c ^ d]
    want += [not want[0], want[1]]
    w4, w5, w6, w7 = want
    want += [w4 | w5, w5 | w6, w6 | w7, w7 | w4, a | d]
    assert list(v[4:]) == [bool(x) for x in want]
    # level by level and in two halves: the same table
    w2 = sk.wires(13)
    w2.upload(np.arange(4, dtype=np.int32), got[:4])
    prog.run(w2, 0, 2)
    prog.run(w2, 2, 2)  # an empty range
    prog.run(w2, 2, 4)
    sk.sync()
    assert np.array_equal(w2.download(), got)


def test_shard_chunk_smaller_than_the_world(toy):
    """Five bootstraps over four ranks (chunks 2, 2, 1, 0): every rank's stage + the scatter give the single-rank table."""
    import torch
    ck, sk = toy
    p = ck.params
    rng = np.random.default_rng(6)
    bits = rng.integers(0, 2, size=6).astype(bool)
    ops = np.full(5, oracle.NAND, np.int32)
    in0, in1 = np.arange(5, dtype=np.int32), np.arange(1, 6, dtype=np.int32)
    in2, out = np.full(5, -1, np.int32), np.arange(6, 11, dtype=np.int32)
    prog = helm_amd.Program(sk, ops, in0, in1, in2, out, np.array([0, 5], np.int64))
    ref = sk.wires(11)
    ref.upload(np.arange(6, dtype=np.int32), ck.encrypt(bits))
    fresh = ref.download()
    prog.run(ref)
    sk.sync()
    world = 4
    rows = prog.chunk_rows(0, world)
    assert rows == 2
    gathered = torch.zeros((rows * world, p.n + 1), dtype=torch.int32, device="cuda")
    tables = []
    for rank in range(world):
        w = sk.wires(11)
        w.upload(np.arange(11, dtype=np.int32), fresh)
        prog.shard_prepare(rank, world)
        stage = torch.zeros((rows, p.n + 1), dtype=torch.int32, device="cuda")
        prog.run_level_shard(w, 0, rank, world, stage.data_ptr())
        sk.sync()
        gathered[rank * rows:(rank + 1) * rows] = stage
        tables.append(w)
    torch.cuda.synchronize()
    for w in tables:
        prog.scatter_level(w, 0, world, gathered.data_ptr())
        sk.sync()
        assert np.array_equal(w.download(), ref.download())
    assert list(ck.decrypt(ref.download())[6:]) == [not (bits[i] & bits[i + 1]) for i in range(5)]


def test_circuit_without_a_bootstrap(toy):
    """Outputs that are inputs, inverted inputs and constants: no bootstrap, the evaluator still returns every wire."""
    from helm_amd import GateCircuit, PtxtType
    ck, sk = toy
    text = "module m(a, b, y0, y1);\n input a, b;\n output y0, y1;\n wire n1;\n not g0(a, n1);\n buf g1(n1, y0);\n buf g2(b, y1);\nendmodule\n"
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(text, False)
    c = Circuit(gates, inputs, outputs, dffs)
    c.sort_circuit()
    c.compute_levels()
    gc = GateCircuit(ck, sk, c)
    enc = gc.encrypt_inputs(wire_set, {"a": PtxtType.Bool(True), "b": PtxtType.Bool(True)})
    res = gc.decrypt_outputs(gc.evaluate_encrypted(enc, 1, "bool"), False)
    assert {k: bool(v.value) for k, v in res.items()} == {"y0": False, "y1": True}


def test_levels_with_intra_level_hazards_are_refused(toy):
    """The gates of a level run concurrently (reference src/circuit.rs:531); the reference gets its guarantee that none reads
    what another writes from compute_levels (circuit.rs:174-239).  A host with a wrong level map gets HELM_ERR_INVALID - not
    silently non-deterministic ciphertexts - from helm_hip_program_create and helm_hip_eval_gate_level; a gate that updates
    its OWN row in place (the READY latch, circuit.rs:482-504) is legal."""
    ck, sk = toy
    A, X, N = oracle.AND, oracle.XOR, oracle.NOT
    m1 = np.full(2, -1, np.int32)

    def create(ops, i0, i1, out, off):
        return helm_amd.Program(sk, np.array(ops, np.int32), np.array(i0, np.int32), np.array(i1, np.int32),
                                np.full(len(ops), -1, np.int32), np.array(out, np.int32), np.array(off, np.int64))
    # read-after-write: gate 1 reads wire 2, which gate 0 of the same level writes
    with pytest.raises(helm_amd.HelmError, match="read-after-write"):
        create([A, X], [0, 2], [1, 1], [2, 3], [0, 2])
    # the same two gates in two levels are fine
    create([A, X], [0, 2], [1, 1], [2, 3], [0, 1, 2]).destroy()
    # write-after-write: both gates write wire 2
    with pytest.raises(helm_amd.HelmError, match="write-after-write"):
        create([A, X], [0, 0], [1, 1], [2, 2], [0, 2])
    # a free gate reading a bootstrapped gate's output row inside the level
    with pytest.raises(helm_amd.HelmError, match="read-after-write"):
        create([A, N], [0, 2], [1, -1], [2, 3], [0, 2])
    # the hazard is per level: level 1 may rewrite a row level 0 wrote, and read it
    create([A, X], [0, 2], [1, 0], [2, 2], [0, 1, 2]).destroy()
    w = sk.wires(6)
    w.upload(np.arange(2, dtype=np.int32), ck.encrypt(np.array([True, False])))
    for dense in (True, False):  # both forms of the per-level check (full row map / compacted rows of a narrow level)
        ww = w if dense else sk.wires(4096)
        with pytest.raises(helm_amd.HelmError, match="read-after-write"):
            ww.eval_gate_level([A, X], [0, 2], [1, 1], m1, [2, 3])
        with pytest.raises(helm_amd.HelmError, match="write-after-write"):
            ww.eval_gate_level([A, X], [0, 0], [1, 1], m1, [2, 2])
    # in place: out = AND(in0, out) reads and writes its own row - every kernel reads before it writes
    w.set_trivial([2], [True])
    w.eval_gate_level([A], [0], [2], [-1], [2])
    sk.sync()
    assert bool(ck.decrypt(w.download([2]))[0]) is True
    w.eval_gate_level([A], [1], [2], [-1], [2])
    sk.sync()
    assert bool(ck.decrypt(w.download([2]))[0]) is False
