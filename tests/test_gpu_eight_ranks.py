"""Eight ranks (and two, three) through the library's communicator, ONE process, one thread and one engine context per rank.

The box allows six GPU processes, the scaling run has eight ranks: helm_amd.comm.Comm.in_process_group gives every rank
thread a communicator whose all-gather is device-to-device copies between the ranks' gather buffers, so the WHOLE native
sharded pass - helm_hip_program_run_sharded_comm: the cut of every launch by bootstrap weight (shard_rule.h), each rank's slot
offset in the gather buffer, padded chunks, levels with fewer gates than ranks, replicated launches, the scatter table, and
the overlapped form (exchange stream, ring of gather buffers, dependency events) - runs at world size 8 exactly as bench.py's
eight processes run it; only ncclAllGather itself is swapped (it runs at world size 1 in tests/test_gpu_rccl_world1.py).
Every rank must end with the wire table of helm_hip_program_run on one context, bit for bit (the sharded unit is the level of
reference src/circuit.rs:531)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

AND, DFF, MUX, NAND, NOR, NOT, OR, XNOR, XOR, BUF, ONE, ZERO = 0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12


def _random_program(seed, n_inputs=24, n_levels=14):
    """A levelised random netlist with every gate type: levels of 1 .. 60 gates (several narrower than eight ranks), free
    gates clustered at the front of some levels (what a cut by gate count would get wrong), MUX gates (two bootstraps)."""
    rng = np.random.default_rng(seed)
    ops, i0, i1, i2, out, off = [], [], [], [], [], [0]
    n_rows = n_inputs
    for l in range(n_levels):
        cnt = int(rng.choice([1, 2, 3, 5, 7, 9, 20, 33, 60]))
        kinds = rng.choice([AND, MUX, NAND, NOR, NOT, OR, XNOR, XOR, BUF, ONE, ZERO, DFF], size=cnt,
                           p=[.12, .14, .1, .08, .12, .08, .08, .12, .06, .03, .03, .04])
        if l % 3 == 1:
            kinds = np.sort(kinds)[::-1]  # constants, BUF, ... first, bootstrapping gates last
        for k in kinds:
            a, b, c = (int(x) for x in rng.integers(0, n_rows, size=3))
            ops.append(int(k))
            i0.append(-1 if k in (ONE, ZERO) else a)
            i1.append(b if k in (AND, MUX, NAND, NOR, OR, XNOR, XOR) else -1)
            i2.append(c if k == MUX else -1)
        out.extend(range(n_rows, n_rows + cnt))
        n_rows += cnt
        off.append(len(ops))
    as32 = lambda a: np.array(a, dtype=np.int32)
    return (as32(ops), as32(i0), as32(i1), as32(i2), as32(out), np.array(off, dtype=np.int64)), n_rows, n_inputs


def _plain(arrays, bits, n_rows):
    ops, i0, i1, i2, out, off = arrays
    v = np.zeros(n_rows, dtype=bool)
    v[:len(bits)] = bits
    f = {AND: lambda a, b: a & b, NAND: lambda a, b: ~(a & b), OR: lambda a, b: a | b, NOR: lambda a, b: ~(a | b),
         XOR: lambda a, b: a ^ b, XNOR: lambda a, b: ~(a ^ b)}
    for g in range(len(ops)):
        k = ops[g]
        if k in f:
            v[out[g]] = f[k](v[i0[g]], v[i1[g]])
        elif k == MUX:
            v[out[g]] = v[i0[g]] if v[i2[g]] else v[i1[g]]
        elif k == NOT:
            v[out[g]] = ~v[i0[g]]
        elif k in (BUF, DFF):
            v[out[g]] = v[i0[g]]
        else:
            v[out[g]] = k == ONE
    return v


def _run_world(world, arrays, n_rows, in_rows, enc, ck, replicate_below, overlap, passes=1):
    """-> ([wire table of every rank], [collectives issued by every rank], sharded launches)"""
    import helm_amd
    from helm_amd.comm import Comm
    comms = Comm.in_process_group([0] * world)
    tables, stats, errors, sharded = [None] * world, [None] * world, [], [None] * world
    # contexts, programs and tables are made one after the other on this thread; the rank threads run the passes
    ranks = []
    for r in range(world):
        sk = helm_amd.ServerKey(ck, device=0)
        prog = helm_amd.Program(sk, *arrays)
        w = sk.wires(n_rows)
        w.upload(in_rows, enc)
        prog.shard_prepare(r, world)
        sk.sync()
        ranks.append((sk, prog, w))

    def rank_main(r):
        try:
            sk, prog, w = ranks[r]
            assert comms[r].info() == {"rank": r, "world_size": world, "device": 0, "rccl_version": 0}
            for _ in range(passes):
                prog.run_sharded_comm(w, comms[r], replicate_below, overlap)
            sk.sync()
        except BaseException as e:  # noqa: BLE001 - reported by the test; the group's barrier is broken so nobody waits for ever
            errors.append((r, repr(e)))
            comms[r].abort_group()
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
        assert not t.is_alive(), "a rank thread is stuck"
    assert not errors, errors
    for r, (sk, prog, w) in enumerate(ranks):
        tables[r] = w.download()
        stats[r] = comms[r].stats()["collectives"]
        sharded[r] = sum(1 for l in range(prog.n_levels) if prog.level_pbs(l) > replicate_below)
        # the cut the engine made: within one gate of an even share of the bootstraps, the same on every rank
        for l in range(prog.n_levels):
            b = prog.chunk_bounds(l, world)
            assert b[0] == 0 and b[-1] == arrays[5][l + 1] - arrays[5][l] and prog.chunk_rows(l, world) == int(np.max(np.diff(b)))
        prog.destroy()
        sk.close()
    for c in comms:
        c.destroy()
    return tables, stats, sharded[0]


@pytest.fixture(scope="module")
def setup():
    import helm_amd
    ck = helm_amd.ClientKey.generate("toy_k2", seed=11)
    arrays, n_rows, n_in = _random_program(7)
    bits = np.random.default_rng(2).integers(0, 2, size=n_in).astype(bool)
    in_rows = np.arange(n_in, dtype=np.int32)
    enc = ck.encrypt(bits)
    sk = helm_amd.ServerKey(ck, device=0)
    prog = helm_amd.Program(sk, *arrays)
    ref = sk.wires(n_rows)
    ref.upload(in_rows, enc)
    prog.run(ref)
    sk.sync()
    want = ref.download()
    # the single-context table is right to begin with: every wire decrypts to the plaintext evaluation
    assert np.array_equal(ck.decrypt(want), _plain(arrays, bits, n_rows))
    narrow = sum(1 for l in range(prog.n_levels) if 0 < prog.level_pbs(l) and arrays[5][l + 1] - arrays[5][l] < 8)
    assert narrow >= 2, "the test program must hold levels with fewer gates than ranks"
    prog.destroy()
    sk.close()
    return ck, arrays, n_rows, in_rows, enc, want


@pytest.mark.parametrize("world,replicate_below,overlap", [(8, 0, False), (8, 0, True), (8, 4, False), (3, 0, True), (2, 0, False)])
def test_eight_ranks_through_the_library_communicator(setup, world, replicate_below, overlap):
    ck, arrays, n_rows, in_rows, enc, want = setup
    passes = 2 if overlap else 1  # the ring of gather buffers and the events are reused by the second pass
    tables, stats, n_sharded = _run_world(world, arrays, n_rows, in_rows, enc, ck, replicate_below, overlap, passes)
    for r, t in enumerate(tables):
        assert np.array_equal(t, want), f"world {world}: rank {r}'s wire table differs from the single-context evaluation"
    assert n_sharded > 0 and all(s == passes * n_sharded for s in stats)


def test_state_writing_program_falls_back_to_the_in_order_exchange(setup):
    """A row written twice per pass (a flip-flop rewritten by a later level): the overlapped schedule does not apply
    (helm_hip_program_overlap_applies == 0) and overlap = 1 must quietly take the in-order exchange - same table."""
    import helm_amd
    ck, arrays, n_rows, in_rows, enc, _ = setup
    ops, i0, i1, i2, out, off = [a.copy() for a in arrays]
    # one more level: XOR of two early rows written INTO a row an earlier level already wrote
    ops = np.append(ops, XOR).astype(np.int32)
    i0, i1, i2 = np.append(i0, 0).astype(np.int32), np.append(i1, 1).astype(np.int32), np.append(i2, -1).astype(np.int32)
    out = np.append(out, out[0]).astype(np.int32)
    off = np.append(off, len(ops)).astype(np.int64)
    arr2 = (ops, i0, i1, i2, out, off)
    sk = helm_amd.ServerKey(ck, device=0)
    prog = helm_amd.Program(sk, *arr2)
    assert not prog.overlap_applies()
    ref = sk.wires(n_rows)
    ref.upload(in_rows, enc)
    prog.run(ref)
    sk.sync()
    want = ref.download()
    prog.destroy()
    sk.close()
    tables, _, _ = _run_world(3, arr2, n_rows, in_rows, enc, ck, 0, True)
    for r, t in enumerate(tables):
        assert np.array_equal(t, want), f"rank {r}"


@pytest.mark.parametrize("world,capacity_rows", [(8, 4), (3, 16)])
def test_lut_level_sharded_over_eight_rank_threads(world, capacity_rows):
    """The 64-bit-torus engine's sharded bootstrap batch (helm_si_set_exchange_comm: LUT mode, the radix operators and WoP
    gates all go through it) at world size 8: 41 three-input LUTs in chunks of ceil(41 / 8) rows per rank, a gather buffer of
    4 rows per rank (several exchange rounds per batch, the last one padded), every rank's table == the unsharded level."""
    import helm_amd
    from helm_amd.comm import Comm
    ck = helm_amd.SiClientKey.generate("si_toy_1024", seed=1)
    B = 41
    bits = np.random.default_rng(4).integers(0, 2, size=3 * B).astype(np.uint64)
    enc = ck.encrypt(bits)
    in_idx = np.arange(3 * B, dtype=np.int32).reshape(3, B).T.copy()
    ar, tb, out = np.full(B, 3, np.int32), np.full(B, 0x96, np.uint64), np.arange(3 * B, 4 * B, dtype=np.int32)
    sk0 = helm_amd.SiServerKey(ck, device=0)
    w0 = sk0.wires(4 * B)
    w0.upload(np.arange(3 * B), enc)
    w0.eval_lut_level(ar, in_idx, tb, out)
    sk0.sync()
    want = w0.download()
    assert np.array_equal(ck.decrypt(want[3 * B:]), bits[:B] ^ bits[B:2 * B] ^ bits[2 * B:])
    sk0.close()
    comms = Comm.in_process_group([0] * world)
    ranks, errors = [], []
    for r in range(world):
        sk = helm_amd.SiServerKey(ck, device=0)
        w = sk.wires(4 * B)
        w.upload(np.arange(3 * B), enc)
        sk.set_exchange_comm(comms[r], min_batch=1, capacity_rows=capacity_rows)
        sk.sync()
        ranks.append((sk, w))

    def rank_main(r):
        try:
            sk, w = ranks[r]
            w.eval_lut_level(ar, in_idx, tb, out)
            sk.sync()
        except BaseException as e:  # noqa: BLE001
            errors.append((r, repr(e)))
            comms[r].abort_group()
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
        assert not t.is_alive(), "a rank thread is stuck"
    assert not errors, errors
    for r, (sk, w) in enumerate(ranks):
        assert np.array_equal(w.download(), want), f"world {world}: rank {r}'s table differs from the unsharded level"
        batches, rows = sk.exchange_stats()
        assert batches == -(-B // (capacity_rows * world)) and B <= rows < B + world * batches, (batches, rows)
        sk.set_exchange_comm(None)
        sk.close()
    for c in comms:
        c.destroy()


@pytest.mark.parametrize("world,overlap", [(3, True), (8, False)])
def test_gate_circuit_shard_over_rank_threads(world, overlap):
    """The evaluator API on top of it (helm_host_gate_circuit_shard_over / _set_exchange_overlap; reference GateCircuit::
    evaluate_encrypted, src/circuit.rs:506-549): every rank thread owns a GateCircuit on its own context, shards it over the
    group's communicator and evaluates the c880-class netlist; every wire of every rank == the one-GPU evaluation."""
    import os
    import helm_amd
    from helm_amd import Circuit, GateCircuit, PtxtType, verilog_parser
    from helm_amd.comm import Comm
    here = os.path.dirname(os.path.abspath(__file__))
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_file(os.path.join(here, "netlists", "alu-c880-class.v"), False)
    ck = helm_amd.ClientKey.generate("toy_k2", seed=5)
    vals = {w: PtxtType.Bool(bool(v)) for w, v in zip(inputs, np.random.default_rng(8).integers(0, 2, len(inputs)))}

    def circuit():
        c = Circuit(gates, inputs, outputs, dffs)
        c.sort_circuit()
        c.compute_levels()
        return c
    sk0 = helm_amd.ServerKey(ck, device=0)
    gc0 = GateCircuit(ck, sk0, circuit())
    enc0 = gc0.encrypt_inputs(wire_set, vals)
    one = gc0.evaluate_encrypted(enc0, 1, "bool")
    want = {w: np.array(one[w]).copy() for w in one.keys()}
    in_cts = {w: np.array(enc0[w]).copy() for w in inputs}
    comms = Comm.in_process_group([0] * world)
    ranks, errors, results = [], [], [None] * world
    for r in range(world):
        sk = helm_amd.ServerKey(ck, device=0)
        gc = GateCircuit(ck, sk, circuit())
        enc = gc.encrypt_inputs(wire_set, vals)
        for w in inputs:  # the same input ciphertexts on every rank (fresh encryptions differ)
            enc[w] = in_cts[w]
        gc.shard_over(comms[r], 0, overlap=overlap)
        ranks.append((sk, gc, enc))

    def rank_main(r):
        try:
            sk, gc, enc = ranks[r]
            out = gc.evaluate_encrypted(enc, 1, "bool")
            results[r] = {w: np.array(out[w]).copy() for w in out.keys()}
        except BaseException as e:  # noqa: BLE001
            errors.append((r, repr(e)))
            comms[r].abort_group()
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
        assert not t.is_alive(), "a rank thread is stuck"
    assert not errors, errors
    for r in range(world):
        assert set(results[r]) == set(want)
        for w in want:
            assert np.array_equal(results[r][w], want[w]), f"world {world}: rank {r}, wire {w}"
        assert comms[r].stats()["collectives"] > 0
    for sk, gc, _ in ranks:
        gc.shard_over(None)
        sk.close()
    sk0.close()
    for c in comms:
        c.destroy()
