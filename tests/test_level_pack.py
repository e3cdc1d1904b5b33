"""Launch packing (helm_amd/csrc/host/level_pack.cpp): the level loop of reference src/circuit.rs:524-543
re-timed so that launches hold whole lockstep rounds.  Checked here on the CPU: it is a permutation, every
gate runs after its producers, launches are whole multiples of the quantum while enough gates are ready,
state-writing gates (DFFs) keep their place, and a launch-by-launch plaintext evaluation gives the level
schedule's values on every wire."""
import numpy as np
import pytest

from helm_amd import Circuit, verilog_parser
from helm_amd.distributed import level_arrays, pack_levels
from helm_amd.netlists import aes128, aes128_reference_encrypt

AND, DFF, LUT, MUX, NAND, NOR, NOT, OR, XNOR, XOR, BUF, ONE, ZERO = range(13)
COST = {AND: 1, NAND: 1, OR: 1, NOR: 1, XOR: 1, XNOR: 1, MUX: 2}


def _plain_run(ops, i0, i1, i2, out, off, values):
    """Evaluate launch by launch: every gate of a launch reads the table as it was before the launch."""
    v = values.copy()
    for l in range(len(off) - 1):
        s = slice(off[l], off[l + 1])
        o, a, b, c = ops[s], v[np.maximum(i0[s], 0)], v[np.maximum(i1[s], 0)], v[np.maximum(i2[s], 0)]
        r = np.zeros(len(o), dtype=np.uint8)
        for code, f in ((AND, a & b), (OR, a | b), (NAND, 1 - (a & b)), (NOR, 1 - (a | b)), (XOR, a ^ b), (XNOR, 1 - (a ^ b)),
                        (MUX, np.where(c == 1, a, b)), (NOT, 1 - a), (BUF, a), (DFF, a), (ONE, np.ones_like(a)), (ZERO, np.zeros_like(a))):
            r = np.where(o == code, f, r)
        v[out[s]] = r
    return v


def _tiled_aes(blocks):
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(aes128(), False)
    c = Circuit(gates, inputs, outputs, dffs)
    c.sort_circuit()
    c.compute_levels()
    names = list(inputs) + sorted(wire_set)
    index = {w: i for i, w in enumerate(names)}
    ops, i0, i1, i2, out, off = level_arrays(c, index)
    nw, nl = len(names), len(off) - 1
    t = lambda a: np.concatenate([np.concatenate([np.where(a[off[l]:off[l + 1]] >= 0, a[off[l]:off[l + 1]] + b * nw, -1)
                                                  for b in range(blocks)]) for l in range(nl)]).astype(np.int32)
    opsT = np.concatenate([np.tile(ops[off[l]:off[l + 1]], blocks) for l in range(nl)]).astype(np.int32)
    return opsT, t(i0), t(i1), t(i2), t(out), (off * blocks).astype(np.int64), index, nw


def _check_schedule(orig, packed, quantum):
    ops, i0, i1, i2, out, off = orig
    pops, pi0, pi1, pi2, pout, poff, was_packed = packed
    assert was_packed
    assert sorted(zip(pops.tolist(), pi0.tolist(), pi1.tolist(), pi2.tolist(), pout.tolist())) == \
        sorted(zip(ops.tolist(), i0.tolist(), i1.tolist(), i2.tolist(), out.tolist()))
    launch_of_wire = {}
    for l in range(len(poff) - 1):
        for g in range(poff[l], poff[l + 1]):
            launch_of_wire[int(pout[g])] = l
    for l in range(len(poff) - 1):
        for g in range(poff[l], poff[l + 1]):
            for w in (pi0[g], pi1[g], pi2[g]):
                if w >= 0 and int(w) in launch_of_wire:
                    assert launch_of_wire[int(w)] < l, f"gate at {g} reads wire {w} before it is produced"
    cost = np.array([COST.get(int(o), 0) for o in pops])
    per_launch = np.array([int(cost[poff[l]:poff[l + 1]].sum()) for l in range(len(poff) - 1)])
    assert per_launch.sum() == sum(COST.get(int(o), 0) for o in ops)
    return per_launch


def test_aes_batch_packs_into_whole_rounds():
    blocks, quantum = 8, 256
    ops, i0, i1, i2, out, off, index, nw = _tiled_aes(blocks)
    packed = pack_levels(ops, i0, i1, i2, out, off, quantum)
    per_launch = _check_schedule((ops, i0, i1, i2, out, off), packed, quantum)
    partial = [int(x) for x in per_launch if x % quantum]
    # the level schedule ends every one of its 207 levels with a partial round; packed, only the drain does
    assert len(partial) <= 4, partial
    assert len(per_launch) < 1.25 * (len(off) - 1)
    # plaintext values of every wire agree with the level schedule, and with a software AES
    rng = np.random.default_rng(1)
    vals = np.zeros(nw * blocks, dtype=np.uint8)
    keys_pt = []
    for b in range(blocks):
        key, pt = bytes(rng.integers(0, 256, 16, dtype=np.uint8)), bytes(rng.integers(0, 256, 16, dtype=np.uint8))
        keys_pt.append((key, pt))
        kv, pv = int.from_bytes(key, "big"), int.from_bytes(pt, "big")
        for i in range(128):
            vals[b * nw + index[f"key[{i}]"]] = (kv >> i) & 1
            vals[b * nw + index[f"pt[{i}]"]] = (pv >> i) & 1
    v_level = _plain_run(ops, i0, i1, i2, out, off, vals)
    v_packed = _plain_run(*packed[:6], vals)
    assert np.array_equal(v_level, v_packed)
    for b, (key, pt) in enumerate(keys_pt):
        ct = sum(int(v_packed[b * nw + index[f"ct[{i}]"]]) << i for i in range(128)).to_bytes(16, "big")
        assert ct == aes128_reference_encrypt(key, pt)


def test_single_block_keeps_its_levels():
    """Levels narrower than the quantum: nothing to pack, the launches are the levels."""
    ops, i0, i1, i2, out, off, _, _ = _tiled_aes(1)
    packed = pack_levels(ops, i0, i1, i2, out, off, 1024)
    assert np.array_equal(packed[5], off) and all(np.array_equal(a, b) for a, b in zip(packed[:5], (ops, i0, i1, i2, out)))


def test_mux_counts_two_and_launches_stay_whole():
    # 3 independent chains of MUX / AND gates over shared inputs 0..3; quantum 4
    rng = np.random.default_rng(5)
    n_in, width, depth = 4, 7, 6
    ops, i0, i1, i2, out, off = [], [], [], [], [], [0]
    prev = list(range(n_in))
    nxt = n_in
    for d in range(depth):
        cur = []
        for g in range(width):
            ops.append(MUX if (g + d) % 3 == 0 else AND)
            a, b, c = (int(x) for x in rng.choice(prev, 3))
            i0.append(a); i1.append(b); i2.append(c if ops[-1] == MUX else -1)
            out.append(nxt); cur.append(nxt); nxt += 1
        prev = cur + list(range(n_in))
        off.append(len(ops))
    arrs = [np.array(x, np.int32) for x in (ops, i0, i1, i2, out)] + [np.array(off, np.int64)]
    packed = pack_levels(*arrs, 4)
    per_launch = _check_schedule(tuple(arrs), packed, 4)
    assert all(x % 4 == 0 for x in per_launch[:-1] if x >= 4)
    vals = np.zeros(nxt, dtype=np.uint8)
    vals[:n_in] = [1, 0, 1, 1]
    assert np.array_equal(_plain_run(*arrs, vals), _plain_run(*packed[:6], vals))


def test_dffs_stay_behind_the_packed_launches():
    """A 2-bit counter: the DFFs overwrite q0/q1, which level-1 gates read - they must run last."""
    text = "input en;\noutput q0, q1;\ndff g0(d0, q0);\ndff g1(d1, q1);\nxor g2(q0, en, d0);\nand g3(q0, en, c0);\nxor g4(q1, c0, d1);\n"
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(text, False)
    c = Circuit(gates, inputs, outputs, dffs)
    c.sort_circuit()
    c.compute_levels()
    names = list(dict.fromkeys(list(inputs) + sorted(wire_set)))
    index = {w: i for i, w in enumerate(names)}
    arrs = level_arrays(c, index)
    packed = pack_levels(*arrs, 2)
    assert packed[6]
    pops, poff = packed[0], packed[5]
    last = slice(poff[-2], poff[-1])
    assert set(pops[last].tolist()) == {DFF} and DFF not in pops[:poff[-2]].tolist()
    vals = np.zeros(len(names), dtype=np.uint8)
    vals[index["en"]] = 1
    a, b = vals.copy(), vals.copy()
    for _ in range(3):  # three clock cycles: 01, 10, 11
        a = _plain_run(*arrs, a)
        b = _plain_run(*packed[:6], b)
        assert np.array_equal(a, b)
    assert (a[index["q0"]], a[index["q1"]]) == (1, 1)


def test_state_writer_that_feeds_later_gates_is_left_alone():
    # gate 0 (level 0) reads wire 2; gate 1 (level 1) overwrites wire 2; gate 2 (level 2) reads the new value
    ops = np.array([NOT, NOT, NOT], np.int32)
    i0 = np.array([2, 3, 2], np.int32)
    m1 = np.full(3, -1, np.int32)
    out = np.array([3, 2, 4], np.int32)
    off = np.array([0, 1, 2, 3], np.int64)
    packed = pack_levels(ops, i0, m1, m1, out, off, 4)
    assert not packed[6] and np.array_equal(packed[5], off) and np.array_equal(packed[4], out)


def test_bad_arguments():
    from helm_amd._host import Panic
    a = np.array([AND], np.int32)
    with pytest.raises(Panic):
        pack_levels(a, a * 0, a * 0, a * 0 - 1, a * 0 + 1, np.array([0, 1], np.int64), 0)


def test_random_levelised_dags_property():
    """Random levelised netlists (random widths, fan-in from any earlier level, a sprinkling of MUX and NOT gates):
    for every quantum the packed schedule is a permutation, respects every dependency
    and evaluates to the level schedule's values on every wire."""
    rng = np.random.default_rng(2026)
    for trial in range(25):
        n_in = int(rng.integers(2, 9))
        n_levels = int(rng.integers(1, 9))
        ops, i0, i1, i2, out, off = [], [], [], [], [], [0]
        avail = list(range(n_in))          # wires of earlier levels
        nxt = n_in
        for _ in range(n_levels):
            width = int(rng.integers(1, 40))
            new = []
            prev_level = avail[-max(1, min(len(avail), 30)):]
            for _ in range(width):
                kind = rng.choice([AND, OR, XOR, NAND, NOR, XNOR, MUX, NOT], p=[.18, .12, .2, .1, .1, .1, .12, .08])
                a = int(rng.choice(prev_level))   # at least one input from recent wires keeps the levels honest enough;
                b, c = int(rng.choice(avail)), int(rng.choice(avail))  # level offsets only need producers to be earlier
                ops.append(int(kind)); i0.append(a)
                i1.append(b if kind != NOT else -1); i2.append(c if kind == MUX else -1)
                out.append(nxt); new.append(nxt); nxt += 1
            avail += new
            off.append(len(ops))
        arrs = [np.array(x, np.int32) for x in (ops, i0, i1, i2, out)] + [np.array(off, np.int64)]
        vals = np.zeros(nxt, dtype=np.uint8)
        vals[:n_in] = rng.integers(0, 2, n_in)
        want = _plain_run(*arrs, vals)
        for quantum in (1, 3, 8, 64, 10**6):
            packed = pack_levels(*arrs, quantum)
            per_launch = _check_schedule(tuple(arrs), packed, quantum)
            assert np.array_equal(_plain_run(*packed[:6], vals), want), (trial, quantum)
            # no more launches than levels plus one per quantum of work plus the drain
            assert len(packed[5]) - 1 <= len(off) - 1 + int(per_launch.sum()) // max(1, quantum) + 2


def test_cost_aware_packing_takes_the_engines_best_width():
    """helm_host_pack_levels_costed (round 4): with the engine's cost per launch width (at most 1/4, 2/4, 3/4, 4/4 of a
    round: 0.42 / 0.64 / 0.89 / 1, helm_hip_launch_costs) a launch narrower than a round takes the width with the best
    bootstraps-per-cost and leaves the rest to the next launch - what keeps a rank's chunk of a sharded launch (quantum =
    world x round) on the widths the engine runs well.  Same permutation, same dependency order, same values on every wire;
    at the drain everything goes."""
    blocks, quantum = 4, 2048  # levels of 4 AES blocks are ~630 bootstraps wide: always narrower than this quantum
    cost = [0.42, 0.64, 0.89, 1.0]
    ops, i0, i1, i2, out, off, index, nw = _tiled_aes(blocks)
    plain = pack_levels(ops, i0, i1, i2, out, off, quantum)
    costed = pack_levels(ops, i0, i1, i2, out, off, quantum, quarter_cost=cost)
    per_plain = _check_schedule((ops, i0, i1, i2, out, off), plain, quantum)
    per_costed = _check_schedule((ops, i0, i1, i2, out, off), costed, quantum)
    q = quantum // 4

    def modelled(per_launch):  # cost of a schedule under the table: whole rounds + the width class of the remainder
        total = 0.0
        for w in per_launch:
            full, rem = divmod(int(w), quantum)
            total += full + (cost[min(3, (rem - 1) // q)] if rem else 0.0)
        return total
    # launches sit on the quarter steps far more often, and the schedule is cheaper under the engine's own table
    on_step = lambda per: sum(1 for w in per if w and w % q == 0)
    assert on_step(per_costed) > 2 * on_step(per_plain) and on_step(per_costed) > len(per_costed) // 3
    assert modelled(per_costed) < 0.95 * modelled(per_plain)
    assert len(per_costed) < 1.3 * len(per_plain)
    # nothing is ever left behind for good, and the values agree with the level schedule on every wire
    rng = np.random.default_rng(5)
    vals = np.zeros(nw * blocks, dtype=np.uint8)
    n_in = 256
    for b in range(blocks):
        vals[b * nw:b * nw + n_in] = rng.integers(0, 2, n_in)
    want = _plain_run(ops, i0, i1, i2, out, off, vals)
    assert np.array_equal(_plain_run(*costed[:6], vals), want)
    # uniform costs change nothing; non-positive costs are refused
    same = pack_levels(ops, i0, i1, i2, out, off, quantum, quarter_cost=[1.0, 1.0, 1.0, 1.0])
    assert np.array_equal(same[5], plain[5])
    from helm_amd._host import Panic
    with pytest.raises(Panic, match="positive"):
        pack_levels(ops, i0, i1, i2, out, off, quantum, quarter_cost=[0.0, 0.5, 0.9, 1.0])
