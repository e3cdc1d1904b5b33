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


# ---------------------------------------------------------------------------------------
# committed golden vectors + an independent numpy restatement of apply_lookup_table
# ---------------------------------------------------------------------------------------
def _np_decompose(x, logB, l):
    """closest-representable balanced digits of uint64 array x -> list of int64 arrays, level 1 first"""
    rep = logB * l
    state = ((x + np.uint64(1 << (63 - rep))) >> np.uint64(64 - rep)).astype(np.uint64)
    out = [None] * l
    for lev in range(l - 1, -1, -1):
        d = state & np.uint64((1 << logB) - 1)
        state = state >> np.uint64(logB)
        carry = (((d - np.uint64(1)) | state) & d) >> np.uint64(logB - 1)
        state = state + carry
        out[lev] = d.astype(np.int64) - (carry.astype(np.int64) << logB)
    return out


def _np_negacyclic(d, row):
    """(sum_a d[a] X^a) * row  mod X^N + 1, wrapping uint64"""
    N = len(row)
    acc = np.zeros(N, dtype=np.uint64)
    du = d.astype(np.uint64)  # two's complement: wrapping arithmetic is a ring homomorphism
    for a in np.nonzero(d)[0]:
        t = du[a] * row
        acc[a:] += t[:N - a]
        acc[:a] -= t[N - a:]
    return acc


def _np_apply_lut(P, bsk, ksk, ct, tv):
    n, k, N, l, logB, ksl, kslogB = P[:7]
    kN = k * N
    ksk = ksk.reshape(kN, ksl, n + 1)
    small = np.zeros(n + 1, dtype=np.uint64)
    small[n] = ct[kN]
    digs = _np_decompose(ct[:kN], kslogB, ksl)
    for j in range(ksl):
        small -= (digs[j].astype(np.uint64)[:, None] * ksk[:, j, :]).sum(axis=0, dtype=np.uint64)
    ms = lambda x: ((int(x) >> (64 - (N.bit_length()) - 1)) + 1 >> 1) & (2 * N - 1)
    rot = lambda poly, a: np.array([poly[(j - a) % (2 * N)] if (j - a) % (2 * N) < N else
                                    np.uint64(0) - poly[(j - a) % (2 * N) - N] for j in range(N)], dtype=np.uint64)
    acc = np.zeros((k + 1, N), dtype=np.uint64)
    acc[k] = rot(tv, (2 * N - ms(small[n])) % (2 * N))
    bsk = bsk.reshape(n, l, k + 1, k + 1, N)
    for i in range(n):
        a = ms(small[i])
        if a == 0:
            continue
        diff = np.stack([rot(acc[r], a) - acc[r] for r in range(k + 1)])
        for r in range(k + 1):
            digs = _np_decompose(diff[r], logB, l)
            for j in range(l):
                for c in range(k + 1):
                    acc[c] += _np_negacyclic(digs[j], bsk[i, j, r, c])
    out = np.zeros(kN + 1, dtype=np.uint64)
    for r in range(k):
        out[r * N] = acc[r][0]
        out[r * N + 1:(r + 1) * N] = np.uint64(0) - acc[r][:0:-1]
    out[kN] = acc[k][0]
    return out


def test_golden_vectors_two_independent_routes():
    g = np.load(os.path.join(HERE, "golden", "shortint_toy.npz"))
    P = [int(x) for x in g["params"]]
    orc = oracle.Oracle64(P, g["bsk"], g["ksk"])
    n_in = len(g["inputs"])
    wires = np.zeros((n_in + len(g["arity"]), P[1] * P[2] + 1), dtype=np.uint64)
    wires[:n_in] = g["inputs"]
    out_idx = np.arange(n_in, len(wires), dtype=np.int32)
    orc.eval_lut_level(wires, g["arity"], g["in_idx"], g["table"], out_idx)
    assert np.array_equal(wires[n_in:], g["expected"])  # the C oracle reproduces the committed vectors
    # numpy route for the three gate kinds that bootstrap: pack, look-up table, keyswitch, blind rotate
    t, delta = P[7] * P[8], (1 << 63) // (P[7] * P[8])
    with np.errstate(over="ignore"):
        for gi in (0, 2, 3):
            ar, ins, tb = int(g["arity"][gi]), g["in_idx"][gi], int(g["table"][gi])
            packed = np.zeros_like(wires[0])
            for q in range(ar):
                packed += np.uint64(1 << (ar - 1 - q)) * g["inputs"][ins[q]]
            f = [(tb >> ((((v >> 1) & 1) * 2 + (v & 1)) if ar == 2 else (v & ((1 << ar) - 1)))) & 1 for v in range(t)]
            box = P[2] // t
            acc = np.repeat(np.array(f, dtype=np.uint64) * np.uint64(delta), box)
            acc[:box // 2] = np.uint64(0) - acc[:box // 2]
            tv = np.roll(acc, -(box // 2))
            assert np.array_equal(tv, orc.make_lut(f))
            assert np.array_equal(_np_apply_lut(P, g["bsk"], g["ksk"], packed, tv), g["expected"][gi]), gi
    # and they decrypt to the truth tables
    bits = [int(b) for b in g["bits"]]
    for gi in range(len(g["arity"])):
        ar, ins, tb = int(g["arity"][gi]), g["in_idx"][gi], int(g["table"][gi])
        x = [bits[i] for i in ins[:max(ar, 1)]]
        want = ((tb >> sum(b << (ar - 1 - q) for q, b in enumerate(x))) & 1) if ar >= 2 else \
            (x[0] if (ar == 0 or tb == 0) else (-x[0]) % t)
        assert orc.decrypt(g["glwe_sk"], g["expected"][gi]) == want, gi


# ---------------------------------------------------------------------------------------
# multi-bit blind rotation (grouping factor g; the arithmetic-mode set of reference src/bin/helm.rs:83)
# ---------------------------------------------------------------------------------------
def _np_multibit_bootstrap(P, bsk, small, tv):
    """numpy restatement of the multi-bit programmable bootstrap: per group of g mask words
    G = sum_S X^(sum_{i in S} a_i) * GGSW_S, acc <- G (x) acc."""
    n, k, N, l, logB = P[:5]
    g = P[9]
    ms = lambda x: ((int(x) >> (64 - (N.bit_length()) - 1)) + 1 >> 1) & (2 * N - 1)
    rot = lambda poly, a: np.array([poly[(j - a) % (2 * N)] if (j - a) % (2 * N) < N else
                                    np.uint64(0) - poly[(j - a) % (2 * N) - N] for j in range(N)], dtype=np.uint64)
    acc = np.zeros((k + 1, N), dtype=np.uint64)
    acc[k] = rot(tv, (2 * N - ms(small[n])) % (2 * N))
    bsk = bsk.reshape(n // g, 1 << g, l, k + 1, k + 1, N)
    for t in range(n // g):
        a = [ms(small[t * g + q]) for q in range(g)]
        G = np.zeros((l, k + 1, k + 1, N), dtype=np.uint64)
        for S in range(1 << g):
            e = sum(a[q] for q in range(g) if (S >> q) & 1) % (2 * N)
            for j in range(l):
                for r in range(k + 1):
                    for c in range(k + 1):
                        G[j, r, c] += rot(bsk[t, S, j, r, c], e)
        new = np.zeros_like(acc)
        for r in range(k + 1):
            digs = _np_decompose(acc[r], logB, l)
            for j in range(l):
                for c in range(k + 1):
                    new[c] += _np_negacyclic(digs[j], G[j, r, c])
        acc = new
    out = np.zeros(k * N + 1, dtype=np.uint64)
    for r in range(k):
        out[r * N] = acc[r][0]
        out[r * N + 1:(r + 1) * N] = np.uint64(0) - acc[r][:0:-1]
    out[k * N] = acc[k][0]
    return out


@pytest.mark.parametrize("g,n", [(2, 4), (3, 6)])
def test_multibit_bootstrap_two_routes_and_every_value(g, n):
    p = helm_amd.SiParams(n=n, k=1, N=256, pbs_l=2, pbs_logB=12, ks_l=4, ks_logB=4, message_modulus=4,
                          carry_modulus=4, grouping_factor=g)
    ck = helm_amd.SiClientKey(p, 1e-9, 1e-15, seed=7)
    assert ck.bsk.size == (n // g) * (1 << g) * p.pbs_l * 4 * p.N  # 2^g GGSWs per group
    orc = oracle.Oracle64(p.as_tuple(), ck.bsk, ck.ksk)
    f = lambda v: (7 * v + 2) % ck.t
    lut = orc.make_lut(f)
    cts = ck.encrypt(np.arange(ck.t, dtype=np.uint64))
    with np.errstate(over="ignore"):
        for v in range(ck.t):
            small = orc.keyswitch(cts[v])
            out = orc.bootstrap(small, lut)
            assert ck.decrypt_message_and_carry(out) == f(v)
            if v in (0, 5, 15):  # the numpy route, bit for bit
                assert np.array_equal(out, _np_multibit_bootstrap(list(p.as_tuple()), ck.bsk, small, lut)), v


def test_multibit_golden_vectors_two_routes():
    """tests/golden/shortint_mb_toy.npz (grouping factor 2, N = 1024): the C oracle reproduces the committed
    ciphertexts; the numpy route re-derives the first 3-input gate from the packed operand."""
    g = np.load(os.path.join(HERE, "golden", "shortint_mb_toy.npz"))
    P = [int(x) for x in g["params"]]
    assert P[9] == 2
    orc = oracle.Oracle64(P, g["bsk"], g["ksk"])
    n_in = len(g["inputs"])
    wires = np.zeros((n_in + len(g["arity"]), P[1] * P[2] + 1), dtype=np.uint64)
    wires[:n_in] = g["inputs"]
    orc.eval_lut_level(wires, g["arity"], g["in_idx"], g["table"], np.arange(n_in, len(wires), dtype=np.int32))
    assert np.array_equal(wires[n_in:], g["expected"])
    t = P[7] * P[8]
    with np.errstate(over="ignore"):
        ar, ins, tb = int(g["arity"][0]), g["in_idx"][0], int(g["table"][0])
        packed = np.zeros_like(wires[0])
        for q in range(ar):
            packed += np.uint64(1 << (ar - 1 - q)) * g["inputs"][ins[q]]
        tv = orc.make_lut([(tb >> (v & ((1 << ar) - 1))) & 1 for v in range(t)])
        small = orc.keyswitch(packed)
        assert np.array_equal(_np_multibit_bootstrap(P, g["bsk"], small, tv), g["expected"][0])
    bits = [int(b) for b in g["bits"]]
    for gi in range(len(g["arity"])):
        ar, ins, tb = int(g["arity"][gi]), g["in_idx"][gi], int(g["table"][gi])
        x = [bits[i] for i in ins[:max(ar, 1)]]
        want = ((tb >> sum(b << (ar - 1 - q) for q, b in enumerate(x))) & 1) if ar >= 2 else \
            (x[0] if (ar == 0 or tb == 0) else (-x[0]) % t)
        assert orc.decrypt(g["glwe_sk"], g["expected"][gi]) == want, gi


# ---- round 5: the second exact route (Goldilocks NTT on the key split into parts) --------------------------------
TOY_SETS = ["si_toy_512", "si_toy_1024", "si_toy_2048", "si_toy_2048_l2", "si_toy_512_k2", "si_toy_512_k3", "si_toy_1024_mb2",
            "si_toy_2048_mb3"]


@pytest.mark.parametrize("name", TOY_SETS)
def test_ntt_route_equals_schoolbook_on_every_toy_set(name):
    """Every kernel class of the 64-bit engine has a toy set (one / two levels, k = 1, 2, 3, multi-bit g = 2, 3): on each,
    keyswitch + bootstrap through the NTT route give the schoolbook route's ciphertexts bit for bit - independent
    arithmetic (wrapping u64 convolution vs exact integers below 2^63 recombined mod 2^64), same algorithm."""
    ck = helm_amd.SiClientKey.generate(name, seed=3)
    a = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk)
    b = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk, use_ntt=True)
    assert a.part_bits == 0 and b.part_bits in (8, 16, 32)
    rng = np.random.default_rng(5)
    vals = rng.integers(0, ck.t, size=4).astype(np.uint64)
    cts = ck.encrypt(vals)
    lut = a.make_lut(lambda v: (3 * v + 1) % ck.t)
    for v, ct in zip(vals, cts):
        want = a.apply_lut(ct, lut)
        assert np.array_equal(b.apply_lut(ct, lut), want), name
        assert ck.decrypt_message_and_carry(want) == (3 * int(v) + 1) % ck.t
    # the batch forms (OpenMP over rows) and the level form
    luts = np.stack([lut, a.make_lut(lambda v: v & 1)])
    idx = np.array([0, 1, 1, 0], dtype=np.int32)
    got = b.apply_luts(cts, luts, idx)
    for q in range(4):
        assert np.array_equal(got[q], a.apply_lut(cts[q], luts[idx[q]]))
    bits = rng.integers(0, 2, size=6).astype(np.uint64)
    wires_a = np.zeros((8, ck.dim + 1), dtype=np.uint64)
    wires_a[:6] = ck.encrypt(bits)
    wires_b = wires_a.copy()
    args = ([3, 2], [[0, 1, 2], [3, 4, -1]] if False else [[0, 1, 2], [3, 4, 5]], [0x96, 0x6], [6, 7])
    a.eval_lut_level(wires_a, *args)
    b.eval_lut_level(wires_b, *args)
    assert np.array_equal(wires_a, wires_b)
    rows = b.eval_lut_rows(wires_b, *args[:3], [1])
    assert np.array_equal(rows[0], wires_a[7])


def test_ntt_route_at_a_full_parameter_set_matches_schoolbook_and_decrypts():
    """PARAM_MESSAGE_1_CARRY_1_KS_PBS's dimensions (k = 3, N = 512: the cheapest full set for the O(N^2) route): one
    bootstrap both ways, identical; and the NTT route alone on a row of PARAM_MESSAGE_2_CARRY_2 (N = 2048, 16-bit parts)
    decrypts to the look-up's value."""
    ck = helm_amd.SiClientKey.generate("shortint_m1c1", seed=4)
    a = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk)
    b = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk, use_ntt=True)
    assert b.part_bits == 32
    ct = ck.encrypt(np.array([3], dtype=np.uint64))[0]
    lut = a.make_lut(lambda v: (v + 2) % ck.t)
    want = a.apply_lut(ct, lut)
    assert np.array_equal(b.apply_lut(ct, lut), want) and ck.decrypt_message_and_carry(want) == (3 + 2) % ck.t
    ck2 = helm_amd.SiClientKey.generate("shortint_m2c2", seed=4)
    c = oracle.Oracle64(ck2.params.as_tuple(), ck2.bsk, ck2.ksk, use_ntt=True)
    assert c.part_bits == 16
    out = c.apply_lut(ck2.encrypt(np.array([11], dtype=np.uint64))[0], c.make_lut(lambda v: (5 * v + 3) % ck2.t))
    assert ck2.decrypt_message_and_carry(out) == (5 * 11 + 3) % ck2.t
