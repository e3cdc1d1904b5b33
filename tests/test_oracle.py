"""The CPU oracle against (a) the known-answer tests the reference's own tests hold for
this path — all at decrypted-plaintext level (tests/gates_test.rs:14-107 truth tables,
tests/circuit_test.rs:47-94 two-bit adder on every wire), (b) its own two independent
exact routes (schoolbook vs Goldilocks NTT), (c) the committed golden vectors."""
import os

import numpy as np
import pytest

import helm_amd
import oracle
from helm_amd import Circuit, PtxtType, verilog_parser
from helm_amd._native import Params

HERE = os.path.dirname(os.path.abspath(__file__))

GATES2 = {
    oracle.AND: lambda a, b: a & b,
    oracle.OR: lambda a, b: a | b,
    oracle.NAND: lambda a, b: 1 - (a & b),
    oracle.NOR: lambda a, b: 1 - (a | b),
    oracle.XOR: lambda a, b: a ^ b,
    oracle.XNOR: lambda a, b: 1 - (a ^ b),
}


def test_decompose_reconstructs_and_is_balanced():
    rng = np.random.default_rng(0)
    for logB, l in [(6, 3), (7, 3), (8, 2), (3, 4), (2, 8), (4, 4)]:
        xs = list(rng.integers(0, 2**32, size=200)) + [0, 1, 2**31, 2**32 - 1, 2**31 - 1, (1 << (32 - logB * l)) - 1]
        for x in xs:
            d = oracle.decompose(int(x), logB, l)
            assert all(-(1 << (logB - 1)) <= int(v) <= (1 << (logB - 1)) for v in d)
            rec = sum(int(d[j]) << (32 - logB * (j + 1)) for j in range(l)) % 2**32
            # closest representable: error at most half of the last level's weight
            err = (rec - int(x) + 2**31) % 2**32 - 2**31
            assert abs(err) <= 1 << (31 - logB * l)


def test_modswitch_rounds_half_up():
    for log2_2N in (10, 11):
        sh = 32 - log2_2N
        assert oracle.modswitch(0, log2_2N) == 0
        assert oracle.modswitch((1 << (sh - 1)) - 1, log2_2N) == 0
        assert oracle.modswitch(1 << (sh - 1), log2_2N) == 1
        assert oracle.modswitch(2**32 - 1, log2_2N) == 0  # wraps to 2N == 0
        assert oracle.modswitch(3 << sh, log2_2N) == 3


def test_lincomb_encodings():  # +-1/8 encoding, circuit.rs:29,33; tfhe boolean gate formulas
    T, F = 1 << 29, 7 << 29
    n = 3
    def ct(v):
        a = np.zeros(n + 1, np.uint32); a[n] = v; return a
    for op, f in GATES2.items():
        for a in (0, 1):
            for b in (0, 1):
                ph = int(oracle.lincomb(n, op, 0, ct(T if a else F), ct(T if b else F))[n])
                assert (ph < 2**31) == bool(f(a, b)), (op, a, b)
                # and the phase sits at distance >= 1/8 from the decision boundaries 0 and 1/2
                assert min(ph % 2**31, 2**31 - ph % 2**31) >= 2**29


@pytest.fixture(scope="module", params=["toy", "toy_k2", "toy_1024"])
def toy(request):
    ck = helm_amd.ClientKey.generate(request.param, seed=7)
    return ck, oracle.Oracle(ck.params.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True)


def test_truth_tables(toy):  # K-3: tests/gates_test.rs:14-107
    ck, orc = toy
    ct = ck.encrypt([False, True])
    for op, f in GATES2.items():
        for a in (0, 1):
            for b in (0, 1):
                assert ck.decrypt(orc.gate(op, ct[a], ct[b])) == bool(f(a, b))
    for s in (0, 1):
        for a in (0, 1):
            for b in (0, 1):
                assert ck.decrypt(orc.gate(oracle.MUX, ct[a], ct[b], ct[s])) == bool(a if s else b)
    assert ck.decrypt(orc.gate(oracle.NOT, ct[1])) is False and ck.decrypt(orc.gate(oracle.NOT, ct[0])) is True


def test_schoolbook_equals_ntt(toy):
    ck, orc = toy
    p = ck.params
    sb = oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk, use_ntt=False)
    rng = np.random.default_rng(5)
    diff = rng.integers(0, 2**32, size=(p.k + 1) * p.N, dtype=np.uint32)
    a1 = rng.integers(0, 2**32, size=(p.k + 1) * p.N, dtype=np.uint32)
    a2 = a1.copy()
    orc.extprod_add(1, diff, a1)
    sb.extprod_add(1, diff, a2)
    assert np.array_equal(a1, a2)
    ct = ck.encrypt([True, False])
    assert np.array_equal(orc.gate(oracle.NAND, ct[0], ct[1]), sb.gate(oracle.NAND, ct[0], ct[1]))


def test_bootstrap_output_noise_is_small(toy):
    ck, orc = toy
    p = ck.params
    tv = np.full(p.N, 1 << 29, dtype=np.uint32)
    for bit in (True, False):
        big = orc.bootstrap_noks(ck.encrypt(bit), tv)
        ph = int(ck.phase(big, big=True)[0])
        want = (1 << 29) if bit else (7 << 29)
        assert abs((ph - want + 2**31) % 2**32 - 2**31) < 1 << 24


def test_golden_vectors():
    g = np.load(os.path.join(HERE, "golden", "gates_toy.npz"))
    params = tuple(int(x) for x in g["params"])
    orc = oracle.Oracle(params, g["bsk"], g["ksk"], use_ntt=True)  # golden was made by the schoolbook route
    n = params[0]
    wires = np.zeros((2 + len(g["ops"]), n + 1), dtype=np.uint32)
    wires[:2] = g["inputs"]
    orc.eval_level(wires, g["ops"], g["in0"], g["in1"], g["in2"], np.arange(2, 2 + len(g["ops"]), dtype=np.int32))
    assert np.array_equal(wires[2:], g["expected"])
    # and the expected ciphertexts decrypt to the truth tables
    sk = g["lwe_sk"]
    exp = []
    for op in (oracle.AND, oracle.OR, oracle.NAND, oracle.NOR, oracle.XOR, oracle.XNOR):
        exp += [GATES2[op](a, b) for a in (0, 1) for b in (0, 1)]
    exp += [(a if s else b) for s in (0, 1) for a in (0, 1) for b in (0, 1)] + [1, 0]
    got = [int(oracle.decrypt_bool(sk, row)) for row in g["expected"]]
    assert got == exp


def test_client_keygen_is_deterministic():
    g = np.load(os.path.join(HERE, "golden", "gates_toy.npz"))
    p = Params(32, *[int(x) for x in g["params"]], 0, 1)
    ck = helm_amd.ClientKey(p, 1e-7, 1e-9, seed=2024)
    assert np.array_equal(ck.lwe_secret, g["lwe_sk"]) and np.array_equal(ck.glwe_secret, g["glwe_sk"])
    assert np.array_equal(ck.ksk, g["ksk"]) and np.array_equal(ck.bsk, g["bsk"])


def _netlist_level_arrays(circuit, index):
    ops, i0, i1, i2, out, off = [], [], [], [], [], [0]
    lm = circuit.level_map()
    for lvl in sorted(lm):
        for gate in lm[lvl]:
            ins = [index[w] for w in gate.input_wires] + [-1, -1, -1]
            ops.append(int(gate.gate_type)); i0.append(ins[0]); i1.append(ins[1]); i2.append(ins[2])
            out.append(index[gate.output_wire])
        off.append(len(ops))
    return [np.array(x, np.int32) for x in (ops, i0, i1, i2, out)] + [off]


def test_encrypted_two_bit_adder_oracle():  # K-1/K-2: circuit_test.rs:47-94 with the oracle as evaluator
    gates_set, wire_set, input_wires, _, _, _, _ = verilog_parser.read_verilog_file(
        os.path.join(HERE, "netlists", "2-bit-adder.v"), False)
    circuit = Circuit(gates_set, input_wires, [], [])
    circuit.sort_circuit()
    circuit.compute_levels()
    ck = helm_amd.ClientKey.generate("toy_k2", seed=3)
    orc = oracle.Oracle(ck.params.as_tuple7(), ck.bsk, ck.ksk)
    names = sorted(wire_set) + list(input_wires)
    index = {w: i for i, w in enumerate(names)}
    ops, i0, i1, i2, out, off = _netlist_level_arrays(circuit, index)
    for inputs in ({w: True for w in input_wires},
                   {"a[0]": True, "a[1]": False, "b[0]": False, "b[1]": True, "cin": False}):
        ptxt = {w: PtxtType.Bool(True) for w in wire_set}
        ptxt.update({w: PtxtType.Bool(v) for w, v in inputs.items()})
        ptxt = circuit.evaluate(ptxt)
        wires = np.zeros((len(names), ck.params.n + 1), dtype=np.uint32)
        for w in wire_set:
            wires[index[w]] = ck.encrypt(False)
        for w, v in inputs.items():
            wires[index[w]] = ck.encrypt(v)
        for l in range(len(off) - 1):
            s = slice(off[l], off[l + 1])
            orc.eval_level(wires, ops[s], i0[s], i1[s], i2[s], out[s])
        for w in names:  # every wire, not only outputs
            assert ck.decrypt(wires[index[w]]) == bool(ptxt[w].value), w


# ---- route 3: the SIMD fp64 route that bench.py times as the CPU baseline (oracle/fp_route.inc) ----------------
_FP_CHECK = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import helm_amd, oracle
assert oracle.fp_lanes() == %d, oracle.fp_lanes()
for name in ("toy", "toy_k2", "toy_1024"):
    ck = helm_amd.ClientKey.generate(name, seed=7)
    p = ck.params
    o_sb = oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk, use_ntt=False)
    o = oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True, use_fp=True)
    rng = np.random.default_rng(5)
    lwe = rng.integers(0, 2**32, size=(13, p.n + 1), dtype=np.uint32)  # 13: a full vector of lanes and a partial one
    lwe[0] = ck.encrypt(True)
    lwe[4, :] = 0                                                     # every rotation zero
    lwe[5, :p.n] = 0                                                  # zero mask, non-zero body
    tv = rng.integers(0, 2**32, size=p.N, dtype=np.uint32)
    got = o.bootstrap_noks_fp(lwe, tv)
    for g in range(len(lwe)):
        assert np.array_equal(got[g], o.bootstrap_noks(lwe[g], tv)), (name, g, "fp vs Goldilocks")
    assert np.array_equal(got[1], o_sb.bootstrap_noks(lwe[1], tv)), (name, "fp vs schoolbook")
    # a whole level with every gate type, MUX included (two bootstraps, one keyswitch)
    bits = rng.integers(0, 2, 6).astype(bool)
    n_g = 21
    ops = np.array([oracle.AND, oracle.OR, oracle.NAND, oracle.NOR, oracle.XOR, oracle.XNOR, oracle.MUX, oracle.NOT,
                    oracle.BUF, oracle.CONST_ONE, oracle.CONST_ZERO] * 2, np.int32)[:n_g]
    i0, i1, i2 = (rng.integers(0, 6, n_g).astype(np.int32) for _ in range(3))
    out = np.arange(6, 6 + n_g, dtype=np.int32)
    w1 = np.zeros((6 + n_g, p.n + 1), np.uint32)
    w1[:6] = ck.encrypt(bits)
    w2 = w1.copy()
    o.eval_level_fp(w1, ops, i0, i1, i2, out, nthreads=2)
    o.eval_level(w2, ops, i0, i1, i2, out, nthreads=2)
    assert np.array_equal(w1, w2), (name, "level")
print("ok", hex(o.fp_prime()))
"""


@pytest.mark.parametrize("lanes", [4, 8])
def test_simd_fp64_route_equals_the_integer_routes(lanes):
    """Route 3 (exact fp64-FMA NTT over a 49/51-bit prime, one gate per SIMD lane) against route 2 (Goldilocks) and
    route 1 (schoolbook) bit for bit: single bootstraps incl. partial lane groups and all-zero rotations, and a level
    with every gate type.  Both builds: AVX-512 (8 lanes) where the CPU has it, AVX2 (4 lanes, forced)."""
    import subprocess
    import sys
    if oracle.fp_lanes() < 0:
        pytest.skip("no AVX2 + FMA on this CPU")
    if lanes == 8 and oracle.fp_lanes() != 8:
        pytest.skip("no AVX-512 on this CPU: the 8-lane build cannot run here")
    env = dict(os.environ, ORC_FP_LANES=str(lanes), OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-c", _FP_CHECK % (os.path.dirname(HERE), lanes)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr


def test_simd_fp64_route_full_size_sets():
    """boolean_default runs in the 49-bit prime 0x24007A8500001 (a different field from the GPU's), helm_cuda
    (helm.rs:141-146: N = 1024, base 2^7) needs the 51-bit one; both equal the Goldilocks route on a mixed level."""
    if oracle.fp_lanes() < 0:
        pytest.skip("no AVX2 + FMA on this CPU")
    for name, prime in (("boolean_default", 0x24007A8500001), ("helm_cuda", 0x6060002B00001)):
        ck = helm_amd.ClientKey.generate(name, seed=5)
        p = ck.params
        o = oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True, use_fp=True)
        assert o.fp_prime() == prime
        rng = np.random.default_rng(3)
        bits = rng.integers(0, 2, 8).astype(bool)
        n_g = 10
        ops = rng.choice([oracle.AND, oracle.XOR, oracle.NOR, oracle.MUX], n_g).astype(np.int32)
        i0, i1, i2 = (rng.integers(0, 8, n_g).astype(np.int32) for _ in range(3))
        out = np.arange(8, 8 + n_g, dtype=np.int32)
        w1 = np.zeros((8 + n_g, p.n + 1), np.uint32)
        w1[:8] = ck.encrypt(bits)
        w2 = w1.copy()
        o.eval_level_fp(w1, ops, i0, i1, i2, out, nthreads=4)
        o.eval_level(w2, ops, i0, i1, i2, out, nthreads=4)
        assert np.array_equal(w1, w2), name
        dec = ck.decrypt(w1[8:])
        for g in range(n_g):
            a, b, c = bits[i0[g]], bits[i1[g]], bits[i2[g]]
            want = {oracle.AND: a & b, oracle.XOR: a ^ b, oracle.NOR: not (a | b), oracle.MUX: a if c else b}[int(ops[g])]
            assert bool(dec[g]) == bool(want), (name, g)
