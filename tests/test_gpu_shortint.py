"""LUT-mode primitives on the GPU (include/helm_shortint.h) against the shortint oracle:
bit-exact keyswitch, programmable bootstrap and whole LUT gates on the toy parameter sets
(every (N, pbs_l) kernel build), truth tables of gates::lut() (reference src/gates.rs:754-785,
tests/circuit_test.rs:287-310), and the full PARAM_MESSAGE_2_CARRY_2 set."""
import numpy as np
import pytest

import helm_amd
import oracle

pytestmark = pytest.mark.gpu

TOYS = ["si_toy_512", "si_toy_1024", "si_toy_2048", "si_toy_2048_l2", "si_toy_1024_mb2", "si_toy_2048_mb3",
        "si_toy_512_k3", "si_toy_512_k2"]  # the last two: k > 1 (k_pbs64k, the kernel of PARAM_MESSAGE_1_CARRY_1)


@pytest.fixture(scope="module", params=TOYS)
def toy(request):
    ck = helm_amd.SiClientKey.generate(request.param, seed=3)
    sk = helm_amd.SiServerKey(ck)
    orc = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk)
    yield ck, sk, orc
    sk.close()


def test_keyswitch_batch_bit_exact(toy):
    ck, sk, orc = toy
    cts = ck.encrypt(np.arange(7) % ck.t)
    got = sk.keyswitch_batch(cts)
    for g in range(len(cts)):
        assert np.array_equal(got[g], orc.keyswitch(cts[g]))
    # the small ciphertexts still carry the message
    ph = ck.phase(got, small=True)
    assert np.array_equal(((ph + ck.delta // 2) // ck.delta) % ck.t, np.arange(7) % ck.t)


def test_pbs_batch_bit_exact(toy):
    ck, sk, orc = toy
    vals = np.arange(ck.t, dtype=np.uint64)
    small = sk.keyswitch_batch(ck.encrypt(vals))
    luts = np.stack([orc.make_lut(lambda x: (5 * x + 3) % ck.t), orc.make_lut(lambda x: x & 1)])
    assert np.array_equal(luts[0], sk.make_lut(lambda x: (5 * x + 3) % ck.t))  # generate_lookup_table parity
    idx = (np.arange(ck.t) % 2).astype(np.int32)
    got = sk.pbs_batch(small, luts, idx)
    for g in range(ck.t):
        want = orc.bootstrap(small[g], luts[idx[g]])
        assert np.array_equal(got[g], want), f"ciphertext {g}"
    dec = ck.decrypt_message_and_carry(got)
    assert list(dec) == [((5 * v + 3) % ck.t) if i == 0 else (v & 1) for v, i in zip(range(ck.t), idx)]


def test_lut_level_bit_exact_and_truth_tables(toy):
    ck, sk, orc = toy
    rng = np.random.default_rng(11)
    # inputs: rows 0..3 hold bits, gates of every arity gates::lut() distinguishes
    bits = np.array([1, 0, 1, 1], dtype=np.uint64)
    gates = [  # (arity, inputs, table)
        (3, [0, 1, 2], 0x96), (3, [0, 1, 2], 0xE8), (3, [3, 2, 1], 0x1B), (4, [0, 1, 2, 3], 0x6996),
        (2, [0, 1], 0x6), (2, [2, 3], 0x8), (2, [1, 0], 0xD), (1, [0], 0x0), (1, [2], 0x2), (0, [3], 0x0),
    ]
    n_in, count, max_in = len(bits), len(gates), 4
    arity = np.array([g[0] for g in gates], dtype=np.int32)
    in_idx = np.full((count, max_in), -1, dtype=np.int32)
    for g, (_, ins, _) in enumerate(gates):
        in_idx[g, :len(ins)] = ins
    table = np.array([g[2] for g in gates], dtype=np.uint64)
    out_idx = np.arange(n_in, n_in + count, dtype=np.int32)
    host = np.zeros((n_in + count, ck.dim + 1), dtype=np.uint64)
    host[:n_in] = ck.encrypt(bits)
    w = sk.wires(n_in + count)
    w.upload(np.arange(n_in), host[:n_in])
    w.eval_lut_level(arity, in_idx, table, out_idx)
    got = w.download()
    orc.eval_lut_level(host, arity, in_idx, table, out_idx)
    assert np.array_equal(got, host)
    dec = ck.decrypt_message_and_carry(got[n_in:])
    for g, (ar, ins, tb) in enumerate(gates):
        x = [int(bits[i]) for i in ins]
        if ar >= 3:
            want = (tb >> sum(b << (ar - 1 - q) for q, b in enumerate(x))) & 1  # first input = MSB (gates.rs:159-167)
        elif ar == 2:
            want = (tb >> (x[0] * 2 + x[1])) & 1
        elif ar == 1:
            want = x[0] if tb == 0 else (-x[0]) % ck.t  # smart_neg (gates.rs:769)
        else:
            want = x[0]
        assert int(dec[g]) == want, f"gate {g}"
    del rng


def test_lincomb_and_apply_luts_in_place(toy):
    ck, sk, orc = toy
    w = sk.wires(8)
    w.upload([0, 1, 2], ck.encrypt([1, 2, 3]))
    w.set_trivial([3], [2])
    # row 4 = 2*r0 + r1 + 3 ; row 0 = r0 + r2 (in place) ; row 5 = -r3 + 1
    w.lincomb([[0, 1], [0, 2], [3, -1]], [[2, 1], [1, 1], [-1, 0]], [4, 0, 5], const_add=[3, 0, 1])
    assert list(ck.decrypt_message_and_carry(w.download([4, 0, 5]))) == [7, 4, (-2 + 1) % ck.t]
    lut = sk.make_lut(lambda x: (x * x) % ck.t)
    w.apply_luts([4, 0], lut, [4, 6])
    assert list(ck.decrypt_message_and_carry(w.download([4, 6]))) == [49 % ck.t, 16 % ck.t]


def test_errors():
    p, a, b = helm_amd.si_named_params("si_toy_512")
    bad = helm_amd.SiParams(*p.as_tuple())
    bad.k = 2
    with pytest.raises(helm_amd.HelmError, match="unsupported"):
        helm_amd.SiServerKey(params=bad)
    sk = helm_amd.SiServerKey(params=p)
    w = sk.wires(4)
    with pytest.raises(helm_amd.HelmError, match="not loaded"):
        w.apply_luts([0], np.zeros(p.N, dtype=np.uint64), [1])
    with pytest.raises(helm_amd.HelmError, match="out of range"):
        w.lincomb([[9]], [[1]], [0])
    sk.close()


def test_full_parameter_set_m2c2():
    """PARAM_MESSAGE_2_CARRY_2_KS_PBS: one bootstrap bit-exact against the O(N^2) oracle, every
    plaintext value through a LUT, and the 3-input packing of gates::lut()."""
    ck = helm_amd.SiClientKey.generate("shortint_m2c2", seed=1)
    sk = helm_amd.SiServerKey(ck)
    orc = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk)
    vals = np.arange(ck.t, dtype=np.uint64)
    cts = ck.encrypt(vals)
    w = sk.wires(64)
    w.upload(np.arange(ck.t), cts)
    lut = sk.make_lut(lambda x: (7 * x + 5) % ck.t)
    w.apply_luts(np.arange(ck.t), lut, np.arange(ck.t) + ck.t)
    got = w.download(np.arange(ck.t) + ck.t)
    assert list(ck.decrypt_message_and_carry(got)) == [(7 * v + 5) % ck.t for v in range(ck.t)]
    # determinism + bit-exactness of one full-size bootstrap (12.4 G exact u64 multiply-adds on the CPU)
    assert np.array_equal(got[5], orc.apply_lut(cts[5], lut))
    # 8 x majority / parity of three encrypted bits
    bits = np.array([[a, b, c] for a in (0, 1) for b in (0, 1) for c in (0, 1)], dtype=np.uint64)
    w.upload(np.arange(32, 56), ck.encrypt(bits.reshape(-1)))
    in_idx = np.arange(32, 56, dtype=np.int32).reshape(8, 3)
    for tb, fn in ((0xE8, lambda a, b, c: (a + b + c) >= 2), (0x96, lambda a, b, c: (a + b + c) & 1)):
        w.eval_lut_level(np.full(8, 3, np.int32), in_idx, np.full(8, tb, np.uint64), np.arange(56, 64))
        dec = ck.decrypt(w.download(np.arange(56, 64)))
        assert list(dec) == [int(fn(*r)) for r in bits.tolist()]
    sk.close()


def test_full_parameter_set_m1c1():
    """PARAM_MESSAGE_1_CARRY_1_KS_PBS, the set the reference BINARY installs for LUT mode (src/bin/helm.rs:301:
    n = 684, k = 3, N = 512, one level of 18 bits [dimensions recalled]): apply_lookup_table rows bit for bit against the
    O(N^2) oracle, every plaintext value through a LUT, and the bivariate form gates::lut() uses for 2-input gates
    (src/gates.rs:761-764)."""
    ck = helm_amd.SiClientKey.generate("shortint_m1c1", seed=1)
    assert (ck.params.n, ck.params.k, ck.params.N, ck.t) == (684, 3, 512, 4)
    sk = helm_amd.SiServerKey(ck)
    assert sk.field_bits() == 46   # round 6: the generated key's exact products fit the 46-bit CRT pair
    orc = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk)
    vals = np.arange(ck.t, dtype=np.uint64)
    cts = ck.encrypt(vals)
    w = sk.wires(3 * ck.t + 16)
    w.upload(np.arange(ck.t), cts)
    lut = sk.make_lut(lambda x: (3 * x + 1) % ck.t)
    assert np.array_equal(lut, orc.make_lut(lambda x: (3 * x + 1) % ck.t))
    w.apply_luts(np.arange(ck.t), lut, np.arange(ck.t) + ck.t)
    got = w.download(np.arange(ck.t) + ck.t)
    assert list(ck.decrypt_message_and_carry(got)) == [(3 * v + 1) % ck.t for v in range(ck.t)]
    for g in (0, 3):
        assert np.array_equal(got[g], orc.apply_lut(cts[g], lut)), f"row {g} of the batch differs from the oracle"
    w.apply_luts([2], lut, [2 * ck.t])  # a batch of one
    assert np.array_equal(w.download([2 * ck.t])[0], got[2])
    # all four input pairs of XOR / AND / OR as 2-input LUT gates, one level, against the oracle and the truth tables
    base = 3 * ck.t
    bits = np.array([[a, b] for a in (0, 1) for b in (0, 1)], dtype=np.uint64)
    host = np.zeros((base + 16, ck.dim + 1), dtype=np.uint64)
    host[base:base + 8] = ck.encrypt(bits.reshape(-1))
    w.upload(np.arange(base, base + 8), host[base:base + 8])
    in_idx = np.arange(base, base + 8, dtype=np.int32).reshape(4, 2)
    for tb, fn in ((0x6, lambda a, b: a ^ b), (0x8, lambda a, b: a & b), (0xE, lambda a, b: a | b)):
        out = np.arange(base + 8, base + 12, dtype=np.int32)
        w.eval_lut_level(np.full(4, 2, np.int32), in_idx, np.full(4, tb, np.uint64), out)
        orc.eval_lut_level(host, np.full(4, 2, np.int32), in_idx, np.full(4, tb, np.uint64), out)
        g = w.download(out)
        assert np.array_equal(g, host[out]), hex(tb)
        assert list(ck.decrypt(g)) == [fn(int(a), int(b)) for a, b in bits.tolist()]
    sk.close()


def _pbs_rows_bit_exact(ck, sk, orc, seed, small):
    luts = np.stack([orc.make_lut(lambda x: (5 * x + 3) % ck.t), orc.make_lut(lambda x: x & 1)])
    idx = ((np.arange(ck.t) + seed) % 2).astype(np.int32)
    got = sk.pbs_batch(small, luts, idx)
    for g in range(ck.t):
        assert np.array_equal(got[g], orc.bootstrap(small[g], luts[idx[g]])), f"ciphertext {g}"
    return got


def test_k_pbs64k_crt_pair_follows_the_loaded_key(monkeypatch):
    """Round 6: k > 1 contexts (k_pbs64k, the kernel of reference src/bin/helm.rs:301's set) compute in the 46-bit CRT pair
    2736^4 + 1, 2872^4 + 1 when B/2 x the largest l1-norm of a column of the LOADED key stays below p p' / 2 (an exact
    guarantee for that key and every input), in the 49-bit pair otherwise: a generated key fits, a key with every coefficient
    at the largest magnitude does not, HELM_SI_FIELD=49 keeps the 49-bit pair.  Bit-exact against the oracle under each; the
    same ciphertexts from either pair; loading another key into the same context moves the pair back and forth; a set with
    20-bit digits (si_toy_512_k2) never takes the 46-bit pair (digit x b^3 must stay an exact double)."""
    from helm_amd._native import hip, hip_check, as_u64p
    ck = helm_amd.SiClientKey.generate("si_toy_512_k3", seed=5)
    sk = helm_amd.SiServerKey(ck)
    assert sk.field_bits() == 46
    small = sk.keyswitch_batch(ck.encrypt(np.arange(ck.t, dtype=np.uint64)))   # the same inputs for every key and pair
    got46 = _pbs_rows_bit_exact(ck, sk, oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk), 1, small)
    worst = np.full_like(ck.bsk, 0x7FFFFFFFFFFFFFFF)
    hip_check(hip.helm_si_load_bootstrap_key(sk._h, as_u64p(worst), worst.size))   # the same context: the tables follow the key
    assert sk.field_bits() == 49
    _pbs_rows_bit_exact(ck, sk, oracle.Oracle64(ck.params.as_tuple(), worst, ck.ksk), 2, small)
    own = np.ascontiguousarray(ck.bsk, dtype=np.uint64).reshape(-1)
    hip_check(hip.helm_si_load_bootstrap_key(sk._h, as_u64p(own), own.size))
    assert sk.field_bits() == 46
    assert np.array_equal(_pbs_rows_bit_exact(ck, sk, oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk), 1, small), got46)
    sk.close()
    monkeypatch.setenv("HELM_SI_FIELD", "49")
    sk49 = helm_amd.SiServerKey(ck)
    assert sk49.field_bits() == 49
    assert np.array_equal(_pbs_rows_bit_exact(ck, sk49, oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk), 1, small), got46)
    sk49.close()
    monkeypatch.delenv("HELM_SI_FIELD")
    ck2 = helm_amd.SiClientKey.generate("si_toy_512_k2", seed=5)
    sk2 = helm_amd.SiServerKey(ck2)
    assert sk2.field_bits() == 49 and ck2.params.pbs_logB == 20
    sk2.close()


def test_full_parameter_set_multibit3():
    """The reference's arithmetic-mode set (helm.rs:83, PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS: n = 888,
    g = 3, 296 group steps, 310 MB key) at full size: apply_lookup_table rows (keyswitch + multi-bit blind rotation
    + sample extract) bit for bit against the integer oracle, one from a batch of one and two from a batch that
    fills workgroups differently; every plaintext value through a LUT after decryption."""
    ck = helm_amd.SiClientKey.generate("shortint_m2c2_multibit3", seed=1)
    assert ck.params.grouping_factor == 3 and ck.params.n == 888
    sk = helm_amd.SiServerKey(ck)
    orc = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk)
    vals = np.arange(ck.t, dtype=np.uint64)
    cts = ck.encrypt(vals)
    w = sk.wires(3 * ck.t)
    w.upload(np.arange(ck.t), cts)
    lut = sk.make_lut(lambda x: (3 * x + 1) % ck.t)
    assert np.array_equal(lut, orc.make_lut(lambda x: (3 * x + 1) % ck.t))
    w.apply_luts(np.arange(ck.t), lut, np.arange(ck.t) + ck.t)
    got = w.download(np.arange(ck.t) + ck.t)
    assert list(ck.decrypt_message_and_carry(got)) == [(3 * v + 1) % ck.t for v in range(ck.t)]
    for g in (0, 11):
        assert np.array_equal(got[g], orc.apply_lut(cts[g], lut)), f"row {g} of the batch differs from the oracle"
    w.apply_luts([6], lut, [2 * ck.t])  # a batch of one
    assert np.array_equal(w.download([2 * ck.t])[0], orc.apply_lut(cts[6], lut))
    assert np.array_equal(w.download([2 * ck.t])[0], got[6])  # and the same ciphertext as inside the batch of 16
    sk.close()


@pytest.mark.parametrize("fixture", ["shortint_toy.npz", "shortint_mb_toy.npz"])  # classical / multi-bit blind rotation
def test_golden_vectors_on_gpu(fixture):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture))
    p = helm_amd.SiParams(*[int(x) for x in g["params"]])
    sk = helm_amd.SiServerKey(params=p, bsk=g["bsk"], ksk=g["ksk"])
    n_in = len(g["inputs"])
    w = sk.wires(n_in + len(g["arity"]))
    w.upload(np.arange(n_in), g["inputs"])
    out_idx = np.arange(n_in, n_in + len(g["arity"]), dtype=np.int32)
    w.eval_lut_level(g["arity"], g["in_idx"], g["table"], out_idx)
    assert np.array_equal(w.download(out_idx), g["expected"])
    sk.close()


@pytest.mark.parametrize("name", ["si_toy_512", "si_toy_2048", "si_toy_2048_l2", "si_toy_2048_mb3"])  # ks_l = 3, 4, 5
def test_matrix_core_keyswitch_bit_exact(name, monkeypatch):
    """Batches of 160 ciphertexts or more keyswitch on the matrix cores (eight int8 GEMMs over the key's byte planes,
    the level count padded to a power of two); narrower ones and HELM_HIP_KS_MFMA=0 use the vector-ALU kernel.  Both
    give the oracle's words on a batch that is wide enough for the matrix-core path and not a multiple of its tiles."""
    ck = helm_amd.SiClientKey.generate(name, seed=9)
    orc = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk)
    sk = helm_amd.SiServerKey(ck)
    monkeypatch.setenv("HELM_HIP_KS_MFMA", "0")
    sk_valu = helm_amd.SiServerKey(ck)
    rng = np.random.default_rng(4)
    count = 203
    cts = ck.encrypt(rng.integers(0, ck.t, count).astype(np.uint64))
    cts[1] = 0
    cts[2] = rng.integers(0, 2**64, size=cts.shape[1], dtype=np.uint64)  # not a ciphertext at all: every digit pattern
    got = sk.keyswitch_batch(cts)
    assert np.array_equal(got, sk_valu.keyswitch_batch(cts))
    for g in (0, 1, 2, 63, 64, 159, 160, count - 1):
        assert np.array_equal(got[g], orc.keyswitch(cts[g])), (name, g)
    narrow = sk.keyswitch_batch(cts[:7])  # the vector-ALU path of the same context
    assert np.array_equal(narrow, got[:7])
    sk.close()
    sk_valu.close()
