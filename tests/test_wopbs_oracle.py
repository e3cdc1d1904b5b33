"""CPU tests of the WoP-PBS oracle (oracle/wopbs_oracle.c) and of the client key material it runs on
(helm_amd/csrc/helm_client_wop.cpp): the two exact routes agree bit for bit, the bootstrap agrees with the
shortint oracle, and every stage decrypts to what the published algorithm says (the reference has no test or
fixture for this path: src/gates.rs:721-742, 787-864 are never called)."""
import numpy as np
import pytest

import helm_amd
import oracle
from helm_amd import wopbs

U64 = np.uint64
pytestmark = pytest.mark.filterwarnings("ignore:overflow encountered")


def negacyclic_mul_bits(a, s_bits):
    """a * s in Z[X]/(X^N+1) mod 2^64, s binary (numpy, O(N^2 / 64))"""
    N = a.size
    out = np.zeros(N, dtype=U64)
    for u in np.nonzero(s_bits)[0]:
        out[u:] += a[: N - u]
        out[:u] -= a[N - u:]
    return out


def glwe_phase(glwe, sk_bits, k, N):
    """body - sum A_c S_c"""
    ph = glwe[k * N:(k + 1) * N].copy()
    for c in range(k):
        ph -= negacyclic_mul_bits(glwe[c * N:(c + 1) * N], sk_bits[c * N:(c + 1) * N])
    return ph


def signed(x):
    return np.asarray(x, dtype=U64).astype(np.int64)


def toy(pbs_name="si_toy_512", wop_name="wop_toy_512", seed=7, moduli=None):
    """moduli = (message_modulus, carry_modulus) overrides both sets (they must share one encoding)"""
    from helm_amd.shortint import si_named_params
    sp, a, b = si_named_params(pbs_name)
    wp, c, d = wopbs.wop_named_params(wop_name)
    if moduli is not None:
        sp.message_modulus, sp.carry_modulus = moduli
        wp.message_modulus, wp.carry_modulus = moduli
    ck = helm_amd.SiClientKey(sp, a, b, seed=seed)
    wk = wopbs.WopClientKey(ck, wp, c, d, seed=seed + 1)
    return ck, wk


def encrypt_wop_big(wk, values, rng):
    """value * delta under the WoP-side big key, noiseless mask-only randomness (numpy)"""
    sk = wk.glwe_secret.astype(bool)
    cts = rng.integers(0, 1 << 64, size=(len(values), wk.dim + 1), dtype=U64)
    body = (cts[:, :-1] * sk).sum(axis=1, dtype=U64) + np.array(values, dtype=U64) * U64(wk.delta)
    cts[:, -1] = body
    return cts


@pytest.mark.parametrize("N,k,l,logB,part", [(512, 1, 2, 15, 32), (256, 1, 3, 5, 32), (512, 1, 1, 23, 16), (256, 2, 2, 10, 32)])
def test_external_product_routes_agree(N, k, l, logB, part):
    rng = np.random.default_rng(N + l)
    k1 = k + 1
    std = rng.integers(0, 1 << 64, size=2 * l * k1 * k1 * N, dtype=U64)
    diff = rng.integers(0, 1 << 64, size=k1 * N, dtype=U64)
    a = oracle.Ggsw(N, k, l, logB, std, use_ntt=True)
    b = oracle.Ggsw(N, k, l, logB, std, use_ntt=False)
    assert a.part_bits() == part and b.part_bits() == 0
    for i in range(2):
        acc_a = rng.integers(0, 1 << 64, size=k1 * N, dtype=U64)
        acc_b = acc_a.copy()
        a.extprod_add(i, diff, acc_a)
        b.extprod_add(i, diff, acc_b)
        assert np.array_equal(acc_a, acc_b)


def test_bootstrap_agrees_with_the_shortint_oracle():
    ck = helm_amd.SiClientKey.generate("si_toy_512", seed=3)
    p = ck.params
    o64 = oracle.Oracle64(p.as_tuple(), ck.bsk, ck.ksk)
    lut = o64.make_lut(lambda x: (3 * x + 1) % ck.t)
    small = o64.keyswitch(ck.encrypt(np.array([5], dtype=U64))[0])
    want = o64.bootstrap(small, lut)
    for use_ntt in (True, False):
        g = oracle.Ggsw(p.N, p.k, p.pbs_l, p.pbs_logB, ck.bsk, use_ntt=use_ntt)
        assert np.array_equal(g.bootstrap(small, lut), want)


def test_keyswitch_and_packing_keyswitch_decrypt():
    ck, wk = toy()
    P = wk.params
    rng = np.random.default_rng(1)
    ow = oracle.OracleW(P.as_tuple(), wk.bsk, wk.ksk, wk.pfpksk)
    ct = encrypt_wop_big(wk, [5], rng)[0]
    small = ow.keyswitch(wk.ksk, wk.dim, P.n, P.ks_l, P.ks_logB, ct)
    assert abs(int(signed(wk.phase(small, small=True) - U64(5 * wk.delta))[0])) < 1 << 52
    # pfpksk: key k (identity) packs the phase into coefficient 0, key r < k multiplies it by -S_r
    glwe_words = (P.k + 1) * P.N
    key_words = (wk.dim + 1) * P.pfks_l * glwe_words
    sk = wk.glwe_secret
    for r in range(P.k + 1):
        out = np.zeros(glwe_words, dtype=U64)
        oracle.libw().orcw_pfpks(wk.dim, glwe_words, P.pfks_l, P.pfks_logB,
                                 oracle._u64(np.ascontiguousarray(wk.pfpksk[r * key_words:(r + 1) * key_words])),
                                 oracle._u64(ct), oracle._u64(out))
        ph = glwe_phase(out, sk, P.k, P.N)
        want = np.zeros(P.N, dtype=U64)
        if r == P.k:
            want[0] = U64(5 * wk.delta)
        else:
            want = U64(0) - sk[r * P.N:(r + 1) * P.N] * U64(5 * wk.delta)
        assert np.abs(signed(ph - want)).max() < 1 << 48


def test_extract_bits_decrypt():
    ck, wk = toy()
    P = wk.params
    ow = oracle.OracleW(P.as_tuple(), wk.bsk, wk.ksk, wk.pfpksk)
    rng = np.random.default_rng(2)
    nb = 4
    for v in (0b1011, 0b0110, 0b1111, 0):
        ct = encrypt_wop_big(wk, [v], rng)[0]
        bits = ow.extract_bits(ct, wk.delta_log, nb)
        ph = wk.phase(bits.reshape(nb, -1), small=True)
        got = [int((int(x) + (1 << 62)) >> 63) & 1 for x in ph]  # row 0 = most significant
        assert got == [(v >> (nb - 1 - i)) & 1 for i in range(nb)]
        assert np.abs(signed(ph - (np.array(got, dtype=U64) << U64(63)))).max() < 1 << 58


def test_circuit_bootstrap_makes_a_ggsw_of_the_bit():
    ck, wk = toy()
    P = wk.params
    ow = oracle.OracleW(P.as_tuple(), wk.bsk, wk.ksk, wk.pfpksk)
    sk = wk.glwe_secret
    lwe_sk = wk.lwe_secret.astype(bool)
    rng = np.random.default_rng(3)
    for bit in (0, 1):
        a = rng.integers(0, 1 << 64, size=P.n + 1, dtype=U64)
        a[-1] = (a[:-1] * lwe_sk).sum(dtype=U64) + U64(bit << 63) + U64(12345)
        ggsw = ow.circuit_bootstrap(a)
        for j in range(P.cbs_l):
            g = U64(bit << (64 - P.cbs_logB * (j + 1)))
            for r in range(P.k + 1):
                ph = glwe_phase(ggsw[j, r], sk, P.k, P.N)
                want = np.zeros(P.N, dtype=U64)
                if r == P.k:
                    want[0] = g
                else:
                    want = U64(0) - sk[r * P.N:(r + 1) * P.N] * g
                assert np.abs(signed(ph - want)).max() < 1 << 44, (bit, j, r)


@pytest.mark.parametrize("bits", [3, 9, 11])
def test_vertical_packing_selects_the_entry(bits):
    """bits <= log2 N: blind rotation only; above: a CMUX tree first.  The GGSWs are made by the client recipe
    (fresh encryptions of the bits), so this checks the packing alone."""
    ck, wk = toy()
    P = wk.params
    ow = oracle.OracleW(P.as_tuple(), wk.bsk, wk.ksk, wk.pfpksk)
    rng = np.random.default_rng(bits)
    lwe_sk = wk.lwe_secret.astype(bool)
    table = rng.integers(0, ck.t, size=max(1 << bits, P.N), dtype=U64) * U64(wk.delta)
    table[(1 << bits):] = 0
    for v in (0, (1 << bits) - 1, int(rng.integers(0, 1 << bits))):
        ggsws = []
        for i in range(bits):  # most significant first
            bit = (v >> (bits - 1 - i)) & 1
            a = rng.integers(0, 1 << 64, size=P.n + 1, dtype=U64)
            a[-1] = (a[:-1] * lwe_sk).sum(dtype=U64) + U64(bit << 63)
            ggsws.append(ow.circuit_bootstrap(a))
        out = ow.vertical_packing(np.stack(ggsws), table)
        ph = wk.phase(out)[0]
        assert abs(int(signed(ph - table[v]))) < 1 << 56
        if bits == 3 and v == 0:  # the schoolbook route computes the same ciphertext
            slow = oracle.OracleW(P.as_tuple(), wk.bsk, wk.ksk, wk.pfpksk, use_ntt=False)
            assert np.array_equal(slow.vertical_packing(np.stack(ggsws), table), out)


@pytest.mark.parametrize("bits_per_block", [1, 2])
def test_wide_lut_end_to_end_decrypts_to_the_truth_table(bits_per_block):
    ck, wk = toy(moduli=(2, 2))  # the reference's LUT-mode encoding (helm.rs:301: PARAM_MESSAGE_1_CARRY_1)
    P = wk.params
    o64 = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk)
    ow = oracle.OracleW(P.as_tuple(), wk.bsk, wk.ksk, wk.pfpksk, pbs=o64, ksk_to_wopbs=wk.ksk_to_wopbs,
                        ksk_to_pbs=wk.ksk_to_pbs)
    m = 5
    truth = np.array([bin(x * 0x2D).count("1") & 1 for x in range(1 << m)], dtype=U64)
    # the product's table generator agrees with the restatement
    assert np.array_equal(wopbs.make_table(P, m, bits_per_block, truth), ow.make_table(m, bits_per_block, truth))
    for x in (0b10110, 0b01001):
        bits_in = [(x >> (m - 1 - q)) & 1 for q in range(m)]  # first input = most significant
        cts = ck.encrypt(np.array(bits_in, dtype=U64))
        out = ow.wide_lut(cts, truth, bits_per_block)
        assert int(ck.decrypt_message_and_carry(out[None, :])[0]) == int(truth[x])


def test_make_table_follows_the_reference_index_rule():
    p, _, _ = wopbs.wop_named_params("wopbs_m1c1")
    truth = np.array([0, 1, 1, 0, 1, 0, 0, 1], dtype=U64)  # 3-input parity
    t1 = wopbs.make_table(p, 3, 1, truth)
    delta = (1 << 63) // 4
    assert t1.size == p.N and list(t1[:8] // U64(delta)) == list(truth) and not t1[8:].any()
    # two bits per block (what tfhe extracts after the cleaning bootstrap): carries add into the next block
    t2 = wopbs.make_table(p, 3, 2, truth)
    for v in range(64):
        fields = [(v >> (2 * j)) & 3 for j in range(3)]
        x = (fields[0] + 2 * fields[1] + 4 * fields[2]) % 8
        assert int(t2[v]) == int(truth[x]) * delta


def test_parameter_and_table_validation_needs_no_device():
    import ctypes as C
    from helm_amd import _native as nv
    p, lwe, glwe = wopbs.wop_named_params("wopbs_m2c2")
    assert p.as_tuple() == (769, 1, 2048, 2, 15, 5, 2, 2, 15, 3, 5, 4, 4)
    p1, _, _ = wopbs.wop_named_params("wopbs_m1c1")
    assert (p1.n, p1.message_modulus, p1.carry_modulus) == (653, 2, 2)
    with pytest.raises(nv.HelmError):
        wopbs.wop_named_params("no_such_set")
    assert nv.hip.helm_wop_table_words(C.byref(p), 3) == 2048 and nv.hip.helm_wop_table_words(C.byref(p), 13) == 8192
    out = np.zeros(2048, dtype=U64)
    truth = np.zeros(8, dtype=U64)
    assert nv.hip.helm_wop_make_table(C.byref(p), 0, 1, nv.as_u64p(truth), 8, nv.as_u64p(out)) != 0
    assert nv.hip.helm_wop_make_table(C.byref(p), 3, 0, nv.as_u64p(truth), 8, nv.as_u64p(out)) != 0
    assert nv.hip.helm_wop_make_table(C.byref(p), 3, 1, None, 8, nv.as_u64p(out)) != 0
    # no device here: creating the context fails loudly (no CPU fallback), with a null PBS side as an invalid argument
    h = nv.vp()
    assert nv.hip.helm_wop_ctx_create(None, C.byref(p), C.byref(h)) == -1
    # the two parameter sets must share one encoding
    ck = helm_amd.SiClientKey.generate("si_toy_512", seed=5)  # message_modulus * carry_modulus = 16
    with pytest.raises(nv.HelmError):
        wopbs.WopClientKey(ck, p1, lwe, glwe, seed=6)
