"""GPU parity of the WoP-PBS wide-LUT path (include/helm_wopbs.h) against oracle/wopbs_oracle.c: every stage and the
whole of high_precision_lut() (reference src/gates.rs:787-815) bit for bit on the same keys and inputs, and at the full
parameter sets the decrypted result against the truth table.  All calls go through the C ABI."""
import numpy as np
import pytest

import helm_amd
import oracle
from helm_amd import wopbs
from helm_amd.shortint import si_named_params

pytestmark = [pytest.mark.gpu, pytest.mark.filterwarnings("ignore:overflow encountered")]
U64 = np.uint64


def make_keys(pbs_name, wop_name, seed, moduli=None):
    sp, a, b = si_named_params(pbs_name)
    wp, c, d = wopbs.wop_named_params(wop_name)
    if moduli is not None:
        sp.message_modulus, sp.carry_modulus = moduli
        wp.message_modulus, wp.carry_modulus = moduli
    ck = helm_amd.SiClientKey(sp, a, b, seed=seed)
    wk = wopbs.WopClientKey(ck, wp, c, d, seed=seed + 1)
    sk = helm_amd.SiServerKey(ck)
    wsk = wopbs.WopServerKey(sk, wk)
    o64 = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk)
    ow = oracle.OracleW(wp.as_tuple(), wk.bsk, wk.ksk, wk.pfpksk, pbs=o64, ksk_to_wopbs=wk.ksk_to_wopbs,
                        ksk_to_pbs=wk.ksk_to_pbs)
    return ck, wk, sk, wsk, ow


def encrypt_wop_big(wk, values, rng):
    sk = wk.glwe_secret.astype(bool)
    cts = rng.integers(0, 1 << 64, size=(len(values), wk.dim + 1), dtype=U64)
    cts[:, -1] = (cts[:, :-1] * sk).sum(axis=1, dtype=U64) + np.array(values, dtype=U64) * U64(wk.delta)
    return cts


def encrypt_bits_small(wk, bits, rng):
    sk = wk.lwe_secret.astype(bool)
    cts = rng.integers(0, 1 << 64, size=(len(bits), wk.params.n + 1), dtype=U64)
    cts[:, -1] = (cts[:, :-1] * sk).sum(axis=1, dtype=U64) + (np.array(bits, dtype=U64) << U64(63)) + U64(999)
    return cts


@pytest.fixture(scope="module", params=[("si_toy_512", "wop_toy_512"), ("si_toy_1024", "wop_toy_1024")])
def toy(request):
    keys = make_keys(*request.param, seed=11)
    yield keys
    keys[3].close()
    keys[2].close()


def test_extract_bits_bit_exact(toy):
    ck, wk, sk, wsk, ow = toy
    rng = np.random.default_rng(1)
    vals = [0b1011, 0b0110, 0b1111, 0, 0b1000, 0b0001, 0b0101]
    cts = encrypt_wop_big(wk, vals, rng)
    for nb in (1, 4):
        got = wsk.extract_bits(cts, wk.delta_log, nb)  # [row][bit, least significant first][n+1]
        for r in range(len(vals)):
            want = ow.extract_bits(cts[r], wk.delta_log, nb)[::-1]  # the oracle lists the most significant first
            assert np.array_equal(got[r], want), (nb, r)
        ph = wk.phase(got.reshape(-1, wk.params.n + 1), small=True)
        dec = ((ph + U64(1 << 62)) >> U64(63)).reshape(len(vals), nb)
        assert [[int(b) for b in row] for row in dec] == [[(v >> i) & 1 for i in range(nb)] for v in vals]


def test_circuit_bootstrap_bit_exact(toy):
    ck, wk, sk, wsk, ow = toy
    rng = np.random.default_rng(2)
    bits = [0, 1, 1, 0, 1]
    small = encrypt_bits_small(wk, bits, rng)
    got = wsk.circuit_bootstrap(small)
    for r in range(len(bits)):
        assert np.array_equal(got[r], ow.circuit_bootstrap(small[r])), r
    # a batch wide enough for the matrix-core packing keyswitch (64 bootstraps and more): same ciphertexts
    many = encrypt_bits_small(wk, list(rng.integers(0, 2, size=40)), rng)
    wide = wsk.circuit_bootstrap(many)
    for r in (0, 17, 39):
        assert np.array_equal(wide[r], ow.circuit_bootstrap(many[r])), r
    assert np.array_equal(wsk.circuit_bootstrap(many[:5]), wide[:5])  # narrow batch: the vector-ALU kernel


@pytest.mark.parametrize("bits", [2, 7, 11, 12])
def test_vertical_packing_bit_exact(toy, bits):
    """bits <= log2 N: blind rotation only; above: a CMUX tree (two levels deep at N = 512 / 1024 for 11 / 12 bits)"""
    ck, wk, sk, wsk, ow = toy
    rng = np.random.default_rng(bits)
    P = wk.params
    count = 3
    values = [0, (1 << bits) - 1, int(rng.integers(0, 1 << bits))]
    words = max(1 << bits, P.N)
    tables = rng.integers(0, ck.t, size=(count, words), dtype=U64) * U64(wk.delta)
    tables[:, (1 << bits):] = 0
    flat_bits = [(v >> i) & 1 for v in values for i in range(bits)]  # least significant first
    ggsw = wsk.circuit_bootstrap(encrypt_bits_small(wk, flat_bits, rng))
    ggsw = ggsw.reshape(count, bits, *ggsw.shape[1:])
    got = wsk.vertical_packing(ggsw, tables)
    for g in range(count):
        want = ow.vertical_packing(ggsw[g][::-1], tables[g])  # the oracle takes the most significant first
        assert np.array_equal(got[g], want), g
        assert abs(int((wk.phase(got[g])[0] - tables[g, values[g]]).astype(np.int64))) < 1 << 56


@pytest.mark.parametrize("n_inputs,bits_per_block", [(3, 1), (5, 2), (3, 4)])
def test_wide_lut_gates_bit_exact(toy, n_inputs, bits_per_block):
    """helm_wop_eval_luts = high_precision_lut() per gate: ciphertexts equal the oracle's, and decrypt to the table
    entry the reference's index rule selects (basis message_modulus = 4 here: sum bit_j 4^j, src/gates.rs:845-848)."""
    ck, wk, sk, wsk, ow = toy
    rng = np.random.default_rng(n_inputs * 10 + bits_per_block)
    count = 4
    truth = rng.integers(0, 2, size=(count, 4 ** n_inputs), dtype=U64)
    xs = rng.integers(0, 1 << n_inputs, size=count)
    bits_in = np.array([[(x >> (n_inputs - 1 - q)) & 1 for q in range(n_inputs)] for x in xs], dtype=U64)
    w = sk.wires(count * (n_inputs + 1))
    cts = ck.encrypt(bits_in.reshape(-1))
    w.upload(np.arange(count * n_inputs), cts)
    in_idx = np.arange(count * n_inputs, dtype=np.int32).reshape(count, n_inputs)
    out_idx = np.arange(count * n_inputs, count * (n_inputs + 1), dtype=np.int32)
    wsk.eval_luts(w, in_idx, truth, out_idx, bits_per_block=bits_per_block)
    got = w.download(out_idx)
    for g in range(count):
        idx = sum(int(b) * 4 ** j for j, b in enumerate(bits_in[g][::-1]))
        assert int(ck.decrypt_message_and_carry(got[g:g + 1])[0]) == int(truth[g, idx])
        if g < 2:
            want = ow.wide_lut(cts[g * n_inputs:(g + 1) * n_inputs], truth[g], bits_per_block)
            assert np.array_equal(got[g], want), g


def test_reference_encoding_two_bits_per_block():
    """The encoding the reference's LUT mode names (helm.rs:301: message_modulus = carry_modulus = 2): six-input
    gates, two bits per block as tfhe extracts after the cleaning bootstrap = 12 index bits = a CMUX tree over
    eight polynomials at N = 512; first input = most significant (gates.rs:795-799)."""
    ck, wk, sk, wsk, ow = make_keys("si_toy_512", "wop_toy_512", seed=21, moduli=(2, 2))
    rng = np.random.default_rng(5)
    m, count = 6, 8
    truth = rng.integers(0, 2, size=1 << m, dtype=U64)
    xs = rng.integers(0, 1 << m, size=count)
    bits_in = np.array([[(x >> (m - 1 - q)) & 1 for q in range(m)] for x in xs], dtype=U64)
    w = sk.wires(count * (m + 1))
    cts = ck.encrypt(bits_in.reshape(-1))
    w.upload(np.arange(count * m), cts)
    in_idx = np.arange(count * m, dtype=np.int32).reshape(count, m)
    out_idx = np.arange(count * m, count * (m + 1), dtype=np.int32)
    wsk.eval_luts(w, in_idx, truth, out_idx)  # default: log2(message_modulus * carry_modulus) = 2 bits per block
    got = w.download(out_idx)
    assert [int(v) for v in ck.decrypt_message_and_carry(got)] == [int(truth[x]) for x in xs]
    assert np.array_equal(got[0], ow.wide_lut(cts[:m], truth, 2))
    t = wsk.timing()
    assert t["gates"] == count and t["bootstraps"] == count * (m + m + 2 * m * wk.params.cbs_l + 1)
    wsk.close()
    sk.close()


def test_full_parameter_sets_m2c2():
    """PARAM_MESSAGE_2_CARRY_2_KS_PBS beside WOPBS_PARAM_MESSAGE_2_CARRY_2_KS_PBS [dimensions recalled] at full size:
    a circuit bootstrap and two whole gates bit for bit against the oracle, and batches of wide gates - up to eight
    inputs, one to four bits per block, with and without a CMUX tree - decrypted against their truth tables."""
    ck, wk, sk, wsk, ow = make_keys("shortint_m2c2", "wopbs_m2c2", seed=31)
    P = wk.params
    assert (P.n, P.N, P.pbs_l, P.cbs_l, P.pfks_l) == (769, 2048, 2, 3, 2)
    rng = np.random.default_rng(9)
    small = encrypt_bits_small(wk, [1, 0], rng)
    got = wsk.circuit_bootstrap(small)
    assert np.array_equal(got[0], ow.circuit_bootstrap(small[0]))
    for n_inputs, bits_per_block, count, check in ((3, 1, 8, True), (2, 2, 4, True), (8, 1, 32, False), (3, 4, 6, False),
                                                   (6, 2, 16, False)):
        truth = rng.integers(0, 2, size=(count, 4 ** n_inputs), dtype=U64)
        xs = rng.integers(0, 1 << n_inputs, size=count)
        bits_in = np.array([[(x >> (n_inputs - 1 - q)) & 1 for q in range(n_inputs)] for x in xs], dtype=U64)
        w = sk.wires(count * (n_inputs + 1))
        cts = ck.encrypt(bits_in.reshape(-1))
        w.upload(np.arange(count * n_inputs), cts)
        in_idx = np.arange(count * n_inputs, dtype=np.int32).reshape(count, n_inputs)
        out_idx = np.arange(count * n_inputs, count * (n_inputs + 1), dtype=np.int32)
        wsk.eval_luts(w, in_idx, truth, out_idx, bits_per_block=bits_per_block)
        got = w.download(out_idx)
        want = [int(truth[g, sum(int(b) * 4 ** j for j, b in enumerate(bits_in[g][::-1]))]) for g in range(count)]
        assert [int(v) for v in ck.decrypt_message_and_carry(got)] == want, (n_inputs, bits_per_block)
        if check:
            assert np.array_equal(got[1], ow.wide_lut(cts[n_inputs:2 * n_inputs], truth[1], bits_per_block))
        w.free()
    wsk.close()
    sk.close()


def test_lut_circuit_routes_wide_gates_through_wopbs():
    """LutCircuit with a wide-LUT key: gates whose index does not fit one block (here: more than two inputs under
    message_modulus = carry_modulus = 2, the reference's LUT-mode encoding) go through the WoP-PBS path -
    Gate::evaluate_encrypted_high_precision_lut (gates.rs:721-742) - and every wire equals the plaintext evaluation,
    as the reference's LUT test checks for the narrow path (circuit_test.rs:308-310)."""
    from helm_amd import Circuit, LutCircuit, PtxtType, verilog_parser
    ck, wk, sk, wsk, ow = make_keys("si_toy_512", "wop_toy_512", seed=41, moduli=(2, 2))
    text = """input a, b, c, d, e, f;
output y, z, p, q;
lut g0(0x6996966996696996, a, b, c, d, e, f, y);
lut g1(0xFEE8E880, a, b, c, d, e, t);
lut g2(0x6, t, f, z);
lut g3(0x96, y, z, t, p);
lut g4(0x8000000000000001, y, z, t, p, a, b, q);
"""
    gates_set, wire_set, input_wires, output_wires, dffs, _, _ = verilog_parser.read_verilog_text(text, False)
    circuit = Circuit(gates_set, input_wires, output_wires, dffs)
    circuit.sort_circuit()
    circuit.compute_levels()
    lc = LutCircuit(ck, sk, circuit)
    lc.set_wide_lut_key(wsk, bits_per_block=1)
    for x in (0b101101, 0b000000, 0b111111, 0b010011):
        inputs = {n: PtxtType.Bool((x >> (5 - i)) & 1) for i, n in enumerate("abcdef")}
        ptxt = circuit.evaluate(circuit.initialize_wire_map(wire_set, inputs, "bool"))
        enc = lc.evaluate_encrypted(lc.encrypt_inputs(wire_set, inputs), 1, "bool")
        for wire, want in ptxt.items():
            assert ck.decrypt(enc[wire]) == int(bool(want)), (x, wire)
    # g2 is the one narrow gate (1 bootstrap); the four wide ones: m cleaning + m * cbs_l circuit + 1 final each
    L = wk.params.cbs_l
    assert lc.pbs_per_cycle() == 1 + sum(m + m * L + 1 for m in (6, 5, 3, 6))
    # without the key the wide gates fall to gates::lut()'s single-block packing, which cannot hold their index:
    # the engine refuses instead of returning a wrong ciphertext
    from helm_amd._host import Panic
    lc.set_wide_lut_key(None)
    with pytest.raises(Panic, match="do not fit the plaintext space"):
        lc.evaluate_encrypted(lc.encrypt_inputs(wire_set, {n: PtxtType.Bool(1) for n in "abcdef"}), 1, "bool")
    wsk.close()
    sk.close()


def test_argument_validation_and_edge_cases(toy):
    """Empty batches are no-ops; a one-input gate works (one block, one GGSW); missing keys, foreign wire tables, too
    many index bits and bad rows fail loudly with HELM_ERR_* (no silent fallback)."""
    from helm_amd import _native as nv
    ck, wk, sk, wsk, ow = toy
    w = sk.wires(8)
    w.upload(np.arange(2), ck.encrypt(np.array([1, 0], dtype=U64)))
    # empty batch
    wsk.eval_luts(w, np.zeros((0, 3), np.int32), np.zeros((0, 64), U64), np.zeros(0, np.int32), bits_per_block=1)
    # one input: NOT through the wide path
    wsk.eval_luts(w, np.array([[0], [1]], np.int32), np.array([1, 0], dtype=U64), np.array([2, 3], np.int32),
                  bits_per_block=1)
    assert [int(v) for v in ck.decrypt_message_and_carry(w.download(np.array([2, 3])))] == [0, 1]
    # in place: output row = input row of the same gate
    wsk.eval_luts(w, np.array([[2]], np.int32), np.array([1, 0], dtype=U64), np.array([2], np.int32), bits_per_block=1)
    assert int(ck.decrypt_message_and_carry(w.download(np.array([2])))[0]) == 1
    # an output row that another gate of the same call reads: the result would depend on the chunking - refused
    with pytest.raises(nv.HelmError, match="input row of another"):
        wsk.eval_luts(w, np.array([[0], [4]], np.int32), np.array([0, 1], dtype=U64), np.array([4, 5], np.int32), bits_per_block=1)
    with pytest.raises(nv.HelmError, match="out of range"):
        wsk.eval_luts(w, np.array([[0, 9]], np.int32), np.zeros(16, U64), np.array([4], np.int32), bits_per_block=1)
    with pytest.raises(nv.HelmError, match="bits_per_block"):
        wsk.eval_luts(w, np.array([[0, 1]], np.int32), np.zeros(16, U64), np.array([4], np.int32), bits_per_block=5)
    too_many = wsk.logN + 7
    with pytest.raises(nv.HelmError, match="index bits"):
        hip_tables = np.zeros(1 << too_many, U64)
        nv.hip_check(nv.hip.helm_wop_eval_luts(wsk._h, w._h, nv.as_i32p(np.zeros(too_many, np.int32)), too_many, 1,
                                               nv.as_u64p(hip_tables), nv.as_i32p(np.array([4], np.int32)), 1))
    # a table of another context
    other = helm_amd.SiServerKey(ck)
    w2 = other.wires(4)
    with pytest.raises(nv.HelmError, match="another context"):
        wsk.eval_luts(w2, np.array([[0]], np.int32), np.array([0, 1], dtype=U64), np.array([1], np.int32), bits_per_block=1)
    # a context without its keys
    bare = wopbs.WopServerKey(other, params=wk.params)
    with pytest.raises(nv.HelmError, match="not loaded"):
        bare.eval_luts(w2, np.array([[0]], np.int32), np.array([0, 1], dtype=U64), np.array([1], np.int32), bits_per_block=1)
    with pytest.raises(nv.HelmError, match="expected"):
        bare.load_key(wopbs.KEY_PFPKSK, np.zeros(10, U64))
    bare.close()
    other.close()
