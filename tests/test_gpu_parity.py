"""GPU parity: the HIP path (through the C ABI) must be BIT-EXACT with the CPU
oracle on the same keys and inputs, and decrypt to the plaintext truth tables the
reference's own tests pin (reference tests/gates_test.rs:82-107)."""
import numpy as np
import pytest

import helm_amd
import oracle

pytestmark = pytest.mark.gpu

GATES2 = {
    oracle.AND: lambda a, b: a & b,
    oracle.OR: lambda a, b: a | b,
    oracle.NAND: lambda a, b: 1 - (a & b),
    oracle.NOR: lambda a, b: 1 - (a | b),
    oracle.XOR: lambda a, b: a ^ b,
    oracle.XNOR: lambda a, b: 1 - (a ^ b),
}


@pytest.fixture(scope="module", params=["toy", "toy_k2", "toy_1024"])
def small(request):
    ck = helm_amd.ClientKey.generate(request.param, seed=11)
    sk = helm_amd.ServerKey(ck)
    orc = oracle.Oracle(ck.params.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True)
    yield ck, sk, orc
    sk.close()


def test_ntt_roundtrip(small):
    ck, sk, _ = small
    rng = np.random.default_rng(1)
    polys = rng.integers(0, 2**32, size=(64, ck.params.N), dtype=np.uint32)
    polys[0] = 0
    polys[1] = 0x80000000
    polys[2] = 0x7FFFFFFF
    out = sk.ntt_roundtrip(polys)
    assert np.array_equal(out, polys)


def test_pbs_batch_bit_exact(small):
    ck, sk, orc = small
    p = ck.params
    rng = np.random.default_rng(2)
    lwe = rng.integers(0, 2**32, size=(6, p.n + 1), dtype=np.uint32)
    lwe[0] = ck.encrypt(True)
    lwe[1] = ck.encrypt(False)
    lwe[2, :] = 0          # all a_i = 0: every CMUX skipped
    tvs = rng.integers(0, 2**32, size=(2, p.N), dtype=np.uint32)
    tvs[0] = 1 << 29
    idx = np.array([0, 0, 1, 1, 0, 1], dtype=np.int32)
    got = sk.pbs_batch(lwe, tvs, idx)
    for g in range(len(lwe)):
        exp = orc.bootstrap_noks(lwe[g], tvs[idx[g]])
        assert np.array_equal(got[g], exp), f"ciphertext {g} differs"
    ph = ck.phase(got[:2], big=True).astype(np.int64)
    assert abs(ph[0] - (1 << 29)) < (1 << 24) and abs(ph[1] - (7 << 29)) < (1 << 24)


def test_keyswitch_batch_bit_exact(small):
    ck, sk, orc = small
    p = ck.params
    rng = np.random.default_rng(3)
    big = rng.integers(0, 2**32, size=(5, p.k * p.N + 1), dtype=np.uint32)
    big[0] = 0
    big[1] = 0xFFFFFFFF
    got = sk.keyswitch_batch(big)
    for g in range(len(big)):
        assert np.array_equal(got[g], orc.keyswitch(big[g]))


def test_gate_level_bit_exact_and_truth_tables(small):
    ck, sk, orc = small
    p = ck.params
    # wires 0,1 = enc(false), enc(true); 2 = trivial true; outputs from 3
    ct = ck.encrypt([False, True])
    ops, i0, i1, i2, exp_bits = [], [], [], [], []
    for op, f in GATES2.items():
        for a in (0, 1):
            for b in (0, 1):
                ops.append(op); i0.append(a); i1.append(b); i2.append(-1); exp_bits.append(f(a, b))
    for s in (0, 1):
        for a in (0, 1):
            for b in (0, 1):
                ops.append(oracle.MUX); i0.append(a); i1.append(b); i2.append(s); exp_bits.append(a if s else b)
    for a in (0, 1):
        ops.append(oracle.NOT); i0.append(a); i1.append(-1); i2.append(-1); exp_bits.append(1 - a)
        ops.append(oracle.BUF); i0.append(a); i1.append(-1); i2.append(-1); exp_bits.append(a)
        ops.append(oracle.DFF); i0.append(a); i1.append(-1); i2.append(-1); exp_bits.append(a)
    ops += [oracle.CONST_ONE, oracle.CONST_ZERO]; i0 += [-1, -1]; i1 += [-1, -1]; i2 += [-1, -1]; exp_bits += [1, 0]
    # AND with a trivial operand
    ops.append(oracle.AND); i0.append(1); i1.append(2); i2.append(-1); exp_bits.append(1)
    n_g = len(ops)
    outs = np.arange(3, 3 + n_g, dtype=np.int32)
    n_w = 3 + n_g
    w = sk.wires(n_w)
    w.upload([0, 1], ct)
    w.set_trivial([2], [1])
    w.eval_gate_level(ops, i0, i1, i2, outs)
    got = w.download()
    ref = np.zeros((n_w, p.n + 1), dtype=np.uint32)
    ref[0:2] = ct
    ref[2, p.n] = 1 << 29
    orc.eval_level(ref, ops, i0, i1, i2, outs)
    assert np.array_equal(got[:3], ref[:3])
    for g in range(n_g):
        assert np.array_equal(got[3 + g], ref[3 + g]), f"gate {g} (op {ops[g]}) differs from the oracle"
    dec = ck.decrypt(got[3:])
    assert list(dec.astype(int)) == exp_bits


@pytest.mark.parametrize("name", ["boolean_default", "helm_cuda"])
def test_full_parameter_sets(name):
    """Full-size parameter sets: tfhe boolean DEFAULT (helm.rs:241) and the set
    hard-coded at helm.rs:141-146.  A few gates bit-exact vs the oracle, plus
    decrypted truth tables for all of them."""
    ck = helm_amd.ClientKey.generate(name, seed=5)
    p = ck.params
    sk = helm_amd.ServerKey(ck)
    orc = oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True)
    ct = ck.encrypt([False, True])
    ops, i0, i1, i2, exp_bits = [], [], [], [], []
    for op, f in GATES2.items():
        for a in (0, 1):
            for b in (0, 1):
                ops.append(op); i0.append(a); i1.append(b); i2.append(-1); exp_bits.append(f(a, b))
    for s in (0, 1):
        for a in (0, 1):
            for b in (0, 1):
                ops.append(oracle.MUX); i0.append(a); i1.append(b); i2.append(s); exp_bits.append(a if s else b)
    n_g = len(ops)
    outs = np.arange(2, 2 + n_g, dtype=np.int32)
    w = sk.wires(2 + n_g)
    w.upload([0, 1], ct)
    w.eval_gate_level(ops, i0, i1, i2, outs)
    got = w.download()
    dec = ck.decrypt(got[2:])
    assert list(dec.astype(int)) == exp_bits
    # bit-exact on a subset (the CPU oracle takes ~0.1-0.3 s per bootstrap)
    sub = [0, 5, 10, 15, 20, 23, 24, 31]
    ref = np.zeros_like(got)
    ref[0:2] = ct
    orc.eval_level(ref, [ops[g] for g in sub], [i0[g] for g in sub], [i1[g] for g in sub], [i2[g] for g in sub],
                   [outs[g] for g in sub])
    for g in sub:
        assert np.array_equal(got[2 + g], ref[2 + g]), f"gate {g} differs from the oracle"
    sk.close()


def test_handles_outlive_their_context():
    """Wire tables, programs and circuits freed AFTER their context (destructor order of a host
    language is arbitrary) must be harmless: the context releases their device memory."""
    import gc as _gc
    from helm_amd import Circuit, GateCircuit, PtxtType, verilog_parser
    ck = helm_amd.ClientKey.generate("toy", seed=1)
    sk = helm_amd.ServerKey(ck)
    w = sk.wires(8)
    w.upload([0, 1], ck.encrypt([True, False]))
    prog = helm_amd.Program(sk, [oracle.AND], [0], [1], [-1], [2], [0, 1])
    prog.run(w)
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(
        "input a, b;\noutput y;\nand g0(a, b, y);\n", False)
    c = Circuit(gates, inputs, outputs, dffs)
    c.sort_circuit()
    c.compute_levels()
    g = GateCircuit(ck, sk, c)
    enc = g.evaluate_encrypted(g.encrypt_inputs(wire_set, {"a": PtxtType.Bool(True), "b": PtxtType.Bool(True)}), 1, "bool")
    assert ck.decrypt(enc["y"])
    sck = helm_amd.SiClientKey.generate("si_toy_512", seed=1)
    ssk = helm_amd.SiServerKey(sck)
    sw = ssk.wires(4)
    sk.close()
    ssk.close()
    h_w, h_p, h_sw = w._h, prog._h, sw._h
    from helm_amd import _native as nv
    assert nv.hip.helm_hip_wires_free(None, h_w) == 0 and nv.hip.helm_hip_program_destroy(None, h_p) == 0
    assert nv.hip.helm_si_wires_free(None, h_sw) == 0
    w._h = prog._h = sw._h = None
    del enc, g
    _gc.collect()


def test_upload_rejects_duplicate_rows():
    """Rows of one upload are written concurrently: naming a wire twice is an error, not a race
    (a DFF output listed both as input and as state used to reach the table twice)."""
    ck = helm_amd.ClientKey.generate("toy", seed=1)
    sk = helm_amd.ServerKey(ck)
    w = sk.wires(4)
    with pytest.raises(helm_amd.HelmError, match="twice"):
        w.upload([1, 2, 1], ck.encrypt([True, False, True]))
    sck = helm_amd.SiClientKey.generate("si_toy_512", seed=1)
    ssk = helm_amd.SiServerKey(sck)
    sw = ssk.wires(4)
    with pytest.raises(helm_amd.HelmError, match="twice"):
        sw.upload([0, 0], sck.encrypt([1, 2]))
    sk.close()
    ssk.close()


@pytest.mark.parametrize("variant,name", [(4, "toy_k2"), (5, "toy_k2"),
                                          (6, "toy_k2"), (7, "toy_k2"),                    # duo in step / staggered
                                          (9, "toy_k2"),                                   # trio: three per workgroup, four waves each
                                          (4, "toy"), (5, "toy"),                          # k = 1, l = 2
                                          (6, "toy"), (7, "toy"),
                                          (4, "toy_1024"), (5, "toy_1024"),                # N = 1024: wide, lockstep
                                          (6, "toy_1024"), (7, "toy_1024"),                # ... and k_pbs_duo's compact layout
                                          (9, "toy_1024"), (8, "toy_1024"),                # k_pbs_tri10: three / two per workgroup, (polynomial, transform half) waves
                                          (4, "toy_1024_l2"), (5, "toy_1024_l2"), (6, "toy_1024_l2"), (9, "toy_1024_l2"), (8, "toy_1024_l2")])   # N = 1024, l = 2
def test_every_build_of_k_pbs_bit_exact(variant, name, monkeypatch):
    """HELM_HIP_PBS_VARIANT forces one build of the blind-rotate kernel for a whole launch (wide, lockstep,
    duo in step / staggered, trio - every build the size dispatch can select); each must reproduce the oracle bit for bit.  Nine
    ciphertexts: two full workgroups and one with a single bootstrap in the lockstep build, four full
    workgroups and a half-empty one in the two-per-workgroup duo build (the other waves leave before the
    first barrier); an all-zero mask keeps every rotation at zero."""
    monkeypatch.setenv("HELM_HIP_PBS_VARIANT", str(variant))
    ck = helm_amd.ClientKey.generate(name, seed=11)
    sk = helm_amd.ServerKey(ck)  # the variant is read when the context is created
    orc = oracle.Oracle(ck.params.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True)
    p = ck.params
    rng = np.random.default_rng(20 + variant)
    cts = 11 if variant == 9 else 9  # trio: three full workgroups and one with two of its three bootstraps
    lwe = rng.integers(0, 2**32, size=(cts, p.n + 1), dtype=np.uint32)
    lwe[0] = ck.encrypt(True)
    lwe[3, :] = 0
    tvs = rng.integers(0, 2**32, size=(2, p.N), dtype=np.uint32)
    idx = rng.integers(0, 2, size=cts).astype(np.int32)
    got = sk.pbs_batch(lwe, tvs, idx)
    for g in range(len(lwe)):
        assert np.array_equal(got[g], orc.bootstrap_noks(lwe[g], tvs[idx[g]])), f"variant {variant}, ciphertext {g}"
    sk.close()


def test_launch_split_over_builds_bit_exact(monkeypatch):
    """Default dispatch of a wide launch: the full rounds (4 bootstraps per CU) go to the lockstep
    build, the remainder to the wide build.  Same ciphertexts as the lockstep build alone, and as the
    oracle on a sample."""
    ck = helm_amd.ClientKey.generate("toy_k2", seed=12)
    p = ck.params
    rng = np.random.default_rng(5)
    sk = helm_amd.ServerKey(ck)
    import torch
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    count = 4 * n_cus + 7
    lwe = rng.integers(0, 2**32, size=(count, p.n + 1), dtype=np.uint32)
    tvs = rng.integers(0, 2**32, size=(3, p.N), dtype=np.uint32)
    idx = rng.integers(0, 3, size=count).astype(np.int32)
    got = sk.pbs_batch(lwe, tvs, idx)
    sk.close()
    monkeypatch.setenv("HELM_HIP_PBS_VARIANT", "5")
    sk3 = helm_amd.ServerKey(ck)
    assert np.array_equal(got, sk3.pbs_batch(lwe, tvs, idx))
    sk3.close()
    orc = oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True)
    for g in (0, 1, 4 * n_cus - 1, count - 7, count - 1):
        assert np.array_equal(got[g], orc.bootstrap_noks(lwe[g], tvs[idx[g]])), g


def test_lockstep_build_n1024_bit_exact(monkeypatch):
    """N = 1024 sets (the reference's CUDA parameters, helm.rs:141-146): the lockstep build (two waves per
    bootstrap on one SIMD, four bootstraps per workgroup) against the oracle; 9 ciphertexts = two full
    workgroups and one with a single bootstrap."""
    monkeypatch.setenv("HELM_HIP_PBS_VARIANT", "5")
    ck = helm_amd.ClientKey.generate("toy_1024", seed=13)
    sk = helm_amd.ServerKey(ck)
    orc = oracle.Oracle(ck.params.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True)
    p = ck.params
    rng = np.random.default_rng(31)
    lwe = rng.integers(0, 2**32, size=(9, p.n + 1), dtype=np.uint32)
    lwe[0] = ck.encrypt(True)
    lwe[3, :] = 0
    tvs = rng.integers(0, 2**32, size=(2, p.N), dtype=np.uint32)
    idx = rng.integers(0, 2, size=9).astype(np.int32)
    got = sk.pbs_batch(lwe, tvs, idx)
    for g in range(len(lwe)):
        assert np.array_equal(got[g], orc.bootstrap_noks(lwe[g], tvs[idx[g]])), g
    sk.close()


def _nine_bootstraps_bit_exact(ck, sk, orc, seed):
    p = ck.params
    rng = np.random.default_rng(seed)
    lwe = rng.integers(0, 2**32, size=(9, p.n + 1), dtype=np.uint32)
    lwe[0] = ck.encrypt(True)
    lwe[3, :] = 0
    tvs = rng.integers(0, 2**32, size=(2, p.N), dtype=np.uint32)
    idx = rng.integers(0, 2, size=9).astype(np.int32)
    got = sk.pbs_batch(lwe, tvs, idx)
    for g in range(len(lwe)):
        assert np.array_equal(got[g], orc.bootstrap_noks(lwe[g], tvs[idx[g]])), g


@pytest.mark.parametrize("variant", [1, 2, 3, 8])
def test_retired_builds_are_refused_by_name(variant, monkeypatch):
    """Round 6 removed the builds the size dispatch had stopped selecting (latency, balanced, throughput, sym); asking for
    one is an error with the list of what exists, not a silent substitute."""
    monkeypatch.setenv("HELM_HIP_PBS_VARIANT", str(variant))
    ck = helm_amd.ClientKey.generate("toy_k2", seed=1)
    with pytest.raises(helm_amd.HelmError, match="retired in round 6"):
        helm_amd.ServerKey(ck)


@pytest.mark.parametrize("variant", [4, 5, 6, 7, 8, 9])
def test_n1024_builds_bit_exact_in_the_51_bit_field_too(variant, monkeypatch):
    """Round 5: N = 1024 sets run in the lazy field FpI (p = 5440^4 + 1) when the loaded key's own bound allows it - the toy and
    the cited set do, so test_every_build_of_k_pbs_bit_exact[*-toy_1024] now covers FpI's kernels.  The 51-bit field stays
    the fallback for keys that do not fit: its six N = 1024 builds (wide, lockstep, duo in step / staggered, tri10 with two / three bootstraps per workgroup)
    against the oracle under HELM_HIP_FIELD=51.  Parameters: reference src/bin/helm.rs:141-146."""
    monkeypatch.setenv("HELM_HIP_FIELD", "51")
    monkeypatch.setenv("HELM_HIP_PBS_VARIANT", str(variant))
    ck = helm_amd.ClientKey.generate("toy_1024", seed=11)
    sk = helm_amd.ServerKey(ck)
    assert sk.field_bits() == 51
    _nine_bootstraps_bit_exact(ck, sk, oracle.Oracle(ck.params.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True), 40 + variant)
    sk.close()


def test_n1024_field_follows_the_loaded_key():
    """helm_hip_load_bootstrap_key decides the field of an N = 1024 context from the key at hand: B/2 x the largest l1-norm of
    a key column below FpI's half -> the lazy field (exact for every input under that key); a key whose norms are larger -
    here every coefficient at the largest magnitude, and a real key with the polynomials of ONE step replaced - keeps the
    51-bit field, whose half covers the worst case.  Bit-exact against the oracle under each key; loading another key into
    the same context moves the field back and forth."""
    ck = helm_amd.ClientKey.generate("toy_1024", seed=12)
    p = ck.params
    sk = helm_amd.ServerKey(ck)
    assert sk.field_bits() == 50
    _nine_bootstraps_bit_exact(ck, sk, oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True), 3)
    per_step = p.pbs_l * (p.k + 1) ** 2 * p.N
    worst = np.full_like(ck.bsk, 0x7FFFFFFF)
    one_step = ck.bsk.copy().reshape(p.n, per_step)
    one_step[5] = 0x80000000                      # -2^31 everywhere in step 5 only
    from helm_amd._native import hip, hip_check, as_u32p
    for bsk in (worst, one_step.reshape(-1)):
        bsk = np.ascontiguousarray(bsk, dtype=np.uint32).reshape(-1)
        hip_check(hip.helm_hip_load_bootstrap_key(sk._h, as_u32p(bsk), bsk.size))   # the same context: the tables follow the key
        assert sk.field_bits() == 51
        _nine_bootstraps_bit_exact(ck, sk, oracle.Oracle(p.as_tuple7(), bsk, ck.ksk, use_ntt=True), 4)
    own = np.ascontiguousarray(ck.bsk, dtype=np.uint32).reshape(-1)
    hip_check(hip.helm_hip_load_bootstrap_key(sk._h, as_u32p(own), own.size))
    assert sk.field_bits() == 50
    _nine_bootstraps_bit_exact(ck, sk, oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True), 5)
    sk.close()
    # the cited set at full size: its generated key fits the lazy field
    ck = helm_amd.ClientKey.generate("helm_cuda", seed=7)
    sk = helm_amd.ServerKey(ck)
    assert sk.field_bits() == 50
    sk.close()


@pytest.mark.parametrize("name", ["boolean_default", "helm_cuda"])
def test_full_size_lockstep_rounds_bit_exact(name):
    """The benchmark's dominant kernel at the benchmark's parameter set (tfhe boolean DEFAULT, helm.rs:241)
    and at the reference's cited CUDA set (helm.rs:141-146), through one eval_gate_level per dispatch shape:
      4 CU + 7        one full lockstep round + a remainder that goes to the wide (N = 512) / all-levels
                      (N = 1024) build
      5 CU + 5        lockstep round + a remainder of more than one bootstrap per CU (k_pbs_duo: staggered at N = 512, the
                      compact layout in step at N = 1024)
      6 CU + CU/2 + 2 between two and three per CU left over: k_pbs_trio (three bootstraps per workgroup, four waves each;
                      k = 2 only - at N = 1024 the whole launch runs in lockstep, last workgroup partial)
      7 CU + CU/2 + 2 more than three per CU left over: the whole launch in lockstep, last workgroup partial
    Rows of the first and last lockstep workgroup, of the partial workgroup and of the remainder build are
    compared bit for bit with the oracle (same gate formulas as tests/gates_test.rs:82-107 decrypts); every
    output is checked after decryption."""
    import torch
    cu = torch.cuda.get_device_properties(0).multi_processor_count
    ck = helm_amd.ClientKey.generate(name, seed=7)
    p = ck.params
    sk = helm_amd.ServerKey(ck)
    orc = oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True)
    rng = np.random.default_rng(0xB007)
    n_in = 24
    bits = rng.integers(0, 2, n_in)
    ct = ck.encrypt(bits.astype(bool))
    two_in = [oracle.AND, oracle.OR, oracle.NAND, oracle.NOR, oracle.XOR, oracle.XNOR]
    for count in (4 * cu + 7, 5 * cu + 5, 6 * cu + cu // 2 + 2, 7 * cu + cu // 2 + 2):
        ops = rng.choice(two_in, size=count).astype(np.int32)
        i0 = rng.integers(0, n_in, count).astype(np.int32)
        i1 = rng.integers(0, n_in, count).astype(np.int32)
        i2 = np.full(count, -1, np.int32)
        outs = np.arange(n_in, n_in + count, dtype=np.int32)
        w = sk.wires(n_in + count)
        w.upload(np.arange(n_in), ct)
        w.eval_gate_level(ops, i0, i1, i2, outs)
        got = w.download()
        w.free()
        want_bits = [GATES2[int(o)](int(bits[a]), int(bits[b])) for o, a, b in zip(ops, i0, i1)]
        assert list(ck.decrypt(got[n_in:]).astype(int)) == want_bits, f"{name}: launch of {count} decrypts wrong"
        full = count // (4 * cu) * (4 * cu)
        if count - full > (3 if p.k == 2 else 2) * cu:
            full = count
        sample = sorted({0, 1, 2, 3, 5, 4 * cu // 2 + 1, full - 4, full - 3, full - 2, full - 1,  # lockstep part
                         min(full, count - 1), count - 3, count - 2, count - 1,                    # remainder build
                         (count - 1) // 4 * 4, count // 2,                                         # last (partial) workgroup
                         min(count - 1, full + (count - full) // 2),                               # (of the remainder build)
                         min(count - 1, full + max(0, count - full - 1) // 3 * 3)})
        ref = np.zeros_like(got)
        ref[:n_in] = ct
        orc.eval_level(ref, ops[sample], i0[sample], i1[sample], i2[sample], outs[sample])
        for g in sample:
            assert np.array_equal(got[n_in + g], ref[n_in + g]), f"{name}: launch of {count}, gate {g} differs from the oracle"
    # MUX at full size: two bootstraps + recombination fused into the keyswitch
    ops = np.full(8, oracle.MUX, np.int32)
    i0, i1, i2 = (rng.integers(0, n_in, 8).astype(np.int32) for _ in range(3))
    outs = np.arange(n_in, n_in + 8, dtype=np.int32)
    w = sk.wires(n_in + 8)
    w.upload(np.arange(n_in), ct)
    w.eval_gate_level(ops, i0, i1, i2, outs)
    got = w.download()
    ref = np.zeros_like(got)
    ref[:n_in] = ct
    orc.eval_level(ref, ops, i0, i1, i2, outs)
    assert np.array_equal(got, ref)
    assert list(ck.decrypt(got[n_in:]).astype(int)) == [int(bits[a] if bits[s] else bits[b]) for a, b, s in zip(i0, i1, i2)]
    sk.close()


@pytest.mark.parametrize("name", ["toy_k2", "toy_1024"])
def test_vector_alu_keyswitch_fallback_bit_exact(name, monkeypatch):
    """The matrix-core keyswitch serves ks_l in {1, 2, 4, 8}; other level counts (and HELM_HIP_KS_MFMA=0) run the
    vector-ALU kernel, with its key rows split over workgroup slices on narrow launches.  Both paths must give the
    oracle's words: a narrow batch (sliced, atomic adds) and a batch wide enough for plain stores."""
    monkeypatch.setenv("HELM_HIP_KS_MFMA", "0")
    ck = helm_amd.ClientKey.generate(name, seed=17)
    sk = helm_amd.ServerKey(ck)
    monkeypatch.delenv("HELM_HIP_KS_MFMA")
    sk_mfma = helm_amd.ServerKey(ck)
    orc = oracle.Oracle(ck.params.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True)
    p = ck.params
    rng = np.random.default_rng(8)
    for count in (5, 2100):
        big = rng.integers(0, 2**32, size=(count, p.k * p.N + 1), dtype=np.uint32)
        got = sk.keyswitch_batch(big)
        assert np.array_equal(got, sk_mfma.keyswitch_batch(big))  # the two kernels agree on every word
        for g in (0, 1, count // 2, count - 1):
            assert np.array_equal(got[g], orc.keyswitch(big[g])), (name, count, g)
    sk.close()
    sk_mfma.close()


@pytest.mark.parametrize("name", ["boolean_default", "helm_cuda"])
def test_whole_launch_every_row_bit_exact(name):
    """EVERY output row of one launch that takes the size dispatch through a full lockstep round AND a two-per-CU remainder
    (4 CU + CU + 5 gates: lockstep k_pbs + k_pbs_duo staggered at N = 512, + k_pbs_tri10 with two per workgroup at N = 1024), of one that
    ends in a one-per-CU remainder (4 CU + 7: k_pbs_wide) and of one of three per CU (3 CU + 5 bootstraps' worth of gates:
    k_pbs_trio at N = 512, k_pbs_tri10 at N = 1024, the last workgroup partial), all gate types incl. MUX, NOT and constants, against the oracle's
    SIMD route (oracle/fp_route.inc: exact fp64 NTT over a different prime than the GPU's, the route bench.py's cpu_baseline
    times) - not a sample: the matrix-core keyswitch's tiles, every workgroup position and both kernels of the launch."""
    import torch
    cu = torch.cuda.get_device_properties(0).multi_processor_count
    ck = helm_amd.ClientKey.generate(name, seed=9)
    p = ck.params
    sk = helm_amd.ServerKey(ck)
    orc = oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk, use_ntt=False, use_fp=True)
    rng = np.random.default_rng(0xA11)
    n_in = 32
    bits = rng.integers(0, 2, n_in)
    ct = ck.encrypt(bits.astype(bool))
    kinds = [oracle.AND, oracle.OR, oracle.NAND, oracle.NOR, oracle.XOR, oracle.XNOR, oracle.MUX, oracle.NOT]
    for count in (5 * cu + 5, 4 * cu + 7, -(3 * cu + 5)):
        if count < 0:
            # a launch of exactly 3 CU + 5 BOOTSTRAPS (a MUX is two, a NOT none): gates drawn until the count is met
            ops = []
            while sum(2 if o == oracle.MUX else 0 if o == oracle.NOT else 1 for o in ops) < -count - 1:
                ops.append(int(rng.choice(kinds, p=[.16, .12, .16, .12, .16, .12, .1, .06])))
            ops.append(oracle.AND)
            ops = np.array(ops, dtype=np.int32)
            count = len(ops)
            assert sum(2 if o == oracle.MUX else 0 if o == oracle.NOT else 1 for o in ops) in (3 * cu + 5, 3 * cu + 6)
        else:
            ops = rng.choice(kinds, size=count, p=[.16, .12, .16, .12, .16, .12, .1, .06]).astype(np.int32)
        i0 = rng.integers(0, n_in, count).astype(np.int32)
        i1 = np.where(ops == oracle.NOT, -1, rng.integers(0, n_in, count)).astype(np.int32)
        i2 = np.where(ops == oracle.MUX, rng.integers(0, n_in, count), -1).astype(np.int32)
        outs = np.arange(n_in, n_in + count, dtype=np.int32)
        w = sk.wires(n_in + count)
        w.upload(np.arange(n_in), ct)
        w.eval_gate_level(ops, i0, i1, i2, outs)
        got = w.download()
        w.free()
        ref = np.zeros_like(got)
        ref[:n_in] = ct
        orc.eval_level_fp(ref, ops, i0, i1, i2, outs)
        bad = np.nonzero(np.any(got != ref, axis=1))[0]
        assert len(bad) == 0, f"{name}: launch of {count}: rows {bad[:8]} differ from the oracle"
        f2 = dict(GATES2)
        want = [(int(bits[a]) if int(bits[c]) else int(bits[b])) if o == oracle.MUX else (1 - int(bits[a])) if o == oracle.NOT
                else f2[int(o)](int(bits[a]), int(bits[b])) for o, a, b, c in zip(ops, i0, i1, i2)]
        assert list(ck.decrypt(got[n_in:]).astype(int)) == want
    sk.close()


@pytest.mark.parametrize("params,levels_checked", [("boolean_default", None), ("helm_cuda", 32)])
def test_one_aes128_evaluation_every_wire_bit_exact(params, levels_checked):
    """BASELINE config 4 as ONE circuit at the full parameter set (what reference src/bin/helm.rs:256-262 runs): the 207 levels of
    the AES-128 netlist (FIPS-197 C.1 key and plaintext) on the GPU - launches of 80-256 bootstraps: k_pbs_wide, the
    single-circuit kernel, and the keyswitch's narrow-launch form - and ALL 207 levels (32 k gates; HELM_TEST_SAMPLED_AES=1: the
    first 64) on the oracle's SIMD route from the same input ciphertexts: EVERY wire bit for bit, and the 128 output bits
    decrypt to the FIPS-197 ciphertext.  helm_cuda (the set reference src/bin/helm.rs:141-146 cites): the first 32 levels, so
    that k_pbs_wide at N = 1024 in the lazy field the loaded key selects (FpI) meets the oracle on a real netlist as well."""
    import os
    from helm_amd import Circuit, verilog_parser
    from helm_amd.distributed import level_arrays
    from helm_amd.netlists import aes128, aes128_reference_encrypt
    ck = helm_amd.ClientKey.generate(params, seed=4)
    p = ck.params
    sk = helm_amd.ServerKey(ck)
    if params == "helm_cuda":
        assert sk.field_bits() == 50   # the generated key fits the lazy field of N = 1024 contexts
    orc = oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk, use_ntt=False, use_fp=True)
    gates, wire_set, inputs, outputs, dffs, _, _ = verilog_parser.read_verilog_text(aes128(), False)
    c = Circuit(gates, inputs, outputs, dffs)
    c.sort_circuit()
    c.compute_levels()
    names = list(inputs) + sorted(wire_set)
    index = {w: i for i, w in enumerate(names)}
    ops, i0, i1, i2, out, off = level_arrays(c, index)
    key, pt = bytes(range(16)), bytes.fromhex("00112233445566778899aabbccddeeff")
    kv, pv = int.from_bytes(key, "big"), int.from_bytes(pt, "big")
    rows = np.array([index[f"{w}[{i}]"] for w in ("key", "pt") for i in range(128)], np.int32)
    bits = np.array([(v >> i) & 1 for v in (kv, pv) for i in range(128)], dtype=bool)
    cts = ck.encrypt(bits)
    w = sk.wires(len(names))
    w.upload(rows, cts)
    prog = helm_amd.Program(sk, ops, i0, i1, i2, out, off)
    prog.run(w)
    sk.sync()
    got = w.download()
    host = np.zeros_like(got)
    host[rows] = cts
    n_check = levels_checked or (64 if os.environ.get("HELM_TEST_SAMPLED_AES") == "1" else len(off) - 1)
    for l in range(n_check):
        s = slice(off[l], off[l + 1])
        orc.eval_level_fp(host, ops[s], i0[s], i1[s], i2[s], out[s])
    written = np.concatenate([rows, out[:off[n_check]]])
    bad = [names[r] for r in written if not np.array_equal(got[r], host[r])]
    assert not bad, f"{len(bad)} wires differ from the oracle, first: {bad[:5]}"
    dec = ck.decrypt(got[[index[f"ct[{i}]"] for i in range(128)]])
    assert sum(int(dec[i]) << i for i in range(128)).to_bytes(16, "big") == aes128_reference_encrypt(key, pt)
    assert prog.total_pbs() > 30000 and len(off) - 1 > 200
    sk.close()
