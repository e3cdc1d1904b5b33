"""Key-layout contract of the Rust shim (INTEGRATION.md "Key layout"; helm_amd/csrc/host/key_import.cpp):
tfhe-rs 0.4 container order [RECALLED] <-> this ABI.  Testable without the crate: the conversions are exact
inverses, the keyswitching-key conversion really reorders (levels reversed), a key left in the other order
keyswitches WRONG, and one converted back evaluates gates bit for bit like the original."""
import ctypes as C

import numpy as np
import pytest

import helm_amd
import oracle
from helm_amd import _native as nv


def _conv(fn, params, src):
    dst = np.zeros_like(src)
    ptr = nv.as_u32p if src.dtype == np.uint32 else nv.as_u64p
    rc = fn(C.byref(params), ptr(src), ptr(dst), src.size)
    if rc != 0:
        raise helm_amd.HelmError(nv.host.helm_keys_last_error().decode())
    return dst


def test_boolean_key_round_trip_and_level_order():
    ck = helm_amd.ClientKey.generate("toy_k2", seed=3)
    p = ck.params
    bsk, ksk = np.array(ck.bsk), np.array(ck.ksk)
    t_bsk = _conv(nv.host.helm_keys_bsk32_to_tfhe, p, bsk)
    t_ksk = _conv(nv.host.helm_keys_ksk32_to_tfhe, p, ksk)
    assert np.array_equal(t_bsk, bsk)                                # same container order [RECALLED]
    assert not np.array_equal(t_ksk, ksk)                            # levels reversed inside every block
    blocks = ksk.reshape(p.k * p.N, p.ks_l, p.n + 1)
    assert np.array_equal(t_ksk.reshape(blocks.shape), blocks[:, ::-1, :])
    assert np.array_equal(_conv(nv.host.helm_keys_bsk32_from_tfhe, p, t_bsk), bsk)
    assert np.array_equal(_conv(nv.host.helm_keys_ksk32_from_tfhe, p, t_ksk), ksk)
    # the order matters: the same words in the other order keyswitch to garbage, converted back they decrypt
    good = oracle.Oracle(p.as_tuple7(), bsk, _conv(nv.host.helm_keys_ksk32_from_tfhe, p, t_ksk))
    bad = oracle.Oracle(p.as_tuple7(), bsk, t_ksk)
    a, b = ck.encrypt([True, False])
    assert ck.decrypt(good.gate(oracle.OR, a, b)) is True and ck.decrypt(good.gate(oracle.AND, a, b)) is False
    outs = [bad.gate(op, x, y) for op in (oracle.OR, oracle.AND, oracle.XOR) for x in (a, b) for y in (a, b)]
    ph = ck.phase(np.array(outs)).astype(np.int64)
    dist = np.minimum(np.abs(ph - (1 << 29)), np.abs(ph - (7 << 29)))
    assert (dist > (1 << 26)).any()                                  # not the clean +-1/8 a correct keyswitch yields
    with pytest.raises(helm_amd.HelmError, match="wrong number of words"):
        _conv(nv.host.helm_keys_ksk32_from_tfhe, p, ksk[:-1].copy())


def test_shortint_key_round_trip_and_multibit_is_refused():
    ck = helm_amd.SiClientKey.generate("si_toy_512", seed=3)
    p = ck.params
    bsk, ksk = np.array(ck.bsk), np.array(ck.ksk)
    t_ksk = _conv(nv.host.helm_keys_ksk64_to_tfhe, p, ksk)
    assert not np.array_equal(t_ksk, ksk)
    assert np.array_equal(_conv(nv.host.helm_keys_ksk64_from_tfhe, p, t_ksk), ksk)
    assert np.array_equal(_conv(nv.host.helm_keys_bsk64_from_tfhe, p, _conv(nv.host.helm_keys_bsk64_to_tfhe, p, bsk)), bsk)
    mb = helm_amd.SiClientKey.generate("si_toy_1024_mb2", seed=3)
    with pytest.raises(helm_amd.HelmError, match="multi-bit"):
        _conv(nv.host.helm_keys_bsk64_from_tfhe, mb.params, np.array(mb.bsk))


@pytest.mark.gpu
def test_key_through_tfhe_order_and_back_bootstraps_bit_identically():
    """The KAT the shim's import path has to pass: (bsk, ksk) -> tfhe order -> back -> helm_hip_load_*_key gives
    the same ciphertext bits as loading the original key, and both equal the oracle."""
    ck = helm_amd.ClientKey.generate("toy_k2", seed=8)
    p = ck.params
    bsk, ksk = np.array(ck.bsk), np.array(ck.ksk)
    back_bsk = _conv(nv.host.helm_keys_bsk32_from_tfhe, p, _conv(nv.host.helm_keys_bsk32_to_tfhe, p, bsk))
    back_ksk = _conv(nv.host.helm_keys_ksk32_from_tfhe, p, _conv(nv.host.helm_keys_ksk32_to_tfhe, p, ksk))
    ops = [oracle.AND, oracle.XOR, oracle.NOR, oracle.MUX, oracle.NAND]
    i0, i1, i2 = [0, 1, 0, 1, 1], [1, 1, 0, 0, 0], [-1, -1, -1, 0, -1]
    out = np.arange(2, 7, dtype=np.int32)
    ct = ck.encrypt([True, False])
    tables = []
    for b, k in ((bsk, ksk), (back_bsk, back_ksk)):
        sk = helm_amd.ServerKey(params=p, bsk=b, ksk=k)
        w = sk.wires(7)
        w.upload([0, 1], ct)
        w.eval_gate_level(ops, i0, i1, i2, out)
        tables.append(w.download())
        sk.close()
    assert np.array_equal(tables[0], tables[1])
    ref = np.zeros_like(tables[0])
    ref[:2] = ct
    oracle.Oracle(p.as_tuple7(), bsk, ksk).eval_level(ref, ops, i0, i1, i2, out)
    assert np.array_equal(tables[1], ref)
    v = [True, False]
    want = [v[0] & v[1], v[1] ^ v[1], not (v[0] | v[0]), v[1] if v[0] else v[0], not (v[1] & v[0])]  # MUX: sel ? in0 : in1
    assert list(ck.decrypt(tables[1][2:])) == want


def test_wopbs_key_level_order_converter():
    """The keys of include/helm_wopbs.h whose levels tfhe keeps last-to-first ([RECALLED]): converter is its own
    inverse, not the identity, refuses wrong sizes and in-place use."""
    from helm_amd import _native as nv
    rng = np.random.default_rng(4)
    blocks, levels, row = 6, 3, 10
    src = rng.integers(0, 1 << 64, size=blocks * levels * row, dtype=np.uint64)
    mid, back = np.zeros_like(src), np.zeros_like(src)
    assert nv.host.helm_keys_levels64_reverse(blocks, levels, row, nv.as_u64p(src), nv.as_u64p(mid), src.size) == 0
    assert nv.host.helm_keys_levels64_reverse(blocks, levels, row, nv.as_u64p(mid), nv.as_u64p(back), src.size) == 0
    assert np.array_equal(back, src) and not np.array_equal(mid, src)
    v = src.reshape(blocks, levels, row)
    assert np.array_equal(mid.reshape(blocks, levels, row), v[:, ::-1, :])
    assert nv.host.helm_keys_levels64_reverse(blocks, levels, row, nv.as_u64p(src), nv.as_u64p(mid), src.size - 1) != 0
    assert nv.host.helm_keys_levels64_reverse(blocks, levels, row, nv.as_u64p(src), nv.as_u64p(src), src.size) != 0
