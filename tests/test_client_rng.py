"""Client randomness (helm_amd/csrc/rng.hpp): OS-entropy ChaCha20 by default (as tfhe's gen_keys(),
reference src/bin/helm.rs:241,301), the seeded deterministic generator only on request
(as the reference's own tests fix a seed, tests/circuit_test.rs:119)."""
import numpy as np
import pytest

import helm_amd
from helm_amd import _native as nv


def test_chacha20_block_function_known_answer():
    assert nv.host.helm_client_rng_selftest() == 0  # RFC 8439 section 2.3.2


def test_default_keys_come_from_os_entropy():
    a = helm_amd.ClientKey.generate("toy")
    b = helm_amd.ClientKey.generate("toy")
    assert not np.array_equal(a.glwe_secret, b.glwe_secret)  # 512 secret bits: equal with probability 2^-512
    assert not np.array_equal(a.bsk[:4096], b.bsk[:4096])
    # fresh, independent encryption randomness: same plaintext, different ciphertexts, both decrypt
    c = a.encrypt([True, True, False])
    assert not np.array_equal(c[0], c[1])
    assert list(a.decrypt(c)) == [True, True, False]
    sa = helm_amd.SiClientKey.generate("si_toy_512")
    sb = helm_amd.SiClientKey.generate("si_toy_512")
    assert not np.array_equal(sa.glwe_secret, sb.glwe_secret)
    assert list(sa.decrypt(sa.encrypt([1, 0, 1]))) == [1, 0, 1]


def test_seeded_keys_reproduce_and_zero_is_reserved():
    a = helm_amd.ClientKey.generate("toy", seed=7)
    b = helm_amd.ClientKey.generate("toy", seed=7)
    assert np.array_equal(a.lwe_secret, b.lwe_secret) and np.array_equal(a.bsk, b.bsk) and np.array_equal(a.ksk, b.ksk)
    assert np.array_equal(a.encrypt([True, False]), b.encrypt([True, False]))
    with pytest.raises(ValueError):
        helm_amd.ClientKey.generate("toy", seed=0)
    with pytest.raises(ValueError):
        helm_amd.SiClientKey.generate("si_toy_512", seed=0)


def test_secret_key_bits_are_balanced_under_os_entropy():
    k = helm_amd.ClientKey.generate("boolean_default")
    ones = int(np.sum(k.glwe_secret)) + int(np.sum(k.lwe_secret))
    total = k.params.k * k.params.N + k.params.n
    assert abs(ones - total / 2) < 6 * (total ** 0.5) / 2  # six sigma
    # noise of fresh encryptions has the set's standard deviation (2^32 * 1.3e-5 = 56 k)
    ph = k.phase(k.encrypt(np.ones(4096, dtype=bool))).astype(np.int64) - (1 << 29)
    assert 0.9 < float(np.std(ph)) / (1.3071021089943935e-5 * 2**32) < 1.1
