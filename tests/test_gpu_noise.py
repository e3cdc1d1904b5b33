"""Noise margins of the GPU bootstraps at the FULL parameter sets, measured.

Bit-exactness against the oracle says the kernels compute the right function; it says nothing about whether a
parameter set - four of the five here are recalled from memory, two of them with an interpolated / extrapolated noise
value - leaves room to decrypt.  The reference pins only decrypted values (tests/gates_test.rs:82-107,
tests/circuit_test.rs:308-310) under tfhe's own sets (src/bin/helm.rs:83, 241, 301); here the phase error of >= 2,048 gate
bootstraps (boolean) and >= 512 look-ups (64-bit sets) is measured on the GPU's ciphertexts and compared with the
standard TFHE variance formulas (torus units, sigma = the set's noise standard deviations, binary secret keys, balanced
signed digits of variance (B^2 + 2) / 12):

  blind rotation   V_br = n [ l (k+1) N (B^2 + 2)/12 sigma_glwe^2  +  (1 + k N / 2) / (24 B^(2l)) ]
     multi-bit, groups of g:  (n / g) [ 2^g l (k+1) N (B^2 + 2)/12 sigma_glwe^2  +  (1 + k N / 2) / (12 B^(2l)) ]
  keyswitch        V_ks = k N l_ks (B_ks^2 + 2)/12 sigma_lwe^2  +  k N / (24 B_ks^(2 l_ks))
  modulus switch   V_ms = (1 + n / 2) / (48 N^2)

(the first term of each is the key's noise through the digits, the second the rounding of the decomposition through the
secret key; Chillotti-Gama-Georgieva-Izabachene, J. Cryptology 2020, with the average-case digit variance).  The measured
variances must sit within [0.6, 1.5] of these, and the decryption-failure probability they imply for the WORST operand the
evaluators form (two gate outputs for a boolean gate, 4x + 2y + z of three look-up outputs for a 3-input LUT, 2x + y for a
bivariate one) must be below the bound stated per set.  As a check of the formulas themselves: for
PARAM_MESSAGE_2_CARRY_2_KS_PBS they predict 2^-40.3, the failure probability tfhe quotes for its shortint sets."""
import math

import numpy as np
import pytest

import helm_amd
from helm_amd.shortint import si_named_params

pytestmark = pytest.mark.gpu
NAND = 4  # HELM_GATE_NAND


def predicted(p, s_lwe, s_glwe, g=1):
    B, Bk = 2.0 ** p.pbs_logB, 2.0 ** p.ks_logB
    dvar = (B * B + 2) / 12
    if g == 1:
        br = p.n * (p.pbs_l * (p.k + 1) * p.N * dvar * s_glwe ** 2 + (1 + p.k * p.N / 2) / (24 * B ** (2 * p.pbs_l)))
    else:
        br = (p.n / g) * ((2 ** g) * p.pbs_l * (p.k + 1) * p.N * dvar * s_glwe ** 2 + (1 + p.k * p.N / 2) / (12 * B ** (2 * p.pbs_l)))
    ks = p.k * p.N * p.ks_l * ((Bk * Bk + 2) / 12) * s_lwe ** 2 + p.k * p.N / (24 * Bk ** (2 * p.ks_l))
    ms = (1 + p.n / 2) / (48 * p.N * p.N)
    return br, ks, ms


def log2_pfail(margin, variance):
    """two-sided Gaussian tail beyond `margin`, as log2 (erfc underflows below ~2^-1000: clamp)"""
    x = margin / math.sqrt(2 * variance)
    v = math.erfc(x)
    return math.log2(v) if v > 0 else -1000.0


def signed_err(phase, expected, bits):
    d = (phase.astype(np.uint64) - expected.astype(np.uint64)) & np.uint64((1 << bits) - 1)
    d = d.astype(np.int64) if bits == 64 else np.where(d >= (1 << (bits - 1)), d.astype(np.int64) - (1 << bits), d.astype(np.int64))
    return d.astype(np.float64) / float(1 << bits)


def test_boolean_default_gate_bootstraps_sit_inside_the_noise_budget():
    """2,048 NAND gates on fresh encryptions at tfhe's boolean DEFAULT_PARAMETERS (helm.rs:241): phase error of the gate
    outputs (blind rotation + keyswitch) against the formulas; a second level on those outputs decrypts correctly; the
    failure probability of a gate fed by two gate outputs (XOR: the doubled sum has the same margin-to-noise ratio)."""
    name, B = "boolean_default", 2048
    p, s_lwe, s_glwe = helm_amd.named_params(name)
    ck = helm_amd.ClientKey.generate(name, seed=17)
    sk = helm_amd.ServerKey(ck)
    rng = np.random.default_rng(1)
    bits = rng.integers(0, 2, size=2 * B).astype(bool)
    w = sk.wires(4 * B)
    w.upload(np.arange(2 * B), ck.encrypt(bits))
    ops = np.full(B, NAND, np.int32)
    m1 = np.full(B, -1, np.int32)
    out1 = np.arange(2 * B, 3 * B, dtype=np.int32)
    w.eval_gate_level(ops, np.arange(B, dtype=np.int32), np.arange(B, 2 * B, dtype=np.int32), m1, out1)
    sk.sync()
    ct = w.download(out1)
    want = ~(bits[:B] & bits[B:])
    assert np.array_equal(ck.decrypt(ct), want)
    err = signed_err(ck.phase(ct), np.where(want, 1 << 29, 7 << 29).astype(np.uint32), 32)
    v_meas = float(np.mean(err ** 2))
    br, ks, ms = predicted(p, s_lwe, s_glwe)
    ratio = v_meas / (br + ks)
    # second level: gates on gate outputs (the steady state of a circuit)
    out2 = np.arange(3 * B, 4 * B, dtype=np.int32)
    w.eval_gate_level(ops, out1, np.roll(out1, 1), m1, out2)
    sk.sync()
    assert np.array_equal(ck.decrypt(w.download(out2)), ~(want & np.roll(want, 1)))
    lp = log2_pfail(0.125, 2 * v_meas + ms)
    print(f"\n{name}: measured std 2^{math.log2(math.sqrt(v_meas)):.2f}, predicted 2^{math.log2(math.sqrt(br + ks)):.2f} "
          f"(blind rotation {br:.3e} + keyswitch {ks:.3e}), ratio {ratio:.3f}; max |err| {np.abs(err).max():.2e} of a margin of 0.125; "
          f"modulus switch {ms:.3e}; failure probability of a gate on two gate outputs 2^{lp:.1f}")
    assert abs(float(np.mean(err))) < 4 * math.sqrt(v_meas / B), "the phase error is biased"
    assert 0.6 < ratio < 1.5, f"measured variance {v_meas:.3e} vs predicted {br + ks:.3e}"
    assert lp < -100, "tfhe's boolean DEFAULT set leaves a wide margin; a figure above 2^-100 means broken noise"
    sk.close()


@pytest.mark.parametrize("name,arity,bound_log2,approximate", [
    ("shortint_m2c2", 3, -35.0, False),            # PARAM_MESSAGE_2_CARRY_2_KS_PBS: the formulas give 2^-40.3, tfhe's own target
    ("shortint_m1c1", 2, -35.0, True),             # helm.rs:301; GLWE noise interpolated
    ("shortint_m2c2_multibit3", 3, -20.0, True),   # helm.rs:83; LWE noise extrapolated: the keyswitch rounding alone caps it near 2^-27
])
def test_shortint_sets_look_ups_sit_inside_the_noise_budget(name, arity, bound_log2, approximate):
    """512 look-ups on fresh encryptions at the full set: variance of the bootstrap's output under the big key, of the
    keyswitch's increment under the small key, each against its formula; failure probability of a look-up whose operand is
    the weighted sum the evaluators form from PREVIOUS look-up outputs (gates.rs:773-778: 4x + 2y + z; :761-764: 2x + y)."""
    B = 512
    p, s_lwe, s_glwe = si_named_params(name)
    g = max(1, p.grouping_factor)
    ck = helm_amd.SiClientKey.generate(name, seed=23)
    sk = helm_amd.SiServerKey(ck)
    t = p.message_modulus * p.carry_modulus
    delta = 1 << (63 - int(math.log2(t)))  # 2^63 / t
    rng = np.random.default_rng(2)
    bits = rng.integers(0, 2, size=arity * B).astype(np.uint64)
    w = sk.wires((arity + 1) * B)
    w.upload(np.arange(arity * B), ck.encrypt(bits))
    in_idx = np.arange(arity * B, dtype=np.int32).reshape(arity, B).T.copy()
    table = 0x96 if arity == 3 else 0x6  # parity
    out = np.arange(arity * B, (arity + 1) * B, dtype=np.int32)
    w.eval_lut_level(np.full(B, arity, np.int32), in_idx, np.full(B, table, np.uint64), out)
    sk.sync()
    ct = w.download(out)
    want = np.bitwise_xor.reduce(bits.reshape(arity, B), axis=0)
    assert np.array_equal(ck.decrypt(ct), want)
    e_big = signed_err(ck.phase(ct), want * np.uint64(delta), 64)
    small = sk.keyswitch_batch(ct)
    e_small = signed_err(ck.phase(small, small=True), want * np.uint64(delta), 64)
    v_br = float(np.mean(e_big ** 2))
    v_ks = float(np.mean((e_small - e_big) ** 2))  # the keyswitch adds its own error to the one it carries over
    br, ks, ms = predicted(p, s_lwe, s_glwe, g)
    weight = 21 if arity == 3 else 5               # (4^2 + 2^2 + 1) or (2^2 + 1) look-up outputs in one operand
    half_box = 1.0 / (4 * t)                       # delta / 2 in torus units
    lp = log2_pfail(half_box, weight * v_br + v_ks + ms)
    print(f"\n{name}{' [approximate set]' if approximate else ''}: blind rotation measured {v_br:.3e} / predicted {br:.3e} "
          f"(ratio {v_br / br:.3f}); keyswitch measured {v_ks:.3e} / predicted {ks:.3e} (ratio {v_ks / ks:.3f}); modulus switch {ms:.3e}; "
          f"max |err| after keyswitch {np.abs(e_small).max():.2e} of half a box {half_box:.2e}; "
          f"failure probability of a {arity}-input look-up on look-up outputs 2^{lp:.1f} (bound 2^{bound_log2:.0f})")
    assert 0.6 < v_br / br < 1.5, f"blind rotation: measured {v_br:.3e} vs predicted {br:.3e}"
    assert 0.6 < v_ks / ks < 1.5, f"keyswitch: measured {v_ks:.3e} vs predicted {ks:.3e}"
    assert np.abs(e_small).max() < half_box, "a look-up left its box"
    assert lp < bound_log2
    sk.close()
