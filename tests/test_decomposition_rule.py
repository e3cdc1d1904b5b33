"""The HIP kernels extract signed digits with one addition per level (decompose_step in
helm_amd/csrc/helm_hip.hip):  next = (state + B/2 - 1 + bit(2 logB - 1)) >> logB,  digit = state - next * B.
This must be tfhe's SignedDecomposer rule, which the oracle restates (oracle/tfhe_oracle.c orc_decompose,
SURVEY.md App. B): checked here against the oracle's C function over whole state spaces, ties included
(d == B/2 with the next bit clear / set), for every (logB, levels) shape the engines accept."""
import ctypes as C

import numpy as np
import pytest

import oracle


def kernel_rule(x, logB, levels):
    """numpy model of decompose<L>() / decompose_step(), uint32 arithmetic; digits[0] = most significant"""
    rep = logB * levels
    state = ((x.astype(np.uint64) + (1 << (31 - rep))) & 0xFFFFFFFF) >> (32 - rep)
    half_m1 = (1 << (logB - 1)) - 1
    out = np.zeros((levels, len(x)), dtype=np.int64)
    for lev in range(levels - 1, -1, -1):
        sb = (state >> (2 * logB - 1)) & 1
        nxt = ((state + half_m1 + sb) & 0xFFFFFFFF) >> logB
        out[lev] = state.astype(np.int64) - (nxt.astype(np.int64) << logB)
        state = nxt
    return out


@pytest.mark.parametrize("logB,levels", [(6, 3), (7, 3), (8, 2), (3, 4), (4, 4), (10, 2), (15, 2), (23, 1), (31, 1)])
def test_kernel_digit_rule_is_the_oracles(logB, levels):
    L = oracle.lib()
    L.orc_decompose.argtypes = [C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_int32)]
    L.orc_decompose.restype = None
    rep = logB * levels
    rng = np.random.default_rng(logB * 100 + levels)
    # every value of the representable part where that is small, plus random low bits; else random words and
    # the neighbourhoods of ties
    if rep <= 14:
        x = (np.arange(1 << rep, dtype=np.uint64) << (32 - rep)) | rng.integers(0, 1 << (32 - rep), size=1 << rep, dtype=np.uint64)
    else:
        x = rng.integers(0, 1 << 32, size=1 << 14, dtype=np.uint64)
        ties = (rng.integers(0, 1 << rep, size=1 << 12, dtype=np.uint64) | (1 << (logB - 1))) & ~np.uint64((1 << (logB - 1)) - 1)
        x = np.concatenate([x, (ties << (32 - rep)) & 0xFFFFFFFF, np.array([0, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF], dtype=np.uint64)])
    got = kernel_rule(x, logB, levels)
    buf = (C.c_int32 * levels)()
    for i, v in enumerate(x):
        L.orc_decompose(int(v), logB, levels, buf)
        assert list(got[:, i]) == list(buf), (hex(int(v)), logB, levels)
    # and the digits recompose to the closest representable value
    weights = np.array([1 << (32 - logB * (j + 1)) for j in range(levels)], dtype=np.int64)
    recomposed = (got * weights[:, None]).sum(axis=0) % (1 << 32)
    closest = ((x.astype(np.int64) + (1 << (31 - rep))) >> (32 - rep) << (32 - rep)) % (1 << 32)
    assert np.array_equal(recomposed, closest)
