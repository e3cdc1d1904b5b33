"""Plaintext model of helm::RadixEngine::propagate / carries (helm_amd/csrc/host/shortint_circuit.cpp): the same tables,
the same recursion, on integers mod 32 with the negacyclic look-up rule f(v + 16) = -f(v) - checked against ordinary
carry propagation for every width the radix layer uses (1..65 blocks, with and without the carry-out flag) and for every
combination of carry states inside a group.  The GPU tests check the device path on values; this pins the ALGORITHM."""
import itertools

import numpy as np
import pytest


def lut(f):
    """A 16-entry table evaluated on a value mod 32 with the padding-bit rule of a programmable bootstrap."""
    def go(v):
        v %= 32
        return f(v) % 32 if v < 16 else (-f(v - 16)) % 32
    return go


T = [lut(lambda v, p=p: (2 if v >= 4 else 1 if v == 3 else 0) << p) for p in range(4)]
S = [lut(lambda v, p=p: 0 if v == 15 else 32 - (1 << p)) for p in range(4)]      # + 2^p afterwards
Q = [lut(lambda v, i=i: 8 if v >= (2 << i) else 4 if v == (2 << i) - 1 else 0) for i in range(3)]
GC = [lut(lambda v, i=i: 4 if v >= (2 << i) else 0) for i in range(3)]
Q3 = lut(lambda v: 0 if v == 15 else 32 - 4)                                      # + 4 afterwards
GC3 = lut(lambda v: 32 - 2)                                                       # + 2 afterwards
RESOLVE = lut(lambda v: 4 if v >= 8 else 0)
FINAL = lut(lambda v: ((v & 3) + (1 if (v >> 2) >= 2 else 0)) & 3)
COUT = lut(lambda v: 1 if v >= 8 else 0)
MSG = lut(lambda v: v & 3)
rounds = 0


def carries(st, n, need, add_const, want_bit):
    """st[m]: state of item m weighted 2^(m % 4) (minus the pending constant when add_const).  Returns ({m: value}, bit)."""
    global rounds
    if need <= 0:
        return {}, True

    def vsum(first, count):
        return sum(st[first + j] + ((1 << ((first + j) % 4)) if add_const else 0) for j in range(count)) % 32

    if n <= 4:
        rounds += 1
        out = {}
        for m in range(1, need + 1):
            v = vsum(0, m)
            assert v <= 30
            out[m] = GC[m - 1](v) if m - 1 < 3 else (GC3(v) + 2) % 32
        return out, True
    ng = (n + 3) // 4
    rounds += 1
    qv, sv = {}, [None] * ng
    for k in range(ng):
        cnt = min(4, n - 4 * k)
        for i in range(cnt):
            m = 4 * k + i + 1
            want_q = m <= need and (m % 4 != 0 or m == n)
            want_s = i == 3 and k < ng - 1
            if not (want_q or want_s):
                continue
            v = vsum(4 * k, i + 1)
            assert v <= 30
            if want_s:
                sv[k] = S[k % 4](v)
            if want_q:
                qv[m - 1] = Q[i](v) if i < 3 else (Q3(v) + 4) % 32
    gneed = n // 4 - 1 if (need == n and n % 4 == 0) else need // 4
    gc, _ = carries(sv, ng, gneed, True, True)
    out = {}
    for m in range(1, need + 1):
        k = m // 4 - 1 if (m == n and m % 4 == 0) else m // 4
        pos0 = m % 4 == 0 and m != n
        c = (0 if pos0 else qv[m - 1]) + (gc[k] if k >= 1 else 0) + (4 if pos0 else 0)
        assert c in (0, 4, 8, 12)
        out[m] = c
    if want_bit:
        rounds += 1
        out = {m: RESOLVE(c) for m, c in out.items()}
        return out, True
    return out, False


def propagate(sums, flags):
    global rounds
    rounds = 1
    W = len(sums)
    need = W if flags else W - 1
    st = [T[i % 4](sums[i]) for i in range(W)]
    msg = [MSG(x) for x in sums]
    if need == 0:
        return msg, None
    C, bit = carries(st, W, need, False, False)
    rounds += 1
    res = [msg[0]] + [FINAL(msg[i] + C[i] + (4 if bit else 0)) for i in range(1, W)]
    return res, (COUT(C[W] + (4 if bit else 0)) if flags else None)


def reference(sums):
    out, c = [], 0
    for x in sums:
        out.append((x + c) & 3)
        c = (x + c) >> 2
    return out, c


@pytest.mark.parametrize("flags", [False, True])
def test_every_width(flags):
    rng = np.random.default_rng(7)
    for W in list(range(1, 70)):
        for trial in range(40):
            if trial == 0:
                sums = [3] * W                      # one long propagating chain, nothing comes in
            elif trial == 1:
                sums = [4] + [3] * (W - 1)          # a carry that runs through every block
            elif trial == 2:
                sums = [7] + [6] * (W - 1)          # the bounds of the contract
            else:
                sums = [int(rng.integers(0, 8))] + [int(x) for x in rng.choice([0, 1, 2, 3, 3, 3, 4, 5, 6], size=W - 1)]
            got, cout = propagate(sums, flags)
            want, wc = reference(sums)
            assert got == want, (W, sums)
            if flags:
                assert cout == wc, (W, sums)
        # rounds in a row: 2 + one per level of groups of four + one per level beyond the first (c form -> bit)
        def levels(n, bit):
            return 1 if n <= 4 else 1 + levels((n + 3) // 4, True) + (1 if bit else 0)
        n_items = W
        expect = 1 if (W - 1 if not flags else W) == 0 else 2 + levels(n_items, False)
        assert rounds == expect, (W, flags, rounds, expect)


def test_every_state_combination_of_a_group():
    """All 3^4 state combinations of a full group through the weighted sum and both padding-bit tables."""
    for states in itertools.product((0, 1, 2), repeat=4):
        v = sum(s << j for j, s in enumerate(states))
        assert v <= 30
        # the prefix state of the whole group: generate / propagate / absorb
        carry, allp = 0, True
        for s in states:
            carry = 1 if s == 2 else (carry if s == 1 else 0)
            allp = allp and s == 1
        want = 2 if carry else 1 if allp else 0
        for p in range(4):
            assert (S[p](v) + (1 << p)) % 32 == want << p
        assert (Q3(v) + 4) % 32 == 4 * want
        assert (GC3(v) + 2) % 32 == (4 if carry else 0)
        for i in range(3):
            vi = sum(s << j for j, s in enumerate(states[:i + 1]))
            c, ap = 0, True
            for s in states[:i + 1]:
                c = 1 if s == 2 else (c if s == 1 else 0)
                ap = ap and s == 1
            assert Q[i](vi) == 4 * (2 if c else 1 if ap else 0)
            assert GC[i](vi) == (4 if c else 0)


def test_u32_takes_four_rounds_and_u128_six():
    for W, want in ((4, 3), (8, 4), (16, 4), (32, 6), (64, 6)):
        propagate([3] * W, False)
        assert rounds == want, (W, rounds)
