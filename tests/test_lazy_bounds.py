"""Worst-case magnitudes of the boolean kernels' fp64 arithmetic in the round-4 fields (interval arithmetic, CPU only).

Every value in the kernels is an integer held in a double; exactness needs |v| < 2^53 at every addition and at the inputs
of `mulmod` (helm_amd/csrc/ntt_fp64.h).  Random parity tests cannot see a worst-case overflow, so the bounds the code
comments state are recomputed here from the structure of a CMUX step, for the primes the engine uses now:

  FpG  p = 5072^4 + 1 (lazy: no recentring in the forward transform, the products or their hand-over sums)
  FpH  p = 6432^4 + 1 (recentred at every block boundary)
  FpI  p = 5440^4 + 1 (round 5; lazy, N = 1024 sets whose LOADED KEY's exact products fit: helm_hip_load_bootstrap_key)

mulmod(a, w), |w| <= p/2:  |r| <= (0.5 + 0.75 |a| 2^-52) p   (ntt_fp64.h)
fwd_top2_digits:           |x| <= D (1 + b^2) + D (b + b^3),  D = 2^(logB-1)   - exact terms, no reduction
a Cooley-Tukey stage:      |x'| <= |u| + |mulmod(v, w)|
a Gentleman-Sande stage:   |u + v| <= 2 m;  |mulmod(u - v, w)| with |u - v| <= 2 m
"""
import pytest

P = {"FpG": 5072 ** 4 + 1, "FpH": 6432 ** 4 + 1, "FpI": 5440 ** 4 + 1}
B = {"FpG": 5072, "FpH": 6432, "FpI": 5440}
LIMIT = 2.0 ** 53


def mulmod_bound(a, p):
    assert a < LIMIT, "mulmod input not exact"
    return (0.5 + 0.75 * a / 2.0 ** 52) * p


def forward_bound(field, logn, logB, lazy, ba):
    """max |x| after the forward transform of digits: top two stages plain, the rest Cooley-Tukey; the non-lazy field
    recentres (|x| <= p/2 + 1) after block A (ba stages) and block B (logn - ba - 3 stages)"""
    p, b = P[field], B[field]
    d = 2.0 ** (logB - 1)
    m = d * (1 + b * b) + d * (b + b ** 3)
    worst = m
    for stage in range(3, logn + 1):
        m = m + mulmod_bound(m, p)
        worst = max(worst, m)
        assert m < LIMIT
        if not lazy and stage == logn - 3:   # (no recentring after block A since round 4: SKIP_T1 in ntt_forward)
            m = p / 2 + 1
    return m, worst


def inverse_ok(field, logn, ba):
    """inverse from |x| <= p/2: blocks of 3 (or 4, recentred after two stages in the 51-bit field) Gentleman-Sande stages,
    recentred at the transposes"""
    p = P[field]
    for block in (3, logn - ba - 3, ba):
        m = p / 2 + 1
        for s in range(block):
            if block == 4 and s == 2 and field == "FpH":
                m = p / 2 + 1
            assert 2 * m < LIMIT
            mulmod_bound(2 * m, p)
            m = 2 * m  # the pure-sum path; the multiplied path is smaller
    return True


@pytest.mark.parametrize("name,field,logn,k,l,logB", [
    ("boolean_default", "FpG", 9, 2, 3, 6),
    ("toy_k2", "FpG", 9, 2, 3, 6),
    ("boolean_default in the 51-bit field (HELM_HIP_FIELD=51)", "FpH", 9, 2, 3, 6),
    ("toy (k = 1, l = 2, logB = 8)", "FpH", 9, 1, 2, 8),
    ("helm_cuda", "FpH", 10, 1, 3, 7),
    ("largest digits the lazy field is chosen for", "FpG", 9, 1, 2, 12),
    ("helm_cuda in the lazy field of round 5 (the loaded key's own bound)", "FpI", 10, 1, 3, 7),
    ("N = 1024, l = 2 in the lazy field", "FpI", 10, 1, 2, 7),
])
def test_every_sum_of_a_cmux_step_stays_exact(name, field, logn, k, l, logB):
    p = P[field]
    lazy = field in ("FpG", "FpI")
    ba = 3 if logn == 9 else 4
    # the set's exact products fit the field (helm_hip_ctx_create's own check, restated)
    exact = (k + 1) * l * (1 << logn) * 2.0 ** (logB - 1) * 2.0 ** 31
    if field == "FpI":
        # not by the worst case (every key coefficient at 2^31, signs aligned) but by the loaded key's own l1-norms
        # (helm_hip_load_bootstrap_key): a key of uniform masks has mean |coefficient| 2^30 - half the worst case - and
        # the norm of (k+1) l N of them concentrates within 0.1 %
        typical = exact / 2
        assert typical * 1.002 < p / 2 and (l < 3 or exact > p / 2)
    elif name != "largest digits the lazy field is chosen for":
        assert exact * 1.0001 < p / 2
    out, worst = forward_bound(field, logn, logB, lazy, ba)
    prod = mulmod_bound(out, p)                       # one product spectrum x key
    if lazy:
        column = (k + 1) * l * prod                   # all (k+1) l products of a column summed raw (hand-over / ds_add_f64)
    else:
        column = max(l * prod, (k + 1) * l * (p / 2 + 1))  # a wave sums its l raw, the hand-over recentres each first
    assert column < LIMIT, (name, column / p)
    assert inverse_ok(field, logn, ba)
    print(f"\n{name} [{field}]: forward <= {out / p:.2f} p (largest intermediate {worst / p:.2f} p), product <= {prod / p:.2f} p, "
          f"column sum <= {column / p:.2f} p of 2^53 = {LIMIT / p:.2f} p")
    if field == "FpG" and logB == 6:
        assert out / p < 5.0 and column / p < 9.5     # the figures DESIGN.md 4.2 "The field" quotes
    if field == "FpI" and l == 3:
        assert out / p < 6.9 and column / p < 9.1 and LIMIT / p > 10.2   # ntt_fp64.h's comment on FpI


def test_top2_terms_are_exact_and_tiny():
    """digit x b^k for k <= 3 stays far below 2^53 and, for the lazy field's sets, below p/2 (no reduction needed)"""
    for field in P:
        b, p = B[field], P[field]
        assert b ** 4 + 1 == p and (p - 1) % (2 * 1024 if field == "FpG" else 2 * 2048) == 0   # (FpI: 2^24 | p - 1)
        for logB in (6, 7, 8, 12):
            d = 2 ** (logB - 1)
            total = d * (1 + b * b) + d * (b + b ** 3)
            assert total < 2 ** 53
            if logB <= 8:
                assert total < 0.05 * p


def test_shortint_stage1_product_is_exact_and_centred():
    """64-bit engine: digit (|d| <= 2^23, pbs_logB <= 24) x psi^(N/2) = +-b^2 is exact and inside (-p/2, p/2)"""
    for b in (5072, 5096):
        p = b ** 4 + 1
        assert (p - 1) % 4096 == 0                    # N = 2048
        assert 2 ** 23 * b * b < p / 2 < 2 ** 53
    assert (5072 ** 4 + 1) * (5096 ** 4 + 1) / 2 > 2.0 ** 97.35   # the range rounds 1-3 had


def test_top2_formulas_equal_two_cooley_tukey_stages():
    """fwd_top2_digits (ntt_fp64.h) restated in integers: the radix-4 butterfly on (d0, d2, d4, d6) with the constants b, b^2,
    b^3 gives the same residues as stage 1 (twiddle psi^(N/2) = b^2) followed by stage 2 (twiddles psi^(N/4) = b for the
    first half, psi^(3N/4) = b^3 for the second) of the merged negacyclic transform."""
    import random
    rnd = random.Random(5)
    for field in P:
        p, b = P[field], B[field]
        assert pow(b, 4, p) == p - 1 and pow(b, 8, p) == 1          # a primitive eighth root of unity
        for _ in range(200):
            d0, d2, d4, d6 = (rnd.randint(-64, 64) for _ in range(4))
            # two Cooley-Tukey stages
            a0, a4 = d0 + b * b * d4, d0 - b * b * d4
            a2, a6 = d2 + b * b * d6, d2 - b * b * d6
            want = (a0 + b * a2, a0 - b * a2, a4 + b ** 3 * a6, a4 - b ** 3 * a6)
            # the kernel's form: every term digit x (b, b^2, b^3); b^5 = -b
            u, v = d2 * b + d6 * b ** 3, d2 * b ** 3 + d6 * b
            got = (a0 + u, a0 - u, a4 + v, a4 - v)
            assert all((g - w) % p == 0 for g, w in zip(got, want))
            assert all(abs(g) < 2 ** 53 and abs(g) < p // 2 for g in got)   # exact doubles, already centred


def test_lean_inverse_transform_bounds():
    """ntt_inverse's LEAN form (lazy fields, N = 512): recentre slots 0-2 at the first transpose and 0-4 at the second.  A
    three-stage Gentleman-Sande block leaves slot e with the sum of everything (e = 0), products followed by two, one or no
    additions (e = 1; 2, 3; 4-7); after a transpose a lane holds eight values of one slot class, so every slot of the next
    block is bounded by the largest value left unreduced."""
    for p in (5072 ** 4 + 1, 5096 ** 4 + 1):

        def block(m):
            m, peak = list(m), 0.0
            for eb in range(3):
                new = m[:]
                for e0 in range(8):
                    if (e0 >> eb) & 1:
                        continue
                    e1 = e0 | (1 << eb)
                    s = m[e0] + m[e1]
                    assert s < LIMIT
                    peak = max(peak, s)
                    new[e0], new[e1] = s, mulmod_bound(s, p)
                m = new
            return m, peak

        half = p / 2 + 1
        m1, pk1 = block([half] * 8)
        assert [round(v / p, 1) for v in m1[:4]] == [4.0, 2.4, 1.4, 1.3]
        m2, pk2 = block([max(half if e < 3 else m1[e] for e in range(8))] * 8)
        m3, pk3 = block([max(half if e < 5 else m2[e] for e in range(8))] * 8)
        assert max(pk1, pk2, pk3) < 0.78 * LIMIT          # 0.75 (FpG), 0.76 (FpG2)
        assert max(m3) < 7 * p                             # the final recentring sees at most 6.7 p


def test_46_bit_pair_of_the_binarys_lut_set_stays_exact_without_recentring():
    """Round 6: k_pbs64k under PARAM_MESSAGE_1_CARRY_1_KS_PBS's dimensions (reference src/bin/helm.rs:301: k = 3, N = 512, one level
    of 18 bits) in the 46-bit pair FpJ = 2736^4 + 1, FpJ2 = 2872^4 + 1 (2^53 / p = 160, 132), chosen when the LOADED key's
    exact products fit p p' / 2.  Recomputed here: the plain radix-4 top on 17-bit digits is exact; the whole forward
    transform, the products and the column sums of k + 1 = 4 products stay exact with NO recentring; the inverse transform
    (ntt_inverse, WIDE) takes the column sums unreduced (<= 4.5 p) and recentres slot 0 alone at each of its two transposes."""
    import math
    for b in (2736, 2872):
        p = b ** 4 + 1
        assert (p - 1) % 1024 == 0 and pow(b, 4, p) == p - 1          # 2N = 1,024 divides p - 1; b a primitive eighth root
        assert LIMIT / p > 128
        d = 2.0 ** 17                                                 # |digit| <= B / 2, pbs_logB = 18
        m = d * (1 + b * b) + d * (b + b ** 3)                        # fwd_top2_digits: exact terms
        assert d * b ** 3 < LIMIT and m < 2.0 ** 52                   # what HELM_CHECK_BOUNDS counts against
        for stage in range(3, 10):                                    # seven Cooley-Tukey stages, nothing recentred
            m = m + mulmod_bound(m, p)
            assert m < LIMIT
        assert m / p < 60
        prod = mulmod_bound(m, p)
        assert prod / p < 1.11                                        # (1.01 p in FpJ, 1.10 p in FpJ2)
        col = 4 * prod                                                # k + 1 = 4 rows meet in a column (gather form)
        assert col / p < 4.5                                          # the bound ntt_inverse's WIDE form is entered with

        def block(vals):
            vals, peak = list(vals), 0.0
            for eb in range(3):
                new = vals[:]
                for e0 in range(8):
                    if (e0 >> eb) & 1:
                        continue
                    e1 = e0 | (1 << eb)
                    s = vals[e0] + vals[e1]
                    assert s < LIMIT
                    peak = max(peak, s)
                    new[e0], new[e1] = s, mulmod_bound(s, p)
                vals = new
            return vals, peak

        half = p / 2 + 1
        m1, pk1 = block([4.5 * p] * 8)                                # inputs: the unreduced column sums
        assert m1[0] / p < 36.1 and max(m1[1:]) / p < 2.6
        m2, pk2 = block([max(half if e < 1 else m1[e] for e in range(8))] * 8)   # slot 0 recentred at the first transpose
        m3, pk3 = block([max(half if e < 1 else m2[e] for e in range(8))] * 8)   # ... and at the second
        assert max(pk1, pk2, pk3) / p < 36.1 and max(pk1, pk2, pk3) < 0.28 * LIMIT
        assert max(m3) / p < 20                                       # the final recentring (CENTRE) sees at most that
    # the pair covers a generated key of the set (about 2^90.0) and not its worst case (2^91): the key decides
    pp_half = (2736 ** 4 + 1) * (2872 ** 4 + 1) / 2
    typical = 4 * 1 * 512 * 2.0 ** 62 * 2.0 ** 17
    worst = 4 * 1 * 512 * 2.0 ** 63 * 2.0 ** 17
    assert typical * 1.3 < pp_half < worst and abs(math.log2(pp_half) - 90.62) < 0.01
