#!/usr/bin/env python3
"""How the fields of round 4 were found: primes p = b^4 + 1 (b = 8 m, so that 2^12 | p - 1) in a size window.

In such a field b is a primitive eighth root of unity (b^4 = -1) and b, b^2, b^3 are short integers: the roots of unity of
the first two stages of a negacyclic transform.  On decomposition digits those stages then need no modular reduction
(helm_amd/csrc/ntt_fp64.h: fwd_top2_digits; DESIGN.md 4.2 "The field").  Windows:
  lazy 49-bit field of the boolean kernels: p/2 above tfhe boolean DEFAULT's exact products (3 * 3 * 512 * 32 * 2^31 = 2^48.17)
                                            and 2^53 / p >= 12 (hand-over sums of nine products without recentring)
  51-bit field: p/2 above helm_cuda's exact products (2 * 3 * 1024 * 64 * 2^31), 2^53 / p >= 5.2
Also prints why there is no field with short SIXTEENTH roots (c^8 + 1 prime) in the lazy window.
usage: find_short_root_primes.py"""


def is_prime(n):
    if n < 2:
        return False
    for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % q == 0:
            return n == q
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):  # deterministic below 3.3e24
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def generator(p):
    g = 2
    while pow(g, (p - 1) // 2, p) != p - 1:
        g += 1
    return g  # a quadratic non-residue: g^((p-1)/2N) has order exactly 2N for every power of two 2N | p - 1


def window(lo, hi, what):
    print(f"{what}: b^4 + 1 prime with {lo:.4e} < p < {hi:.4e}")
    m = 1
    while (8 * m) ** 4 + 1 < hi:
        b = 8 * m
        p = b ** 4 + 1
        if p > lo and is_prime(p):
            v = p - 1
            two = 0
            while v % 2 == 0:
                v //= 2
                two += 1
            print(f"  b = {b:5d}  p = {p}  = 2^{__import__('math').log2(p):.3f}   2^53 / p = {2 ** 53 / p:.2f}   2^{two} | p - 1   "
                  f"generator {generator(p)}   b^2 = {b * b}  b^3 = {b ** 3}")
        m += 1


if __name__ == "__main__":
    bool_exact = 3 * 3 * 512 * 32 * 2 ** 31
    window(2 * bool_exact * 1.002, 2 ** 53 / 11.5, "lazy field (boolean kernels, CRT pair of the 64-bit kernels)")
    cuda_exact = 2 * 3 * 1024 * 64 * 2 ** 31
    window(2 * cuda_exact * 1.0001, 2 ** 53 / 5.2, "51-bit field")
    print("c^8 + 1 in the lazy window:", [(c, is_prime(c ** 8 + 1)) for c in range(70, 76, 2)], "(72^8 + 1 is the only candidate, composite)")
