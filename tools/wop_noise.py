"""Noise margin of the WoP-PBS path at the full parameter sets: the phase error of the vertical-packing output (before the
final keyswitch + bootstrap cleans it) against half an encoding step, over `count` random 12-bit look-ups.
Usage: python tools/wop_noise.py [wop set] [count]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helm_amd
from helm_amd import wopbs
from helm_amd.shortint import si_named_params

name = sys.argv[1] if len(sys.argv) > 1 else "wopbs_m1c1"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 64
bits = 12
sp, sa, sb = si_named_params("shortint_m2c2")
wp, wa, wb = wopbs.wop_named_params(name)
sp.message_modulus, sp.carry_modulus = wp.message_modulus, wp.carry_modulus
ck = helm_amd.SiClientKey(sp, sa, sb, seed=1)
wk = wopbs.WopClientKey(ck, wp, wa, wb, seed=2)
sk = helm_amd.SiServerKey(ck)
wsk = wopbs.WopServerKey(sk, wk)
rng = np.random.default_rng(0)
t = wp.message_modulus * wp.carry_modulus
values = rng.integers(0, 1 << bits, size=count)
flat = np.array([(v >> i) & 1 for v in values for i in range(bits)], dtype=np.uint64)
# fresh encryptions of the bits under the WoP small key at the set's LWE noise (what bit extraction hands over is
# noisier: a keyswitch output; measured separately below through the whole gate)
lwe_sk = wk.lwe_secret.astype(bool)
cts = rng.integers(0, 1 << 64, size=(flat.size, wp.n + 1), dtype=np.uint64)
noise = np.rint(rng.normal(0.0, wa * 2.0 ** 64, size=flat.size)).astype(np.int64).astype(np.uint64)
cts[:, -1] = (cts[:, :-1] * lwe_sk).sum(axis=1, dtype=np.uint64) + (flat << np.uint64(63)) + noise
ggsw = wsk.circuit_bootstrap(cts)
ggsw = ggsw.reshape(count, bits, *ggsw.shape[1:])
tables = rng.integers(0, t, size=(count, 1 << bits), dtype=np.uint64) * np.uint64(wk.delta)
out = wsk.vertical_packing(ggsw, tables)
err = (wk.phase(out) - tables[np.arange(count), values]).astype(np.int64).astype(np.float64)
half = wk.delta / 2
print(f"{name}: {count} look-ups of {bits} bits: phase error rms 2^{np.log2(np.sqrt((err ** 2).mean())):.1f}, "
      f"max 2^{np.log2(np.abs(err).max()):.1f}, half an encoding step 2^{np.log2(half):.1f} "
      f"-> max / half = {np.abs(err).max() / half:.4f}")
