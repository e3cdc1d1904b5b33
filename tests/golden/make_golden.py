"""Generates tests/golden/gates_toy.npz: a tiny parameter set (n=4, N=512, k=1), its keys,
two input ciphertexts and the expected output ciphertext of every boolean gate, produced
by the CPU oracle's SCHOOLBOOK route (wrapping u32 arithmetic, no transform at all).

The reference (Rust + un-vendored tfhe crate) cannot run here, so these vectors pin the
oracle's NTT route and the HIP path to the schoolbook definition, not to tfhe-rs bits
(see oracle/tfhe_oracle.c header: ciphertext-level parity with tfhe-rs is unpinned).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import helm_amd  # noqa: E402
import oracle  # noqa: E402
from helm_amd._native import Params  # noqa: E402

p = Params(torus_bits=32, n=4, k=1, N=512, pbs_l=2, pbs_logB=8, ks_l=4, ks_logB=4, pbs_order=0, grouping_factor=1)
ck = helm_amd.ClientKey(p, 1e-7, 1e-9, seed=2024)
orc = oracle.Oracle(p.as_tuple7(), ck.bsk, ck.ksk, use_ntt=False)
cts = ck.encrypt([False, True])
ops, i0, i1, i2 = [], [], [], []
for op in (oracle.AND, oracle.OR, oracle.NAND, oracle.NOR, oracle.XOR, oracle.XNOR):
    for a in (0, 1):
        for b in (0, 1):
            ops.append(op); i0.append(a); i1.append(b); i2.append(-1)
for s in (0, 1):
    for a in (0, 1):
        for b in (0, 1):
            ops.append(oracle.MUX); i0.append(a); i1.append(b); i2.append(s)
for a in (0, 1):
    ops.append(oracle.NOT); i0.append(a); i1.append(-1); i2.append(-1)
wires = np.zeros((2 + len(ops), p.n + 1), dtype=np.uint32)
wires[:2] = cts
outs = np.arange(2, 2 + len(ops), dtype=np.int32)
orc.eval_level(wires, ops, i0, i1, i2, outs)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "gates_toy.npz"),
                    params=np.array(p.as_tuple7(), dtype=np.int32), lwe_sk=ck.lwe_secret.copy(),
                    glwe_sk=ck.glwe_secret.copy(), bsk=ck.bsk.copy(), ksk=ck.ksk.copy(), inputs=cts,
                    ops=np.array(ops, np.int32), in0=np.array(i0, np.int32), in1=np.array(i1, np.int32),
                    in2=np.array(i2, np.int32), expected=wires[2:])
print("wrote gates_toy.npz:", len(ops), "gates")
