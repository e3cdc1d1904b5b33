"""The debug build of the boolean engine (csrc/libhelm_hip_check.so, -DHELM_CHECK_BOUNDS) under the parity suite's workloads:
every contract of the lazy modular arithmetic in ntt_fp64.h - operands of modular multiplications and recentrings below
2^53, butterfly sums below 2^53, the plain short-root stages inside (-p/2, p/2), the LEAN inverse transform entered with
recentred inputs (|x| <= p/2), lifted values inside to_torus32's range - is checked and COUNTED by the kernels themselves.
A change of layout or of a caller that overflowed would otherwise only show as a wrong ciphertext somewhere (advisor, round
4).  The library is chosen at import (HELM_HIP_LIB), so the workload runs in a child process.  Gate semantics: reference
src/gates.rs:254-275; the checked arithmetic replaces the tfhe crate's bootstrap behind them."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
import helm_amd, oracle
res = {}
for name, B in (("toy_k2", 40), ("boolean_default", 300), ("helm_cuda", 40)):
    ck = helm_amd.ClientKey.generate(name, seed=3)
    sk = helm_amd.ServerKey(ck, device=0)
    if name == "toy_k2":
        res["selftest"] = sk.bound_violations(reset=True, selftest=True)
    sk.bound_violations(reset=True)
    rng = np.random.default_rng(1)
    bits = rng.integers(0, 2, size=3 * B).astype(bool)
    w = sk.wires(4 * B)
    w.upload(np.arange(3 * B), ck.encrypt(bits))
    ops = np.array([0, 3, 4, 5, 7, 8, 9] * B, dtype=np.int32)[:B]     # AND MUX NAND NOR OR XNOR XOR
    i0, i1, i2 = np.arange(B), np.arange(B, 2 * B), np.where(ops == 3, np.arange(2 * B, 3 * B), -1)
    out = np.arange(3 * B, 4 * B)
    w.eval_gate_level(ops, i0, i1, i2, out)     # B = 300 at boolean_default: one bootstrap per CU and more -> wide AND duo / lockstep builds
    sk.sync()
    got = w.download(out)
    a, b, c = bits[:B], bits[B:2 * B], bits[2 * B:]
    want = {0: a & b, 3: np.where(c, a, b), 4: ~(a & b), 5: ~(a | b), 7: a | b, 8: ~(a ^ b), 9: a ^ b}
    plain = np.array([want[int(o)][g] for g, o in enumerate(ops)])
    ok = bool(np.array_equal(ck.decrypt(got), plain))
    # bit-exact against the oracle on a sample (the check build computes the same integers)
    orc = oracle.Oracle(ck.params.as_tuple7(), ck.bsk, ck.ksk, use_ntt=True)
    host = np.zeros((4 * B, ck.params.n + 1), dtype=np.uint32)
    host[:3 * B] = w.download(np.arange(3 * B))
    s = np.arange(0, B, max(1, B // 12))
    orc.eval_level(host, ops[s], i0[s], i1[s], i2[s], out[s])
    res[name] = {"decrypt_ok": ok, "bit_exact_sample": bool(np.array_equal(host[out[s]], got[s])), "violations": sk.bound_violations()}
    sk.close()
print("RESULT " + json.dumps(res))
"""


def test_no_contract_of_the_lazy_arithmetic_is_broken_and_the_check_can_fire():
    lib = os.path.join(ROOT, "helm_amd", "csrc", "libhelm_hip_check.so")
    assert os.path.exists(lib), "make -C helm_amd/csrc libhelm_hip_check.so"
    env = dict(os.environ, HELM_HIP_LIB=lib)
    p = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert res["selftest"][0] == 1 and sum(res["selftest"][1:]) == 0, res["selftest"]   # the one contract broken on purpose
    for name in ("toy_k2", "boolean_default", "helm_cuda"):
        r = res[name]
        assert r["decrypt_ok"] and r["bit_exact_sample"], (name, r)
        assert r["violations"] == [0] * 8, (name, r["violations"])


def test_the_regular_build_says_it_has_no_counters():
    import helm_amd
    from helm_amd._native import HelmError
    ck = helm_amd.ClientKey.generate("toy_k2", seed=3)
    sk = helm_amd.ServerKey(ck, device=0)
    with pytest.raises(HelmError, match="HELM_CHECK_BOUNDS"):
        sk.bound_violations()
    sk.close()
