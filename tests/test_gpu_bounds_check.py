"""The debug build of both engines (csrc/libhelm_hip_check.so, -DHELM_CHECK_BOUNDS) under the parity suite's workloads:
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
# helm_cuda (round 5: the lazy field FpI, bounds closer to 2^53 than FpG's): 40 gates -> the wide build, 1250 (1,428 bootstraps
# on 256 CUs) -> a lockstep round + k_pbs_duo's compact layout, 1400 (1,600) -> a lockstep round + k_pbs_tri10 (round 6), 1700
# (1,943) -> two lockstep rounds, the last workgroups partial
for name, B in (("toy_k2", 40), ("boolean_default", 300), ("helm_cuda", 40), ("helm_cuda", 1250), ("helm_cuda", 1400), ("helm_cuda", 1700)):
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
    res[f"{name}:{B}"] = {"decrypt_ok": ok, "bit_exact_sample": bool(np.array_equal(host[out[s]], got[s])), "violations": sk.bound_violations(),
                          "field": sk.field_bits()}
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
    for name in ("toy_k2:40", "boolean_default:300", "helm_cuda:40", "helm_cuda:1250", "helm_cuda:1400", "helm_cuda:1700"):
        r = res[name]
        assert r["decrypt_ok"] and r["bit_exact_sample"], (name, r)
        assert r["violations"] == [0] * 8, (name, r["violations"])
        assert r["field"] == (50 if name.startswith("helm_cuda") else 49), (name, r["field"])


CHILD64 = r"""
import json, sys
import numpy as np
sys.path.insert(0, %r)
import helm_amd, oracle
res = {}
# every kernel family of the 64-bit engine: k_pbs64 (N <= 1024), k_pbs64s (N = 2048, one and two levels), the multi-bit walks
# (g = 2, 3), k_pbs64k (k = 2, 3), the three full sets the benchmarks run - and one wide gate through the WoP-PBS path
for name, full in (("si_toy_512", 0), ("si_toy_1024", 0), ("si_toy_2048", 0), ("si_toy_2048_l2", 0), ("si_toy_1024_mb2", 0),
                   ("si_toy_2048_mb3", 0), ("si_toy_512_k3", 0), ("si_toy_512_k2", 0),
                   ("shortint_m2c2", 1), ("shortint_m1c1", 1), ("shortint_m2c2_multibit3", 1)):
    ck = helm_amd.SiClientKey.generate(name, seed=3)
    sk = helm_amd.SiServerKey(ck)
    sk.bound_violations(reset=True)
    B = 2 * ck.t if not full else 48
    vals = (np.arange(B) %% ck.t).astype(np.uint64)
    w = sk.wires(2 * B)
    w.upload(np.arange(B), ck.encrypt(vals))
    lut = sk.make_lut(lambda x: (3 * x + 1) %% ck.t)
    w.apply_luts(np.arange(B), lut, np.arange(B) + B)
    got = w.download(np.arange(B) + B)
    ok = [int(v) for v in ck.decrypt_message_and_carry(got)] == [int((3 * v + 1) %% ck.t) for v in vals]
    exact = None
    if not full:   # bit for bit on the toy sets (the full sets have tests/test_gpu_audit.py on the regular build)
        orc = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk)
        cts = w.download(np.arange(B))
        exact = bool(all(np.array_equal(got[g], orc.apply_lut(cts[g], lut)) for g in (0, B - 1)))
    res[name] = {"decrypt_ok": bool(ok), "bit_exact_sample": exact, "violations": sk.bound_violations()}
    sk.close()
from helm_amd import wopbs
from helm_amd.shortint import si_named_params
sp, a, b = si_named_params("si_toy_512")
wp, c, d = wopbs.wop_named_params("wop_toy_512")
ck = helm_amd.SiClientKey(sp, a, b, seed=5)
wk = wopbs.WopClientKey(ck, wp, c, d, seed=6)
sk = helm_amd.SiServerKey(ck)
wsk = wopbs.WopServerKey(sk, wk)
sk.bound_violations(reset=True)
rng = np.random.default_rng(2)
n_inputs, count = 3, 4
mm = int(ck.params.message_modulus)
truth = rng.integers(0, 2, size=(count, mm ** n_inputs), dtype=np.uint64)
xs = rng.integers(0, 1 << n_inputs, size=count)
bits_in = np.array([[(x >> (n_inputs - 1 - q)) & 1 for q in range(n_inputs)] for x in xs], dtype=np.uint64)
w = sk.wires(count * (n_inputs + 1))
w.upload(np.arange(count * n_inputs), ck.encrypt(bits_in.reshape(-1)))
in_idx = np.arange(count * n_inputs, dtype=np.int32).reshape(count, n_inputs)
out_idx = np.arange(count * n_inputs, count * (n_inputs + 1), dtype=np.int32)
wsk.eval_luts(w, in_idx, truth, out_idx, bits_per_block=1)
got = w.download(out_idx)
want = [int(truth[g, sum(int(bb) * mm ** j for j, bb in enumerate(bits_in[g][::-1]))]) for g in range(count)]
res["wop_toy_512"] = {"decrypt_ok": [int(v) for v in ck.decrypt_message_and_carry(got)] == want, "bit_exact_sample": None,
                      "violations": sk.bound_violations()}
wsk.close()
sk.close()
print("RESULT " + json.dumps(res))
"""


def test_the_64_bit_engine_and_the_wide_lut_path_keep_their_contracts_too():
    """The same counting build over the 64-bit engine's kernel families (every toy set = every (N, levels, k, grouping) build;
    the three full sets of the benchmarks) and one wide gate of the WoP-PBS path: zero violations, values right, toy sets bit
    for bit against the oracle.  Reference behaviour behind it: src/gates.rs:754-785 (LUT gates), :787-864 (wide LUTs)."""
    lib = os.path.join(ROOT, "helm_amd", "csrc", "libhelm_hip_check.so")
    assert os.path.exists(lib), "make -C helm_amd/csrc libhelm_hip_check.so"
    env = dict(os.environ, HELM_HIP_LIB=lib)
    p = subprocess.run([sys.executable, "-c", CHILD64 % ROOT], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert len(res) == 12
    for name, r in res.items():
        assert r["decrypt_ok"] and r["bit_exact_sample"] in (True, None), (name, r)
        assert r["violations"] == [0] * 8, (name, r["violations"])


def test_the_regular_build_says_it_has_no_counters():
    import helm_amd
    from helm_amd._native import HelmError
    ck = helm_amd.ClientKey.generate("toy_k2", seed=3)
    sk = helm_amd.ServerKey(ck, device=0)
    with pytest.raises(HelmError, match="HELM_CHECK_BOUNDS"):
        sk.bound_violations()
    sk.close()
    ck = helm_amd.SiClientKey.generate("si_toy_512", seed=3)
    sk = helm_amd.SiServerKey(ck)
    with pytest.raises(HelmError, match="HELM_CHECK_BOUNDS"):
        sk.bound_violations()
    sk.close()
