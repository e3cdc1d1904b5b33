"""Full-size parity of the 64-bit-torus engine, bit for bit against the oracle's Goldilocks-NTT route (round 5).

(1) Whole levels at the three full parameter sets - the launch shapes bench.py times (1,024 three-input LUTs under
    PARAM_MESSAGE_2_CARRY_2 and the multi-bit set, 2,048 two-input LUTs under PARAM_MESSAGE_1_CARRY_1) - evaluated by the
    GPU as ONE level; the oracle recomputes the rows at the positions where a kernel could go wrong without a small batch
    noticing: first and last workgroup, both sides of every boundary of the matrix-core keyswitch's 64-ciphertext tiles and
    of the round structure (a CU-count multiple), the two ciphertexts of a CU where two are resident, and a batch of one.
(2) Whole EVALUATIONS through the evaluator API with the engine's audit hook (helm_si_set_audit): every linear step and every
    look-up batch the evaluator issues hands its operand rows and results to the host; the oracle recomputes each batch from
    the GPU's own operands.  Every batch equal => every wire equal to what the oracle would compute for the whole circuit:
    the 8-bit LUT-3-1 adder (BASELINE config 3; reference tests/circuit_test.rs:266-311) and chi-squared u32 (config 5,
    reference src/bin/helm.rs:83's set): every linear step, and by default every row of the first three look-up batches and of
    every third one after them (a third of the 2,845 look-ups; HELM_TEST_FULL_AUDIT=1: all of them, about 70 s of oracle
    time on 16 cores - the default pays for the full AES-128 evaluation of tests/test_gpu_parity.py instead;
    HELM_TEST_SAMPLED_AUDIT=1: the first, the last and every eighth row of each batch).
Reference: gates::lut() src/gates.rs:754-785, the FheUintN operators src/gates.rs:331-701."""
import os
import threading

import numpy as np
import pytest

import helm_amd
import oracle
from helm_amd import ArithCircuit, Circuit, EvalCircuit, LutCircuit, PtxtType, verilog_parser

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
NET = os.path.join(HERE, "netlists")


def _positions(B, per_round):
    """Rows worth checking in a batch of B: ends, tile edges of the 64-ciphertext keyswitch tiles, round boundaries."""
    pos = {0, 1, B - 2, B - 1, B // 2}
    for edge in list(range(64, B, 64))[:3] + list(range(64, B, 64))[-2:] + list(range(per_round, B, per_round)):
        pos |= {edge - 1, edge}
    rng = np.random.default_rng(B)
    pos |= {int(x) for x in rng.integers(0, B, 6)}
    return np.array(sorted(p for p in pos if 0 <= p < B), dtype=np.int32)


@pytest.mark.parametrize("name,B,arity,table", [("shortint_m2c2", 1024, 3, 0xE8), ("shortint_m2c2_multibit3", 1024, 3, 0x96),
                                                ("shortint_m1c1", 2048, 2, 0x6)])
def test_whole_level_at_the_full_sets_bit_exact(name, B, arity, table):
    ck = helm_amd.SiClientKey.generate(name, seed=1)
    sk = helm_amd.SiServerKey(ck)
    orc = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk, use_ntt=True)
    bits = np.random.default_rng(0).integers(0, 2, size=arity * B).astype(np.uint64)
    host = np.zeros(((arity + 1) * B + 2, ck.dim + 1), dtype=np.uint64)
    host[:arity * B] = ck.encrypt(bits)
    w = sk.wires(len(host))
    w.upload(np.arange(arity * B), host[:arity * B])
    in_idx = np.arange(arity * B, dtype=np.int32).reshape(arity, B).T.copy()
    ar, tb, out = np.full(B, arity, np.int32), np.full(B, table, np.uint64), np.arange(arity * B, (arity + 1) * B, dtype=np.int32)
    w.eval_lut_level(ar, in_idx, tb, out)
    sk.sync()
    got = w.download(out)
    # decrypted: the whole level
    b = bits.reshape(arity, B)
    idx = sum(b[q].astype(np.int64) << (arity - 1 - q) for q in range(arity))
    assert np.array_equal(ck.decrypt(got), (table >> idx) & 1)
    # bit for bit: the positions that matter
    per_round = int(sk.round_capacity())
    pos = _positions(B, per_round)
    want = orc.eval_lut_rows(host, ar, in_idx, tb, pos)
    for q, g in enumerate(pos):
        assert np.array_equal(got[g], want[q]), f"{name}: row {g} of the {B}-LUT level differs from the oracle"
    # the batch of one (its own dispatch shape: one workgroup, the sliced vector-ALU keyswitch)
    one_out = np.array([(arity + 1) * B], dtype=np.int32)
    w.eval_lut_level(ar[:1], in_idx[7:8], tb[:1], one_out)
    sk.sync()
    assert np.array_equal(w.download(one_out)[0], orc.eval_lut_rows(host, ar[:1], in_idx[7:8], tb[:1], [0])[0])
    assert np.array_equal(w.download(one_out)[0], got[7])  # and the same ciphertext as row 7 of the wide level
    sk.close()


class Auditor:
    """fn for SiServerKey.set_audit: recomputes every batch on the oracle from the GPU's own operand rows."""

    def __init__(self, ck, orc, every=1, batch_every=1):
        """every: rows of a look-up batch recomputed (1 = all; k = the first, the last and every k-th).  batch_every: look-up
        batches recomputed IN FULL (1 = all; k = the first three, then every k-th); linear steps are always checked."""
        self.ck, self.orc, self.every, self.batch_every = ck, orc, every, batch_every
        self.lut_batches = 0
        self.delta = np.uint64(orc.delta)
        self.lock = threading.Lock()
        self.bad, self.luts_checked, self.luts_seen, self.lin_checked, self.batches = [], 0, 0, 0, 0

    def __call__(self, rec):
        with self.lock:
            self.batches += 1
            n = self.batches
        if rec["kind"] == "lincomb":
            cnt, terms, _ = rec["in_rows"].shape
            acc = np.zeros_like(rec["out_rows"])
            with np.errstate(over="ignore"):
                for t in range(terms):
                    use = rec["in_idx"][:, t] >= 0
                    acc[use] += rec["coef"][use, t].astype(np.uint64)[:, None] * rec["in_rows"][use, t]
                if rec["const_add"] is not None:
                    acc[:, -1] += rec["const_add"].astype(np.uint64) * self.delta
            ok = np.all(acc == rec["out_rows"], axis=1)
            with self.lock:
                self.lin_checked += cnt
                self.bad += [("lincomb", n, int(g)) for g in np.nonzero(~ok)[0]]
            return True
        cnt = len(rec["lut_idx"])
        with self.lock:
            self.lut_batches += 1
            nb = self.lut_batches
            self.luts_seen += cnt
        if self.batch_every > 1 and nb > 3 and nb % self.batch_every:
            return True
        rows = np.arange(cnt) if self.every == 1 else np.unique(np.concatenate([[0, cnt - 1], np.arange(n % self.every, cnt, self.every)]))
        want = self.orc.apply_luts(rec["in_rows"][rows], rec["luts"], rec["lut_idx"][rows])
        ok = np.all(want == rec["out_rows"][rows], axis=1)
        with self.lock:
            self.luts_checked += len(rows)
            self.bad += [("luts", n, int(rows[g])) for g in np.nonzero(~ok)[0]]
        return True


def _circuit(path, is_arith):
    gs, ws, ins, outs, d, _, _ = verilog_parser.read_verilog_file(path, is_arith)
    c = Circuit(gs, ins, outs, d)
    c.sort_circuit()
    c.compute_levels()
    return c, ws


def test_lut_adder_every_operation_bit_exact_at_the_full_set():
    """BASELINE config 3 under PARAM_MESSAGE_2_CARRY_2: every linear step and every look-up of the whole evaluation."""
    ck, sk = helm_amd.gen_keys_shortint("shortint_m2c2", seed=1)
    orc = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk, use_ntt=True)
    aud = Auditor(ck, orc)
    sk.set_audit(aud)
    c, ws = _circuit(os.path.join(NET, "8-bit-adder-lut-3-1.v"), False)
    a, b, cin = 0xB7, 0x6E, 1
    inputs = {f"a[{i}]": PtxtType.Bool((a >> i) & 1) for i in range(8)}
    inputs.update({f"b[{i}]": PtxtType.Bool((b >> i) & 1) for i in range(8)})
    inputs["cin"] = PtxtType.Bool(cin)
    ptxt = c.evaluate(c.initialize_wire_map(ws, inputs, "bool"))
    lc = LutCircuit(ck, sk, c)
    enc = EvalCircuit.evaluate_encrypted(lc, EvalCircuit.encrypt_inputs(lc, ws, inputs), 1, "bool")
    sk.set_audit(None)
    assert not aud.bad, aud.bad[:5]
    assert aud.luts_checked == aud.luts_seen == lc.pbs_per_cycle() == 16 and aud.lin_checked >= 16
    for wire, want in ptxt.items():
        assert ck.decrypt(enc[wire]) == int(bool(want)), wire
    sk.close()


def test_chi_squared_u32_every_batch_bit_exact_under_the_references_set():
    """BASELINE config 5 under PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3's dimensions (helm.rs:83): the radix operators'
    whole op stream - carry-save products, grouped carry propagation, merged rounds of the two sub-circuits."""
    ck, sk = helm_amd.gen_keys_shortint("shortint_m2c2_multibit3", seed=1)
    orc = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk, use_ntt=True)
    full = os.environ.get("HELM_TEST_FULL_AUDIT") == "1"
    aud = Auditor(ck, orc, every=8 if os.environ.get("HELM_TEST_SAMPLED_AUDIT") == "1" else 1, batch_every=1 if full else 3)
    sk.set_audit(aud)  # before the evaluator forks its lanes
    c, ws = _circuit(os.path.join(NET, "chi_squared_arith.v"), True)
    ac = ArithCircuit(ck, sk, c)
    enc = ac.encrypt_inputs(ws, {"N0": PtxtType.U32(2), "N1": PtxtType.U32(7), "N2": PtxtType.U32(9)})
    out = ac.evaluate_encrypted(enc, 1, "u32")
    sk.set_audit(None)
    dec = {k: int(v.value) for k, v in ac.decrypt_outputs(out, True).items()}
    assert dec == {"alpha": 529, "beta1": 242, "beta2": 275, "beta3": 1250}
    assert not aud.bad, aud.bad[:5]
    assert aud.luts_seen == ac.pbs_per_cycle() > 2000, (aud.luts_seen, ac.pbs_per_cycle())
    assert aud.luts_checked == aud.luts_seen or not full or os.environ.get("HELM_TEST_SAMPLED_AUDIT") == "1"
    assert aud.luts_checked >= aud.luts_seen // 24 and aud.lin_checked > 0
    if not full and os.environ.get("HELM_TEST_SAMPLED_AUDIT") != "1":
        assert aud.luts_checked >= aud.luts_seen // 4, (aud.luts_checked, aud.luts_seen)
    sk.close()


def test_fheuint16_known_answers_every_operation_bit_exact():
    """K-7 (reference tests/gates_test.rs:127-310: FheUint16 10 + 20, 20 - 10, 10 x 20 and the scalar forms) under
    PARAM_MESSAGE_2_CARRY_2's full dimensions: every linear step and every look-up of the evaluation against the oracle."""
    ck, sk = helm_amd.gen_keys_shortint("shortint_m2c2", seed=1)
    orc = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk, use_ntt=True)
    aud = Auditor(ck, orc)
    sk.set_audit(aud)
    text = """input [15:0] A, B;
output [15:0] S, D, P, Q, R;
add g0(A, B, S);
sub g1(B, A, D);
mult g2(A, B, P);
add g3(A, 7, Q);
sub g4(B, 3, R);
"""
    gs, ws, ins, outs, d, _, _ = verilog_parser.read_verilog_text(text, True)
    c = Circuit(gs, ins, outs, d)
    c.sort_circuit()
    c.compute_levels()
    ac = ArithCircuit(ck, sk, c)
    out = ac.decrypt_outputs(ac.evaluate_encrypted(ac.encrypt_inputs(ws, {"A": PtxtType.U16(10), "B": PtxtType.U16(20)}), 1, "u16"), True)
    sk.set_audit(None)
    assert (out["S"], out["D"], out["P"], out["Q"], out["R"]) == (PtxtType.U16(30), PtxtType.U16(10), PtxtType.U16(200), PtxtType.U16(17),
                                                                 PtxtType.U16(17))
    assert not aud.bad, aud.bad[:5]
    assert aud.luts_checked == aud.luts_seen == ac.pbs_per_cycle() > 100 and aud.lin_checked > 0
    sk.close()


def test_every_radix_operator_u8_every_operation_bit_exact():
    """All thirteen FheUintN operators of arithmetic mode (reference src/gates.rs:331-701: + - * / << >> with encrypted and plain
    right-hand sides, copy) on FheUint8 operands at PARAM_MESSAGE_2_CARRY_2's full dimensions - the restoring division and the
    shifts by an encrypted amount have round structures of their own - every linear step and every look-up against the oracle."""
    ck, sk = helm_amd.gen_keys_shortint("shortint_m2c2", seed=1)
    orc = oracle.Oracle64(ck.params.as_tuple(), ck.bsk, ck.ksk, use_ntt=True)
    aud = Auditor(ck, orc)
    sk.set_audit(aud)
    text = """input [7:0] A, B;
output [7:0] S, D, P, Q, L, R, SA, SS, SM, SQ, SL, SR, C;
add g0(A, B, S);
sub g1(A, B, D);
mult g2(A, B, P);
div g3(A, B, Q);
shl g4(A, B, L);
shr g5(A, B, R);
add g6(A, 77, SA);
sub g7(A, 77, SS);
mult g8(A, 77, SM);
div g9(A, 11, SQ);
shl g10(A, 5, SL);
shr g11(A, 5, SR);
copy g12(A, C);
"""
    gs, ws, ins, outs, d, _, _ = verilog_parser.read_verilog_text(text, True)
    c = Circuit(gs, ins, outs, d)
    c.sort_circuit()
    c.compute_levels()
    ac = ArithCircuit(ck, sk, c)
    a, b = 201, 6
    out = {k: v.value for k, v in ac.decrypt_outputs(ac.evaluate_encrypted(ac.encrypt_inputs(ws, {"A": PtxtType.U8(a), "B": PtxtType.U8(b)}), 1, "u8"), True).items()}
    sk.set_audit(None)
    assert out == {"S": (a + b) % 256, "D": (a - b) % 256, "P": a * b % 256, "Q": a // b, "L": (a << b) % 256, "R": a >> b,
                   "SA": (a + 77) % 256, "SS": (a - 77) % 256, "SM": a * 77 % 256, "SQ": a // 11, "SL": (a << 5) % 256, "SR": a >> 5, "C": a}, out
    assert not aud.bad, aud.bad[:5]
    assert aud.luts_checked == aud.luts_seen == ac.pbs_per_cycle() > 200 and aud.lin_checked > 0
    sk.close()
