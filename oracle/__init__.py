"""CPU oracle (TEST INFRASTRUCTURE ONLY).

ctypes bindings of oracle/liborc.so, the plain-C restatement of the TFHE
gate-bootstrap that HELM reaches through `tfhe::boolean::ServerKey`
(reference src/gates.rs:254-275).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package; helm_amd/ never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

AND, DFF, LUT, MUX, NAND, NOR, NOT, OR, XNOR, XOR, BUF, CONST_ONE, CONST_ZERO = range(13)


class Params(C.Structure):
    _fields_ = [(f, C.c_int32) for f in ("n", "k", "N", "pbs_l", "pbs_logB", "ks_l", "ks_logB")]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so", "liborc64.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        u32p = C.POINTER(C.c_uint32)
        i32p = C.POINTER(C.c_int32)
        L.orc_bsk_ntt_new.restype = C.c_void_p
        L.orc_bsk_ntt_new.argtypes = [C.POINTER(Params), u32p]
        L.orc_bsk_ntt_free.argtypes = [C.c_void_p]
        L.orc_gate.argtypes = [C.POINTER(Params), u32p, C.c_void_p, u32p, C.c_int, u32p, u32p, u32p, u32p]
        L.orc_eval_level.argtypes = [C.POINTER(Params), u32p, C.c_void_p, u32p, u32p, i32p, i32p, i32p, i32p, i32p,
                                     C.c_int, C.c_int]
        L.orc_bootstrap_noks.argtypes = [C.POINTER(Params), u32p, C.c_void_p, u32p, u32p, u32p]
        L.orc_keyswitch.argtypes = [C.POINTER(Params), u32p, u32p, u32p]
        L.orc_gate_lincomb.argtypes = [C.c_int, C.c_int, C.c_int, u32p, u32p, u32p, u32p]
        L.orc_extprod_add.argtypes = [C.POINTER(Params), u32p, C.c_void_p, C.c_int, u32p, u32p]
        L.orc_phase.restype = C.c_uint32
        L.orc_phase.argtypes = [C.c_int, u32p, u32p]
        L.orc_modswitch.restype = C.c_uint32
        L.orc_modswitch.argtypes = [C.c_uint32, C.c_int]
        L.orc_decompose.argtypes = [C.c_uint32, C.c_int, C.c_int, i32p]
        L.orc_max_threads.restype = C.c_int
        pp = C.POINTER(u32p)
        L.orc_bsk_fp_new.restype = C.c_void_p
        L.orc_bsk_fp_new.argtypes = [C.POINTER(Params), u32p]
        L.orc_bsk_fp_free.argtypes = [C.c_void_p]
        L.orc_bsk_fp_prime.restype = C.c_double
        L.orc_bsk_fp_prime.argtypes = [C.c_void_p]
        L.orc_bootstrap_noks_fp.argtypes = [C.c_void_p, pp, u32p, pp, C.c_int]
        L.orc_eval_level_fp.argtypes = [C.POINTER(Params), C.c_void_p, u32p, u32p, i32p, i32p, i32p, i32p, i32p,
                                        C.c_int, C.c_int]
        L.orc_fp_lanes.restype = C.c_int
        L.orc_ntt_route.restype = C.c_char_p
        _LIB = L
    return _LIB


def _u32(a):
    if a is None:
        return None
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _i32(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Oracle:
    """Server-side evaluation with a given (bsk, ksk), both in the standard
    domain layouts documented in include/helm_hip.h."""

    def __init__(self, params, bsk_std, ksk, use_ntt=True, use_fp=False):
        """Routes: schoolbook (use_ntt=False), Goldilocks NTT (default), and with use_fp=True additionally the
        SIMD fp64 route (eval_level_fp / bootstrap_noks_fp: the timed CPU baseline)."""
        self.p = Params(*[int(x) for x in params])
        self.bsk = np.ascontiguousarray(bsk_std, dtype=np.uint32)
        self.ksk = np.ascontiguousarray(ksk, dtype=np.uint32)
        self._ntt = lib().orc_bsk_ntt_new(C.byref(self.p), _u32(self.bsk)) if use_ntt else None
        self._fp = lib().orc_bsk_fp_new(C.byref(self.p), _u32(self.bsk)) if use_fp else None
        if use_fp and not self._fp:
            raise RuntimeError("fp64 route unavailable: no AVX2 + FMA, or the set's exact products exceed the 51-bit prime")

    def __del__(self):
        if getattr(self, "_ntt", None):
            lib().orc_bsk_ntt_free(self._ntt)
            self._ntt = None
        if getattr(self, "_fp", None):
            lib().orc_bsk_fp_free(self._fp)
            self._fp = None

    def fp_prime(self):
        return int(lib().orc_bsk_fp_prime(self._fp))

    def bootstrap_noks_fp(self, lwes, tv):
        """[count, n+1] -> [count, k*N+1] by the SIMD fp64 route (one test vector for all)."""
        lwes = np.ascontiguousarray(lwes, dtype=np.uint32).reshape(-1, self.p.n + 1)
        tv = np.ascontiguousarray(tv, dtype=np.uint32)
        out = np.zeros((len(lwes), self.p.k * self.p.N + 1), dtype=np.uint32)
        u32p = C.POINTER(C.c_uint32)
        ins = (u32p * len(lwes))(*[_u32(lwes[g]) for g in range(len(lwes))])
        outs = (u32p * len(lwes))(*[_u32(out[g]) for g in range(len(lwes))])
        lib().orc_bootstrap_noks_fp(self._fp, ins, _u32(tv), outs, len(lwes))
        return out

    def eval_level_fp(self, wires, opcode, in0, in1, in2, outw, nthreads=0):
        """eval_level by the SIMD fp64 route; in place on `wires`."""
        assert wires.dtype == np.uint32 and wires.flags["C_CONTIGUOUS"]
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (opcode, in0, in1, in2, outw)]
        lib().orc_eval_level_fp(C.byref(self.p), self._fp, _u32(self.ksk), _u32(wires), *[_i32(a) for a in arrs],
                                len(arrs[0]), int(nthreads))

    def gate(self, op, in0, in1=None, in2=None):
        out = np.zeros(self.p.n + 1, dtype=np.uint32)
        lib().orc_gate(C.byref(self.p), _u32(self.bsk), self._ntt, _u32(self.ksk), int(op), _u32(in0), _u32(in1),
                       _u32(in2), _u32(out))
        return out

    def bootstrap_noks(self, lwe, tv):
        out = np.zeros(self.p.k * self.p.N + 1, dtype=np.uint32)
        lib().orc_bootstrap_noks(C.byref(self.p), _u32(self.bsk), self._ntt, _u32(lwe), _u32(tv), _u32(out))
        return out

    def keyswitch(self, big):
        out = np.zeros(self.p.n + 1, dtype=np.uint32)
        lib().orc_keyswitch(C.byref(self.p), _u32(self.ksk), _u32(big), _u32(out))
        return out

    def eval_level(self, wires, opcode, in0, in1, in2, outw, nthreads=0):
        """In-place on `wires` ([n_wires, n+1] uint32)."""
        assert wires.dtype == np.uint32 and wires.flags["C_CONTIGUOUS"]
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (opcode, in0, in1, in2, outw)]
        lib().orc_eval_level(C.byref(self.p), _u32(self.bsk), self._ntt, _u32(self.ksk), _u32(wires),
                             *[_i32(a) for a in arrs], len(arrs[0]), int(nthreads))

    def extprod_add(self, i, diff, acc):
        bsk_i = self.bsk.reshape(self.p.n, -1)[i]
        lib().orc_extprod_add(C.byref(self.p), _u32(np.ascontiguousarray(bsk_i)), self._ntt, int(i), _u32(diff),
                              _u32(acc))


def lincomb(n, op, which, in0, in1, in2=None):
    out = np.zeros(n + 1, dtype=np.uint32)
    lib().orc_gate_lincomb(n, int(op), int(which), _u32(in0), _u32(in1), _u32(in2), _u32(out))
    return out


def phase(sk_bits, ct):
    sk = np.ascontiguousarray(sk_bits, dtype=np.uint32)
    return int(lib().orc_phase(len(sk), _u32(sk), _u32(ct)))


def decrypt_bool(sk_bits, ct):
    return phase(sk_bits, ct) < (1 << 31)


def modswitch(x, log2_2N):
    return int(lib().orc_modswitch(int(x), int(log2_2N)))


def decompose(x, logB, l):
    d = np.zeros(l, dtype=np.int32)
    lib().orc_decompose(int(x), int(logB), int(l), _i32(d))
    return d


def ntt_route_name():
    """What the timed NTT route of liborc.so is (bench.py's cpu_baseline quotes it)."""
    return lib().orc_ntt_route().decode()


def fp_lanes():
    return int(lib().orc_fp_lanes())


def max_threads():
    return int(lib().orc_max_threads())


# ---------------------------------------------------------------------------------------
# shortint (LUT / arithmetic mode) oracle: oracle/shortint_oracle.c
# ---------------------------------------------------------------------------------------
class Params64(C.Structure):
    _fields_ = [(f, C.c_int32) for f in ("n", "k", "N", "pbs_l", "pbs_logB", "ks_l", "ks_logB",
                                         "message_modulus", "carry_modulus", "grouping_factor")]


_LIB64 = None


def lib64():
    global _LIB64
    if _LIB64 is None:
        path = os.path.join(_HERE, "liborc64.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE, "liborc64.so"])
        L = C.CDLL(path)
        u64p = C.POINTER(C.c_uint64)
        i32p = C.POINTER(C.c_int32)
        P = C.POINTER(Params64)
        L.orc64_make_lut.argtypes = [P, u64p, u64p]
        L.orc64_bootstrap.argtypes = [P, u64p, u64p, u64p, u64p]
        L.orc64_keyswitch.argtypes = [P, u64p, u64p, u64p]
        L.orc64_apply_lut.argtypes = [P, u64p, u64p, u64p, u64p, u64p]
        L.orc64_eval_lut_level.argtypes = [P, u64p, u64p, u64p, i32p, i32p, C.c_int, u64p, i32p, C.c_int]
        L.orc64_decrypt.restype = C.c_uint64
        L.orc64_decrypt.argtypes = [P, u64p, u64p]
        L.orc64_modswitch.restype = C.c_uint64
        L.orc64_modswitch.argtypes = [C.c_uint64, C.c_int]
        L.orc64_decompose.argtypes = [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        # the key behind a handle: either exact route (schoolbook / Goldilocks NTT on the key split into parts)
        L.orc64_key_new.restype = C.c_void_p
        L.orc64_key_new.argtypes = [P, u64p, C.c_int]
        L.orc64_key_free.argtypes = [C.c_void_p]
        L.orc64_key_part_bits.argtypes = [C.c_void_p]
        L.orc64_bootstrap_k.argtypes = [C.c_void_p, u64p, u64p, u64p]
        L.orc64_apply_lut_k.argtypes = [C.c_void_p, u64p, u64p, u64p, u64p]
        L.orc64_eval_lut_rows_k.argtypes = [C.c_void_p, u64p, u64p, i32p, i32p, C.c_int, u64p, i32p, C.c_int, u64p]
        L.orc64_apply_luts_k.argtypes = [C.c_void_p, u64p, u64p, u64p, i32p, C.c_int, u64p]
        _LIB64 = L
    return _LIB64


def _u64(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


class Oracle64:
    """Server-side shortint evaluation with a given (bsk, ksk) in the standard-domain
    layouts documented in include/helm_shortint.h."""

    def __init__(self, params9, bsk_std, ksk, use_ntt=False):
        """use_ntt: negacyclic products by the Goldilocks NTT on the key split into 32- / 16- / 8-bit parts (exact, the
        route that makes whole levels at the full parameter sets checkable) instead of schoolbook convolution; the two
        routes give identical ciphertexts (tests/test_oracle_shortint.py)."""
        self.p = Params64(*[int(x) for x in params9])
        self.bsk = np.ascontiguousarray(bsk_std, dtype=np.uint64)
        self.ksk = np.ascontiguousarray(ksk, dtype=np.uint64)
        self.dim = self.p.k * self.p.N
        self.t = self.p.message_modulus * self.p.carry_modulus
        self.delta = (1 << 63) // self.t
        self.use_ntt = bool(use_ntt)
        self._key = lib64().orc64_key_new(C.byref(self.p), _u64(self.bsk), 1 if use_ntt else 0)
        self.part_bits = int(lib64().orc64_key_part_bits(self._key))

    def __del__(self):
        k, self._key = getattr(self, "_key", None), None
        if k:
            lib64().orc64_key_free(k)

    def make_lut(self, f):
        vals = np.array([f(v) for v in range(self.t)] if callable(f) else list(f), dtype=np.uint64)
        out = np.zeros(self.p.N, dtype=np.uint64)
        lib64().orc64_make_lut(C.byref(self.p), _u64(vals), _u64(out))
        return out

    def keyswitch(self, big):
        big = np.ascontiguousarray(big, dtype=np.uint64)
        out = np.zeros(self.p.n + 1, dtype=np.uint64)
        lib64().orc64_keyswitch(C.byref(self.p), _u64(self.ksk), _u64(big), _u64(out))
        return out

    def bootstrap(self, small, lut):
        small = np.ascontiguousarray(small, dtype=np.uint64)
        lut = np.ascontiguousarray(lut, dtype=np.uint64)
        out = np.zeros(self.dim + 1, dtype=np.uint64)
        lib64().orc64_bootstrap_k(self._key, _u64(small), _u64(lut), _u64(out))
        return out

    def apply_lut(self, big, lut):
        big = np.ascontiguousarray(big, dtype=np.uint64)
        lut = np.ascontiguousarray(lut, dtype=np.uint64)
        out = np.zeros(self.dim + 1, dtype=np.uint64)
        lib64().orc64_apply_lut_k(self._key, _u64(self.ksk), _u64(big), _u64(lut), _u64(out))
        return out

    def apply_luts(self, in_rows, luts, lut_index):
        """A batch of independent look-ups (what helm_si_apply_luts does on the GPU): row q -> apply_lut(in_rows[q],
        luts[lut_index[q]]); OpenMP over the rows."""
        in_rows = np.ascontiguousarray(in_rows, dtype=np.uint64)
        luts = np.ascontiguousarray(np.atleast_2d(luts), dtype=np.uint64)
        idx = np.ascontiguousarray(lut_index, dtype=np.int32)
        assert in_rows.shape == (len(idx), self.dim + 1) and luts.shape[1] == self.p.N
        out = np.zeros_like(in_rows)
        lib64().orc64_apply_luts_k(self._key, _u64(self.ksk), _u64(in_rows), _u64(luts), _i32(idx), len(idx), _u64(out))
        return out

    def eval_lut_rows(self, wires, arity, in_idx, table, gates):
        """The output rows of the chosen `gates` of a LUT level over `wires` (not modified): lets a test check rows of a level
        the GPU evaluated as a whole - first / last workgroup, tile edges - without paying for the others."""
        assert wires.dtype == np.uint64 and wires.flags["C_CONTIGUOUS"]
        arity = np.ascontiguousarray(arity, dtype=np.int32)
        in_idx = np.ascontiguousarray(np.atleast_2d(in_idx), dtype=np.int32)
        table = np.ascontiguousarray(table, dtype=np.uint64)
        gates = np.ascontiguousarray(gates, dtype=np.int32)
        out = np.zeros((len(gates), self.dim + 1), dtype=np.uint64)
        lib64().orc64_eval_lut_rows_k(self._key, _u64(self.ksk), _u64(wires), _i32(arity), _i32(in_idx), in_idx.shape[1],
                                      _u64(table), _i32(gates), len(gates), _u64(out))
        return out

    def eval_lut_level(self, wires, arity, in_idx, table, out_idx):
        """In place on `wires` ([rows, k*N+1] uint64)."""
        assert wires.dtype == np.uint64 and wires.flags["C_CONTIGUOUS"]
        arity = np.ascontiguousarray(arity, dtype=np.int32)
        in_idx = np.ascontiguousarray(np.atleast_2d(in_idx), dtype=np.int32)
        table = np.ascontiguousarray(table, dtype=np.uint64)
        out_idx = np.ascontiguousarray(out_idx, dtype=np.int32)
        if self.use_ntt:
            wires[out_idx] = self.eval_lut_rows(wires, arity, in_idx, table, np.arange(len(arity), dtype=np.int32))
            return
        lib64().orc64_eval_lut_level(C.byref(self.p), _u64(self.bsk), _u64(self.ksk), _u64(wires), _i32(arity),
                                     _i32(in_idx), in_idx.shape[1], _u64(table), _i32(out_idx), len(arity))

    def decrypt(self, glwe_sk_bits, ct):
        """message and carry"""
        sk = np.ascontiguousarray(glwe_sk_bits, dtype=np.uint64)
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        return int(lib64().orc64_decrypt(C.byref(self.p), _u64(sk), _u64(ct)))


# ---------------------------------------------------------------------------------------------------------
# WoP-PBS wide-LUT path (oracle/wopbs_oracle.c)
# ---------------------------------------------------------------------------------------------------------
class ParamsW(C.Structure):
    _fields_ = [(f, C.c_int32) for f in ("n", "k", "N", "pbs_l", "pbs_logB", "ks_l", "ks_logB", "pfks_l", "pfks_logB",
                                         "cbs_l", "cbs_logB", "message_modulus", "carry_modulus")]


_LIBW = None


def libw():
    global _LIBW
    if _LIBW is None:
        path = os.path.join(_HERE, "liborcw.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE, "liborcw.so"])
        L = C.CDLL(path)
        u64p = C.POINTER(C.c_uint64)
        P = C.POINTER(ParamsW)
        vp = C.c_void_p
        L.orcw_keyswitch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, u64p, u64p, u64p]
        L.orcw_pfpks.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, u64p, u64p, u64p]
        L.orcw_ggsw_new.restype = vp
        L.orcw_ggsw_new.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, u64p, C.c_int, C.c_int]
        L.orcw_ggsw_free.argtypes = [vp]
        L.orcw_ggsw_part_bits.argtypes = [vp]
        L.orcw_extprod_add.argtypes = [vp, C.c_int, u64p, u64p]
        L.orcw_cmux.argtypes = [vp, C.c_int, u64p, u64p]
        L.orcw_bootstrap.argtypes = [vp, C.c_int, u64p, u64p, u64p]
        L.orcw_extract_bits.argtypes = [P, vp, u64p, C.c_int, C.c_int, u64p, u64p]
        L.orcw_circuit_bootstrap.argtypes = [P, vp, u64p, u64p, u64p]
        L.orcw_vertical_packing.argtypes = [vp, C.c_int, u64p, u64p]
        _LIBW = L
    return _LIBW


class Ggsw:
    """A stack of GGSWs ([count][l][k+1][k+1][N], standard domain) ready for external products: the Goldilocks
    NTT route (default) or the schoolbook route (use_ntt=False)."""

    def __init__(self, N, k, l, logB, ggsw_std, use_ntt=True):
        self.std = np.ascontiguousarray(ggsw_std, dtype=np.uint64).reshape(-1)
        self.N, self.k, self.l, self.logB = N, k, l, logB
        per = l * (k + 1) * (k + 1) * N
        assert self.std.size % per == 0
        self.count = self.std.size // per
        self._h = libw().orcw_ggsw_new(N, k, l, logB, _u64(self.std), self.count, int(use_ntt))

    def __del__(self):
        if getattr(self, "_h", None):
            libw().orcw_ggsw_free(self._h)
            self._h = None

    def part_bits(self):
        return libw().orcw_ggsw_part_bits(self._h)

    def extprod_add(self, i, diff, acc):
        diff = np.ascontiguousarray(diff, dtype=np.uint64)
        assert acc.dtype == np.uint64 and acc.flags["C_CONTIGUOUS"]
        libw().orcw_extprod_add(self._h, i, _u64(diff), _u64(acc))

    def bootstrap(self, small, tv):
        """small: count+1 words (the stack is a bootstrapping key), tv: N words -> k*N+1 words"""
        small = np.ascontiguousarray(small, dtype=np.uint64)
        tv = np.ascontiguousarray(tv, dtype=np.uint64)
        out = np.zeros(self.k * self.N + 1, dtype=np.uint64)
        libw().orcw_bootstrap(self._h, small.size - 1, _u64(small), _u64(tv), _u64(out))
        return out

    def vertical_packing(self, bits, lut):
        """the stack holds `bits` GGSWs, index 0 = MOST significant bit (tfhe's list order)"""
        assert self.count == bits
        lut = np.ascontiguousarray(lut, dtype=np.uint64)
        assert lut.size == max(1 << bits, self.N)
        out = np.zeros(self.k * self.N + 1, dtype=np.uint64)
        libw().orcw_vertical_packing(self._h, bits, _u64(lut), _u64(out))
        return out


class OracleW:
    """high_precision_lut() (reference src/gates.rs:787-815) with given keys: `pbs` is the Oracle64 of the PBS side
    (cleaning bootstrap, final bootstrap), the WoP side's keys in the layouts of include/helm_wopbs.h."""

    def __init__(self, params13, bsk, ksk, pfpksk, pbs=None, ksk_to_wopbs=None, ksk_to_pbs=None, use_ntt=True):
        self.p = ParamsW(*[int(x) for x in params13])
        p = self.p
        self.dim = p.k * p.N
        self.bsk = Ggsw(p.N, p.k, p.pbs_l, p.pbs_logB, bsk, use_ntt)
        self.ksk = np.ascontiguousarray(ksk, dtype=np.uint64)
        self.pfpksk = np.ascontiguousarray(pfpksk, dtype=np.uint64)
        self.pbs = pbs
        # the PBS side's bootstraps on the same exact route as this side's (pinned equal to Oracle64.bootstrap, the
        # schoolbook one, by tests/test_wopbs_oracle.py)
        self.pbs_bsk = None if pbs is None else Ggsw(pbs.p.N, pbs.p.k, pbs.p.pbs_l, pbs.p.pbs_logB, pbs.bsk, use_ntt)
        self.to_wop = None if ksk_to_wopbs is None else np.ascontiguousarray(ksk_to_wopbs, dtype=np.uint64)
        self.to_pbs = None if ksk_to_pbs is None else np.ascontiguousarray(ksk_to_pbs, dtype=np.uint64)
        self.use_ntt = use_ntt
        self.t = p.message_modulus * p.carry_modulus
        self.delta = (1 << 63) // self.t
        self.delta_log = self.delta.bit_length() - 1

    def keyswitch(self, key, in_dim, out_dim, l, logB, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        out = np.zeros(out_dim + 1, dtype=np.uint64)
        libw().orcw_keyswitch(in_dim, out_dim, l, logB, _u64(key), _u64(ct), _u64(out))
        return out

    def extract_bits(self, big, delta_log, nb):
        """-> [nb][n+1], row 0 = MOST significant extracted bit (tfhe's order)"""
        big = np.ascontiguousarray(big, dtype=np.uint64)
        out = np.zeros((nb, self.p.n + 1), dtype=np.uint64)
        libw().orcw_extract_bits(C.byref(self.p), self.bsk._h, _u64(self.ksk), delta_log, nb, _u64(big), _u64(out))
        return out

    def circuit_bootstrap(self, small):
        """-> [cbs_l][k+1][(k+1) N]"""
        small = np.ascontiguousarray(small, dtype=np.uint64)
        k1 = self.p.k + 1
        out = np.zeros((self.p.cbs_l, k1, k1 * self.p.N), dtype=np.uint64)
        libw().orcw_circuit_bootstrap(C.byref(self.p), self.bsk._h, _u64(self.pfpksk), _u64(small), _u64(out))
        return out

    def vertical_packing(self, ggsws_msb_first, lut):
        g = Ggsw(self.p.N, self.p.k, self.p.cbs_l, self.p.cbs_logB, np.ascontiguousarray(ggsws_msb_first), self.use_ntt)
        return g.vertical_packing(g.count, lut)

    def make_table(self, n_blocks, bits_per_block, truth):
        """generate_high_precision_lut_radix_helm (src/gates.rs:817-864), block 0, in plain Python."""
        basis, total = self.p.message_modulus, n_blocks * bits_per_block
        modulus = basis ** n_blocks
        out = np.zeros(max(1 << total, self.p.N), dtype=np.uint64)
        for v in range(1 << total):
            fields = [(v >> (j * bits_per_block)) & ((1 << bits_per_block) - 1) for j in range(n_blocks)]
            x = sum(f * basis ** j for j, f in enumerate(fields)) % modulus
            f_val = (int(truth[x]) & 1) if x < len(truth) else 0
            out[v] = ((f_val % modulus) % basis) * self.delta
        return out

    def wide_lut(self, inputs, truth, bits_per_block):
        """inputs: PBS-side big ciphertexts, first = most significant block; -> PBS-side big ciphertext"""
        assert self.pbs is not None and self.to_wop is not None and self.to_pbs is not None
        P, S = self.p, self.pbs.p
        ident = self.pbs.make_lut(lambda x: x)
        blocks = list(inputs)[::-1]  # radix block 0 = last input (gates.rs:795-799)
        bits_msb_first = []
        for blk in reversed(blocks):  # most significant block first (WopbsKey::wopbs)
            clean = self.pbs_bsk.bootstrap(self.pbs.keyswitch(blk), ident)
            wbig = self.keyswitch(self.to_wop, S.k * S.N, self.dim, S.ks_l, S.ks_logB, clean)
            bits_msb_first.extend(self.extract_bits(wbig, self.delta_log, bits_per_block))
        ggsws = np.stack([self.circuit_bootstrap(b) for b in bits_msb_first])
        table = self.make_table(len(blocks), bits_per_block, truth)
        vp = self.vertical_packing(ggsws, table)
        small = self.keyswitch(self.to_pbs, self.dim, S.n, S.ks_l, S.ks_logB, vp)
        return self.pbs_bsk.bootstrap(small, ident)
