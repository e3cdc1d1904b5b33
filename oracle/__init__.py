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
        _LIB64 = L
    return _LIB64


def _u64(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


class Oracle64:
    """Server-side shortint evaluation with a given (bsk, ksk) in the standard-domain
    layouts documented in include/helm_shortint.h."""

    def __init__(self, params9, bsk_std, ksk):
        self.p = Params64(*[int(x) for x in params9])
        self.bsk = np.ascontiguousarray(bsk_std, dtype=np.uint64)
        self.ksk = np.ascontiguousarray(ksk, dtype=np.uint64)
        self.dim = self.p.k * self.p.N
        self.t = self.p.message_modulus * self.p.carry_modulus
        self.delta = (1 << 63) // self.t

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
        lib64().orc64_bootstrap(C.byref(self.p), _u64(self.bsk), _u64(small), _u64(lut), _u64(out))
        return out

    def apply_lut(self, big, lut):
        big = np.ascontiguousarray(big, dtype=np.uint64)
        lut = np.ascontiguousarray(lut, dtype=np.uint64)
        out = np.zeros(self.dim + 1, dtype=np.uint64)
        lib64().orc64_apply_lut(C.byref(self.p), _u64(self.bsk), _u64(self.ksk), _u64(big), _u64(lut), _u64(out))
        return out

    def eval_lut_level(self, wires, arity, in_idx, table, out_idx):
        """In place on `wires` ([rows, k*N+1] uint64)."""
        assert wires.dtype == np.uint64 and wires.flags["C_CONTIGUOUS"]
        arity = np.ascontiguousarray(arity, dtype=np.int32)
        in_idx = np.ascontiguousarray(np.atleast_2d(in_idx), dtype=np.int32)
        table = np.ascontiguousarray(table, dtype=np.uint64)
        out_idx = np.ascontiguousarray(out_idx, dtype=np.int32)
        lib64().orc64_eval_lut_level(C.byref(self.p), _u64(self.bsk), _u64(self.ksk), _u64(wires), _i32(arity),
                                     _i32(in_idx), in_idx.shape[1], _u64(table), _i32(out_idx), len(arity))

    def decrypt(self, glwe_sk_bits, ct):
        """message and carry"""
        sk = np.ascontiguousarray(glwe_sk_bits, dtype=np.uint64)
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        return int(lib64().orc64_decrypt(C.byref(self.p), _u64(sk), _u64(ct)))
