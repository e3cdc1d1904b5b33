"""WoP-PBS wide-LUT path, object layer over include/helm_wopbs.h: WopClientKey (CPU key material next to a
SiClientKey), WopServerKey (GPU context next to a SiServerKey).

Mirrors the roles of tfhe::shortint::wopbs::WopbsKey / tfhe::integer::wopbs::WopbsKey as HELM's
high_precision_lut() uses them (reference src/gates.rs:787-815); Gate::evaluate_encrypted_high_precision_lut
(src/gates.rs:721-742) is WopServerKey.eval_luts() for one gate.
"""
import ctypes as C

import numpy as np

from . import _native as nv
from ._native import WopParams, WopTiming, hip, host, hip_check
from .shortint import _seed_arg

KEY_BSK, KEY_KSK, KEY_KSK_TO_WOPBS, KEY_KSK_TO_PBS, KEY_PFPKSK, KEY_LWE_SECRET, KEY_GLWE_SECRET = range(7)


def wop_named_params(name):
    """-> (WopParams, lwe_noise_std, glwe_noise_std)"""
    p = WopParams()
    a, b = C.c_double(), C.c_double()
    if host.helm_wop_client_named_params(name.encode(), C.byref(p), C.byref(a), C.byref(b)) != 0:
        raise nv.HelmError(f"unknown WoP-PBS parameter set {name!r}")
    return p, a.value, b.value


class WopClientKey:
    """WopbsKey::new_wopbs_key(cks, sks, params): the WoP-side secret keys and the five evaluation keys."""

    def __init__(self, pbs_key, params, lwe_std, glwe_std, seed=None):
        self.pbs_key = pbs_key
        self.params = params
        h = nv.vp()
        rc = host.helm_wop_client_keygen(pbs_key._h, C.byref(params), lwe_std, glwe_std, _seed_arg(seed), C.byref(h))
        if rc != 0:
            raise nv.HelmError(f"helm_wop_client_keygen failed ({rc})")
        self._h = h
        self.dim = params.k * params.N
        self.t = params.message_modulus * params.carry_modulus
        self.delta = (1 << 63) // self.t
        self.delta_log = self.delta.bit_length() - 1

    @classmethod
    def generate(cls, pbs_key, name="wopbs_m2c2", seed=None):
        p, a, b = wop_named_params(name)
        return cls(pbs_key, p, a, b, seed)

    def __del__(self):
        if getattr(self, "_h", None):
            host.helm_wop_client_key_free(self._h)
            self._h = None

    def part(self, which):
        ptr, n = nv.u64p(), C.c_size_t()
        if host.helm_wop_client_key_part(self._h, which, C.byref(ptr), C.byref(n)) != 0:
            raise nv.HelmError("unknown key part")
        return np.ctypeslib.as_array(ptr, shape=(n.value,))

    bsk = property(lambda self: self.part(KEY_BSK))
    ksk = property(lambda self: self.part(KEY_KSK))
    ksk_to_wopbs = property(lambda self: self.part(KEY_KSK_TO_WOPBS))
    ksk_to_pbs = property(lambda self: self.part(KEY_KSK_TO_PBS))
    pfpksk = property(lambda self: self.part(KEY_PFPKSK))
    lwe_secret = property(lambda self: self.part(KEY_LWE_SECRET))
    glwe_secret = property(lambda self: self.part(KEY_GLWE_SECRET))

    def phase(self, lwe, small=False):
        """b - <a, s> under the WoP-side big (default) or small key."""
        sk = (self.lwe_secret if small else self.glwe_secret).astype(bool)
        a = np.atleast_2d(np.ascontiguousarray(lwe, dtype=np.uint64))
        return a[:, -1] - (a[:, :-1] * sk).sum(axis=1, dtype=np.uint64)


def make_table(params, n_blocks, bits_per_block, truth):
    """generate_high_precision_lut_radix_helm (reference src/gates.rs:817-864), block 0."""
    truth = np.ascontiguousarray(truth, dtype=np.uint64)
    out = np.zeros(hip.helm_wop_table_words(C.byref(params), n_blocks * bits_per_block), dtype=np.uint64)
    hip_check(hip.helm_wop_make_table(C.byref(params), n_blocks, bits_per_block, nv.as_u64p(truth), truth.size,
                                      nv.as_u64p(out)))
    return out


class WopServerKey:
    """GPU context of the wide-LUT path next to a SiServerKey (the PBS side); keys resident in HBM."""

    def __init__(self, server_key, client_key=None, params=None):
        self.server_key = server_key
        self.params = client_key.params if client_key is not None else params
        h = nv.vp()
        hip_check(hip.helm_wop_ctx_create(server_key._h, C.byref(self.params), C.byref(h)))
        self._h = h
        self.dim = self.params.k * self.params.N
        self.logN = self.params.N.bit_length() - 1
        if client_key is not None:
            sp = server_key.params
            for which in (KEY_BSK, KEY_KSK, KEY_KSK_TO_WOPBS, KEY_KSK_TO_PBS, KEY_PFPKSK):
                self.load_key(which, client_key.part(which), sp.ks_l, sp.ks_logB)

    def load_key(self, which, words, l=0, logB=0):
        words = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
        hip_check(hip.helm_wop_load_key(self._h, which, nv.as_u64p(words), words.size, l, logB))

    def close(self):
        if getattr(self, "_h", None):
            hip.helm_wop_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def table_words(self, bits):
        return hip.helm_wop_table_words(C.byref(self.params), bits)

    def eval_luts(self, wires, in_idx, truth_tables, out_idx, bits_per_block=None):
        """high_precision_lut() for a batch of gates with the same number of inputs.  in_idx [count][n_inputs] rows
        of `wires` (first input = most significant), truth_tables [count][2^n_inputs] (or one table for all)."""
        in_idx = np.ascontiguousarray(np.atleast_2d(in_idx), dtype=np.int32)
        count, n_inputs = in_idx.shape
        if bits_per_block is None:
            bits_per_block = (self.params.message_modulus * self.params.carry_modulus).bit_length() - 1
        truth_tables = np.atleast_2d(np.asarray(truth_tables, dtype=np.uint64))
        if truth_tables.shape[0] == 1 and count > 1:
            truth_tables = np.repeat(truth_tables, count, axis=0)
        made = {}
        tables = np.empty((count, self.table_words(n_inputs * bits_per_block)), dtype=np.uint64)
        for g in range(count):
            key = truth_tables[g].tobytes()
            if key not in made:
                made[key] = make_table(self.params, n_inputs, bits_per_block, truth_tables[g])
            tables[g] = made[key]
        out_idx = np.ascontiguousarray(out_idx, dtype=np.int32)
        hip_check(hip.helm_wop_eval_luts(self._h, wires._h, nv.as_i32p(in_idx), n_inputs, bits_per_block,
                                         nv.as_u64p(tables), nv.as_i32p(out_idx), count))

    # ---- stage primitives (tests) --------------------------------------------------------------------------
    def extract_bits(self, big, delta_log, nb):
        big = np.ascontiguousarray(np.atleast_2d(big), dtype=np.uint64)
        out = np.zeros((big.shape[0], nb, self.params.n + 1), dtype=np.uint64)
        hip_check(hip.helm_wop_extract_bits_batch(self._h, nv.as_u64p(big), delta_log, nb, nv.as_u64p(out), big.shape[0]))
        return out

    def circuit_bootstrap(self, small):
        small = np.ascontiguousarray(np.atleast_2d(small), dtype=np.uint64)
        k1 = self.params.k + 1
        out = np.zeros((small.shape[0], self.params.cbs_l, k1, k1 * self.params.N), dtype=np.uint64)
        hip_check(hip.helm_wop_circuit_bootstrap_batch(self._h, nv.as_u64p(small), nv.as_u64p(out), small.shape[0]))
        return out

    def vertical_packing(self, ggsw, tables):
        """ggsw [count][bits][cbs_l][k+1][(k+1) N], index 0 = least significant bit; tables [count][table_words]."""
        ggsw = np.ascontiguousarray(ggsw, dtype=np.uint64)
        count, bits = ggsw.shape[0], ggsw.shape[1]
        tables = np.ascontiguousarray(tables, dtype=np.uint64).reshape(count, -1)
        assert tables.shape[1] == self.table_words(bits)
        out = np.zeros((count, self.dim + 1), dtype=np.uint64)
        hip_check(hip.helm_wop_vertical_packing_batch(self._h, nv.as_u64p(ggsw), bits, nv.as_u64p(tables), nv.as_u64p(out),
                                                      count))
        return out

    def timing(self, reset=False):
        t = WopTiming()
        hip_check(hip.helm_wop_get_timing(self._h, C.byref(t), int(reset)))
        return {f: getattr(t, f) for f, _ in t._fields_}
