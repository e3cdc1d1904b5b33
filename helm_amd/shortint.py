"""Shortint (LUT / arithmetic mode) object layer over include/helm_shortint.h:
SiClientKey (CPU), SiServerKey (GPU engine context with keys resident in HBM),
SiWires (HBM-resident table of big-LWE ciphertexts).

Mirrors the roles of tfhe::shortint::{ClientKey, ServerKey} as HELM uses them
(reference src/bin/helm.rs:301, src/circuit.rs:75-79, src/gates.rs:754-785).
"""
import ctypes as C

import numpy as np

from . import _native as nv
from ._native import SiParams, Timing, hip, host, hip_check, client_check


def si_named_params(name):
    """-> (SiParams, lwe_noise_std, glwe_noise_std)"""
    p = SiParams()
    a, b = C.c_double(), C.c_double()
    rc = host.helm_si_client_named_params(name.encode(), C.byref(p), C.byref(a), C.byref(b))
    if rc != 0:
        raise nv.HelmError(f"unknown shortint parameter set {name!r}")
    return p, a.value, b.value


def _seed_arg(seed):
    """None -> 0 = OS entropy (ChaCha20 under a getrandom key; the default, like tfhe's gen_keys()).
    An integer selects the DETERMINISTIC, INSECURE test generator (helm_amd/csrc/rng.hpp)."""
    if seed is None:
        return 0
    seed = int(seed)
    if seed == 0:
        raise ValueError("seed=0 is reserved for OS entropy: pass seed=None, or a non-zero test seed")
    return seed


class SiClientKey:
    def __init__(self, params, lwe_std, glwe_std, seed=None):
        self.params = params
        h = nv.vp()
        rc = host.helm_si_client_keygen(C.byref(params), lwe_std, glwe_std, _seed_arg(seed), C.byref(h))
        if rc != 0:
            raise nv.HelmError(f"helm_si_client_keygen failed ({rc})")
        self._h = h
        self.dim = params.k * params.N
        self.t = params.message_modulus * params.carry_modulus
        self.delta = (1 << 63) // self.t

    @classmethod
    def generate(cls, name="shortint_m2c2", seed=None):
        p, a, b = si_named_params(name)
        return cls(p, a, b, seed)

    def __del__(self):
        if getattr(self, "_h", None):
            host.helm_si_client_key_free(self._h)
            self._h = None

    def _view(self, fn, count):
        return np.ctypeslib.as_array(fn(self._h), shape=(count,))

    @property
    def bsk(self):
        return self._view(host.helm_si_client_bsk, host.helm_si_client_bsk_words(self._h))

    @property
    def ksk(self):
        return self._view(host.helm_si_client_ksk, host.helm_si_client_ksk_words(self._h))

    @property
    def lwe_secret(self):
        return self._view(host.helm_si_client_lwe_secret, self.params.n)

    @property
    def glwe_secret(self):
        return self._view(host.helm_si_client_glwe_secret, self.dim)

    def encrypt(self, values):
        """int or sequence -> [count, k*N+1] uint64 (ClientKey::encrypt(u64))."""
        scalar = np.isscalar(values)
        v = np.ascontiguousarray(np.atleast_1d(np.asarray(values)).astype(np.uint64))
        out = np.zeros((len(v), self.dim + 1), dtype=np.uint64)
        rc = host.helm_si_client_encrypt(self._h, nv.as_u64p(v), len(v), nv.as_u64p(out))
        assert rc == 0
        return out[0] if scalar else out

    def decrypt_message_and_carry(self, lwe):
        a = np.ascontiguousarray(lwe, dtype=np.uint64)
        one = a.ndim == 1
        a2 = a.reshape(-1, self.dim + 1)
        out = np.zeros(len(a2), dtype=np.uint64)
        rc = host.helm_si_client_decrypt(self._h, nv.as_u64p(a2), len(a2), nv.as_u64p(out))
        assert rc == 0
        return int(out[0]) if one else out

    def decrypt(self, lwe):
        """ClientKey::decrypt: the message part (mod message_modulus)."""
        v = self.decrypt_message_and_carry(lwe)
        return v % self.params.message_modulus

    def phase(self, lwe, small=False):
        dim = self.params.n if small else self.dim
        a2 = np.ascontiguousarray(lwe, dtype=np.uint64).reshape(-1, dim + 1)
        out = np.zeros(len(a2), dtype=np.uint64)
        rc = host.helm_si_client_phase(self._h, nv.as_u64p(a2), len(a2), int(small), nv.as_u64p(out))
        assert rc == 0
        return out


class SiServerKey:
    """GPU engine context (64-bit torus) with both keys resident in HBM."""

    def __init__(self, client_key=None, params=None, bsk=None, ksk=None, device=0):
        self.params = client_key.params if client_key is not None else params
        h = nv.vp()
        hip_check(hip.helm_si_ctx_create(device, C.byref(self.params), C.byref(h)))
        self._h = h
        self.dim = self.params.k * self.params.N
        if client_key is not None:
            bsk, ksk = client_key.bsk, client_key.ksk
        if bsk is not None:
            bsk = np.ascontiguousarray(bsk, dtype=np.uint64).reshape(-1)
            hip_check(hip.helm_si_load_bootstrap_key(self._h, nv.as_u64p(bsk), bsk.size))
        if ksk is not None:
            ksk = np.ascontiguousarray(ksk, dtype=np.uint64).reshape(-1)
            hip_check(hip.helm_si_load_keyswitch_key(self._h, nv.as_u64p(ksk), ksk.size))

    def fork(self):
        """A lane (helm_si_ctx_fork): shares this key's device-resident keys and wire tables, own stream and scratch."""
        lane = SiServerKey.__new__(SiServerKey)
        lane.params, lane.dim = self.params, self.dim
        h = nv.vp()
        hip_check(hip.helm_si_ctx_fork(self._h, C.byref(h)))
        lane._h = h
        lane._primary = self  # keep alive; lanes are closed before their primary
        self._lanes = getattr(self, "_lanes", []) + [lane]
        return lane

    def close(self):
        for lane in getattr(self, "_lanes", []):
            lane.close()
        self._lanes = []
        if getattr(self, "_h", None):
            hip.helm_si_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def sync(self):
        hip_check(hip.helm_si_sync(self._h))

    def round_capacity(self):
        """Bootstraps the device holds at once under this set (helm_si_round_capacity): CUs x resident workgroups per CU."""
        v = int(hip.helm_si_round_capacity(self._h))
        if v < 0:
            hip_check(v)
        return v

    def field_bits(self):
        """49 or 46: the CRT pair of prime fields this context's bootstrap kernels compute in (helm_si_field_bits: it follows
        the loaded key for k > 1 contexts)."""
        return int(hip.helm_si_field_bits(self._h))

    def set_stream(self, stream_ptr):
        nv.require_one_hip_runtime(type(self).__name__ + ".set_stream")  # the handle is another framework's
        hip_check(hip.helm_si_set_stream(self._h, nv.vp(stream_ptr)))

    def bound_violations(self, reset=True):
        """Check build only (HELM_HIP_LIB=libhelm_hip_check.so): violations of the lazy arithmetic's contracts counted by the
        64-bit engine's kernels (and the WoP-PBS path's) since the last reset, slots as ServerKey.bound_violations."""
        c = (C.c_uint32 * 8)()
        hip_check(hip.helm_si_bound_violations(self._h, c, 1 if reset else 0))
        return [int(v) for v in c]

    def set_audit(self, fn):
        """helm_si_set_audit: while set, every linear step and every look-up batch of this key - LUT levels, the radix
        operators of arithmetic mode - hands `fn` a dict with its operand rows (read before the call ran), its result rows
        and its arguments: kind "luts" -> in_rows, out_rows, lut_idx, luts; kind "lincomb" -> in_rows [count, terms, row],
        out_rows, in_idx, coef, const_add (or None).  fn returning False (or raising) fails the call.  fn = None: off.
        Set it BEFORE the evaluators fork their lanes."""
        if fn is None:
            hip_check(hip.helm_si_set_audit(self._h, nv.SI_AUDIT_FN(0), None))
            self._audit = None
            return
        brow = self.params.k * self.params.N + 1
        N = self.params.N

        def trampoline(_user, recp):
            try:
                r = recp.contents
                cnt = int(r.count)
                arr = lambda p, shape, dt: np.ctypeslib.as_array(p, shape=shape).astype(dt, copy=True)
                if r.kind == 0:
                    rec = {"kind": "luts", "in_rows": arr(r.in_rows, (cnt, brow), np.uint64), "out_rows": arr(r.out_rows, (cnt, brow), np.uint64),
                           "lut_idx": arr(r.lut_idx, (cnt,), np.int32), "luts": arr(r.luts, (int(r.n_luts), N), np.uint64)}
                else:
                    t = int(r.terms)
                    rec = {"kind": "lincomb", "in_rows": arr(r.in_rows, (cnt, t, brow), np.uint64), "out_rows": arr(r.out_rows, (cnt, brow), np.uint64),
                           "in_idx": arr(r.in_idx, (cnt, t), np.int32), "coef": arr(r.coef, (cnt, t), np.int64),
                           "const_add": arr(r.const_add, (cnt,), np.int64) if r.const_add else None}
                return 0 if fn(rec) is not False else 1
            except BaseException:  # noqa: BLE001 - reported through the status code
                import traceback
                traceback.print_exc()
                return 2
        cb = nv.SI_AUDIT_FN(trampoline)
        hip_check(hip.helm_si_set_audit(self._h, cb, None))
        self._audit = cb  # kept alive as long as it is set

    def set_exchange_comm(self, comm, min_batch=None, capacity_rows=4096):
        """Shard every bootstrap batch of at least `min_batch` ciphertexts over the ranks of `comm`
        (helm_amd.comm.Comm: the library's own RCCL communicator; helm_si_set_exchange_comm) - the
        all-gather is ncclAllGather inside libhelm_hip.so on the engine's stream, no torch in the data
        path.  comm = None switches sharding off.  A world-size-1 communicator keeps every batch on the
        stage -> all-gather -> scatter path."""
        if comm is None:
            hip_check(hip.helm_si_set_exchange_comm(self._h, None, 1, 1))
            self._exchange = None
            return
        if min_batch is None:
            # a batch that fits one wave of workgroups gains nothing from sharding
            min_batch = int(hip.helm_si_round_capacity(self._h)) + 1
        hip_check(hip.helm_si_set_exchange_comm(self._h, comm._h, int(min_batch), int(capacity_rows)))
        self._exchange = (comm,)  # kept alive for the engine

    def set_exchange(self, dist, rank, world, min_batch=None, capacity_rows=4096, force=False):
        """Shard every bootstrap batch of at least `min_batch` ciphertexts over the `world`
        ranks of torch.distributed `dist` (backend nccl = RCCL; helm_si_set_exchange).  The
        engine is put on torch's current stream so that the all-gather is ordered behind the
        kernels that fill the staging rows.  world <= 1 switches sharding off, unless `force`
        (world = 1 then still sends every batch through stage -> all-gather -> scatter)."""
        if world <= 1 and not force:
            hip_check(hip.helm_si_set_exchange(self._h, 0, 1, 1, None, None, 1, nv.SI_EXCHANGE_FN(0), None))
            self._exchange = None
            return
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        self.set_stream(torch.cuda.current_stream().cuda_stream)
        row = self.dim + 1
        stage = torch.empty((capacity_rows, row), dtype=torch.int64, device=dev)
        gather = torch.empty((capacity_rows * world, row), dtype=torch.int64, device=dev)

        def exchange(_user, rows):
            try:
                dist.all_gather_into_tensor(gather[:rows * world], stage[:rows])
                return 0
            except Exception:  # an exception cannot cross the C frames
                import traceback
                traceback.print_exc()
                return -1

        fn = nv.SI_EXCHANGE_FN(exchange)
        if min_batch is None:
            # a batch that fits one wave of workgroups (one bootstrap per CU) gains nothing from sharding
            min_batch = torch.cuda.get_device_properties(dev).multi_processor_count + 1
        hip_check(hip.helm_si_set_exchange(self._h, rank, world, int(min_batch), nv.vp(stage.data_ptr()),
                                           nv.vp(gather.data_ptr()), capacity_rows, fn, None))
        self._exchange = (fn, stage, gather)  # kept alive for the engine

    def exchange_stats(self):
        b, r = C.c_int64(0), C.c_int64(0)
        hip_check(hip.helm_si_exchange_stats(self._h, C.byref(b), C.byref(r)))
        return int(b.value), int(r.value)

    def wires(self, n_rows):
        return SiWires(self, n_rows)

    def make_lut(self, f):
        """generate_lookup_table(f): f is a callable or a value table over [0, msg*carry)."""
        t = self.params.message_modulus * self.params.carry_modulus
        vals = np.array([f(v) for v in range(t)] if callable(f) else list(f), dtype=np.uint64)
        assert len(vals) == t
        out = np.zeros(self.params.N, dtype=np.uint64)
        hip_check(hip.helm_si_make_lut(self._h, nv.as_u64p(vals), nv.as_u64p(out)))
        return out

    def keyswitch_batch(self, big):
        p = self.params
        big = np.ascontiguousarray(big, dtype=np.uint64).reshape(-1, self.dim + 1)
        out = np.zeros((len(big), p.n + 1), dtype=np.uint64)
        hip_check(hip.helm_si_keyswitch_batch(self._h, nv.as_u64p(big), nv.as_u64p(out), len(big)))
        return out

    def pbs_batch(self, small, luts, lut_idx=None):
        p = self.params
        small = np.ascontiguousarray(small, dtype=np.uint64).reshape(-1, p.n + 1)
        luts = np.ascontiguousarray(luts, dtype=np.uint64).reshape(-1, p.N)
        if lut_idx is None:
            lut_idx = np.zeros(len(small), dtype=np.int32)
        lut_idx = np.ascontiguousarray(lut_idx, dtype=np.int32)
        out = np.zeros((len(small), self.dim + 1), dtype=np.uint64)
        hip_check(hip.helm_si_pbs_batch(self._h, nv.as_u64p(small), nv.as_u64p(luts), len(luts), nv.as_i32p(lut_idx),
                                        nv.as_u64p(out), len(small)))
        return out

    def timing_enable(self, on=True):
        hip_check(hip.helm_si_timing_enable(self._h, int(on)))

    def timing(self, reset=False):
        t = Timing()
        hip_check(hip.helm_si_get_timing(self._h, C.byref(t), int(reset)))
        return t


class SiWires:
    """HBM-resident ciphertext table: rows of k*N+1 uint64."""

    def __init__(self, server_key, n_rows):
        self.sk = server_key
        self.n_rows = int(n_rows)
        h = nv.vp()
        hip_check(hip.helm_si_wires_alloc(server_key._h, self.n_rows, C.byref(h)))
        self._h = h

    def free(self):
        if getattr(self, "_h", None):
            hip.helm_si_wires_free(getattr(self.sk, "_h", None), self._h)  # NULL owner: host struct only
        self._h = None

    def __del__(self):
        self.free()

    def upload(self, idx, lwe):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        lwe = np.ascontiguousarray(lwe, dtype=np.uint64).reshape(len(idx), self.sk.dim + 1)
        hip_check(hip.helm_si_wires_upload(self.sk._h, self._h, nv.as_i32p(idx), nv.as_u64p(lwe), len(idx)))

    def download(self, idx=None):
        if idx is None:
            idx = np.arange(self.n_rows)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        out = np.zeros((len(idx), self.sk.dim + 1), dtype=np.uint64)
        hip_check(hip.helm_si_wires_download(self.sk._h, self._h, nv.as_i32p(idx), nv.as_u64p(out), len(idx)))
        return out

    def set_trivial(self, idx, values):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(values), idx.shape).astype(np.uint64))
        hip_check(hip.helm_si_wires_set_trivial(self.sk._h, self._h, nv.as_i32p(idx), nv.as_u64p(v), len(idx)))

    def lincomb(self, in_idx, coef, out_idx, const_add=None):
        """out[g] = sum_t coef[g,t] * row in_idx[g,t] + const_add[g] * delta."""
        in_idx = np.ascontiguousarray(np.atleast_2d(in_idx), dtype=np.int32)
        coef = np.ascontiguousarray(np.atleast_2d(coef), dtype=np.int64)
        out_idx = np.ascontiguousarray(out_idx, dtype=np.int32)
        assert in_idx.shape == coef.shape and in_idx.shape[0] == len(out_idx)
        ca = None
        if const_add is not None:
            ca = np.ascontiguousarray(np.broadcast_to(np.asarray(const_add), out_idx.shape).astype(np.int64))
        hip_check(hip.helm_si_lincomb(self.sk._h, self._h, nv.as_i32p(in_idx), nv.as_i64p(coef),
                                      nv.as_i64p(ca) if ca is not None else None, nv.as_i32p(out_idx),
                                      in_idx.shape[1], len(out_idx)))

    def apply_luts(self, in_idx, luts, out_idx, lut_idx=None):
        in_idx = np.ascontiguousarray(in_idx, dtype=np.int32)
        out_idx = np.ascontiguousarray(out_idx, dtype=np.int32)
        luts = np.ascontiguousarray(luts, dtype=np.uint64).reshape(-1, self.sk.params.N)
        if lut_idx is None:
            lut_idx = np.zeros(len(in_idx), dtype=np.int32)
        lut_idx = np.ascontiguousarray(lut_idx, dtype=np.int32)
        hip_check(hip.helm_si_apply_luts(self.sk._h, self._h, nv.as_i32p(in_idx), nv.as_i32p(lut_idx),
                                         nv.as_i32p(out_idx), len(in_idx), nv.as_u64p(luts), len(luts)))

    def eval_lut_level(self, arity, in_idx, table, out_idx):
        """gates::lut() for a level: in_idx [count, max_in] (-1 padded), table = truth tables as bit masks."""
        arity = np.ascontiguousarray(arity, dtype=np.int32)
        in_idx = np.ascontiguousarray(np.atleast_2d(in_idx), dtype=np.int32)
        table = np.ascontiguousarray(table, dtype=np.uint64)
        out_idx = np.ascontiguousarray(out_idx, dtype=np.int32)
        hip_check(hip.helm_si_eval_lut_level(self.sk._h, self._h, nv.as_i32p(arity), nv.as_i32p(in_idx),
                                             in_idx.shape[1], nv.as_u64p(table), nv.as_i32p(out_idx), len(arity)))
