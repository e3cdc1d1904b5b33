"""Thin object layer over the C ABIs: ClientKey (CPU), ServerKey (GPU engine
context with keys loaded), DeviceWires (HBM-resident wire table), Program
(levelised netlist on the device).

Mirrors the roles of tfhe::boolean::{ClientKey, ServerKey} as HELM uses them
(reference src/bin/helm.rs:241, src/circuit.rs:69-73).
"""
import ctypes as C

import numpy as np

from . import _native as nv
from ._native import Params, HipTiming as Timing, hip, host, hip_check, client_check


def named_params(name):
    """-> (Params, lwe_noise_std, glwe_noise_std)"""
    p = Params()
    a, b = C.c_double(), C.c_double()
    client_check(host.helm_client_named_params(name.encode(), C.byref(p), C.byref(a), C.byref(b)))
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


class ClientKey:
    """Secret keys + the exported server key material (CPU)."""

    def __init__(self, params, lwe_std, glwe_std, seed=None):
        self.params = params
        h = nv.vp()
        client_check(host.helm_client_keygen(C.byref(params), lwe_std, glwe_std, _seed_arg(seed), C.byref(h)))
        self._h = h

    @classmethod
    def generate(cls, name="boolean_default", seed=None):
        p, a, b = named_params(name)
        return cls(p, a, b, seed)

    def __del__(self):
        if getattr(self, "_h", None):
            host.helm_client_key_free(self._h)
            self._h = None

    def _view(self, fn, count):
        ptr = fn(self._h)
        return np.ctypeslib.as_array(ptr, shape=(count,))

    @property
    def bsk(self):
        return self._view(host.helm_client_bsk, host.helm_client_bsk_words(self._h))

    @property
    def ksk(self):
        return self._view(host.helm_client_ksk, host.helm_client_ksk_words(self._h))

    @property
    def lwe_secret(self):
        return self._view(host.helm_client_lwe_secret, self.params.n)

    @property
    def glwe_secret(self):
        return self._view(host.helm_client_glwe_secret, self.params.k * self.params.N)

    def encrypt(self, bits):
        """bool or sequence of bools -> [count, n+1] uint32 (ClientKey::encrypt)."""
        scalar = np.isscalar(bits) or isinstance(bits, (bool, np.bool_))
        b = np.ascontiguousarray(np.atleast_1d(np.asarray(bits)).astype(np.uint8))
        out = np.zeros((len(b), self.params.n + 1), dtype=np.uint32)
        client_check(host.helm_client_encrypt_bool(self._h, nv.as_u8p(b), len(b), nv.as_u32p(out)))
        return out[0] if scalar else out

    def decrypt(self, lwe):
        """[count, n+1] (or one row) -> bool array (ClientKey::decrypt)."""
        a = np.ascontiguousarray(lwe, dtype=np.uint32)
        one = a.ndim == 1
        a2 = a.reshape(-1, self.params.n + 1)
        out = np.zeros(len(a2), dtype=np.uint8)
        client_check(host.helm_client_decrypt_bool(self._h, nv.as_u32p(a2), len(a2), nv.as_u8p(out)))
        return bool(out[0]) if one else out.astype(bool)

    def phase(self, lwe, big=False):
        dim = self.params.k * self.params.N if big else self.params.n
        a2 = np.ascontiguousarray(lwe, dtype=np.uint32).reshape(-1, dim + 1)
        out = np.zeros(len(a2), dtype=np.uint32)
        client_check(host.helm_client_phase(self._h, nv.as_u32p(a2), len(a2), int(big), nv.as_u32p(out)))
        return out


class ServerKey:
    """GPU engine context with the bootstrapping and keyswitching keys resident
    in HBM.  Construction fails (HelmError) when no gfx950 device is usable."""

    def __init__(self, client_key=None, params=None, bsk=None, ksk=None, device=0):
        self.params = client_key.params if client_key is not None else params
        h = nv.vp()
        hip_check(hip.helm_hip_ctx_create(device, C.byref(self.params), C.byref(h)))
        self._h = h
        self.device = device
        if client_key is not None:
            bsk, ksk = client_key.bsk, client_key.ksk
        if bsk is not None:
            bsk = np.ascontiguousarray(bsk, dtype=np.uint32).reshape(-1)
            hip_check(hip.helm_hip_load_bootstrap_key(self._h, nv.as_u32p(bsk), bsk.size))
        if ksk is not None:
            ksk = np.ascontiguousarray(ksk, dtype=np.uint32).reshape(-1)
            hip_check(hip.helm_hip_load_keyswitch_key(self._h, nv.as_u32p(ksk), ksk.size))

    def close(self):
        if getattr(self, "_h", None):
            hip.helm_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_stream(self, stream_ptr):
        nv.require_one_hip_runtime(type(self).__name__ + ".set_stream")  # the handle is another framework's
        hip_check(hip.helm_hip_set_stream(self._h, nv.vp(stream_ptr)))

    def sync(self):
        hip_check(hip.helm_hip_sync(self._h))

    def launch_quantum(self):
        """Bootstraps of one full round of the lockstep build (4 per compute unit)."""
        q = int(hip.helm_hip_launch_quantum(self._h))
        if q <= 0:
            hip_check(q or -1)
        return q

    def wires(self, n_wires):
        return DeviceWires(self, n_wires)

    # primitive batch ops -------------------------------------------------
    def pbs_batch(self, lwe_in, test_vectors, tv_index=None):
        p = self.params
        lwe_in = np.ascontiguousarray(lwe_in, dtype=np.uint32).reshape(-1, p.n + 1)
        tvs = np.ascontiguousarray(test_vectors, dtype=np.uint32).reshape(-1, p.N)
        if tv_index is None:
            tv_index = np.zeros(len(lwe_in), dtype=np.int32)
        tv_index = np.ascontiguousarray(tv_index, dtype=np.int32)
        out = np.zeros((len(lwe_in), p.k * p.N + 1), dtype=np.uint32)
        hip_check(hip.helm_hip_pbs_batch(self._h, nv.as_u32p(lwe_in), nv.as_u32p(tvs), len(tvs), nv.as_i32p(tv_index),
                                         nv.as_u32p(out), len(lwe_in)))
        return out

    def keyswitch_batch(self, big):
        p = self.params
        big = np.ascontiguousarray(big, dtype=np.uint32).reshape(-1, p.k * p.N + 1)
        out = np.zeros((len(big), p.n + 1), dtype=np.uint32)
        hip_check(hip.helm_hip_keyswitch_batch(self._h, nv.as_u32p(big), nv.as_u32p(out), len(big)))
        return out

    def ntt_roundtrip(self, polys):
        polys = np.ascontiguousarray(polys, dtype=np.uint32).reshape(-1, self.params.N)
        out = np.zeros_like(polys)
        hip_check(hip.helm_hip_ntt_roundtrip(self._h, nv.as_u32p(polys), nv.as_u32p(out), len(polys)))
        return out

    def launch_costs(self):
        """Relative cost of a launch of at most 1/4, 2/4, 3/4, 4/4 of launch_quantum() bootstraps (helm_hip_launch_costs):
        what pack_levels(..., quarter_cost=) sizes launches narrower than a round with."""
        c = (C.c_double * 4)()
        hip_check(hip.helm_hip_launch_costs(self._h, c))
        return [float(x) for x in c]

    def field_bits(self):
        """49: the blind-rotate kernels compute in the lazy field p = 5072^4 + 1 (short eighth roots of unity: two forward
        stages on digits without modular reductions), 51: in the 51-bit field, 50: N = 1024 in the lazy field p = 5440^4 + 1,
        chosen when the loaded key's own bound fits (helm_hip_field_bits)."""
        v = int(hip.helm_hip_field_bits(self._h))
        if v < 0:
            hip_check(v)
        return v

    def bound_violations(self, reset=True, selftest=False):
        """Check build only (HELM_HIP_LIB=libhelm_hip_check.so): violations of the lazy arithmetic's contracts counted by the
        kernels since the last reset -> [mulmod, reduce, butterfly, lean-inverse input, lift, 0, 0, 0]."""
        c = (C.c_uint32 * 8)()
        hip_check(hip.helm_hip_bound_violations(self._h, c, 1 if reset else 0, 1 if selftest else 0))
        return [int(v) for v in c]

    def short_root_stages(self):
        """Leading forward-transform stages on digits done as one radix-4 butterfly of plain operations (helm_hip_short_root_stages)."""
        v = int(hip.helm_hip_short_root_stages(self._h))
        if v < 0:
            hip_check(v)
        return v

    def kernel_clock_ghz(self):
        """Shader clock held during the most recent k_pbs launch (None before the first one)."""
        g, ms = C.c_double(), C.c_double()
        if hip.helm_hip_get_clock(self._h, C.byref(g), C.byref(ms)) != 0:
            return None
        return g.value

    def timing_enable(self, on=True):
        hip_check(hip.helm_hip_timing_enable(self._h, int(on)))

    def timing(self, reset=False):
        t = Timing()
        hip_check(hip.helm_hip_get_timing(self._h, C.byref(t), int(reset)))
        return t


class DeviceWires:
    """HBM-resident wire table: n_wires rows of n+1 words."""

    def __init__(self, server_key, n_wires):
        self.sk = server_key
        self.n_wires = int(n_wires)
        h = nv.vp()
        hip_check(hip.helm_hip_wires_alloc(server_key._h, self.n_wires, C.byref(h)))
        self._h = h

    def free(self):
        if getattr(self, "_h", None):
            # the owner may be gone already (destructor order is arbitrary): the C side then only deletes the
            # host struct - the context released the device memory
            hip.helm_hip_wires_free(getattr(self.sk, "_h", None), self._h)
        self._h = None

    def __del__(self):
        self.free()

    def upload(self, idx, lwe):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        lwe = np.ascontiguousarray(lwe, dtype=np.uint32).reshape(len(idx), self.sk.params.n + 1)
        hip_check(hip.helm_hip_wires_upload(self.sk._h, self._h, nv.as_i32p(idx), nv.as_u32p(lwe), len(idx)))

    def download(self, idx=None):
        if idx is None:
            idx = np.arange(self.n_wires)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        out = np.zeros((len(idx), self.sk.params.n + 1), dtype=np.uint32)
        hip_check(hip.helm_hip_wires_download(self.sk._h, self._h, nv.as_i32p(idx), nv.as_u32p(out), len(idx)))
        return out

    def set_trivial(self, idx, values):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(values), idx.shape).astype(np.uint8))
        hip_check(hip.helm_hip_wires_set_trivial(self.sk._h, self._h, nv.as_i32p(idx), nv.as_u8p(v), len(idx)))

    def device_ptr(self):
        p = nv.vp()
        n = C.c_int64()
        hip_check(hip.helm_hip_wires_device_ptr(self.sk._h, self._h, C.byref(p), C.byref(n)))
        return p.value

    def eval_gate_level(self, opcode, in0, in1, in2, out):
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (opcode, in0, in1, in2, out)]
        hip_check(hip.helm_hip_eval_gate_level(self.sk._h, self._h, *[nv.as_i32p(a) for a in arrs], len(arrs[0])))


class Program:
    """A levelised netlist uploaded once (level_map of reference circuit.rs:174-239)."""

    def __init__(self, server_key, opcode, in0, in1, in2, out, level_offsets):
        self.sk = server_key
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (opcode, in0, in1, in2, out)]
        off = np.ascontiguousarray(level_offsets, dtype=np.int64)
        self.n_levels = len(off) - 1
        self.level_offsets = off
        h = nv.vp()
        hip_check(hip.helm_hip_program_create(server_key._h, *[nv.as_i32p(a) for a in arrs], nv.as_i64p(off),
                                              self.n_levels, C.byref(h)))
        self._h = h

    def destroy(self):
        if getattr(self, "_h", None):
            hip.helm_hip_program_destroy(getattr(self.sk, "_h", None), self._h)  # NULL owner: host struct only
        self._h = None

    def __del__(self):
        self.destroy()

    def run(self, wires, level_begin=0, level_end=None):
        if level_end is None:
            level_end = self.n_levels
        hip_check(hip.helm_hip_program_run(self.sk._h, self._h, wires._h, level_begin, level_end))

    def level_pbs(self, level):
        return int(hip.helm_hip_program_level_pbs(self._h, level))

    def total_pbs(self):
        return sum(self.level_pbs(l) for l in range(self.n_levels))

    def chunk_rows(self, level, world):
        return int(hip.helm_hip_program_chunk_rows(self._h, level, world))

    def chunk_bounds(self, level, world):
        """The cut of a launch for `world` ranks, by bootstrap weight: rank r owns gates bounds[r] .. bounds[r+1]."""
        b = np.zeros(world + 1, dtype=np.int64)
        hip_check(hip.helm_hip_program_chunk_bounds(self._h, level, world, nv.as_i64p(b)))
        return b

    def overlap_applies(self):
        return bool(hip.helm_hip_program_overlap_applies(self._h))

    def shard_prepare(self, rank, world):
        hip_check(hip.helm_hip_program_shard_prepare(self.sk._h, self._h, rank, world))

    def run_level_shard(self, wires, level, rank, world, staging_ptr):
        hip_check(hip.helm_hip_program_run_level_shard(self.sk._h, self._h, wires._h, level, rank, world,
                                                       nv.vp(staging_ptr)))

    def run_sharded(self, wires, rank, world, stage_ptr, gather_ptr, capacity_rows, exchange, replicate_below=256):
        """The whole sharded pass inside the library (helm_hip_program_run_sharded): `exchange(stage_ptr, gather_ptr,
        rows_per_rank) -> 0` all-gathers on the engine's stream.  The callback object must outlive the call."""
        def _cb(_user, stage, gather, rows):
            try:
                return int(exchange(stage, gather, rows) or 0)
            except Exception:  # an exception cannot cross the C frames
                import traceback
                traceback.print_exc()
                return -1
        fn = nv.HIP_EXCHANGE_FN(_cb)
        hip_check(hip.helm_hip_program_run_sharded(self.sk._h, self._h, wires._h, int(rank), int(world), int(replicate_below),
                                                   nv.vp(stage_ptr), nv.vp(gather_ptr), int(capacity_rows), fn, None))

    def run_sharded_comm(self, wires, comm, replicate_below=256, overlap=False):
        """The whole sharded pass with the collective inside the library too (helm_hip_program_run_sharded_comm): every
        launch of more than `replicate_below` bootstraps is computed into this rank's slot of the program's gather
        buffer, all-gathered in place with ncclAllGather through `comm` (helm_amd.comm.Comm) on the engine's stream and
        scattered into the replicated wire table.  overlap: all-gather + scatter on the engine's exchange stream while the
        launches that do not need them run (same wire table)."""
        hip_check(hip.helm_hip_program_run_sharded_comm(self.sk._h, self._h, wires._h, comm._h, int(replicate_below),
                                                        1 if overlap else 0))

    def scatter_level(self, wires, level, world, gathered_ptr):
        hip_check(hip.helm_hip_program_scatter_level(self.sk._h, self._h, wires._h, level, world,
                                                     nv.vp(gathered_ptr)))
