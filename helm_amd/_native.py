"""ctypes bindings of the in-tree native libraries.

libhelm_hip.so  (include/helm_hip.h)    HIP kernels + C ABI  -- the hot path
libhelm_host.so (include/helm_client.h) CPU client + netlist front end

There is no Python or CPU fallback for the hot path: a missing library raises
ImportError here, and a missing GPU makes helm_hip_ctx_create() fail with
HELM_ERR_NO_DEVICE, which surfaces as HelmError.
"""
import ctypes as C
import os

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


class HelmError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [(f, C.c_int32) for f in ("torus_bits", "n", "k", "N", "pbs_l", "pbs_logB", "ks_l", "ks_logB",
                                         "pbs_order", "grouping_factor")]

    def as_tuple7(self):
        """(n, k, N, pbs_l, pbs_logB, ks_l, ks_logB)"""
        return (self.n, self.k, self.N, self.pbs_l, self.pbs_logB, self.ks_l, self.ks_logB)

    def __repr__(self):
        return "Params(" + ", ".join(f"{f}={getattr(self, f)}" for f, _ in self._fields_) + ")"


class Timing(C.Structure):
    _fields_ = [("pbs_ms", C.c_double), ("ks_ms", C.c_double), ("linear_ms", C.c_double),
                ("pbs_launches", C.c_int64), ("pbs_count", C.c_int64),
                ("ks_launches", C.c_int64), ("ks_count", C.c_int64)]


class HipTiming(C.Structure):  # helm_hip_timing
    _fields_ = Timing._fields_ + [("pbs_main_ms", C.c_double), ("pbs_main_launches", C.c_int64),
                                  ("pbs_main_count", C.c_int64), ("exchange_ms", C.c_double),
                                  ("exchange_count", C.c_int64), ("exchange_bytes", C.c_int64)]


class SiParams(C.Structure):
    _fields_ = [(f, C.c_int32) for f in ("n", "k", "N", "pbs_l", "pbs_logB", "ks_l", "ks_logB",
                                         "message_modulus", "carry_modulus", "grouping_factor")]

    def as_tuple(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)

    def __repr__(self):
        return "SiParams(" + ", ".join(f"{f}={getattr(self, f)}" for f, _ in self._fields_) + ")"


class WopParams(C.Structure):  # helm_wop_params
    _fields_ = [(f, C.c_int32) for f in ("n", "k", "N", "pbs_l", "pbs_logB", "ks_l", "ks_logB", "pfks_l", "pfks_logB",
                                         "cbs_l", "cbs_logB", "message_modulus", "carry_modulus")]

    def as_tuple(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)

    def __repr__(self):
        return "WopParams(" + ", ".join(f"{f}={getattr(self, f)}" for f, _ in self._fields_) + ")"


class WopTiming(C.Structure):  # helm_wop_timing
    _fields_ = [(f, C.c_double) for f in ("clean_ms", "to_wopbs_ms", "extract_ms", "cbs_pbs_ms", "pfpks_ms",
                                          "convert_ms", "packing_ms", "to_pbs_ms")] + \
               [("gates", C.c_int64), ("bootstraps", C.c_int64)]


u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p
HIP_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, vp, vp, vp, C.c_int64)  # helm_hip_exchange_fn


def mapped_hip_runtimes():
    """Real paths of the libamdhip64 copies mapped into this process (Linux: /proc/self/maps)."""
    found = []
    try:
        with open("/proc/self/maps") as maps:
            for line in maps:
                fields = line.split(None, 5)  # address perms offset dev inode [path]
                path = fields[5].strip() if len(fields) == 6 else ""
                if path.startswith("/") and os.path.basename(path).startswith("libamdhip64.so"):
                    path = os.path.realpath(path)
                    if path not in found:
                        found.append(path)
    except OSError:
        pass
    return found


def _bind_one_hip_runtime():
    """One HIP runtime per process, whatever the import order.

    libhelm_hip.so NEEDs `libamdhip64.so.7` with a RUNPATH into the ROCm installation; PyTorch's libtorch_hip.so NEEDs
    `libamdhip64.so` and finds the wheel's own copy next to itself.  Loaded in that order the process holds two runtimes,
    and a torch stream handed to helm_hip_set_stream (or one of our device ranges wrapped in a torch tensor) crosses from
    one to the other: round 5's `std::bad_variant_access` abort.  So, BEFORE libhelm_hip.so is opened:
      * a runtime is already mapped (torch, or the host application's): open that very file RTLD_GLOBAL - its SONAME
        `libamdhip64.so.7` then satisfies our NEEDED entry and the RUNPATH is never searched;
      * none is mapped and PyTorch is installed (found without importing it): open the wheel's copy RTLD_GLOBAL - we run
        on it, and a later `import torch` finds the file it looks for already mapped (same inode) and reuses it;
      * neither: the RUNPATH's copy, as linked.
    HELM_HIP_RUNTIME=<path> names the copy explicitly; HELM_HIP_RUNTIME=system keeps the RUNPATH's even when PyTorch is
    installed (the caller then must not import torch afterwards and exchange handles: the guards raise HelmError)."""
    forced = os.environ.get("HELM_HIP_RUNTIME", "")
    if forced == "system":
        return None
    mapped = mapped_hip_runtimes()
    if forced:
        choice = forced
    elif mapped:
        choice = mapped[0]
    else:
        choice = None
        try:
            import importlib.util
            spec = importlib.util.find_spec("torch")  # locates the package, does not import it
        except (ImportError, ValueError):
            spec = None
        for root in (spec.submodule_search_locations or []) if spec is not None else []:
            cand = os.path.join(root, "lib", "libamdhip64.so")
            if os.path.exists(cand):
                choice = cand
                break
    if choice is None:
        return None
    try:
        C.CDLL(choice, mode=C.RTLD_GLOBAL)
    except OSError as e:
        if forced:
            raise ImportError(f"HELM_HIP_RUNTIME={forced}: {e}")
        return None  # the wheel's copy does not load here (a CPU-only torch build): the RUNPATH's copy it is
    return os.path.realpath(choice)


def _load(name):
    path = name if os.path.isabs(name) else os.path.join(_CSRC, name)
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `make -C {_CSRC}` "
                          "(or __graft_entry__.build()); helm_amd has no fallback path")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


hip_runtime = _bind_one_hip_runtime()  # the libamdhip64 this process was bound to before libhelm_hip.so (None: as linked)
# HELM_HIP_LIB: another build of libhelm_hip.so (same-box A/B of compile-time switches, tools/ab_variants.py) - a file
# name under csrc/ or an absolute path; loaded first, so libhelm_host.so's references bind to it as well.  The
# Makefile's library stays untouched (round 2's script copied the alternative over it).
_alt = os.environ.get("HELM_HIP_LIB")
hip = _load(_alt if _alt else "libhelm_hip.so")
host = _load("libhelm_host.so")

# every symbol include/helm_hip.h declares: (restype, argtypes)
HIP_API = {
    "helm_hip_last_error": (C.c_char_p, []),
    "helm_hip_device_count": (C.c_int, []),
    "helm_hip_runtime_copies": (C.c_int, [C.c_char_p, C.c_size_t]),
    "helm_hip_ctx_create": (C.c_int, [C.c_int, C.POINTER(Params), C.POINTER(vp)]),
    "helm_hip_ctx_destroy": (C.c_int, [vp]),
    "helm_hip_get_params": (C.c_int, [vp, C.POINTER(Params)]),
    "helm_hip_set_stream": (C.c_int, [vp, vp]),
    "helm_hip_sync": (C.c_int, [vp]),
    "helm_hip_launch_quantum": (C.c_int64, [vp]),
    "helm_hip_launch_costs": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "helm_hip_field_bits": (C.c_int, [vp]),
    "helm_hip_short_root_stages": (C.c_int, [vp]),
    "helm_hip_bound_violations": (C.c_int, [vp, C.POINTER(C.c_uint32), C.c_int, C.c_int]),
    "helm_hip_load_bootstrap_key": (C.c_int, [vp, u32p, C.c_size_t]),
    "helm_hip_load_keyswitch_key": (C.c_int, [vp, u32p, C.c_size_t]),
    "helm_hip_wires_alloc": (C.c_int, [vp, C.c_int64, C.POINTER(vp)]),
    "helm_hip_wires_free": (C.c_int, [vp, vp]),
    "helm_hip_wires_upload": (C.c_int, [vp, vp, i32p, u32p, C.c_int64]),
    "helm_hip_wires_download": (C.c_int, [vp, vp, i32p, u32p, C.c_int64]),
    "helm_hip_wires_set_trivial": (C.c_int, [vp, vp, i32p, u8p, C.c_int64]),
    "helm_hip_wires_copy": (C.c_int, [vp, vp, i32p, vp, i32p, C.c_int64]),
    "helm_hip_program_run_sharded": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int64, vp, vp, C.c_int64, HIP_EXCHANGE_FN, vp]),
    "helm_hip_program_run_sharded_comm": (C.c_int, [vp, vp, vp, vp, C.c_int64, C.c_int]),
    "helm_hip_program_overlap_applies": (C.c_int, [vp]),
    "helm_hip_program_chunk_bounds": (C.c_int, [vp, C.c_int64, C.c_int, C.POINTER(C.c_int64)]),
    "helm_hip_wires_device_ptr": (C.c_int, [vp, vp, C.POINTER(vp), i64p]),
    "helm_hip_eval_gate_level": (C.c_int, [vp, vp, i32p, i32p, i32p, i32p, i32p, C.c_int64]),
    "helm_hip_program_create": (C.c_int, [vp, i32p, i32p, i32p, i32p, i32p, i64p, C.c_int64, C.POINTER(vp)]),
    "helm_hip_program_destroy": (C.c_int, [vp, vp]),
    "helm_hip_program_run": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64]),
    "helm_hip_program_chunk_rows": (C.c_int64, [vp, C.c_int64, C.c_int]),
    "helm_hip_program_shard_prepare": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "helm_hip_program_run_level_shard": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, C.c_int, vp]),
    "helm_hip_program_scatter_level": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, vp]),
    "helm_hip_program_level_pbs": (C.c_int64, [vp, C.c_int64]),
    "helm_hip_pbs_batch": (C.c_int, [vp, u32p, u32p, C.c_int64, i32p, u32p, C.c_int64]),
    "helm_hip_keyswitch_batch": (C.c_int, [vp, u32p, u32p, C.c_int64]),
    "helm_hip_ntt_roundtrip": (C.c_int, [vp, u32p, u32p, C.c_int64]),
    "helm_hip_timing_enable": (C.c_int, [vp, C.c_int]),
    "helm_hip_get_timing": (C.c_int, [vp, C.POINTER(HipTiming), C.c_int]),
    "helm_hip_get_clock": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

# helm_si_exchange_fn: int (*)(void *user, int64_t rows_per_rank)
SI_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, vp, C.c_int64)


class SiAuditRecord(C.Structure):  # helm_si_audit_record
    _fields_ = [("kind", C.c_int32), ("terms", C.c_int32), ("count", C.c_int64), ("n_luts", C.c_int64),
                ("in_rows", C.POINTER(C.c_uint64)), ("out_rows", C.POINTER(C.c_uint64)), ("lut_idx", C.POINTER(C.c_int32)),
                ("luts", C.POINTER(C.c_uint64)), ("in_idx", C.POINTER(C.c_int32)), ("coef", C.POINTER(C.c_int64)),
                ("const_add", C.POINTER(C.c_int64))]


SI_AUDIT_FN = C.CFUNCTYPE(C.c_int, vp, C.POINTER(SiAuditRecord))  # helm_si_audit_fn

# every symbol include/helm_shortint.h declares
SI_API = {
    "helm_si_ctx_create": (C.c_int, [C.c_int, C.POINTER(SiParams), C.POINTER(vp)]),
    "helm_si_ctx_destroy": (C.c_int, [vp]),
    "helm_si_ctx_fork": (C.c_int, [vp, C.POINTER(vp)]),
    "helm_si_get_params": (C.c_int, [vp, C.POINTER(SiParams)]),
    "helm_si_field_bits": (C.c_int, [vp]),
    "helm_si_set_stream": (C.c_int, [vp, vp]),
    "helm_si_sync": (C.c_int, [vp]),
    "helm_si_load_bootstrap_key": (C.c_int, [vp, u64p, C.c_size_t]),
    "helm_si_load_keyswitch_key": (C.c_int, [vp, u64p, C.c_size_t]),
    "helm_si_wires_alloc": (C.c_int, [vp, C.c_int64, C.POINTER(vp)]),
    "helm_si_wires_free": (C.c_int, [vp, vp]),
    "helm_si_wires_upload": (C.c_int, [vp, vp, i32p, u64p, C.c_int64]),
    "helm_si_wires_download": (C.c_int, [vp, vp, i32p, u64p, C.c_int64]),
    "helm_si_wires_set_trivial": (C.c_int, [vp, vp, i32p, u64p, C.c_int64]),
    "helm_si_wires_copy": (C.c_int, [vp, vp, i32p, vp, i32p, C.c_int64]),
    "helm_si_lincomb": (C.c_int, [vp, vp, i32p, i64p, i64p, i32p, C.c_int32, C.c_int64]),
    "helm_si_make_lut": (C.c_int, [vp, u64p, u64p]),
    "helm_si_apply_luts": (C.c_int, [vp, vp, i32p, i32p, i32p, C.c_int64, u64p, C.c_int64]),
    "helm_si_eval_lut_level": (C.c_int, [vp, vp, i32p, i32p, C.c_int32, u64p, i32p, C.c_int64]),
    "helm_si_set_exchange": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int64, vp, vp, C.c_int64, SI_EXCHANGE_FN, vp]),
    "helm_si_set_exchange_comm": (C.c_int, [vp, vp, C.c_int64, C.c_int64]),
    "helm_si_set_audit": (C.c_int, [vp, SI_AUDIT_FN, vp]),
    "helm_si_bound_violations": (C.c_int, [vp, C.POINTER(C.c_uint32), C.c_int]),
    "helm_si_exchange_stats": (C.c_int, [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "helm_si_exchange_world": (C.c_int, [vp]),
    "helm_si_round_capacity": (C.c_int64, [vp]),
    "helm_si_set_priority": (C.c_int, [vp, C.c_int]),
    "helm_si_keyswitch_batch": (C.c_int, [vp, u64p, u64p, C.c_int64]),
    "helm_si_pbs_batch": (C.c_int, [vp, u64p, u64p, C.c_int64, i32p, u64p, C.c_int64]),
    "helm_si_timing_enable": (C.c_int, [vp, C.c_int]),
    "helm_si_get_timing": (C.c_int, [vp, C.POINTER(Timing), C.c_int]),
}

# every symbol include/helm_comm.h declares (the library's own RCCL communicator)
COMM_ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, vp, vp, vp, C.c_size_t, vp)  # helm_comm_all_gather_fn
COMM_API = {
    "helm_comm_available": (C.c_int, []),
    "helm_comm_precheck": (C.c_int, [C.c_int]),
    "helm_comm_create_in_process": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_double, C.POINTER(vp)]),
    "helm_comm_abort_group": (C.c_int, [vp]),
    "helm_comm_get_unique_id": (C.c_int, [u8p]),
    "helm_comm_create": (C.c_int, [C.c_int, u8p, C.c_int, C.c_int, C.POINTER(vp)]),
    "helm_comm_create_with_transport": (C.c_int, [C.c_int, C.c_int, C.c_int, COMM_ALL_GATHER_FN, vp, C.POINTER(vp)]),
    "helm_comm_destroy": (C.c_int, [vp]),
    "helm_comm_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "helm_comm_stats": (C.c_int, [vp, i64p, i64p]),
    "helm_comm_all_gather": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
    "helm_comm_all_reduce_f64": (C.c_int, [vp, C.POINTER(C.c_double), C.c_int]),
    "helm_comm_barrier": (C.c_int, [vp]),
}

WOP_CLIENT_API = {
    "helm_wop_client_named_params": (C.c_int, [C.c_char_p, C.POINTER(WopParams), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "helm_wop_client_keygen": (C.c_int, [vp, C.POINTER(WopParams), C.c_double, C.c_double, C.c_uint64, C.POINTER(vp)]),
    "helm_wop_client_key_free": (None, [vp]),
    "helm_wop_client_key_part": (C.c_int, [vp, C.c_int, C.POINTER(u64p), C.POINTER(C.c_size_t)]),
}

# every symbol include/helm_wopbs.h declares: device side (libhelm_hip.so) and client side (libhelm_host.so)
WOP_API = {
    "helm_wop_ctx_create": (C.c_int, [vp, C.POINTER(WopParams), C.POINTER(vp)]),
    "helm_wop_ctx_destroy": (C.c_int, [vp]),
    "helm_wop_get_params": (C.c_int, [vp, C.POINTER(WopParams)]),
    "helm_wop_load_key": (C.c_int, [vp, C.c_int, u64p, C.c_size_t, C.c_int32, C.c_int32]),
    "helm_wop_table_words": (C.c_size_t, [C.POINTER(WopParams), C.c_int32]),
    "helm_wop_make_table": (C.c_int, [C.POINTER(WopParams), C.c_int32, C.c_int32, u64p, C.c_size_t, u64p]),
    "helm_wop_eval_luts": (C.c_int, [vp, vp, i32p, C.c_int32, C.c_int32, u64p, i32p, C.c_int64]),
    "helm_wop_extract_bits_batch": (C.c_int, [vp, u64p, C.c_int32, C.c_int32, u64p, C.c_int64]),
    "helm_wop_circuit_bootstrap_batch": (C.c_int, [vp, u64p, u64p, C.c_int64]),
    "helm_wop_vertical_packing_batch": (C.c_int, [vp, u64p, C.c_int32, u64p, u64p, C.c_int64]),
    "helm_wop_get_timing": (C.c_int, [vp, C.POINTER(WopTiming), C.c_int]),
}
SI_CLIENT_API = {
    "helm_si_client_named_params": (C.c_int, [C.c_char_p, C.POINTER(SiParams), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "helm_si_client_keygen": (C.c_int, [C.POINTER(SiParams), C.c_double, C.c_double, C.c_uint64, C.POINTER(vp)]),
    "helm_si_client_key_free": (None, [vp]),
    "helm_si_client_params": (C.c_int, [vp, C.POINTER(SiParams)]),
    "helm_si_client_noise": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "helm_si_client_bsk_words": (C.c_size_t, [vp]),
    "helm_si_client_ksk_words": (C.c_size_t, [vp]),
    "helm_si_client_bsk": (u64p, [vp]),
    "helm_si_client_ksk": (u64p, [vp]),
    "helm_si_client_lwe_secret": (u64p, [vp]),
    "helm_si_client_glwe_secret": (u64p, [vp]),
    "helm_si_client_encrypt": (C.c_int, [vp, u64p, C.c_int64, u64p]),
    "helm_si_client_decrypt": (C.c_int, [vp, u64p, C.c_int64, u64p]),
    "helm_si_client_phase": (C.c_int, [vp, u64p, C.c_int64, C.c_int, u64p]),
}

CLIENT_API = {
    "helm_client_named_params": (C.c_int, [C.c_char_p, C.POINTER(Params), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "helm_client_last_error": (C.c_char_p, []),
    "helm_client_rng_selftest": (C.c_int, []),
    "helm_client_keygen": (C.c_int, [C.POINTER(Params), C.c_double, C.c_double, C.c_uint64, C.POINTER(vp)]),
    "helm_client_key_free": (None, [vp]),
    "helm_client_params": (C.c_int, [vp, C.POINTER(Params)]),
    "helm_client_bsk_words": (C.c_size_t, [vp]),
    "helm_client_ksk_words": (C.c_size_t, [vp]),
    "helm_client_bsk": (u32p, [vp]),
    "helm_client_ksk": (u32p, [vp]),
    "helm_client_lwe_secret": (u32p, [vp]),
    "helm_client_glwe_secret": (u32p, [vp]),
    "helm_client_encrypt_bool": (C.c_int, [vp, u8p, C.c_int64, u32p]),
    "helm_client_decrypt_bool": (C.c_int, [vp, u32p, C.c_int64, u8p]),
    "helm_client_phase": (C.c_int, [vp, u32p, C.c_int64, C.c_int, u32p]),
}

KEYS_API = {
    "helm_keys_last_error": (C.c_char_p, []),
    "helm_keys_bsk32_from_tfhe": (C.c_int, [C.POINTER(Params), u32p, u32p, C.c_size_t]),
    "helm_keys_bsk32_to_tfhe": (C.c_int, [C.POINTER(Params), u32p, u32p, C.c_size_t]),
    "helm_keys_ksk32_from_tfhe": (C.c_int, [C.POINTER(Params), u32p, u32p, C.c_size_t]),
    "helm_keys_ksk32_to_tfhe": (C.c_int, [C.POINTER(Params), u32p, u32p, C.c_size_t]),
    "helm_keys_bsk64_from_tfhe": (C.c_int, [C.POINTER(SiParams), u64p, u64p, C.c_size_t]),
    "helm_keys_bsk64_to_tfhe": (C.c_int, [C.POINTER(SiParams), u64p, u64p, C.c_size_t]),
    "helm_keys_ksk64_from_tfhe": (C.c_int, [C.POINTER(SiParams), u64p, u64p, C.c_size_t]),
    "helm_keys_ksk64_to_tfhe": (C.c_int, [C.POINTER(SiParams), u64p, u64p, C.c_size_t]),
    "helm_keys_levels64_reverse": (C.c_int, [C.c_size_t, C.c_int32, C.c_size_t, u64p, u64p, C.c_size_t]),
}

for _lib, _api in ((hip, HIP_API), (host, CLIENT_API), (hip, SI_API), (host, SI_CLIENT_API), (host, KEYS_API),
                   (hip, WOP_API), (host, WOP_CLIENT_API), (hip, COMM_API)):
    for _name, (_res, _args) in _api.items():
        _fn = getattr(_lib, _name)  # AttributeError here = header/library mismatch
        _fn.restype = _res
        _fn.argtypes = _args


def require_one_hip_runtime(where):
    """HelmError naming every copy when this process maps more than one libamdhip64 - called where a handle of another
    framework (a torch stream, a torch tensor over our device memory) is about to cross into the library.  The library's
    entry points check the same thing natively (helm_hip_runtime_copies); this is the early, Python-side message."""
    buf = C.create_string_buffer(4096)
    n = hip.helm_hip_runtime_copies(buf, len(buf))
    if n > 1:
        raise HelmError(f"{where}: this process maps {n} HIP runtimes ({buf.value.decode().replace(chr(10), ', ')}); a stream "
                        "or device pointer of one is not valid in the other.  Import helm_amd before anything dlopens a "
                        "second copy, or set HELM_HIP_RUNTIME to the copy the host framework uses (INTEGRATION.md, "
                        "\"One HIP runtime per process\")")


def hip_check(rc):
    if rc != 0:
        raise HelmError(f"helm_hip error {rc}: {hip.helm_hip_last_error().decode()}")


def client_check(rc):
    if rc != 0:
        raise HelmError(f"helm_client error {rc}: {host.helm_client_last_error().decode()}")


def as_u32p(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


def as_u64p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def as_i32p(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(i32p)


def as_i64p(a):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(i64p)


def as_u8p(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u8p)
