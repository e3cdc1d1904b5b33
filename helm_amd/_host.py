"""ctypes bindings of include/helm_host.h (libhelm_host.so)."""
import ctypes as C

from ._native import host, vp, u32p, u64p, HelmError, Params  # noqa: F401

cp = C.c_char_p
cpp = C.POINTER(C.c_void_p)  # char** returned as raw pointer so we can free it

HOST_API = {
    "helm_host_last_error": (cp, []),
    "helm_host_free": (None, [vp]),
    "helm_host_read_verilog_file": (C.c_int, [cp, C.c_int, C.POINTER(vp)]),
    "helm_host_read_verilog_text": (C.c_int, [cp, C.c_int, C.POINTER(vp)]),
    "helm_host_netlist_free": (None, [vp]),
    "helm_host_netlist_list": (vp, [vp, C.c_int]),
    "helm_host_netlist_flags": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "helm_host_read_input_wires": (C.c_int, [cp, cp, C.POINTER(vp)]),
    "helm_host_write_output_wires": (C.c_int, [cp, cp]),
    "helm_host_parse_input_wire": (C.c_int, [cp, cp, C.POINTER(vp)]),
    "helm_host_hex_to_bitstring": (C.c_int, [cp, C.POINTER(vp)]),
    "helm_host_circuit_new": (C.c_int, [vp, cp, cp, cp, C.POINTER(vp)]),
    "helm_host_circuit_free": (None, [vp]),
    "helm_host_circuit_sort_circuit": (C.c_int, [vp]),
    "helm_host_circuit_compute_levels": (C.c_int, [vp]),
    "helm_host_circuit_get_ordered_gates": (vp, [vp]),
    "helm_host_circuit_level_map": (vp, [vp]),
    "helm_host_circuit_initialize_wire_map": (C.c_int, [vp, cp, cp, cp, C.POINTER(vp)]),
    "helm_host_circuit_evaluate": (C.c_int, [vp, cp, C.POINTER(vp)]),
    "helm_host_preprocess": (C.c_int, [cp, C.c_int, C.POINTER(vp)]),
    "helm_host_pack_levels": (C.c_int, [C.POINTER(C.c_int32)] * 5 + [C.POINTER(C.c_int64), C.c_int64, C.c_int64] +
                              [C.POINTER(C.c_int64)] * 3),
    "helm_host_pack_levels_costed": (C.c_int, [C.POINTER(C.c_int32)] * 5 + [C.POINTER(C.c_int64), C.c_int64, C.c_int64,
                                                C.POINTER(C.c_double)] + [C.POINTER(C.c_int64)] * 3),
    "helm_host_shard_bounds": (C.c_int64, [C.POINTER(C.c_int32), C.c_int64, C.c_int, C.POINTER(C.c_int64)]),
    "helm_host_enc_map_new": (C.c_int, [vp, C.POINTER(vp)]),
    "helm_host_enc_map_free": (None, [vp]),
    "helm_host_enc_map_insert": (C.c_int, [vp, cp, u32p]),
    "helm_host_enc_map_get": (C.c_int, [vp, cp, u32p]),
    "helm_host_enc_map_contains_key": (C.c_int, [vp, cp]),
    "helm_host_enc_map_keys": (vp, [vp]),
    "helm_host_gate_circuit_new": (C.c_int, [vp, vp, vp, C.POINTER(vp)]),
    "helm_host_gate_circuit_free": (None, [vp]),
    "helm_host_gate_circuit_encrypt_inputs": (C.c_int, [vp, cp, cp, C.POINTER(vp)]),
    "helm_host_gate_circuit_evaluate_encrypted": (C.c_int, [vp, vp, C.c_int64, cp, C.POINTER(vp)]),
    "helm_host_gate_circuit_init_ready": (C.c_int, [vp, C.POINTER(vp)]),
    "helm_host_gate_circuit_evaluate_ready": (C.c_int, [vp, vp, vp]),
    "helm_host_gate_circuit_decrypt_outputs": (C.c_int, [vp, vp, C.c_int, C.POINTER(vp)]),
    "helm_host_gate_circuit_log": (vp, [vp]),
    "helm_host_gate_circuit_pbs_per_cycle": (C.c_int64, [vp]),
    "helm_host_gate_circuit_memo_hits": (C.c_int64, [vp]),
    "helm_host_gate_circuit_shard_over": (C.c_int, [vp, vp, C.c_int64]),
    "helm_host_gate_circuit_set_exchange_overlap": (C.c_int, [vp, C.c_int]),
    "helm_host_si_circuit_new": (C.c_int, [C.c_int, vp, vp, vp, C.POINTER(vp)]),
    "helm_host_si_circuit_free": (None, [vp]),
    "helm_host_si_circuit_encrypt_inputs": (C.c_int, [vp, cp, cp, C.POINTER(vp)]),
    "helm_host_si_circuit_evaluate_encrypted": (C.c_int, [vp, vp, C.c_int64, cp, C.POINTER(vp)]),
    "helm_host_si_circuit_init_ready": (C.c_int, [vp, C.POINTER(vp)]),
    "helm_host_si_circuit_evaluate_ready": (C.c_int, [vp, vp, vp]),
    "helm_host_si_circuit_decrypt_outputs": (C.c_int, [vp, vp, C.c_int, C.POINTER(vp)]),
    "helm_host_si_circuit_set_wopbs": (C.c_int, [vp, vp, C.c_int]),
    "helm_host_si_circuit_add_lane": (C.c_int, [vp, vp]),
    "helm_host_si_circuit_set_lazy_carries": (C.c_int, [vp, C.c_int]),
    "helm_host_si_circuit_set_round_capacity": (C.c_int, [vp, C.c_int64]),
    "helm_host_si_circuit_set_memo": (C.c_int, [vp, C.c_int]),
    "helm_host_si_circuit_reset_memo": (C.c_int, [vp]),
    "helm_host_si_circuit_set_timing_lines": (C.c_int, [vp, C.c_int]),
    "helm_host_si_circuit_log": (vp, [vp]),
    "helm_host_radix_scratch_rows": (C.c_int64, [vp, C.c_int32, vp, C.c_int64]),
    "helm_host_radix_level": (C.c_int, [vp, vp, C.c_int32, vp, C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "helm_host_si_circuit_pbs_per_cycle": (C.c_int64, [vp]),
    "helm_host_si_circuit_pbs_rounds_per_cycle": (C.c_int64, [vp]),
    "helm_host_si_circuit_memo_hits": (C.c_int64, [vp]),
    "helm_host_si_enc_map_new": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "helm_host_si_enc_map_free": (None, [vp]),
    "helm_host_si_enc_map_blocks": (C.c_int, [vp]),
    "helm_host_si_enc_map_row_words": (C.c_int, [vp]),
    "helm_host_si_enc_map_insert": (C.c_int, [vp, cp, u64p]),
    "helm_host_si_enc_map_get": (C.c_int, [vp, cp, u64p]),
    "helm_host_si_enc_map_contains_key": (C.c_int, [vp, cp]),
    "helm_host_si_enc_map_keys": (vp, [vp]),
}
for _name, (_res, _args) in HOST_API.items():
    _fn = getattr(host, _name)
    _fn.restype = _res
    _fn.argtypes = _args


class Panic(HelmError):
    """The reference would have panicked here; the message is the panic message."""


def check(rc):
    if rc != 0:
        raise Panic(host.helm_host_last_error().decode())


def take(ptr):
    """malloc'd char* -> str (and free it)."""
    if not ptr:
        return ""
    s = C.string_at(ptr).decode()
    host.helm_host_free(ptr)
    return s


def out_text(fn, *args):
    p = vp()
    check(fn(*args, C.byref(p)))
    return take(p.value)


def nl(names):
    return "\n".join(names).encode()
