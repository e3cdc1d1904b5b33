"""Circuit, EvalCircuit, GateCircuit — mirror of reference src/circuit.rs over the C++
host library (helm_amd/csrc/host).  The level loop, the device program and all
ciphertext handling live in C++; this file only marshals names and values."""
import ctypes as C

import numpy as np

from . import _host as H
from . import _native as nv
from .gates import parse_gate_lines, text_to_map, map_to_text


class Circuit:
    """reference src/circuit.rs:60-67, 104-381"""

    def __init__(self, gates, input_wires, output_wires, dff_outputs):
        h = H.vp()
        H.check(H.host.helm_host_circuit_new(gates._h, H.nl(input_wires), H.nl(output_wires), H.nl(dff_outputs),
                                             C.byref(h)))
        self._h = h
        self.input_wires, self.output_wires, self.dff_outputs = list(input_wires), list(output_wires), list(dff_outputs)

    def __del__(self):
        if getattr(self, "_h", None):
            H.host.helm_host_circuit_free(self._h)
            self._h = None

    def sort_circuit(self):
        H.check(H.host.helm_host_circuit_sort_circuit(self._h))

    def compute_levels(self):
        H.check(H.host.helm_host_circuit_compute_levels(self._h))

    def get_ordered_gates(self):
        return parse_gate_lines(H.take(H.host.helm_host_circuit_get_ordered_gates(self._h)), with_level=True)

    def level_map(self):
        m = {}
        for g in parse_gate_lines(H.take(H.host.helm_host_circuit_level_map(self._h)), with_level=True):
            m.setdefault(g.level, []).append(g)
        return m

    def initialize_wire_map(self, wire_set, user_inputs, ptxt_type):
        return text_to_map(H.out_text(H.host.helm_host_circuit_initialize_wire_map, self._h, H.nl(sorted(wire_set)),
                                      map_to_text(user_inputs).encode(), ptxt_type.encode()))

    def evaluate(self, wire_map):
        return text_to_map(H.out_text(H.host.helm_host_circuit_evaluate, self._h, map_to_text(wire_map).encode()))


class EncWireMap:
    """HashMap<String, Ciphertext> whose values live in an HBM wire table."""

    def __init__(self, server_key=None, _handle=None, _n=None):
        if _handle is None:
            h = H.vp()
            H.check(H.host.helm_host_enc_map_new(server_key._h, C.byref(h)))
            _handle, _n = h, server_key.params.n
        self._h, self._n = _handle, _n

    def __del__(self):
        if getattr(self, "_h", None):
            H.host.helm_host_enc_map_free(self._h)
            self._h = None

    def insert(self, wire, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint32)
        assert ct.shape == (self._n + 1,)
        H.check(H.host.helm_host_enc_map_insert(self._h, wire.encode(), nv.as_u32p(ct)))

    __setitem__ = insert

    def __getitem__(self, wire):
        out = np.zeros(self._n + 1, dtype=np.uint32)
        H.check(H.host.helm_host_enc_map_get(self._h, wire.encode(), nv.as_u32p(out)))
        return out

    def contains_key(self, wire):
        return bool(H.host.helm_host_enc_map_contains_key(self._h, wire.encode()))

    __contains__ = contains_key

    def keys(self):
        return [k for k in H.take(H.host.helm_host_enc_map_keys(self._h)).splitlines() if k]

    def __len__(self):
        return len(self.keys())


class EvalCircuit:
    """trait EvalCircuit<C>, reference src/circuit.rs:35-58 (static-call style of the tests:
    EvalCircuit.evaluate_encrypted(circuit, map, cycle, datatype))."""

    @staticmethod
    def encrypt_inputs(c, wire_set, input_wire_map): return c.encrypt_inputs(wire_set, input_wire_map)
    @staticmethod
    def evaluate_encrypted(c, enc_wire_map, current_cycle, ptxt_type): return c.evaluate_encrypted(enc_wire_map, current_cycle, ptxt_type)
    @staticmethod
    def init_ready(c): return c.init_ready()
    @staticmethod
    def evaluate_ready(c, enc_wire_map, valid_outputs): return c.evaluate_ready(enc_wire_map, valid_outputs)
    @staticmethod
    def decrypt_outputs(c, enc_wire_map, verbose): return c.decrypt_outputs(enc_wire_map, verbose)


class GateCircuit(EvalCircuit):
    """reference src/circuit.rs:69-73, 449-577"""

    def __init__(self, client_key, server_key, circuit):
        h = H.vp()
        H.check(H.host.helm_host_gate_circuit_new(client_key._h, server_key._h, circuit._h, C.byref(h)))
        self._h = h
        self._ck, self._sk, self.circuit = client_key, server_key, circuit  # keep alive
        self._n = server_key.params.n

    def __del__(self):
        if getattr(self, "_h", None):
            H.host.helm_host_gate_circuit_free(self._h)
            self._h = None

    def _map(self, fn, *args):
        h = H.vp()
        H.check(fn(self._h, *args, C.byref(h)))
        return EncWireMap(_handle=h, _n=self._n)

    def encrypt_inputs(self, wire_set, input_wire_map):
        return self._map(H.host.helm_host_gate_circuit_encrypt_inputs, H.nl(sorted(wire_set)),
                         map_to_text(input_wire_map).encode())

    def evaluate_encrypted(self, enc_wire_map, current_cycle, ptxt_type="bool"):
        return self._map(H.host.helm_host_gate_circuit_evaluate_encrypted, enc_wire_map._h, int(current_cycle),
                         ptxt_type.encode())

    def init_ready(self):
        return self._map(H.host.helm_host_gate_circuit_init_ready)

    def evaluate_ready(self, enc_wire_map, valid_outputs):
        H.check(H.host.helm_host_gate_circuit_evaluate_ready(self._h, enc_wire_map._h, valid_outputs._h))

    def decrypt_outputs(self, enc_wire_map, verbose=False):
        return text_to_map(H.out_text(H.host.helm_host_gate_circuit_decrypt_outputs, self._h, enc_wire_map._h,
                                      int(verbose)))

    def log(self):
        return H.take(H.host.helm_host_gate_circuit_log(self._h))

    def pbs_per_cycle(self):
        return int(H.host.helm_host_gate_circuit_pbs_per_cycle(self._h))

    def memo_hits(self):
        """evaluate_encrypted calls answered from the same-cycle memo (reference src/gates.rs:55-59)."""
        return int(H.host.helm_host_gate_circuit_memo_hits(self._h))

    def shard_over(self, comm, replicate_below=256, overlap=False):
        """Multi-GPU (one process per GPU): split every launch of more than `replicate_below` bootstraps over the ranks of
        `comm` (helm_amd.comm.Comm, the engine's own RCCL communicator) and all-gather the output ciphertexts inside the
        engine - the level of reference src/circuit.rs:531 is the sharded unit.  Same keys, circuit and inputs on every
        rank; every rank gets the wire map of a one-GPU evaluation.  comm = None: back to one GPU.  overlap: the exchange
        of a launch runs beside the launches that do not need its outputs."""
        H.check(H.host.helm_host_gate_circuit_shard_over(self._h, comm._h if comm is not None else None, int(replicate_below)))
        H.check(H.host.helm_host_gate_circuit_set_exchange_overlap(self._h, 1 if overlap else 0))
        self._comm = comm  # keep alive


class SiEncWireMap:
    """HashMap<String, CtxtShortInt> / HashMap<String, FheType> whose values live in an HBM table of
    big-LWE rows: `blocks` rows per wire (1 in LUT mode, 4..64 for FheUint8..128)."""

    def __init__(self, server_key=None, blocks=1, _handle=None):
        if _handle is None:
            h = H.vp()
            H.check(H.host.helm_host_si_enc_map_new(server_key._h, int(blocks), C.byref(h)))
            _handle = h
        self._h = _handle
        self.blocks = int(H.host.helm_host_si_enc_map_blocks(self._h))
        self.row_words = int(H.host.helm_host_si_enc_map_row_words(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            H.host.helm_host_si_enc_map_free(self._h)
            self._h = None

    def insert(self, wire, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint64).reshape(self.blocks, self.row_words)
        H.check(H.host.helm_host_si_enc_map_insert(self._h, wire.encode(), nv.as_u64p(ct)))

    __setitem__ = insert

    def __getitem__(self, wire):
        out = np.zeros((self.blocks, self.row_words), dtype=np.uint64)
        H.check(H.host.helm_host_si_enc_map_get(self._h, wire.encode(), nv.as_u64p(out)))
        return out[0] if self.blocks == 1 else out

    def contains_key(self, wire):
        return bool(H.host.helm_host_si_enc_map_contains_key(self._h, wire.encode()))

    __contains__ = contains_key

    def keys(self):
        return [k for k in H.take(H.host.helm_host_si_enc_map_keys(self._h)).splitlines() if k]

    def __len__(self):
        return len(self.keys())


class _SiCircuit(EvalCircuit):
    MODE = 0

    def __init__(self, client_key, server_key, circuit):
        h = H.vp()
        # client_key None: evaluation only (the caller encrypts / decrypts and moves rows through SiEncWireMap)
        H.check(H.host.helm_host_si_circuit_new(self.MODE, client_key._h if client_key is not None else None, server_key._h,
                                               circuit._h, C.byref(h)))
        self._h = h
        self._ck, self._sk, self.circuit = client_key, server_key, circuit  # keep alive

    def __del__(self):
        if getattr(self, "_h", None):
            H.host.helm_host_si_circuit_free(self._h)
            self._h = None

    def _map(self, fn, *args):
        h = H.vp()
        H.check(fn(self._h, *args, C.byref(h)))
        return SiEncWireMap(_handle=h)

    def encrypt_inputs(self, wire_set, input_wire_map):
        return self._map(H.host.helm_host_si_circuit_encrypt_inputs, H.nl(sorted(wire_set)),
                         map_to_text(input_wire_map).encode())

    def evaluate_encrypted(self, enc_wire_map, current_cycle, ptxt_type="bool"):
        return self._map(H.host.helm_host_si_circuit_evaluate_encrypted, enc_wire_map._h, int(current_cycle),
                         ptxt_type.encode())

    def init_ready(self):
        return self._map(H.host.helm_host_si_circuit_init_ready)

    def evaluate_ready(self, enc_wire_map, valid_outputs):
        H.check(H.host.helm_host_si_circuit_evaluate_ready(self._h, enc_wire_map._h, valid_outputs._h))

    def decrypt_outputs(self, enc_wire_map, verbose=False):
        return text_to_map(H.out_text(H.host.helm_host_si_circuit_decrypt_outputs, self._h, enc_wire_map._h,
                                      int(verbose)))

    def log(self):
        return H.take(H.host.helm_host_si_circuit_log(self._h))

    def pbs_per_cycle(self):
        return int(H.host.helm_host_si_circuit_pbs_per_cycle(self._h))

    def pbs_rounds_per_cycle(self):
        return int(H.host.helm_host_si_circuit_pbs_rounds_per_cycle(self._h))

    def memo_hits(self):
        """evaluate_encrypted calls answered from the same-cycle memo (reference src/gates.rs:288-292, 307-312)."""
        return int(H.host.helm_host_si_circuit_memo_hits(self._h))


class LutCircuit(_SiCircuit):
    """reference src/circuit.rs:75-79, 969-1120"""
    MODE = 0

    def set_wide_lut_key(self, wop_server_key, bits_per_block=1):
        """LUT gates with more inputs than one block's index bits go through the WoP-PBS path
        (Gate::evaluate_encrypted_high_precision_lut, reference src/gates.rs:721-742).  bits_per_block: 1 when every
        wire holds one bit (LUT mode's wires do); log2(message_modulus * carry_modulus) is what tfhe's degree
        bookkeeping would extract.  None switches the path off."""
        H.check(H.host.helm_host_si_circuit_set_wopbs(self._h, wop_server_key._h if wop_server_key is not None else None,
                                                      int(bits_per_block)))
        self._wop = wop_server_key  # keep alive

    def set_timing_lines(self, on=True):
        """The per-gate `PBS time: {} us` lines of reference src/gates.rs:293-302 (default on) cost one host
        synchronisation per level; False drops the lines and the synchronisation."""
        H.check(H.host.helm_host_si_circuit_set_timing_lines(self._h, int(bool(on))))


class ArithCircuit(_SiCircuit):
    """reference src/circuit.rs:81-85, 1112-1500

    SAME-CYCLE MEMO - read this before calling evaluate_encrypted twice.  As in the reference (src/gates.rs:307-312:
    `if self.cycle == cycle { return cached }`, the operands are not looked at; tests/gates_test.rs:196-223 relies on
    it, tests/circuit_test.rs:314-474 passes cycles 1..4 to defeat it), the memo is keyed on the cycle number ALONE:
    evaluate_encrypted(other_inputs, same_cycle) returns the FIRST call's gate outputs and launches nothing.
    GateCircuit and LutCircuit differ: they only answer from the memo for the very same, unmodified input map.  Pass a
    new cycle for new inputs, or call reset_memo() / set_memo(False).  set_lanes, set_lazy_carries and
    set_round_capacity reset the memo."""
    MODE = 1

    def set_memo(self, on=True):
        """Switch the same-cycle memo (class docstring) off or back on (helm_host_si_circuit_set_memo)."""
        H.check(H.host.helm_host_si_circuit_set_memo(self._h, int(bool(on))))

    def reset_memo(self):
        """Forget the remembered cycle and release the device copy of its wire map."""
        H.check(H.host.helm_host_si_circuit_reset_memo(self._h))

    def set_round_capacity(self, capacity=0):
        """Merged rounds: launches of at most `capacity` ciphertexts (0 = what the device bootstraps at once).  A round
        is cut where it does not fit; every round is checked to list the readers of a row before its in-place writer."""
        H.check(H.host.helm_host_si_circuit_set_round_capacity(self._h, int(capacity)))

    def set_lazy_carries(self, on=True):
        """Carry-save products feeding additions / subtractions (default on; helm_host_si_circuit_set_lazy_carries)."""
        H.check(H.host.helm_host_si_circuit_set_lazy_carries(self._h, int(bool(on))))

    def set_lanes(self, n):
        """Evaluate sub-circuits that share no wire concurrently on `n` contexts in all (the server key and n - 1 lanes
        forked from it) instead of level by level; identical ciphertexts.  n = 1 switches lanes off."""
        H.check(H.host.helm_host_si_circuit_add_lane(self._h, None))
        self._lane_keys = [self._sk.fork() for _ in range(max(0, int(n) - 1))]
        for lane in self._lane_keys:
            H.check(H.host.helm_host_si_circuit_add_lane(self._h, lane._h))
