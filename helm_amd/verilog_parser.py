"""verilog_parser — mirror of reference src/verilog_parser.rs over the C++ host library."""
import ctypes as C

from . import _host as H
from .gates import parse_gate_lines, text_to_map, map_to_text


class GateSet:
    """HashSet<Gate> (first return value of read_verilog_file): owns the parsed netlist."""

    def __init__(self, handle):
        self._h = handle
        self._gates = None

    def __del__(self):
        if getattr(self, "_h", None):
            H.host.helm_host_netlist_free(self._h)
            self._h = None

    def _load(self):
        if self._gates is None:
            self._gates = parse_gate_lines(H.take(H.host.helm_host_netlist_list(self._h, 0)))
        return self._gates

    def __len__(self):
        return len(self._load())

    def __iter__(self):
        return iter(self._load())


def _wrap(handle):
    gs = GateSet(handle)
    lst = lambda w: [x for x in H.take(H.host.helm_host_netlist_list(handle, w)).splitlines() if x]
    a, b = C.c_int(), C.c_int()
    H.host.helm_host_netlist_flags(handle, C.byref(a), C.byref(b))
    return gs, set(lst(1)), lst(2), lst(3), lst(4), bool(a.value), bool(b.value)


def read_verilog_file(file_name, is_arith):
    """-> (gates, wire_set, inputs, outputs, dff_outputs, has_luts, has_arith)
    reference src/verilog_parser.rs:138-276"""
    h = H.vp()
    H.check(H.host.helm_host_read_verilog_file(str(file_name).encode(), int(is_arith), C.byref(h)))
    return _wrap(h)


def read_verilog_text(text, is_arith):
    h = H.vp()
    H.check(H.host.helm_host_read_verilog_text(text.encode(), int(is_arith), C.byref(h)))
    return _wrap(h)


def read_input_wires(file_name, ptxt_type):
    """reference src/verilog_parser.rs:278-317"""
    return text_to_map(H.out_text(H.host.helm_host_read_input_wires, str(file_name).encode(), ptxt_type.encode()))


def write_output_wires(file_name, input_map):
    """reference src/verilog_parser.rs:319-349"""
    if file_name is None:
        return
    H.check(H.host.helm_host_write_output_wires(str(file_name).encode(), map_to_text(input_map).encode()))


def parse_input_wire(wire, ptxt_type):
    """reference src/lib.rs:90-106"""
    return text_to_map(H.out_text(H.host.helm_host_parse_input_wire, wire.encode(), ptxt_type.encode()))["v"]


def hex_to_bitstring(hex_string):
    """reference src/lib.rs:181-194"""
    return H.out_text(H.host.helm_host_hex_to_bitstring, hex_string.encode())
