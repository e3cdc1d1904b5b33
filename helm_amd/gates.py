"""PtxtType / GateType / Gate — Python views of the C++ host objects
(reference src/lib.rs:20-29, src/gates.rs:23-60)."""
import enum
from dataclasses import dataclass, field
from typing import List, Optional


class GateType(enum.IntEnum):
    # declaration order of reference src/gates.rs:23-45 (= helm_gate_op)
    And = 0
    Dff = 1
    Lut = 2
    Mux = 3
    Nand = 4
    Nor = 5
    Not = 6
    Or = 7
    Xnor = 8
    Xor = 9
    Buf = 10
    ConstOne = 11
    ConstZero = 12
    Mult = 13
    Add = 14
    Sub = 15
    Div = 16
    Shl = 17
    Shr = 18
    Copy = 19


@dataclass(frozen=True)
class PtxtType:
    """reference src/lib.rs:20-29. kind in None/Bool/U8/U16/U32/U64/U128."""
    kind: str
    value: int = 0

    @staticmethod
    def Bool(b):
        return PtxtType("Bool", int(bool(b)))

    @staticmethod
    def U8(v): return PtxtType("U8", int(v))
    @staticmethod
    def U16(v): return PtxtType("U16", int(v))
    @staticmethod
    def U32(v): return PtxtType("U32", int(v))
    @staticmethod
    def U64(v): return PtxtType("U64", int(v))
    @staticmethod
    def U128(v): return PtxtType("U128", int(v))

    @staticmethod
    def None_():
        return PtxtType("None", 0)

    def __bool__(self):
        return bool(self.value)


@dataclass
class Gate:
    gate_name: str
    gate_type: GateType
    input_wires: List[str]
    lut_const: Optional[List[int]]
    output_wire: str
    level: int = 0

    def get_input_wires(self): return self.input_wires
    def get_output_wire(self): return self.output_wire
    def get_gate_type(self): return self.gate_type
    def get_gate_name(self): return self.gate_name
    def get_lut_const(self): return self.lut_const

    def __hash__(self):  # identity by name, reference src/gates.rs:62-87
        return hash(self.gate_name)

    def __eq__(self, o):
        return isinstance(o, Gate) and self.gate_name == o.gate_name


def parse_gate_lines(text, with_level=False):
    gates = []
    for line in text.splitlines():
        if not line:
            continue
        f = line.split("\t")
        name, typ, out, lut, ins = f[0], f[1], f[2], f[3], f[4]
        g = Gate(name, GateType[typ], ins.split(",") if ins else [], None if lut == "-" else [int(c) for c in lut], out)
        if with_level:
            g.level = int(f[5])
        gates.append(g)
    return gates


def map_to_text(m):
    return "".join(f"{k}\t{v.kind}\t{v.value}\n" for k, v in m.items())


def text_to_map(text):
    m = {}
    for line in text.splitlines():
        if line:
            k, kind, val = line.split("\t")
            m[k] = PtxtType(kind, int(val))
    return m
