"""Generators for gates-mode netlists in HELM's dialect.

Every generator returns the netlist text.  Gate lines follow the forms
verilog_parser.rs accepts: `<kw> NAME(in0, in1, out);`, `not NAME(in, out);`,
`mux NAME(in0, in1, sel, out);` (Appendix A of SURVEY.md)."""
from . import sbox_bp


class NetlistBuilder:
    def __init__(self, name):
        self.name = name
        self.inputs = []   # (name, width or None)
        self.outputs = []
        self.lines = []
        self._n = 0
        self._g = 0

    def input(self, name, width=None):
        self.inputs.append((name, width))
        return [f"{name}[{i}]" for i in range(width)] if width else name

    def output(self, name, width=None):
        self.outputs.append((name, width))
        return [f"{name}[{i}]" for i in range(width)] if width else name

    def wire(self):
        self._n += 1
        return f"n{self._n}"

    def gate(self, kw, ins, out=None):
        out = out or self.wire()
        self._g += 1
        self.lines.append(f"  {kw} g{self._g}({', '.join(list(ins) + [out])});")
        return out

    def xor(self, a, b, out=None): return self.gate("xor", [a, b], out)
    def xnor(self, a, b, out=None): return self.gate("xnor", [a, b], out)
    def and_(self, a, b, out=None): return self.gate("and", [a, b], out)
    def or_(self, a, b, out=None): return self.gate("or", [a, b], out)
    def nand(self, a, b, out=None): return self.gate("nand", [a, b], out)
    def nor(self, a, b, out=None): return self.gate("nor", [a, b], out)
    def not_(self, a, out=None): return self.gate("not", [a], out)
    def buf(self, a, out=None): return self.gate("buf", [a], out)
    def mux(self, in0, in1, sel, out=None): return self.gate("mux", [in0, in1, sel], out)

    @property
    def n_gates(self):
        return self._g

    def text(self):
        ports = [n for n, _ in self.inputs] + [n for n, _ in self.outputs]
        s = [f"module {self.name}({', '.join(ports)});"]
        for n, w in self.inputs:
            s.append(f"  input [{w - 1}:0] {n};" if w else f"  input {n};")
        for n, w in self.outputs:
            s.append(f"  output [{w - 1}:0] {n};" if w else f"  output {n};")
        s += self.lines
        s.append("endmodule")
        return "\n".join(s) + "\n"


def two_bit_adder():
    """Authored to the constraints the reference's tests pin (tests/circuit_test.rs:25,37-44;
    tests/verilog_parser_test.rs:9-11): inputs a[0] a[1] b[0] b[1] cin, outputs sum[0]
    sum[1] cout, exactly 10 gates / 10 gate-output wires / 15 wires, internal wires i0 and
    i1 that are 0 when every input is 1."""
    return """module adder(a, b, cin, sum, cout);
  input [1:0] a;
  input [1:0] b;
  input cin;
  output [1:0] sum;
  output cout;
  wire i0, i1, i2, i3, i4, i5, i6;
  xor g0(a[0], b[0], i0);
  xor g1(i0, cin, sum[0]);
  and g2(a[0], b[0], i2);
  and g3(i0, cin, i3);
  or g4(i2, i3, i4);
  xor g5(a[1], b[1], i1);
  xor g6(i1, i4, sum[1]);
  and g7(a[1], b[1], i5);
  and g8(i1, i4, i6);
  or g9(i5, i6, cout);
endmodule
"""


def _full_adder(nb, a, b, c, s_out=None, c_out=None):
    x = nb.xor(a, b)
    s = nb.xor(x, c, s_out)
    t = nb.and_(a, b)
    u = nb.and_(x, c)
    co = nb.or_(t, u, c_out)
    return s, co


def ripple_adder(nbits):
    nb = NetlistBuilder(f"adder{nbits}")
    a, b = nb.input("a", nbits), nb.input("b", nbits)
    c = nb.input("cin")
    s = nb.output("sum", nbits)
    co = nb.output("cout")
    for i in range(nbits):
        _, c = _full_adder(nb, a[i], b[i], c, s[i], co if i == nbits - 1 else None)
    return nb.text()


def nand_bank(count):
    """`count` independent NAND gates (the µ-bench of SURVEY.md §8d as a netlist)."""
    nb = NetlistBuilder("nandbank")
    a, b = nb.input("a", count), nb.input("b", count)
    y = nb.output("y", count)
    for i in range(count):
        nb.nand(a[i], b[i], y[i])
    return nb.text()


def alu_c880_class():
    """STAND-IN for ISCAS'85 c880 (an 8-bit ALU: 60 inputs, 26 outputs, 383 gates), which
    cannot be obtained offline.  Same class of circuit and the same I/O and gate counts:
    three 8-bit adders, a comparator, a logic unit, parity and and/or/not selection logic
    over 60 inputs and 26 outputs, built from two-input gates and inverters."""
    nb = NetlistBuilder("alu_c880_class")
    a, b, c, d, e, f = (nb.input(x, 8) for x in "abcdef")
    sel = nb.input("sel", 4)
    cin0, cin1 = nb.input("cin0"), nb.input("cin1")
    m = nb.input("m", 6)
    y, z, w = nb.output("y", 8), nb.output("z", 8), nb.output("w", 8)
    cout0, cout1 = nb.output("cout0"), nb.output("cout1")
    nsel = [nb.not_(s) for s in sel]
    # adders: s0 = a + b + cin0, s1 = c + d + cin1, s2 = e + f + cin0
    s0, s1, s2 = [], [], []
    c0, c1, c2 = cin0, cin1, cin0
    for i in range(8):
        s, c0 = _full_adder(nb, a[i], b[i], c0, None, cout0 if i == 7 else None)
        s0.append(s)
    for i in range(8):
        s, c1 = _full_adder(nb, c[i], d[i], c1, None, cout1 if i == 7 else None)
        s1.append(s)
    for i in range(8):
        s, c2 = _full_adder(nb, e[i], f[i], c2)
        s2.append(s)
    # comparator a == c, a > c (MSB first)
    eqb = [nb.xnor(a[i], c[i]) for i in range(8)]
    eq_prefix = [None] * 8  # eq of bits above i
    gt = None
    run = None
    for i in range(7, -1, -1):
        g = nb.and_(a[i], nb.not_(c[i]))
        if run is not None:
            g = nb.and_(g, run)
        gt = g if gt is None else nb.or_(gt, g)
        run = eqb[i] if run is None else nb.and_(run, eqb[i])
    eq = run
    # parity of s2 and "any d"
    par = s2[0]
    for i in range(1, 8):
        par = nb.xor(par, s2[i])
    anyd = d[0]
    for i in range(1, 8):
        anyd = nb.or_(anyd, d[i])
    k0 = nb.and_(eq, c2)
    k1 = nb.and_(gt, anyd)
    # mode decode: 6 gates
    m01 = nb.and_(m[0], m[1])
    m23 = nb.or_(m[2], m[3])
    m45 = nb.xor(m[4], m[5])
    mode = nb.or_(nb.and_(m01, m23), nb.and_(m45, sel[0]))
    k0 = nb.xor(k0, mode)
    for i in range(8):
        # logic unit
        l_and = nb.and_(a[i], e[i])
        l_or = nb.or_(b[i], f[i])
        l_xor = nb.xor(c[i], e[i])
        l_nor = nb.nor(d[i], f[i])
        # y = sel0 ? s0 : l_xor, conditionally inverted by m0 / m1
        p = nb.and_(sel[0], s0[i])
        q = nb.and_(nsel[0], l_xor)
        r = nb.or_(p, q)
        nb.xnor(r, m[0], y[i]) if i % 2 else nb.xor(r, m[1], y[i])
        # z = sel1 ? s1 : (sel2 ? l_and : l_or), gated by m2 / m3
        p = nb.and_(sel[2], l_and)
        q = nb.and_(nsel[2], l_or)
        r = nb.or_(p, q)
        p = nb.and_(sel[1], s1[i])
        q = nb.and_(nsel[1], r)
        t = nb.or_(p, q)
        t = nb.xor(t, nb.and_(eqb[i], k1))
        nb.nand(t, m[2 + (i % 2)], z[i])
        # w = sel3 ? (s0 xor s1 xor s2) : ((l_nor nand m) xor flag)
        p = nb.xor(nb.xor(s0[i], s1[i]), s2[i])
        q = nb.nand(l_nor, m[4 + (i % 2)])
        q = nb.xor(q, (k0, k1, par, eq)[i % 4])
        p = nb.and_(sel[3], p)
        q = nb.and_(nsel[3], q)
        nb.or_(p, q, w[i])
    return nb.text()


# ---------------------------------------------------------------------------------------
# AES-128 (FIPS-197), one block, key schedule included.
# Bit convention: FIPS byte i (0..15), bit b (0 = LSB) is wire  bus[8*(15-i) + b], so that
# a 3-column CSV row `pt, 00112233445566778899aabbccddeeff, 128` (hex, LSB-first expansion,
# reference src/verilog_parser.rs:287-305) reads as the FIPS byte string.
# ---------------------------------------------------------------------------------------
def _sbox(nb, byte):
    """byte: list of 8 wires, index 0 = LSB. Returns 8 wires (LSB first)."""
    v = {f"U{i}": byte[7 - i] for i in range(8)}
    for dst, op, x, y in sbox_bp.parse():
        v[dst] = nb.gate(op, [v[x], v[y]])
    return [v[f"S{7 - i}"] for i in range(8)]


def _xor_bytes(nb, x, y):
    return [nb.xor(p, q) for p, q in zip(x, y)]


def _xtime(nb, a):
    """multiply by x in GF(2^8) mod x^8+x^4+x^3+x+1; a[0] = LSB"""
    return [a[7], nb.xor(a[0], a[7]), a[1], nb.xor(a[2], a[7]), nb.xor(a[3], a[7]), a[4], a[5], a[6]]


def _mix_column(nb, col):
    u = [_xor_bytes(nb, col[i], col[(i + 1) % 4]) for i in range(4)]
    t = _xor_bytes(nb, u[0], u[2])
    out = []
    for i in range(4):
        at = _xor_bytes(nb, col[i], t)
        out.append(_xor_bytes(nb, at, _xtime(nb, u[i])))
    return out


def aes128():
    nb = NetlistBuilder("aes128")
    key, pt = nb.input("key", 128), nb.input("pt", 128)
    ct = nb.output("ct", 128)

    def byte(bus, i):
        return [bus[8 * (15 - i) + b] for b in range(8)]

    # key schedule: words of 4 bytes
    w = [[byte(key, 4 * j + r) for r in range(4)] for j in range(4)]
    rcon = [0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1B, 0x36]
    for i in range(4, 44):
        temp = w[i - 1]
        if i % 4 == 0:
            rot = temp[1:] + temp[:1]
            sub = [_sbox(nb, b) for b in rot]
            rc = rcon[i // 4 - 1]
            # xor with the round constant: a constant-1 bit is a NOT (no bootstrap)
            sub[0] = [nb.not_(sub[0][b]) if (rc >> b) & 1 else sub[0][b] for b in range(8)]
            temp = sub
        w.append([_xor_bytes(nb, w[i - 4][r], temp[r]) for r in range(4)])

    def round_key(rnd):
        return [w[4 * rnd + j][r] for j in range(4) for r in range(4)]  # byte index 4*j + r

    state = [_xor_bytes(nb, byte(pt, i), k) for i, k in enumerate(round_key(0))]
    for rnd in range(1, 11):
        sb = [_sbox(nb, s) for s in state]
        # ShiftRows: byte (r, c) <- (r, c + r)
        sr = [sb[4 * ((c + r) % 4) + r] for c in range(4) for r in range(4)]
        if rnd < 10:
            mc = []
            for c in range(4):
                mc += _mix_column(nb, sr[4 * c:4 * c + 4])
        else:
            mc = sr
        rk = round_key(rnd)
        if rnd < 10:
            state = [_xor_bytes(nb, mc[i], rk[i]) for i in range(16)]
        else:
            for i in range(16):
                for b in range(8):
                    nb.xor(mc[i][b], rk[i][b], ct[8 * (15 - i) + b])
    return nb.text()


def aes128_reference_encrypt(key: bytes, pt: bytes) -> bytes:
    """Independent software AES-128 (table-free, from the FIPS-197 definitions) used by
    the tests to check the netlist; not derived from the netlist generator."""
    def gmul(a, b):
        r = 0
        while b:
            if b & 1:
                r ^= a
            a = ((a << 1) ^ (0x11B if a & 0x80 else 0)) & 0x1FF
            b >>= 1
        return r & 0xFF

    def inv(a):
        r = 1
        for _ in range(254):
            r = gmul(r, a)
        return r if a else 0

    def sbox(a):
        x, r = inv(a), 0
        for i in range(8):
            bit = ((x >> i) ^ (x >> ((i + 4) % 8)) ^ (x >> ((i + 5) % 8)) ^ (x >> ((i + 6) % 8)) ^
                   (x >> ((i + 7) % 8)) ^ (0x63 >> i)) & 1
            r |= bit << i
        return r

    sb = [sbox(a) for a in range(256)]
    w = [list(key[4 * i:4 * i + 4]) for i in range(4)]
    rc = 1
    for i in range(4, 44):
        t = list(w[i - 1])
        if i % 4 == 0:
            t = [sb[x] for x in t[1:] + t[:1]]
            t[0] ^= rc
            rc = gmul(rc, 2)
        w.append([a ^ b for a, b in zip(w[i - 4], t)])
    st = [pt[i] ^ w[i // 4][i % 4] for i in range(16)]
    for rnd in range(1, 11):
        st = [sb[x] for x in st]
        st = [st[4 * ((c + r) % 4) + r] for c in range(4) for r in range(4)]
        if rnd < 10:
            ns = []
            for c in range(4):
                a = st[4 * c:4 * c + 4]
                ns += [gmul(a[i], 2) ^ gmul(a[(i + 1) % 4], 3) ^ a[(i + 2) % 4] ^ a[(i + 3) % 4] for i in range(4)]
            st = ns
        st = [st[i] ^ w[4 * rnd + i // 4][i % 4] for i in range(16)]
    return bytes(st)
