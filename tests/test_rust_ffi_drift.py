"""The Rust shim under rust/ has never met a compiler in this image (no rustc), so nothing but this test keeps its
`extern "C"` view of the library honest: every declaration of rust/helm-hip-sys/src/lib.rs is compared with the
prototype of the same name in include/*.h - name, arity, the width and kind of every argument and of the return
value, pointer vs integer, pointee width and constness - every `#[repr(C)]` struct with the header's struct of the
same name field by field, every constant with the header's enumerator, and every `sys::` item the shim crate uses
must be declared.  (Reference boundary: `impl EvalCircuit` of src/circuit.rs:35-58 over the `tfhe` crate of
Cargo.toml:18; the shim forwards that trait to this C ABI.)"""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYS = os.path.join(ROOT, "rust", "helm-hip-sys", "src", "lib.rs")
SHIM = sorted(glob.glob(os.path.join(ROOT, "rust", "helm-hip", "src", "*.rs")))
HEADERS = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))


def split_top(text, sep=","):
    """Split at `sep` outside (), [], <>."""
    parts, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([<":
            depth += 1
        elif ch in ")]" or (ch == ">" and not cur.endswith("-")):  # `->` is no bracket
            depth -= 1
        if ch == sep and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [p.strip() for p in parts]


# ---------------------------------------------------------------------------------------------- C side
C_SCALARS = {"int": ("int", 32), "int32_t": ("int", 32), "int64_t": ("int", 64), "uint64_t": ("uint", 64),
             "uint32_t": ("uint", 32), "uint8_t": ("uint", 8), "int8_t": ("int", 8), "size_t": ("usize", 64),
             "double": ("float", 64), "char": ("char", 8), "void": ("void", 0)}


def parse_headers():
    protos, structs, consts, fn_typedefs, enums = {}, {}, {}, set(), set()
    for path in HEADERS:
        text = open(path).read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", " ", text)
        for m in re.finditer(r"#define\s+(HELM_\w+)\s+(-?\d+)", text):
            consts[m.group(1)] = int(m.group(2))
        text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
        text = text.replace('extern "C" {', " ")
        for m in re.finditer(r"typedef\s+\w[\w\s\*]*?\(\s*\*\s*(\w+)\s*\)\s*\([^;]*?\)\s*;", text, flags=re.S):
            fn_typedefs.add(m.group(1))
        for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
            fields = []
            for decl in m.group(1).split(";"):
                decl = decl.strip()
                if not decl:
                    continue
                if "*" in decl:  # pointer fields (helm_si_audit_record): `const T *a, *b`
                    for n in decl.split("*")[1:]:
                        fields.append((n.strip(" ,"), ("ptr", 64)))
                    continue
                ty, names = decl.split(None, 1)
                for n in names.split(","):
                    fields.append((n.strip(), C_SCALARS[ty]))
            structs[m.group(2)] = fields
        for m in re.finditer(r"(?:typedef\s+)?enum\s*\{(.*?)\}\s*(\w*)\s*;", text, flags=re.S):
            if m.group(2):
                enums.add(m.group(2))
            nxt = 0
            for item in m.group(1).split(","):
                item = item.strip()
                if not item:
                    continue
                if "=" in item:
                    name, val = [x.strip() for x in item.split("=")]
                    nxt = int(val, 0)
                else:
                    name = item
                consts[name] = nxt
                nxt += 1
        body = re.sub(r"(typedef\s+)?(struct|enum)\s*\{.*?\}\s*\w*\s*;", " ", text, flags=re.S)
        for stmt in body.split(";"):
            stmt = " ".join(stmt.split())
            m = re.match(r"^(?!typedef)(.*?)\b(helm_\w+)\s*\((.*)\)$", stmt)
            if m and "(" not in m.group(1):
                protos[m.group(2)] = (m.group(1).strip(), m.group(3).strip(), os.path.basename(path))
    return protos, structs, consts, fn_typedefs, enums


def c_type(t, fn_typedefs, enums):
    """-> ('ptr', constness, pointee) | (kind, bits)"""
    t = re.sub(r"\b(\w+)\s*\[[^\]]*\]$", r"* \1", t.strip())  # `uint8_t id[128]` decays to a pointer
    t = re.sub(r"\bstruct\s+", "", t)
    stars = t.count("*")
    is_const = bool(re.search(r"\bconst\b", t))
    words = [w for w in re.sub(r"[\*]", " ", t).split() if w != "const"]
    # drop the parameter name (the last word, when more than a bare type is left)
    base = words[0] if len(words) >= 1 else "void"
    if base in fn_typedefs:
        return ("fnptr",)
    if stars:
        pointee = C_SCALARS.get(base, ("opaque:" + base, 0)) if stars == 1 else ("ptr",)
        return ("ptr", is_const if stars == 1 else False, pointee)
    if base in enums:
        return ("int", 32)
    return C_SCALARS[base]


# ------------------------------------------------------------------------------------------- Rust side
R_SCALARS = {"c_int": ("int", 32), "i32": ("int", 32), "i64": ("int", 64), "u64": ("uint", 64), "u32": ("uint", 32),
             "u8": ("uint", 8), "i8": ("int", 8), "usize": ("usize", 64), "f64": ("float", 64), "c_char": ("char", 8),
             "c_void": ("void", 0)}


def r_type(t):
    t = t.strip()
    if t.startswith("extern \"C\" fn") or t.startswith("Option<extern"):
        return ("fnptr",)
    m = re.match(r"^\*(const|mut)\s+(.*)$", t)
    if m:
        inner = m.group(2).strip()
        if inner.startswith("*"):
            return ("ptr", False, ("ptr",))
        return ("ptr", m.group(1) == "const", R_SCALARS.get(inner, ("opaque:" + inner, 0)))
    return R_SCALARS[t]


def parse_rust():
    text = open(SYS).read()
    text = re.sub(r"//[^\n]*", " ", text)
    fns = {}
    for block in re.findall(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S):
        for stmt in split_top(block, ";"):
            stmt = " ".join(stmt.split())
            m = re.match(r"^pub fn (\w+)\s*\((.*)\)\s*(?:->\s*(.*))?$", stmt)
            if not m:
                continue
            args = [a.split(":", 1) for a in split_top(m.group(2))] if m.group(2).strip() else []
            fns[m.group(1)] = ([(a[0].strip(), a[1].strip()) for a in args], (m.group(3) or "").strip())
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for f in split_top(m.group(2)):
            f = f.strip()
            if not f or f.startswith("_"):
                continue
            name, ty = f.replace("pub ", "").split(":")
            fields.append((name.strip(), R_SCALARS[ty.strip()]))
        structs[m.group(1)] = fields
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (HELM_\w+)\s*:\s*\w+\s*=\s*(-?\d+)\s*;", text)}
    return fns, structs, consts


def same(a, b):
    """C type vs Rust type.  Pointers: kind, constness and pointee width; opaque handles by name; `void *` matches
    `*mut c_void`; a C `void *` user pointer never matches a typed one."""
    if a[0] != b[0]:
        return False
    if a[0] == "ptr":
        return a[1] == b[1] and a[2] == b[2]
    return a == b


def compare(fns, protos, fn_typedefs, enums):
    problems = []
    for name, (args, ret) in sorted(fns.items()):
        if name not in protos:
            problems.append(f"{name}: declared in Rust, in no header under include/")
            continue
        c_ret, c_args, header = protos[name]
        c_list = [] if c_args in ("", "void") else split_top(c_args)
        if len(c_list) != len(args):
            problems.append(f"{name} ({header}): {len(c_list)} parameters in C, {len(args)} in Rust")
            continue
        for i, (c_arg, (r_name, r_ty)) in enumerate(zip(c_list, args)):
            ct, rt = c_type(c_arg, fn_typedefs, enums), r_type(r_ty)
            if not same(ct, rt):
                problems.append(f"{name} ({header}) parameter {i} `{c_arg}` vs `{r_name}: {r_ty}`: {ct} != {rt}")
        cr = c_type(c_ret + " x" if "*" not in c_ret else c_ret, fn_typedefs, enums) if c_ret != "void" else ("void", 0)
        rr = r_type(ret) if ret else ("void", 0)
        if not same(cr, rr):
            problems.append(f"{name} ({header}) return `{c_ret}` vs `{ret}`: {cr} != {rr}")
    return problems


def test_every_rust_declaration_matches_its_header_prototype():
    protos, _, _, fn_typedefs, enums = parse_headers()
    fns, _, _ = parse_rust()
    assert len(fns) == open(SYS).read().count("pub fn helm_") >= 70, "the extern block was not parsed completely"
    problems = compare(fns, protos, fn_typedefs, enums)
    assert not problems, "\n".join(problems)


def test_the_comparison_catches_drift():
    """The checker itself: a narrowed integer, a dropped parameter, lost constness, a pointer turned integer, a changed
    return width and an unknown symbol are all reported."""
    protos, _, _, fn_typedefs, enums = parse_headers()
    fns, _, _ = parse_rust()
    bad = dict(fns)
    a, r = fns["helm_hip_program_run"]
    bad["helm_hip_program_run"] = (a[:3] + [("level_begin", "i32")] + a[4:], r)               # int64_t narrowed
    a, r = fns["helm_hip_wires_upload"]
    bad["helm_hip_wires_upload"] = (a[:-1], r)                                                # count dropped
    a, r = fns["helm_si_load_bootstrap_key"]
    bad["helm_si_load_bootstrap_key"] = ([a[0], ("bsk_std", "*mut u64"), a[2]], r)            # const lost
    a, r = fns["helm_comm_create"]
    bad["helm_comm_create"] = ([a[0], ("id", "u64")] + a[2:], r)                               # pointer -> integer
    a, r = fns["helm_hip_launch_quantum"]
    bad["helm_hip_launch_quantum"] = (a, "c_int")                                              # int64_t return narrowed
    a, r = fns["helm_keys_bsk32_from_tfhe"]
    bad["helm_keys_bsk32_from_tfhe"] = ([a[0], ("tfhe", "*const u64")] + a[2:], r)             # pointee width
    bad["helm_hip_no_such_function"] = ([], "c_int")
    problems = compare(bad, protos, fn_typedefs, enums)
    hit = {p.split(" ")[0].rstrip(":") for p in problems}
    assert hit == {"helm_hip_program_run", "helm_hip_wires_upload", "helm_si_load_bootstrap_key", "helm_comm_create",
                   "helm_hip_launch_quantum", "helm_keys_bsk32_from_tfhe", "helm_hip_no_such_function"}, problems


def test_repr_c_structs_match_field_by_field():
    _, c_structs, _, _, _ = parse_headers()
    _, r_structs, _ = parse_rust()
    checked = 0
    for name, fields in r_structs.items():
        if not fields:  # opaque handles
            continue
        assert name in c_structs, f"{name}: #[repr(C)] struct without a header struct of that name"
        assert fields == c_structs[name], f"{name}: Rust {fields} != C {c_structs[name]}"
        checked += 1
    assert checked >= 4  # helm_hip_params, helm_si_params, helm_wop_params, helm_radix_op


def test_constants_match_the_headers():
    _, _, c_consts, _, _ = parse_headers()
    _, _, r_consts = parse_rust()
    assert len(r_consts) >= 30
    for name, value in r_consts.items():
        assert name in c_consts, f"{name}: Rust constant without a header definition"
        assert c_consts[name] == value, f"{name}: {value} in Rust, {c_consts[name]} in the header"


def test_every_sys_item_the_shim_uses_is_declared():
    fns, structs, consts = parse_rust()
    text = open(SYS).read()
    opaque = set(re.findall(r"pub struct (\w+)", text))
    declared = set(fns) | set(structs) | set(consts) | opaque
    used = set()
    for path in SHIM:
        src = re.sub(r"//[^\n]*", " ", open(path).read())
        used |= set(re.findall(r"\bsys::(\w+)", src))
    assert len(used) >= 60
    missing = sorted(used - declared)
    assert not missing, f"used by rust/helm-hip but not declared in helm-hip-sys: {missing}"
