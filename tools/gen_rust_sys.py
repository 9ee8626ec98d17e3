#!/usr/bin/env python3
"""Generates rust-shim/ligero-hip-sys/src/lib.rs -- the raw Rust binding of include/ligero_hip.h -- FROM THE HEADER, so the
`extern "C"` block a Rust host links against cannot drift from the C ABI: every function prototype, struct (field for field,
callbacks as `Option<unsafe extern "C" fn>`), enum constant and numeric #define of the header, nothing else.

    python tools/gen_rust_sys.py            # rewrites the file
    python tools/gen_rust_sys.py --check    # exit 1 if the committed file is not what the header generates

tests/test_abi.py::test_rust_extern_block_matches_header runs --check and, independently of this parser, compares names,
arity and pointer / integer widths of both files.  (No Rust toolchain exists in the build image: the crate is un-built text;
the generator keeps to constructs whose Rust spelling is mechanical.)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ligero_hip.h")
OUT = os.path.join(ROOT, "rust-shim", "ligero-hip-sys", "src", "lib.rs")

SCALARS = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int8_t": "i8", "int32_t": "i32", "int64_t": "i64",
           "int": "c_int", "unsigned": "c_uint", "float": "f32", "double": "f64", "char": "c_char", "size_t": "usize", "void": "c_void"}


def strip_comments(src: str) -> str:
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def rust_type(ctype: str, known: set) -> str:
    """`const uint64_t*` -> `*const u64`; `lg_ctx**` -> `*mut *mut lg_ctx`; `const char* const*` -> `*const *const c_char`"""
    t = ctype.strip()
    stars = []
    while True:                                   # peel pointer levels from the right: `* const` / `*`
        m = re.match(r"^(.*?)(\*)\s*(const)?\s*$", t)
        if not m:
            break
        t = m.group(1).strip()
        stars.append(m.group(3) is not None)      # constness of the POINTER itself (irrelevant to Rust's raw pointer type)
    const_base = bool(re.search(r"\bconst\b", t))
    base = re.sub(r"\b(const|volatile|struct|enum)\b", "", t).strip()
    if base in SCALARS:
        r = SCALARS[base]
    elif base in known:
        r = base
    else:
        raise ValueError(f"unknown C type {ctype!r}")
    # innermost pointer carries the base's constness; outer pointers point at pointers: const iff that inner pointer was `* const`
    levels = len(stars)
    for lvl in range(levels - 1, -1, -1):         # stars[levels-1] is the innermost `*` (closest to the base)
        pointee_const = const_base if lvl == levels - 1 else stars[lvl + 1]
        r = ("*const " if pointee_const else "*mut ") + r
    if levels == 0 and r == "c_void":
        return "()"
    return r


def split_args(arglist: str):
    out, depth, cur = [], 0, ""
    for ch in arglist:
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out]


def parse_param(p: str, known: set):
    """one parameter / field declarator -> (name, rust type).  Arrays as parameters decay to pointers."""
    m = re.match(r"^(.*?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", p, flags=re.S)      # function pointer
    if m:
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
        ps = [] if args.strip() in ("", "void") else [parse_param(a, known) for a in split_args(args)]
        sig = ", ".join(f"{n}: {t}" for n, t in ps)
        rt = rust_type(ret, known)
        return name, f"Option<unsafe extern \"C\" fn({sig})" + ("" if rt == "()" else f" -> {rt}") + ">"
    m = re.match(r"^(.*?)(\w+)\s*\[\s*(\w*)\s*\]$", p, flags=re.S)              # array parameter: pointer to the element
    if m:
        base, name = m.group(1).strip(), m.group(2)
        const = bool(re.search(r"\bconst\b", base))
        return name, ("*const " if const else "*mut ") + rust_type(re.sub(r"\bconst\b", "", base), known)
    m = re.match(r"^(.*?)(\w+)$", p, flags=re.S)
    return m.group(2), rust_type(m.group(1), known)


RUST_KEYWORDS = {"type", "ref", "in", "fn", "mod", "move", "box", "loop", "match", "self", "use", "where", "yield", "priv"}


def ident(name: str) -> str:
    return "r#" + name if name in RUST_KEYWORDS else name


def parse_header(src: str):
    src = strip_comments(src)
    defines = [(m.group(1), m.group(2)) for m in re.finditer(r"^#define\s+(LG_\w+)\s+((?:0x[0-9a-fA-F]+|\d+)u?)\s*$", src, flags=re.M)]
    # function-like macros of the one shape the header uses -- NAME(arg) (EXPR over + and identifiers) -- become const fns
    skipped_macros = [(m.group(1), m.group(2), m.group(3)) for m in re.finditer(r"^#define\s+(LG_\w+)\((\w+)\)\s+\((.+)\)\s*$", src, flags=re.M)]
    body = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    body = body.replace('extern "C" {', "")
    opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", body)
    known = set(opaque)
    enums, structs = [], []
    for m in re.finditer(r"(typedef\s+)?enum\s*(\w*)\s*\{(.*?)\}\s*(\w*)\s*;", body, flags=re.S):
        name = m.group(4) or m.group(2)
        items, nxt = [], 0
        for it in split_args(m.group(3)):
            if not it:
                continue
            im = re.match(r"^(\w+)\s*(?:=\s*(.+))?$", it, flags=re.S)
            val = im.group(2).strip() if im.group(2) else str(nxt)
            items.append((im.group(1), val))
            try:
                nxt = int(val, 0) + 1
            except ValueError:
                nxt = 0
        enums.append((name, items))
        if name:
            known.add(name)
    struct_re = re.compile(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", flags=re.S)
    for m in struct_re.finditer(body):
        known.add(m.group(3))
    for m in struct_re.finditer(body):
        fields = []
        for decl in [d.strip() for d in m.group(2).split(";") if d.strip()]:
            if "(*" in decl:
                fields.append(parse_param(decl, known))
                continue
            dm = re.match(r"^((?:const\s+)?\w+(?:\s*\*+)?)\s*(.*)$", decl, flags=re.S)
            base, rest = dm.group(1), dm.group(2)
            for d in split_args(rest):
                am = re.match(r"^(\*?)\s*(\w+)\s*(?:\[\s*(\w+)\s*\])?$", d)
                ty = rust_type(base + am.group(1), known)
                if am.group(3):
                    ty = f"[{ty}; {am.group(3)}]"
                fields.append((am.group(2), ty))
        structs.append((m.group(3), fields))
    rest = struct_re.sub("", body)
    rest = re.sub(r"(typedef\s+)?enum\s*\w*\s*\{.*?\}\s*\w*\s*;", "", rest, flags=re.S)
    rest = re.sub(r"typedef\s+struct\s+\w+\s+\w+\s*;", "", rest)
    funcs = []
    for m in re.finditer(r"([\w\s\*]+?)\b(lg_\w+)\s*\(([^;{}]*)\)\s*;", rest, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
        params = [] if args.strip() in ("", "void") else [parse_param(a, known) for a in split_args(args)]
        funcs.append((name, params, rust_type(ret, known)))
    return defines, skipped_macros, opaque, enums, structs, funcs


def generate() -> str:
    defines, skipped, opaque, enums, structs, funcs = parse_header(open(HEADER).read())
    o = []
    o.append("// GENERATED by tools/gen_rust_sys.py from include/ligero_hip.h -- DO NOT EDIT (tests/test_abi.py checks it is current).")
    o.append("//! Raw bindings of `libligero_hip.so`, the MI355X encode-and-commit path behind NP-Eng/ligero's `prove_inner`")
    o.append("//! (`src/ligero/mod.rs:521-551`, `935-955`, `998-1012`).  Every item below mirrors one declaration of")
    o.append("//! `include/ligero_hip.h`, whose comments are the documentation: which reference lines an entry point replaces,")
    o.append("//! argument meaning, ownership, error behaviour.  Field elements cross as `[u64; 4]` Montgomery limbs -- the in-memory")
    o.append("//! form of `ark_bn254::Fr` -- so `*const u64` / `*mut u64` point at `4 * count` limbs.")
    o.append("#![allow(non_camel_case_types, non_upper_case_globals, non_snake_case)]")
    o.append("use core::ffi::{c_char, c_int, c_void};")
    o.append("")
    for name, val in defines:
        v = val.rstrip("u")
        o.append(f"pub const {name}: u32 = {v};")
    macro_fns = skipped
    o.append("")
    for name in opaque:
        o.append("#[repr(C)]")
        o.append(f"pub struct {name} {{")
        o.append("    _private: [u8; 0],")
        o.append("}")
    o.append("")
    for name, items in enums:
        ty = name if name else "c_int"
        if name:
            o.append(f"pub type {name} = c_int;")
        for iname, val in items:
            o.append(f"pub const {iname}: {ty} = {val};")
        o.append("")
    for name, fields in structs:
        o.append("#[repr(C)]")
        o.append("#[derive(Clone, Copy)]")
        o.append(f"pub struct {name} {{")
        for fname, fty in fields:
            o.append(f"    pub {ident(fname)}: {fty},")
        o.append("}")
        o.append("")
    for mname, arg, body in macro_fns:
        expr = re.sub(r"\((\w+)\)", r"\1", body)
        if not re.fullmatch(r"[\w\s+]+", expr):
            raise ValueError(f"macro {mname}: body {body!r} is not a sum of identifiers")
        o.append(f"pub const fn {mname}({arg}: c_int) -> c_int {{")
        o.append(f"    {expr}")
        o.append("}")
        o.append("")
    o.append('#[link(name = "ligero_hip")]')
    o.append('extern "C" {')
    for name, params, ret in funcs:
        sig = ", ".join(f"{ident(n)}: {t}" for n, t in params)
        o.append(f"    pub fn {name}({sig})" + ("" if ret == "()" else f" -> {ret}") + ";")
    o.append("}")
    o.append("")
    return "\n".join(l for l in o if l is not None)


def main():
    text = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            sys.stderr.write(f"{OUT} is stale: run python tools/gen_rust_sys.py\n")
            sys.exit(1)
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(text)
    print(f"wrote {OUT}")


if __name__ == "__main__":
    main()
