#!/usr/bin/env python3
"""Emit integration/rust/ffi.rs from include/neumann_gpu.h: the `mod ffi` a maintainer drops into
vector_engine/src/ (INTEGRATION.md §2).  One Rust item per C declaration: constants, opaque handles, #[repr(C)] structs,
the metric enum and the extern "C" block.  tests/test_capi_cpu.py checks that the committed file is what this script
emits and that it names every symbol the library exports.  (No Rust toolchain in this image: the file is not compiled
here; the C ABI itself is exercised from C++ and ctypes.)

    python tools/gen_rust_ffi.py            # rewrite integration/rust/ffi.rs
    python tools/gen_rust_ffi.py --check    # exit 1 if the committed file is stale
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "neumann_gpu.h")
OUT = os.path.join(ROOT, "integration", "rust", "ffi.rs")

SCALARS = {"void": "c_void", "char": "c_char", "int": "i32", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64",
           "uint64_t": "u64", "uint8_t": "u8", "uint16_t": "u16", "float": "f32", "double": "f64", "size_t": "usize",
           "nmn_status": "nmn_status", "nmn_metric": "nmn_metric"}
KEYWORDS = {"type", "in", "ref", "fn", "mod", "match", "loop", "move", "box", "where", "self", "use"}


def rust_type(ctype, known):
    """`const float*` -> `*const f32`, `nmn_index**` -> `*mut *mut nmn_index`, `const uint64_t* const*` -> `*const *const u64`."""
    t = ctype.strip()
    ptrs = []  # innermost first: True = pointee is const
    while t.endswith("*") or t.endswith("const"):
        if t.endswith("const"):  # `* const`: constness of the pointer itself, irrelevant in an argument
            t = t[:-5].strip()
            continue
        t = t[:-1].strip()
        ptrs.append(None)
    const_base = False
    toks = [x for x in t.split() if x not in ("struct", "enum")]
    if "const" in toks:
        const_base = True
        toks.remove("const")
    base = " ".join(toks)
    if base == "unsigned":
        base = "uint32_t"
    r = SCALARS.get(base) or (base if base in known else None)
    if r is None:
        raise SystemExit(f"gen_rust_ffi: unknown C type {ctype!r}")
    # constness: `const T*` makes the INNERMOST pointer const; `T* const*` (seen as '* const *') makes the outer one const
    out = r
    consts = [const_base] + [False] * (len(ptrs) - 1)
    if "* const*" in ctype.replace(" ", "").replace("*const*", "* const*") or "*const*" in ctype.replace(" ", ""):
        consts = [const_base] + [True] * (len(ptrs) - 1)
    for c in consts[: len(ptrs)]:
        out = ("*const " if c else "*mut ") + out
    if not ptrs and r == "c_void":
        return "()"
    return out


def ident(name):
    return "r#" + name if name in KEYWORDS else name


def split_args(s):
    s = s.strip()
    if s in ("", "void"):
        return []
    return [a.strip() for a in s.split(",")]


def parse(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    consts, opaque, structs, enums, funcs = [], [], [], [], []
    for m in re.finditer(r"^#define\s+(NMN_[A-Z0-9_]+)\s+(.+?)\s*$", text, flags=re.M):
        consts.append((m.group(1), m.group(2).strip()))
    body = re.sub(r"^#.*$", "", text, flags=re.M)
    body = body.replace('extern "C" {', "").replace("extern \"C\"", "")
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", body):
        opaque.append(m.group(2))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", body, flags=re.S):
        fields = []
        for f in m.group(2).split(";"):
            f = " ".join(f.split())
            if not f:
                continue
            fm = re.match(r"(.+?)\s*(\w+)\s*(\[\s*(\w+)\s*\])?$", f)
            ctype, name, arr = fm.group(1), fm.group(2), fm.group(4)
            while ctype.endswith("*") is False and name.startswith("*"):
                name = name[1:]
            fields.append((ctype, name, arr))
        structs.append((m.group(3), fields))
    for m in re.finditer(r"typedef\s+enum\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", body, flags=re.S):
        items = []
        for it in m.group(2).split(","):
            it = " ".join(it.split())
            if it:
                n, _, v = it.partition("=")
                items.append((n.strip(), v.strip()))
        enums.append((m.group(3), items))
    stripped = re.sub(r"typedef\s+(struct|enum)\s+\w+\s*\{.*?\}\s*\w+\s*;", "", body, flags=re.S)
    stripped = re.sub(r"typedef[^;]*;", "", stripped)
    for m in re.finditer(r"([\w\s\*]+?)\b(nmn_\w+)\s*\(([^;{}]*?)\)\s*;", stripped, flags=re.S):
        ret = " ".join(m.group(1).split())
        args = []
        for a in split_args(" ".join(m.group(3).split())):
            am = re.match(r"(.+?)(\w+)$", a)
            args.append((am.group(1).strip(), am.group(2)))
        funcs.append((ret, m.group(2), args))
    return consts, opaque, structs, enums, funcs


def const_line(name, val, consts_by_name):
    v = val.strip()
    if v.startswith("(") and v.endswith(")"):
        v = v[1:-1].strip()
    m = re.fullmatch(r"(-?\d+)(u|U|ull|ULL)?", v)
    if m:
        if m.group(2) and m.group(2).lower() == "ull":
            return f"pub const {name}: u64 = {m.group(1)};"
        if m.group(2):
            return f"pub const {name}: u32 = {m.group(1)};"
        return f"pub const {name}: i32 = {m.group(1)};"
    if re.fullmatch(r'".*"', v):
        return f"pub const {name}: &str = {v};"
    return None  # not a plain literal (expression macros are not part of the ABI)


def emit():
    consts, opaque, structs, enums, funcs = parse(open(HEADER).read())
    known = set(opaque) | {s for s, _ in structs} | {e for e, _ in enums}
    o = []
    o.append("// GENERATED by tools/gen_rust_ffi.py from include/neumann_gpu.h — do not edit by hand.")
    o.append("// `mod ffi` for the reference's crate vector_engine (INTEGRATION.md §2): link with")
    o.append("//   println!(\"cargo:rustc-link-lib=dylib=neumann_gpu\");  in vector_engine/build.rs")
    o.append("#![allow(unsafe_code, non_camel_case_types, dead_code)]")
    o.append("use std::os::raw::{c_char, c_void};")
    o.append("")
    o.append("pub type nmn_status = i32;")
    o.append("")
    by_name = dict(consts)
    for n, v in consts:
        if n == "NEUMANN_GPU_H":
            continue
        line = const_line(n, v, by_name)
        if line:
            o.append(line)
    o.append("")
    for name, items in enums:
        o.append("#[repr(i32)]")
        o.append("#[derive(Clone, Copy, Debug, PartialEq, Eq)]")
        o.append(f"pub enum {name} {{")
        nxt = 0
        for n, v in items:
            val = int(v, 0) if v else nxt
            nxt = val + 1
            o.append(f"    {n} = {val},")
        o.append("}")
        o.append("")
    struct_names = {s for s, _ in structs}
    for name in opaque:
        if name in struct_names:
            continue
        o.append("#[repr(C)]")
        o.append(f"pub struct {name} {{ _private: [u8; 0] }}")
    o.append("")
    for name, fields in structs:
        o.append("#[repr(C)]")
        o.append("#[derive(Clone, Copy)]")
        o.append(f"pub struct {name} {{")
        for ctype, fname, arr in fields:
            rt = rust_type(ctype, known)
            if arr:
                n = by_name.get(arr, arr).rstrip("uU")
                rt = f"[{rt}; {n}]"
            o.append(f"    pub {ident(fname)}: {rt},")
        o.append("}")
        o.append("")
    o.append('#[link(name = "neumann_gpu")]')
    o.append('extern "C" {')
    for ret, name, args in funcs:
        rargs = ", ".join(f"{ident(a)}: {rust_type(t, known)}" for t, a in args)
        rr = rust_type(ret, known)
        tail = "" if rr == "()" else f" -> {rr}"
        line = f"    pub fn {name}({rargs}){tail};"
        if len(line) > 118:
            o.append(f"    pub fn {name}(")
            cur = "        "
            for i, (t, a) in enumerate(args):
                piece = f"{ident(a)}: {rust_type(t, known)}" + ("," if i + 1 < len(args) else "")
                if len(cur) + len(piece) + 1 > 118:
                    o.append(cur.rstrip())
                    cur = "        "
                cur += piece + " "
            o.append(cur.rstrip())
            o.append(f"    ){tail};")
        else:
            o.append(line)
    o.append("}")
    return "\n".join(o) + "\n", [f[1] for f in funcs]


def main():
    text, _ = emit()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("integration/rust/ffi.rs is stale: run python tools/gen_rust_ffi.py", file=sys.stderr)
            return 1
        return 0
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
