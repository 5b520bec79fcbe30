#!/usr/bin/env python3
"""Generate the Rust FFI binding of include/raftgroups.h.

    python tools/gen_rust_bindings.py            # writes bindings/raftgroups.rs and the block in INTEGRATION.md
    python tools/gen_rust_bindings.py --check    # exit 1 if either is out of date (tests/test_abi.py runs this)

Everything is derived from the header: `#[repr(C)]` structs, the `extern "C"` block (one declaration per exported
function, same argument order), the numeric constants. No Rust toolchain exists in the build image, so the output is
source a maintainer compiles on their side; what IS checked here is that it names every export of libraftgroups.so with
the header's arity (tests/test_abi.py).
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "raftgroups.h")
OUT_RS = os.path.join(ROOT, "bindings", "raftgroups.rs")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED: tools/gen_rust_bindings.py -->", "<!-- END GENERATED -->"

SCALARS = {"uint64_t": "u64", "uint32_t": "u32", "uint16_t": "u16", "uint8_t": "u8", "int32_t": "i32", "int64_t": "i64",
           "int": "i32", "unsigned": "u32", "double": "f64", "float": "f32", "size_t": "usize", "char": "c_char", "void": "c_void"}
KEYWORDS = {"match", "type", "ref", "box", "move", "fn", "in", "loop", "self", "use", "mod", "as", "where"}


def strip_comments(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def camel(name):
    return "".join(p.capitalize() for p in name.split("_"))


def rust_type(ctype, typedefs):
    """ctype: e.g. 'const uint64_t *', 'rg_engine **', 'void *'."""
    # `base q0 * q1 * q2 ...`: q0 qualifies the pointee of the innermost pointer, q_k (k >= 1) the k-th pointer itself -- i.e.
    # the pointee of the (k+1)-th. `rg_engine *const *engines` is a pointer to CONST pointers to mutable engines:
    # *const *mut RgEngine (round 5's generator had the two levels swapped).
    parts = ctype.strip().split("*")
    stars = len(parts) - 1
    consts = [bool(re.search(r"\bconst\b", q)) for q in parts]
    base = re.sub(r"\bconst\b|\bstruct\b", " ", parts[0]).strip()
    if base in SCALARS:
        r = SCALARS[base]
    elif base in typedefs:
        r = typedefs[base]
    else:
        raise ValueError(f"unknown C type {ctype!r}")
    for k in range(stars):
        r = ("*const " if consts[k] else "*mut ") + r
    return r


def ident(name):
    return "r#" + name if name in KEYWORDS else name


def parse_decl(decl, typedefs, as_param):
    """'const uint64_t *peer_ids' / 'uint64_t match[RG_MAX_SLOTS]' -> (name, rust type)."""
    decl = decl.strip()
    m = re.match(r"^(.*?)(\w+)\s*(\[\s*(\w+)\s*\])?$", decl, flags=re.S)
    if not m:
        raise ValueError(decl)
    ctype, name, arr, n = m.group(1), m.group(2), m.group(3), m.group(4)
    if not ctype.strip():  # unnamed parameter such as 'void'
        ctype, name = name, ""
    rt = rust_type(ctype, typedefs)
    if arr:
        if as_param:  # arrays decay to pointers
            rt = ("*const " if "const" in ctype else "*mut ") + rt
        else:
            rt = f"[{rt}; {n + ' as usize' if not n.isdigit() else int(n)}]"
    return name, rt


def parse(src):
    src = strip_comments(src)
    consts, structs, funcs, enums = [], [], [], []
    typedefs = {"rg_engine": "RgEngine"}
    for m in re.finditer(r"^[ \t]*#define[ \t]+(RG_\w+)[ \t]+(\(?-?(?:0x[0-9a-fA-F]+|\d+)(?:u|U|ull|ULL)?\)?)[ \t]*$", src, flags=re.M):
        v = m.group(2).strip("()")
        wide = v.lower().endswith("ull")
        v = re.sub(r"(?i)u?l*$", "", v)
        consts.append((m.group(1), v, "u64" if wide else "u32"))
    for m in re.finditer(r"typedef\s+enum\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        items, nxt = [], 0
        for it in m.group(1).split(","):
            it = it.strip()
            if not it:
                continue
            if "=" in it:
                k, v = [x.strip() for x in it.split("=")]
                nxt = int(v, 0)
            else:
                k = it
            items.append((k, nxt))
            nxt += 1
        enums.append((m.group(2), items))
        typedefs[m.group(2)] = "i32"
    fp = re.search(r"typedef\s+(\w+)\s*\(\s*\*\s*(\w+)\s*\)\s*\((.*?)\)\s*;", src, flags=re.S)
    fnptr = None
    if fp:
        fnptr = (fp.group(2), fp.group(1), fp.group(3))
        typedefs[fp.group(2)] = camel(fp.group(2))
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        typedefs[m.group(2)] = camel(m.group(2))
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for stmt in m.group(1).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            # 'uint64_t a, b' / 'uint64_t *m, *n' / 'uint64_t m[8], n[8]': several declarators share the base type, each
            # brings its own stars and array bounds
            bm = re.match(r"^((?:const\s+)?(?:struct\s+)?\w+)\s*(.*)$", stmt)
            base, decls = bm.group(1), bm.group(2)
            for d in decls.split(","):
                fields.append(parse_decl(base + " " + d.strip(), typedefs, False))
        structs.append((m.group(2), fields))
    body = re.sub(r"typedef\s+(struct|enum)\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    body = re.sub(r"typedef[^;{]*;", " ", body)
    body = re.sub(r"^[ \t]*#.*$", " ", body, flags=re.M)
    for m in re.finditer(r"([\w \t\*]+?)\b(rg_\w+)\s*\(([^()]*)\)\s*;", body, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                params.append(parse_decl(a, typedefs, True))
        funcs.append((name, None if ret == "void" else rust_type(ret, typedefs), params))
    if fnptr:
        name, ret, args = fnptr
        ps = [parse_decl(a, typedefs, True) for a in " ".join(args.split()).split(",")]
        fnptr = (camel(name), rust_type(ret, typedefs), ps)
    return consts, enums, structs, funcs, fnptr


def emit(consts, enums, structs, funcs, fnptr):
    o = ["// GENERATED from include/raftgroups.h by tools/gen_rust_bindings.py -- do not edit.",
         "// Link with `-lraftgroups` (build.rs: cargo:rustc-link-lib=dylib=raftgroups). Never compiled in the build image (no",
         "// Rust toolchain there); tests/test_abi.py checks that it names every export of libraftgroups.so with the header's arity.",
         "#![allow(non_camel_case_types, dead_code)]",
         "use std::os::raw::{c_char, c_void};", ""]
    for name, v, ty in consts:
        o.append(f"pub const {name}: {ty} = {v};")
    o.append("")
    for ename, items in enums:
        o.append(f"// {ename}")
        for k, v in items:
            o.append(f"pub const {k}: i32 = {v};")
        o.append("")
    o.append("pub enum RgEngine {} // opaque")
    if fnptr:
        n, ret, ps = fnptr
        o.append(f"pub type {n} = Option<unsafe extern \"C\" fn({', '.join(t for _, t in ps)}) -> {ret}>;")
    o.append("")
    for sname, fields in structs:
        o.append("#[repr(C)]")
        o.append(f"pub struct {camel(sname)} {{")
        for fname, ft in fields:
            o.append(f"    pub {ident(fname)}: {ft},")
        o.append("}")
        o.append("")
    o.append('extern "C" {')
    for name, ret, params in funcs:
        ps = ", ".join(f"{ident(n) if n else '_'}: {t}" for n, t in params)
        o.append(f"    pub fn {name}({ps})" + (f" -> {ret}" if ret else "") + ";")
    o.append("}")
    o += ["",
          "/// Error mapping back to raft::Error (src/errors.rs:6-50)",
          "pub fn check(rc: i32) -> raft::Result<()> {",
          "    match rc {",
          "        0 => Ok(()),",
          "        RG_ERR_STEP_LOCAL_MSG => Err(raft::Error::StepLocalMsg),",
          "        RG_ERR_STEP_PEER_NOT_FOUND => Err(raft::Error::StepPeerNotFound), // checked BEFORE the term, as RawNode::step does",
          "        _ => panic!(\"raftgroups: {}\", unsafe { std::ffi::CStr::from_ptr(rg_last_error()) }.to_string_lossy()),",
          "    }",
          "}", ""]
    return "\n".join(o)


def generate():
    return emit(*parse(open(HEADER, encoding="utf-8").read()))


def doc_with_block(doc, rs):
    block = f"{BEGIN}\n```rust\n{rs}```\n{END}"
    if BEGIN in doc and END in doc:
        return doc[:doc.index(BEGIN)] + block + doc[doc.index(END) + len(END):]
    raise SystemExit("INTEGRATION.md lacks the generated-block markers")


def main():
    rs = generate()
    doc = open(DOC, encoding="utf-8").read()
    new_doc = doc_with_block(doc, rs)
    if "--check" in sys.argv:
        stale = []
        if not os.path.exists(OUT_RS) or open(OUT_RS, encoding="utf-8").read() != rs:
            stale.append(OUT_RS)
        if new_doc != doc:
            stale.append(DOC)
        if stale:
            print("out of date (run tools/gen_rust_bindings.py):", ", ".join(stale))
            return 1
        return 0
    os.makedirs(os.path.dirname(OUT_RS), exist_ok=True)
    open(OUT_RS, "w", encoding="utf-8").write(rs)
    open(DOC, "w", encoding="utf-8").write(new_doc)
    print(f"wrote {OUT_RS} and the generated block of {DOC}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
