#!/usr/bin/env python3
"""Generates the Rust FFI declarations of INTEGRATION.md from include/brush_hip.h.

The header is the single source of truth of the C ABI; the `extern "C"` block a brush maintainer
pastes into `crates/brush-render/src/hip_ffi.rs` is derived from it mechanically, so the two cannot
drift (tests/test_host_cpu.py re-generates the block and compares it with the document).

    python tools/gen_rust_ffi.py            # print the block
    python tools/gen_rust_ffi.py --write    # rewrite the block between the markers in INTEGRATION.md
"""
from __future__ import annotations

import os
import re
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "brush_hip.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN = "<!-- BEGIN GENERATED FFI (tools/gen_rust_ffi.py) -->"
END = "<!-- END GENERATED FFI -->"

SCALARS = {"float": "f32", "uint32_t": "u32", "int32_t": "i32", "int": "i32", "size_t": "usize",
           "uint64_t": "u64", "char": "core::ffi::c_char", "void": "core::ffi::c_void",
           "brush_stream_t": "*mut core::ffi::c_void"}


RUST_KEYWORDS = {"in": "input", "type": "kind", "ref": "reference", "move": "mv"}


def strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"^\s*#.*$", "", src, flags=re.M)  # preprocessor lines


def rust_type(ctype: str) -> str:
    """`const float *` -> `*const f32`, `BrushProfiler **` -> `*mut *mut BrushProfiler` ..."""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ").split()
    const = "const" in t
    base = [w for w in t if w not in ("const", "struct")]
    assert len(base) == 1, ctype
    r = SCALARS.get(base[0], base[0])
    for i in range(stars):
        # only the innermost level carries the C const
        r = ("*const " if (const and i == 0) else "*mut ") + r
    return r


def parse_header(src: str):
    src = strip_comments(src)
    structs = []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"(.+?)([\w\[\], ]+)$", decl)
            ctype, names = fm.group(1), fm.group(2)
            # `float lr_mean, lr_scale` / `float viewmat[16]` / `uint32_t *overflow`
            tm = re.match(r"^((?:const\s+)?\w+)\s*(.*)$", decl)
            ctype, rest = tm.group(1), tm.group(2)
            for name in rest.split(","):
                name = name.strip()
                stars = name.count("*")
                name = name.replace("*", "").strip()
                am = re.match(r"(\w+)\[(\d+)\]$", name)
                rt = rust_type(ctype + " " + "*" * stars)
                if am:
                    fields.append((am.group(1), f"[{rt}; {am.group(2)}]"))
                else:
                    fields.append((name, rt))
        structs.append((m.group(3), fields))
    funcs = []
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
    body = re.sub(r"typedef\s+enum\s+\w+\s*\{.*?\}\s*\w+\s*;", "", body, flags=re.S)
    body = re.sub(r"\benum\s*\{.*?\}\s*;", "", body, flags=re.S)
    body = re.sub(r"typedef[^;{}]*;", "", body)
    body = body.replace('extern "C" {', "")
    for m in re.finditer(r"([\w\s\*]+?)\b(brush_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", body, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                am = re.match(r"^(.*?)(\w+)$", a)
                params.append((am.group(2), rust_type(am.group(1))))
        funcs.append((name, ret, params))
    return structs, funcs


def generate(src: str) -> str:
    structs, funcs = parse_header(src)
    out = ["```rust", "// generated from include/brush_hip.h by tools/gen_rust_ffi.py — do not edit by hand"]
    for name, fields in structs:
        out.append("#[repr(C)]")
        out.append(f"pub struct {name} {{")
        for f, t in fields:
            out.append(f"    pub {f}: {t},")
        out.append("}")
    out.append("#[repr(C)] pub struct BrushProfiler { _private: [u8; 0] }")
    out.append('#[link(name = "brush_hip")]')
    out.append('extern "C" {')
    for name, ret, params in funcs:
        params = [(RUST_KEYWORDS.get(p, p), t) for p, t in params]
        ps = ", ".join(f"{p}: {t}" for p, t in params)
        r = "" if ret == "void" else f" -> {rust_type(ret)}"
        line = f"    pub fn {name}({ps}){r};"
        if len(line) > 118:  # wrap long prototypes
            line = textwrap.fill(line, width=118, subsequent_indent="        ", break_long_words=False)
        out.append(line)
    out.append("}")
    out.append("```")
    return "\n".join(out)


def main():
    block = generate(open(HEADER).read())
    if "--write" in sys.argv:
        doc = open(DOC).read()
        a, b = doc.index(BEGIN), doc.index(END)
        doc = doc[:a] + BEGIN + "\n" + block + "\n" + doc[b:]
        open(DOC, "w").write(doc)
    else:
        print(block)


if __name__ == "__main__":
    main()
