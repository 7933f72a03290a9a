#!/usr/bin/env python3
"""Writes bindings/rust/nova-mi355x-sys/src/ffi.rs from include/nova_mi355x.h: every entry point as an `extern "C"` declaration,
every enum / #define constant as a `pub const`, the transcript callback as a type.  The image has no cargo, no rustc and no
bindgen (probed every round), so this does by hand what `bindgen include/nova_mi355x.h` would; constant VALUES come from gcc
(a generated C program that includes the header prints them), not from a re-implementation of C's expression grammar.
tests/test_integration_shim.py regenerates the file and fails when the committed one is stale.

usage: gen_rust_sys.py [--check] [output]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nova_mi355x.h")
OUT = os.path.join(ROOT, "bindings", "rust", "nova-mi355x-sys", "src", "ffi.rs")

SCALAR = {"int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "uint8_t": "u8", "float": "f32", "char": "c_char",
          "void": "c_void", "unsigned": "c_uint", "unsigned int": "c_uint"}


RUST_KEYWORDS = {"as", "box", "fn", "in", "let", "loop", "match", "mod", "move", "mut", "ref", "self", "type", "use", "where", "yield", "final",
                 "override", "priv", "try", "dyn", "impl", "pub", "static", "super", "trait", "unsafe", "crate", "extern", "abstract", "do"}


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def split_params(s):
    return [p.strip() for p in s.split(",") if p.strip() and p.strip() != "void"]


def rust_type(ctype):
    """`const void* const*` -> `*const *const c_void` etc.; the parameter name is already removed."""
    t = re.sub(r"\s+", " ", ctype.strip())
    if t == "nmx_transcript_fn":
        return "NmxTranscriptFn"
    if t == "nmx_ipa_transcript_fn":
        return "NmxIpaTranscriptFn"
    toks = re.findall(r"const|\*|[A-Za-z_][A-Za-z0-9_]*", t)
    # C declarator, left to right: [const] base [const] { * [const] }
    i, base_const = 0, False
    if toks[i] == "const":
        base_const, i = True, i + 1
    base = toks[i]
    i += 1
    if base == "unsigned" and i < len(toks) and toks[i] == "int":
        i += 1
    if i < len(toks) and toks[i] == "const":
        base_const, i = True, i + 1
    r, pointee_const = SCALAR[base], base_const
    while i < len(toks):
        assert toks[i] == "*", (ctype, toks)
        i += 1
        r = ("*const " if pointee_const else "*mut ") + r
        pointee_const = False
        if i < len(toks) and toks[i] == "const":
            pointee_const, i = True, i + 1
    return r


def parse(src):
    src = strip_comments(src)
    protos = []
    for ret, name, params in re.findall(r"\b(int|size_t|uint64_t|const char\s*\*|void)\s+(nmx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ps = []
        for p in split_params(re.sub(r"\s+", " ", params)):
            m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", p)
            pname = m.group(2) + ("_" if m.group(2) in RUST_KEYWORDS else "")
            ps.append((pname, rust_type(m.group(1))))
        protos.append((name, rust_type(ret) if ret.strip() != "void" else None, ps))
    consts = []
    for body in re.findall(r"\benum\s*\{(.*?)\}", src, flags=re.S):
        consts += re.findall(r"\b(NMX_[A-Z0-9_]+)\b\s*(?:=|,|$)", body, flags=re.M)
    consts += [n for n in re.findall(r"^#define\s+(NMX_[A-Z0-9_]+)\s+\S", src, flags=re.M)]
    seen, ordered = set(), []
    for c in consts:
        if c not in seen:
            seen.add(c)
            ordered.append(c)
    return protos, ordered


def const_values(names):
    prog = '#include <stdio.h>\n#include "nova_mi355x.h"\nint main(void){\n' + "".join(
        f'  printf("{n} %lld\\n", (long long)({n}));\n' for n in names) + "  return 0; }\n"
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, "v.c"), os.path.join(d, "v")
        open(c, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.dirname(HEADER), "-o", exe, c], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    return {ln.split()[0]: int(ln.split()[1]) for ln in out.splitlines()}


def const_type(name):
    if name == "NMX_OK" or name.startswith("NMX_E_"):
        return "c_int"
    if name.startswith("NMX_STAT_") or name == "NMX_PROF_STAGES":
        return "usize"
    if name.startswith(("NMX_F_", "NMX_OP_", "NMX_BRANCH_")) or name in ("NMX_BN254_G1", "NMX_GRUMPKIN", "NMX_PALLAS", "NMX_VESTA",
                                                                         "NMX_NUM_CURVES"):
        return "c_int"
    return "u32"


def wrap_fn(name, ret, params):
    head = f"    pub fn {name}("
    parts = [f"{n}: {t}" for n, t in params]
    tail = ")" + (f" -> {ret}" if ret else "") + ";"
    one = head + ", ".join(parts) + tail
    if len(one) <= 140:
        return one
    lines, cur = [], head
    for k, p in enumerate(parts):
        piece = p + ("," if k + 1 < len(parts) else "")
        if len(cur) + len(piece) + 1 > 140:
            lines.append(cur.rstrip())
            cur = " " * 8 + piece
        else:
            cur += ("" if cur.endswith("(") else " ") + piece
    lines.append(cur + tail)
    return "\n".join(lines)


def generate():
    protos, names = parse(open(HEADER).read())
    vals = const_values(names)
    o = ["// GENERATED by scripts/gen_rust_sys.py from include/nova_mi355x.h -- do not edit; re-run the script after a header change.",
         "// What `bindgen include/nova_mi355x.h` would emit (the build image has neither cargo nor bindgen).  NEVER COMPILED HERE:",
         "// tests/test_integration_shim.py checks it against the header mechanically (names, arity, ABI class of every parameter,",
         "// constant values through gcc).  The reference-side precedent for such a crate is blitzar-sys behind src/provider/blitzar.rs:7-40.",
         "#![allow(non_camel_case_types, dead_code)]",
         "use std::os::raw::{c_char, c_int, c_uint, c_void};",
         ""]
    for n in names:
        t, v = const_type(n), vals[n]
        lit = f"0x{v:x}" if (t == "u32" and v > 9) else str(v)
        o.append(f"pub const {n}: {t} = {lit};")
    o += ["",
          "/// One sum-check round: `coeffs32` = `n_coeffs` field elements of 32 bytes (the compressed round polynomial), the callback",
          "/// absorbs them, squeezes the challenge into `challenge32_out` and returns 0.  It must not synchronise the device.",
          "pub type NmxTranscriptFn = unsafe extern \"C\" fn(ctx: *mut c_void, coeffs32: *const u8, n_coeffs: usize, challenge32_out: *mut u8) -> c_int;",
          "/// One round of the inner-product argument: absorb L and R (affine canonical x || y, and whether each is the identity), squeeze r.",
          "pub type NmxIpaTranscriptFn = unsafe extern \"C\" fn(ctx: *mut c_void, l_xy64: *const u8, l_is_inf: c_int, r_xy64: *const u8, r_is_inf: c_int,",
          "                                                     r32_out: *mut u8) -> c_int;",
          "",
          "#[link(name = \"nova_mi355x\")]",
          "extern \"C\" {"]
    for name, ret, params in protos:
        o.append(wrap_fn(name, ret, params))
    o += ["}", ""]
    return "\n".join(o)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--check"]
    out = args[0] if args else OUT
    text = generate()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(out) and open(out).read() == text else 1)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    open(out, "w").write(text)
    print("wrote", os.path.relpath(out, ROOT), text.count("pub fn"), "entry points,", text.count("pub const"), "constants")
