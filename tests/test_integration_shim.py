"""The Rust side of INTEGRATION.md has never met a compiler (no cargo in this image), so its `extern "C"` blocks are checked
mechanically against include/nova_mi355x.h instead: every `pub fn nmx_*` a maintainer would paste into nova-mi355x-sys must name
an entry point the header declares, with the same number of parameters, the same return type and, parameter by parameter, a Rust
type whose C ABI class is the header's (c_int <-> int, usize <-> size_t, u32/u64 <-> uint32_t/uint64_t, raw pointers <-> pointers,
the transcript callback <-> nmx_transcript_fn).  The reference-side template these follow is src/provider/blitzar.rs:7-40."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nova_mi355x.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")


def _split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def c_class(t):
    t = re.sub(r"\s+", " ", t.strip())
    if t in ("void", ""):
        return "void"
    if "*" in t:
        return "ptr"
    if t.startswith("nmx_transcript_fn") or t.startswith("nmx_ipa_transcript_fn"):
        return "fnptr"
    t = re.sub(r"^const ", "", t)
    t = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", t).strip() if " " in t else t   # drop the parameter name
    return {"int": "i32", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "unsigned": "u32", "unsigned int": "u32"}[t]


def rust_class(t):
    t = t.strip()
    if t.startswith("*const") or t.startswith("*mut"):
        return "ptr"
    if t.startswith(("NmxTranscriptFn", "NmxIpaTranscriptFn", "Option<NmxTranscriptFn")) or "extern \"C\" fn" in t:
        return "fnptr"
    return {"c_int": "i32", "i32": "i32", "u32": "u32", "u64": "u64", "usize": "usize", "c_uint": "u32"}[t]


def header_protos():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for ret, name, params in re.findall(r"\b(int|size_t|uint64_t|const char\s*\*|void)\s+(nmx_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ps = [p for p in _split_params(params) if p and p != "void"]
        protos[name] = ("ptr" if "*" in ret else c_class(ret), [c_class(p) for p in ps])
    return protos


def rust_decls(text=None):
    """Every `pub fn nmx_*(..) -> ..;` declaration (inside an extern block or quoted alone) of the ```rust snippets."""
    doc = open(DOC).read() if text is None else text
    blocks = re.findall(r"```rust\n(.*?)```", doc, flags=re.S) if text is None else [doc]
    decls = []
    for block in blocks:
        block = re.sub(r"//[^\n]*", "", block)
        for name, params, ret in re.findall(r"pub fn (nmx_[a-z0-9_]+)\s*\(([^{};]*?)\)\s*(?:->\s*([^;{]+?))?\s*;", block, flags=re.S):
            ps = [p.split(":", 1)[1] for p in _split_params(params)]
            decls.append((name, rust_class(ret) if ret else "void", [rust_class(p) for p in ps]))
    return decls


def rust_calls(text):
    """(name, number of arguments) of every call `nmx_*(...)` in Rust text that is not a declaration."""
    text = re.sub(r"//[^\n]*", "", text)
    calls = []
    for m in re.finditer(r"(?<!fn )\b(nmx_[a-z0-9_]+)\s*\(", text):
        i, depth, args, cur = m.end(), 1, [], ""
        while depth and i < len(text):
            ch = text[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
                if depth == 0:
                    break
            if ch == "," and depth == 1:
                args.append(cur)
                cur = ""
            else:
                cur += ch
            i += 1
        if cur.strip():
            args.append(cur)
        calls.append((m.group(1), len(args)))
    return calls


def test_header_prototypes_parse():
    protos = header_protos()
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    names = set(re.findall(r"\b(nmx_[a-z0-9_]+)\s*\(", src)) - {"nmx_transcript_fn", "nmx_ipa_transcript_fn"}
    assert names <= set(protos), sorted(names - set(protos))
    assert protos["nmx_init"] == ("i32", ["i32"])
    assert protos["nmx_last_error"] == ("ptr", [])
    assert protos["nmx_min_gpu_n"][0] == "usize"


def test_rust_externs_match_the_header():
    protos = header_protos()
    decls = rust_decls()
    assert len(decls) >= 15, "INTEGRATION.md lost its extern blocks"
    for name, ret, params in decls:
        assert name in protos, f"INTEGRATION.md binds {name}, which include/nova_mi355x.h does not declare"
        c_ret, c_params = protos[name]
        assert ret == c_ret, f"{name}: return {ret} in the Rust shim, {c_ret} in the header"
        assert len(params) == len(c_params), f"{name}: {len(params)} parameters in the Rust shim, {len(c_params)} in the header"
        assert params == c_params, f"{name}: parameter classes {params} in the Rust shim, {c_params} in the header"


CRATE = os.path.join(ROOT, "bindings", "rust", "nova-mi355x-sys")


def test_generated_ffi_is_current_and_complete():
    """bindings/rust/nova-mi355x-sys/src/ffi.rs is what scripts/gen_rust_sys.py makes of today's header, and declares every
    entry point of the header with the header's ABI classes (checked by this file's own, independent parser)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_rust_sys.py"), "--check"])
    assert r.returncode == 0, "bindings/rust/nova-mi355x-sys/src/ffi.rs is stale: run scripts/gen_rust_sys.py"
    protos = header_protos()
    decls = {d[0]: d for d in rust_decls(open(os.path.join(CRATE, "src", "ffi.rs")).read())}
    assert set(decls) == set(protos)
    for name, (c_ret, c_params) in protos.items():
        assert decls[name][1] == c_ret and decls[name][2] == c_params, name


def test_constants_of_the_generated_ffi_match_the_python_binding():
    from nova_amd import _lib
    text = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    consts = {n: int(v, 0) for n, v in re.findall(r"pub const (NMX_[A-Z0-9_]+): \w+ = (-?(?:0x)?[0-9a-f]+);", text)}
    assert consts["NMX_E_NO_DEVICE"] == _lib.E_NO_DEVICE == -2
    assert consts["NMX_SCALARS_MONT"] == 1 and consts["NMX_BASES_MONT"] == 2 and consts["NMX_SCALARS_DEVICE"] == 4
    assert consts["NMX_BITS_AUTO"] == 0xffffffff and consts["NMX_ASYNC"] == 512 and consts["NMX_STAT_COUNT"] == 17
    for name in ("SCALARS_MONT", "BASES_MONT", "SCALARS_DEVICE", "BASES_PRECOMPUTE", "OUT_PARTIAL", "ASYNC", "SCALARS_SHARDED"):
        if hasattr(_lib, name):
            assert getattr(_lib, name) == consts["NMX_" + name], name


def test_calls_in_the_shim_and_the_snippets_have_the_header_arity():
    protos = header_protos()
    lib_rs = open(os.path.join(CRATE, "src", "lib.rs")).read()
    calls = rust_calls(lib_rs)
    assert len(calls) >= 9
    doc = open(DOC).read()
    for block in re.findall(r"```rust\n(.*?)```", doc, flags=re.S):
        calls += rust_calls(block)
    for name, nargs in calls:
        assert name in protos, name
        assert nargs == len(protos[name][1]), f"{name} called with {nargs} arguments, the header takes {len(protos[name][1])}"


def test_the_trait_override_entry_points_are_in_the_shim():
    """What bn256_grumpkin.rs:43-78 / pasta.rs:33-47 need when Nova calls the provider unchanged."""
    bound = {d[0] for d in rust_decls()}
    for need in ("nmx_init", "nmx_last_error", "nmx_min_gpu_n", "nmx_check_layout", "nmx_msm", "nmx_msm_u64", "nmx_msm_batch",
                 "nmx_msm_u64_batch", "nmx_sumcheck_prove_cubic_with_three_inputs", "nmx_sumcheck_prove_quad_prod",
                 "nmx_sumcheck_prove_batch_eval"):
        assert need in bound, need
