"""bindings/rust has never met a compiler (no Rust toolchain in this image), so what CAN be checked is checked here, by
reading it: every function of include/ibu_hip.h is declared in the `extern "C"` block of bindings/rust/src/ffi.rs and
nothing else is; each declaration has the header's number of parameters, and every parameter and return value has the
header's ABI class (pointer / integer of that width / double); every `#[repr(C)]` struct has the size of its ctypes twin,
and the ctypes twins have the sizes gcc gives the header's structs."""
import ctypes as C
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ibu_hip.h")
FFI = os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")


def _strip_c(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^\s*#[^\n]*", "", src, flags=re.M)            # preprocessor lines
    return src


def _split_params(s):
    s = s.replace("->", " RETURNS ")                          # not a closing bracket
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


def c_prototypes():
    src = _strip_c(open(HEADER).read())
    src = re.sub(r"typedef\s+struct[^;{]*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)   # struct bodies
    src = re.sub(r"typedef[^;]*\(\*[^;]*;", "", src)                                # function-pointer typedefs
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(ibu_[a-z0-9_]+)\s*\(([^;{]*)\)\s*;", src):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        plist = [] if params in ("", "void") else _split_params(params)
        protos[name] = (ret, plist)
    return protos


def c_class(t):
    """ABI class of a C parameter / return type."""
    t = t.strip()
    if "*" in t or "[" in t or re.search(r"\b(ibu_write_fn|ibu_flush_fn|ibu_read_fn)\b", t):
        return "ptr"
    t = re.sub(r"\b(const|struct|enum)\b", "", t)
    t = re.sub(r"\b[a-z_][a-z0-9_]*$", "", t.strip()).strip() or t.strip()       # drop the parameter name
    return {"int32_t": "i32", "uint32_t": "i32", "int": "i32", "uint64_t": "i64", "int64_t": "i64", "size_t": "i64", "double": "f64",
            "void": "void", "uint8_t": "i8", "char": "i8"}.get(t, "?" + t)


def rust_class(t):
    t = t.strip()
    if t.startswith("*") or t.startswith("Option<") or t in ("ibu_write_fn", "ibu_flush_fn", "ibu_read_fn"):
        return "ptr"
    return {"i32": "i32", "u32": "i32", "c_int": "i32", "u64": "i64", "i64": "i64", "usize": "i64", "f64": "f64", "u8": "i8", "c_char": "i8",
            "()": "void"}.get(t, "?" + t)


def rust_externs():
    src = re.sub(r"//[^\n]*", "", open(FFI).read())
    block = src[src.index('extern "C" {'):]
    fns = {}
    for m in re.finditer(r"pub fn (ibu_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        params = [p.split(":", 1)[1].strip() for p in _split_params(m.group(2).strip()) if ":" in p]
        fns[m.group(1)] = ((m.group(3) or "()").strip(), params)
    return fns


def test_every_header_function_is_declared_in_rust_and_nothing_else():
    c, r = c_prototypes(), rust_externs()
    assert len(c) >= 70
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))


def test_parameter_counts_and_abi_classes_agree():
    c, r = c_prototypes(), rust_externs()
    bad = []
    for name, (cret, cparams) in sorted(c.items()):
        rret, rparams = r[name]
        if len(cparams) != len(rparams):
            bad.append((name, "parameter count", len(cparams), len(rparams)))
            continue
        if c_class(cret + " x") != rust_class(rret):
            bad.append((name, "return", cret, rret))
        for k, (cp, rp) in enumerate(zip(cparams, rparams)):
            if c_class(cp) != rust_class(rp):
                bad.append((name, f"parameter {k}", cp, rp))
    assert not bad, bad


def _twins():
    from ibu_amd import _lib
    return {"ibu_header_t": _lib.CHeader, "ibu_record_t": _lib.CRecord, "ibu_error_detail_t": _lib.CErrorDetail,
            "ibu_reduce_result_t": _lib.CReduceResult, "ibu_ring_config_t": _lib.CRingConfig, "ibu_stream_stats_t": _lib.CStreamStats,
            "ibu_alloc_probe_t": _lib.CAllocProbe, "ibu_sort_shard_t": _lib.CSortShard, "ibu_decode_sink_t": _lib.CDecodeSink,
            "ibu_key_plan_t": _lib.CKeyPlan, "ibu_processor_vtable_t": _lib.CProcessorVTable, "ibu_numa_info_t": _lib.CNumaInfo}


def test_ctypes_structs_have_the_sizes_the_header_compiles_to(tmp_path):
    twins = _twins()
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "ibu_hip.h"\nint main(void) {\n'
                   + "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in twins) + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    assert {n: int(v) for n, v in got.items()} == {n: C.sizeof(t) for n, t in twins.items()}


def test_repr_c_structs_have_the_size_of_their_ctypes_twins():
    src = re.sub(r"//[^\n]*", "", open(FFI).read())
    sizes = {"u8": 1, "i8": 1, "c_char": 1, "u32": 4, "i32": 4, "f32": 4, "u64": 8, "i64": 8, "f64": 8, "usize": 8}

    def field_size_align(t):
        t = t.strip()
        m = re.fullmatch(r"\[(.+);\s*(\d+)\]", t)
        if m:
            s, a = field_size_align(m.group(1))
            return s * int(m.group(2)), a
        if t.startswith("*") or t.startswith("Option<"):
            return 8, 8
        return sizes[t], sizes[t]

    twins = _twins()
    seen = 0
    for m in re.finditer(r"#\[repr\(C\)\][^{]*?pub struct (\w+)\s*\{(.*?)\n\}", src, flags=re.S):
        name, body = m.group(1), m.group(2)
        if name not in twins:
            continue
        off, amax = 0, 1
        for f in _split_params(body):
            if ":" not in f:
                continue
            s, a = field_size_align(f.split(":", 1)[1])
            off = (off + a - 1) // a * a + s
            amax = max(amax, a)
        size = (off + amax - 1) // amax * amax
        assert size == C.sizeof(twins[name]), (name, size, C.sizeof(twins[name]))
        seen += 1
    assert seen == len(twins), seen


def test_safe_wrappers_call_the_externs_with_the_right_number_of_arguments():
    """bindings/rust/src/lib.rs: every `ffi::ibu_*( ... )` call names a declared extern and passes as many arguments as it takes;
    brackets balance over the whole file (the cheapest proxy for "it parses")."""
    src = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    code = re.sub(r"//[^\n]*", "", src)
    code = re.sub(r'"(?:[^"\\]|\\.)*"', '""', code)           # string literals may hold brackets
    code = re.sub(r"'(?:[^'\\]|\\.)'", "' '", code)           # char literals too (lifetimes like 'a stay: no closing quote)
    for o, c in ("()", "[]", "{}"):
        assert code.count(o) == code.count(c), (o, code.count(o), code.count(c))
    externs = rust_externs()
    calls = 0
    for m in re.finditer(r"ffi::(ibu_[a-z0-9_]+)\s*\(", code):
        name, i, depth = m.group(1), m.end(), 1
        assert name in externs, name
        j = i
        while depth:
            depth += {"(": 1, ")": -1}.get(code[j], 0)
            j += 1
        args = _split_params(code[i:j - 1])
        assert len(args) == len(externs[name][1]), (name, len(args), len(externs[name][1]), code[i:j - 1][:120])
        calls += 1
    assert calls >= 60, calls
