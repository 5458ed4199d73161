"""The C-ABI library loads without a GPU and exports every symbol include/ibu_hip.h declares.
No compute calls here (CPU-only box)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ibu_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)  # drop comments
    src = re.sub(r"typedef[^;]*\(\*[^;]*;", "", src)  # drop function-pointer typedefs
    src = re.sub(r"\{[^{}]*\}", "{}", src)  # drop struct bodies (function-pointer members)
    return sorted(set(re.findall(r"\b(ibu_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_what_we_think():
    names = declared_functions()
    assert len(names) >= 70
    for must in ("ibu_decode_ascii", "ibu_encode_ascii", "ibu_deserialize", "ibu_serialize", "ibu_reduce",
                 "ibu_writer_write_batch", "ibu_load_to_vec", "ibu_mmap_process_parallel", "ibu_shard_range"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from ibu_amd import _lib
    lib = ctypes.CDLL(_lib.SO_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_every_declared_symbol():
    from ibu_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_library_carries_gfx950_code_object():
    from ibu_amd import _lib
    out = subprocess.run(["strings", "-a", _lib.SO_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out and "ibu_k_decode" in out and "ibu_k_encode" in out


def test_no_product_kernel_uses_scratch():
    """Every gfx950 kernel in the library must fit its register budget: `.private_segment_fixed_size` of the code
    objects' metadata (tools/kernel_resources.py, needs no GPU) is 0 for all of them.  A spilling streaming kernel
    writes and re-reads its spills through HBM: round 2's compress-with-census spilled 36 bytes per lane and wrote
    20.3 B/record instead of 13 without any test noticing."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    from ibu_amd import _lib
    ks = kernel_resources.all_kernels(_lib.SO_PATH)
    assert len(ks) > 150 and any("ibu_k_sort_compress<true, 3, false>" in k for k in ks), sorted(ks)[:5]
    spilling = {k: v["private_segment_fixed_size"] for k, v in ks.items() if v.get("private_segment_fixed_size", 0)}
    assert not spilling, spilling
    dynamic = [k for k, v in ks.items() if v.get("uses_dynamic_stack", 0)]
    assert not dynamic, dynamic


def test_header_compiles_as_c_and_cxx(tmp_path):
    c = tmp_path / "t.c"
    c.write_text('#include "ibu_hip.h"\n_Static_assert(sizeof(ibu_header_t)==32 && sizeof(ibu_record_t)==24, "pod");\n'
                 "int main(void){ibu_header_t h; ibu_header_init(&h,16,12); return ibu_header_validate(&h);}\n")
    from ibu_amd import _lib
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o",
                           str(exe), _lib.SO_PATH, f"-Wl,-rpath,{os.path.dirname(_lib.SO_PATH)}"])
    assert subprocess.run([str(exe)]).returncode == 0
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include "ibu_hip.h"\nint main(){return 0;}\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I",
                           os.path.join(ROOT, "include"), str(cpp)])


def test_device_entry_points_fail_loudly_without_gpu():
    """On a box with no GPU the device path must report NoDevice — never quietly compute on the host."""
    import ibu_amd
    if ibu_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ibu_amd.IbuError) as ei:
        ibu_amd.Context(0)
    assert ei.value.kind == "NoDevice"
    # the one-call multi-device form too: no device list -> "all devices" -> none visible -> NoDevice, never a host loop
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "x.ibu")
        w = ibu_amd.Writer.from_path(path, ibu_amd.Header(16, 12))
        w.write_batch(ibu_amd.records_array([(1, 2, 3), (4, 5, 6)]))
        w.finish()
        w.close()
        m = ibu_amd.MmapReader.new(path)
        for devices in ((), (0,), (0, 0)):
            with pytest.raises(ibu_amd.IbuError) as ei:
                m.process_devices(devices, ibu_amd.PROC_REDUCE)
            assert ei.value.kind == "NoDevice", devices
        m.close()


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under ibu_amd/, include/, examples/, tools/ or bindings/ may import,
    link or call it; bench.py may, inside its cpu_baseline leg only."""
    bad = []
    for top in ("ibu_amd", "include", "examples", "tools", "bindings"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", ".rs", "Makefile")):
                    txt = open(os.path.join(dirpath, f), errors="replace").read()
                    if re.search(r"ibu_oracle|from oracle|import oracle|orc_[a-z]", txt):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"from oracle|import oracle|orc\.", bench)]
    lo = bench.index("def cpu_baseline(")
    hi = bench.index("\ndef ", lo + 1)
    assert uses and all(lo < u < hi for u in uses), "bench.py touches the oracle outside cpu_baseline()"
    from ibu_amd import _lib
    deps = subprocess.run(["ldd", _lib.SO_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in deps
