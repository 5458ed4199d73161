#!/usr/bin/env python3
"""Register / scratch / LDS budget of every gfx950 kernel in libibu_hip.so, read from the code objects' own metadata.

  python tools/kernel_resources.py [path/to/libibu_hip.so] [--json] [--scratch-only]

The shared library carries one clang offload bundle per translation unit in its `.hip_fatbin` section; each bundle holds
a gfx950 ELF whose NT_AMDGPU_METADATA note (msgpack, printed as YAML by `llvm-readelf --notes`) lists, per kernel,
`.vgpr_count`, `.agpr_count`, `.sgpr_count`, `.private_segment_fixed_size` (scratch bytes per lane — anything but 0 means
spills or a stack), `.group_segment_fixed_size` (static LDS) and `.max_flat_workgroup_size`.  Needs no GPU:
tests/test_abi_symbols.py uses it to fail the CPU suite when a product kernel has scratch (round 2's folded census spilled
36 bytes per lane unnoticed and wrote 1.56x its algorithmic bytes).
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so_path):
    """Yield the bytes of every gfx950 code object bundled in `so_path`."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so_path, os.path.join(td, "unused.so")])
        blob = open(fat, "rb").read()
    at = blob.find(MAGIC)
    while at >= 0:
        (count,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[at + off:at + off + size]
        at = blob.find(MAGIC, at + 1)


def kernels_of(code_object):
    """[(name, {field: int})] from one code object's metadata note (llvm-readelf prints it as a YAML document)."""
    import yaml
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code_object)
        f.flush()
        notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", f.name], text=True)
    out = []
    for doc in re.findall(r"^\s*---\n(.*?)^\s*\.\.\.", notes, flags=re.S | re.M):
        meta = yaml.safe_load(doc)
        for k in meta.get("amdhsa.kernels", []):
            fields = {key[1:]: int(val) for key, val in k.items()
                      if key in (".vgpr_count", ".agpr_count", ".sgpr_count", ".private_segment_fixed_size", ".group_segment_fixed_size",
                                 ".max_flat_workgroup_size", ".vgpr_spill_count", ".sgpr_spill_count", ".uses_dynamic_stack")}
            out.append((k[".name"], fields))
    return out


def demangle(names):
    if not names:
        return {}
    text = subprocess.check_output(["c++filt"] + list(names), text=True)
    return dict(zip(names, text.splitlines()))


def all_kernels(so_path):
    rows = []
    for co in code_objects(so_path):
        rows += kernels_of(co)
    pretty = demangle([n for n, _ in rows])
    res = {}
    for name, fields in rows:
        p = pretty.get(name, name)
        p = re.sub(r"^void ", "", p).split("(")[0]
        res[p] = fields
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    so = args[0] if args else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ibu_amd", "libibu_hip.so")
    ks = all_kernels(so)
    if "--json" in sys.argv:
        print(json.dumps(ks, indent=1, sort_keys=True))
        return
    print(f"{'kernel':90s} vgpr agpr sgpr scratch  lds")
    for name in sorted(ks):
        k = ks[name]
        if "--scratch-only" in sys.argv and not k.get("private_segment_fixed_size"):
            continue
        print(f"{name[:90]:90s} {k.get('vgpr_count', 0):4d} {k.get('agpr_count', 0):4d} {k.get('sgpr_count', 0):4d} "
              f"{k.get('private_segment_fixed_size', 0):7d} {k.get('group_segment_fixed_size', 0):5d}")


if __name__ == "__main__":
    main()
