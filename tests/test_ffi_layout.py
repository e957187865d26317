"""The Rust binding (integration/rtx_ffi.rs) cannot be compiled in this image (no rustc), so its `#[repr(C)]` structs are
kept in step with include/rtx.h here: a C program prints sizeof/offsetof of every field of RtxSceneDesc, RtxStats and
RtxSceneInfo as gcc lays them out; the same numbers are computed from the Rust source by the C layout rules
(#[repr(C)] = exactly those rules); both must equal integration/layout.json.  Drift in any of the three fails."""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRUCTS = ("RtxSceneDesc", "RtxStats", "RtxSceneInfo")

# size, alignment of the Rust types the binding uses (x86-64 / any LP64 target)
RUST = {"u8": (1, 1), "u32": (4, 4), "i32": (4, 4), "c_int": (4, 4), "u64": (8, 8), "f32": (4, 4), "f64": (8, 8),
        "usize": (8, 8)}


def c_fields(header, name):
    """Field names of `typedef struct name { ... } name;` in declaration order."""
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        # "uint32_t width, height" / "float eye[3], u[3]" / "const float *v0v1v2"
        first, *more = decl.split(",")
        for part in [first.split()[-1]] + [m.strip() for m in more]:
            names.append(re.sub(r"\[\d+\]", "", part).replace("*", "").strip())
    return names


def c_layout(tmp_path):
    header = open(os.path.join(ROOT, "include", "rtx.h")).read()
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "rtx.h"', "int main(void) {"]
    for s in STRUCTS:
        lines.append('printf("%s %%zu %%zu\\n", sizeof(%s), _Alignof(%s));' % (s, s, s))
        for f in c_fields(header, s):
            lines.append('printf("%s.%s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s *)0)->%s));' % (s, f, s, f, s, f))
    lines.append("return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        k, a, b = line.split()
        out[k] = [int(a), int(b)]
    return out


def rust_layout():
    src = open(os.path.join(ROOT, "integration", "rtx_ffi.rs")).read()
    out = {}
    for s in STRUCTS:
        m = re.search(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\)]*\)\]\s*)?pub struct %s \{(.*?)\n\}" % s, src, re.S)
        assert m, "struct %s with #[repr(C)] not found in rtx_ffi.rs" % s
        off, align = 0, 1
        for name, ty in re.findall(r"pub (\w+): ([^,\n]+),", m.group(1)):
            ty = ty.strip()
            arr = re.match(r"\[(\w+); (\d+)\]", ty)
            if ty.startswith("*"):
                size, al = 8, 8
            elif arr:
                size, al = RUST[arr.group(1)][0] * int(arr.group(2)), RUST[arr.group(1)][1]
            else:
                size, al = RUST[ty]
            off = (off + al - 1) // al * al
            out["%s.%s" % (s, name)] = [off, size]
            off += size
            align = max(align, al)
        out[s] = [(off + align - 1) // align * align, align]
    return out


def test_rust_binding_matches_the_c_header(tmp_path):
    table = json.load(open(os.path.join(ROOT, "integration", "layout.json")))["layout"]
    c = c_layout(tmp_path)
    r = rust_layout()
    assert c == table, {k: (c.get(k), table.get(k)) for k in set(c) | set(table) if c.get(k) != table.get(k)}
    assert r == table, {k: (r.get(k), table.get(k)) for k in set(r) | set(table) if r.get(k) != table.get(k)}
    fields = lambda d: [k for k in d if "." in k]
    assert fields(c) == fields(r)                 # the same fields in the same order


def test_binding_declares_only_what_the_library_exports():
    """every `pub fn` of the extern block is a symbol include/rtx.h declares, with the header's ABI version"""
    src = open(os.path.join(ROOT, "integration", "rtx_ffi.rs")).read()
    hdr = open(os.path.join(ROOT, "include", "rtx.h")).read()
    declared = set(re.findall(r"\b(rtxh?_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)))
    block = re.search(r'extern "C" \{(.*?)\n\}', src, re.S).group(1)
    bound = set(re.findall(r"pub fn (\w+)\(", block))
    assert bound and bound <= declared, bound - declared
    assert int(re.search(r"RTX_ABI_VERSION: c_int = (\d+)", src).group(1)) == int(re.search(r"#define RTX_ABI_VERSION (\d+)", hdr).group(1))
    assert "Vec<(f32, f32)>" in src and "repr(Rust)" in src      # the tuple-layout warning stays with the Sample type
