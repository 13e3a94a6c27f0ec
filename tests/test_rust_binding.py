"""The Rust side of the boundary (rust/) cannot be compiled here (no cargo / rustc), so it is checked as text:
rust/dray-hip-sys/src/lib.rs against include/dray.h -- every #[repr(C)] struct (field order, types, size and offsets
as gcc lays the C struct out), every constant, every extern "C" signature -- and the flattener and the render body
against the names the reference's scene types serialise under."""
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dray.h")
LIB_RS = os.path.join(ROOT, "rust", "dray-hip-sys", "src", "lib.rs")
FLATTEN_RS = os.path.join(ROOT, "rust", "hip_backend", "flatten.rs")
RENDER_RS = os.path.join(ROOT, "rust", "hip_backend", "camera_render_hip.rs")
REFERENCE = "/root/reference/src/raytracer"

C_TO_RUST = {"double": "f64", "float": "f32", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64",
             "size_t": "usize", "int": "c_int", "char": "c_char", "void": "c_void"}
PRIM = {"f64": (8, 8), "f32": (4, 4), "i32": (4, 4), "u32": (4, 4), "u64": (8, 8), "i64": (8, 8), "usize": (8, 8),
        "c_int": (4, 4), "u8": (1, 1)}


def strip_c_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def strip_rust_comments(text):
    return re.sub(r"//[^\n]*", " ", text)


def c_type_to_rust(ctype):
    """'const dray_camera*' -> '*const dray_camera', 'dray_scene**' -> '*mut *mut dray_scene', 'double' -> 'f64'."""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ").split()
    const = "const" in t
    base = [w for w in t if w not in ("const", "struct")]
    assert len(base) == 1, ctype
    rust = C_TO_RUST.get(base[0], base[0])
    for k in range(stars):
        rust = ("*const " if (const and k == 0) else "*mut ") + rust
    return rust


def parse_header():
    text = strip_c_comments(open(HEADER).read())
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            m = re.match(r"((?:const\s+)?\w+(?:\s*\*+)?)\s+(.*)$", decl, flags=re.S)
            ctype, names = m.group(1), m.group(2)
            for one in names.split(","):
                mm = re.match(r"\s*(\w+)\s*(?:\[(\d+)\])?\s*$", one)
                rust = c_type_to_rust(ctype)
                if mm.group(2):
                    rust = "[%s; %s]" % (rust, mm.group(2))
                fields.append((mm.group(1), rust))
        structs[name] = fields
    opaque = set(re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", text))
    consts = {}
    for body in re.findall(r"enum\s*(?:\w+\s*)?\{(.*?)\}", text, flags=re.S):
        for item in body.split(","):
            m = re.match(r"\s*(\w+)\s*=\s*(-?\d+)u?\s*$", item)
            if m:
                consts[m.group(1)] = int(m.group(2))
    for name, value in re.findall(r"#define\s+(DRAY_\w+)\s+(\d+)\s*$", text, flags=re.M):
        consts[name] = int(value)
    funcs = {}
    for ret, name, args in re.findall(r"((?:const\s+)?\w+\s*\*?)\s*\b(dray_\w+)\s*\(([^)]*)\)\s*;", text):
        if name in structs or ret.strip() in ("typedef",):
            continue
        params = []
        if args.strip() and args.strip() != "void":
            for a in args.split(","):
                m = re.match(r"\s*(.*?)(\w+)\s*$", a, flags=re.S)
                params.append((m.group(2), c_type_to_rust(m.group(1))))
        r = ret.strip()
        funcs[name] = (params, None if r == "void" else c_type_to_rust(r))
    return structs, opaque, consts, funcs


def parse_lib_rs():
    text = strip_rust_comments(open(LIB_RS).read())
    structs, opaque = {}, set()
    for attrs, name, body in re.findall(r"((?:#\[[^\]]*\]\s*)+)pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        assert "repr(C)" in attrs, "%s is not #[repr(C)]" % name
        fields = []
        for f in body.split(","):
            f = f.strip()
            if not f:
                continue
            m = re.match(r"(?:pub\s+)?(\w+)\s*:\s*(.+)$", f, flags=re.S)
            fields.append((m.group(1), re.sub(r"\s+", " ", m.group(2).strip())))
        if fields == [("_private", "[u8; 0]")]:
            opaque.add(name)
        else:
            structs[name] = fields
    # arrays contain a ';' and a ',' never: re-join is not needed, but nested generics would break the split above
    consts = {n: (t, int(v)) for n, t, v in re.findall(r"pub const (\w+)\s*:\s*(\w+)\s*=\s*(-?\d+)\s*;", text)}
    funcs = {}
    ext = re.search(r'extern "C"\s*\{(.*)\}', text, flags=re.S).group(1)
    for name, args, ret in re.findall(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", ext, flags=re.S):
        params = []
        for a in args.split(","):
            a = a.strip()
            if a:
                pn, pt = a.split(":", 1)
                params.append((pn.strip(), re.sub(r"\s+", " ", pt.strip())))
        funcs[name] = (params, ret.strip() if ret else None)
    return structs, opaque, consts, funcs


def rust_layout(structs):
    """repr(C) layout: each field at the next multiple of its alignment, size rounded up to the struct's alignment."""
    memo = {}

    def size_align(t):
        m = re.match(r"\[(.+); (\d+)\]$", t)
        if m:
            s, a = size_align(m.group(1))
            return s * int(m.group(2)), a
        if t.startswith("*"):
            return 8, 8
        if t in PRIM:
            return PRIM[t]
        return layout(t)[:2]

    def layout(name):
        if name not in memo:
            off, align, offsets = 0, 1, {}
            for fname, t in structs[name]:
                s, a = size_align(t)
                off = (off + a - 1) // a * a
                offsets[fname] = off
                off += s
                align = max(align, a)
            memo[name] = ((off + align - 1) // align * align, align, offsets)
        return memo[name]

    return {n: layout(n) for n in structs}


def c_layout(structs, tmp_path):
    """sizeof / offsetof of every struct of the header as gcc lays them out."""
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "dray.h"', "int main(void) {", 'printf("{");']
    first = True
    for name, fields in structs.items():
        src.append('printf("%s\\"%s\\": {\\"size\\": %%zu", sizeof(%s));' % ("" if first else ", ", name, name))
        first = False
        for fname, _ in fields:
            src.append('printf(", \\"%s\\": %%zu", offsetof(%s, %s));' % (fname, name, fname))
        src.append('printf("}");')
    src += ['printf("}\\n");', "return 0;", "}"]
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    return json.loads(subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout)


def test_sys_crate_matches_header(tmp_path):
    c_structs, c_opaque, c_consts, c_funcs = parse_header()
    r_structs, r_opaque, r_consts, r_funcs = parse_lib_rs()
    # structs: same set, same field names in the same order, same types
    assert set(c_structs) == set(r_structs), set(c_structs) ^ set(r_structs)
    assert c_opaque == r_opaque == {"dray_scene", "dray_context"}
    for name, fields in c_structs.items():
        assert fields == r_structs[name], (name, fields, r_structs[name])
    # layout: the repr(C) rules applied to the Rust declarations give gcc's sizes and offsets
    got = rust_layout(r_structs)
    want = c_layout(c_structs, tmp_path)
    for name in c_structs:
        assert got[name][0] == want[name]["size"], (name, got[name][0], want[name]["size"])
        for fname, off in got[name][2].items():
            assert off == want[name][fname], (name, fname, off, want[name][fname])
    assert got["dray_sphere"][0] == got["dray_box"][0] == 72 and got["dray_path_entry"][0] == 128
    assert got["dray_sphere"][2]["shape"] == got["dray_box"][2]["shape"]
    # constants: every enumerator and #define of the header, same value
    assert set(c_consts) == set(r_consts), set(c_consts) ^ set(r_consts)
    for name, value in c_consts.items():
        assert r_consts[name][1] == value, name
    # functions: same set, same parameter names, types and order, same result
    assert set(c_funcs) == set(r_funcs), set(c_funcs) ^ set(r_funcs)
    for name, (params, ret) in c_funcs.items():
        assert (params, ret) == r_funcs[name], (name, params, ret, r_funcs[name])
    from dray_amd import _ffi
    assert set(_ffi.EXPORTS) == set(c_funcs)   # and the Python binding's list is the same list


def test_crate_files_and_flattener_names():
    cargo = open(os.path.join(ROOT, "rust", "dray-hip-sys", "Cargo.toml")).read()
    assert 'name = "dray-hip-sys"' in cargo and 'links = "dray_hip"' in cargo and "[dependencies]" not in cargo
    build = open(os.path.join(ROOT, "rust", "dray-hip-sys", "build.rs")).read()
    assert "rustc-link-lib=dylib=dray_hip" in build and "DRAY_HIP_LIB_DIR" in build
    code = strip_rust_comments(open(FLATTEN_RS).read())
    assert "serde_json" not in code and "impl ser::Serializer for TreeSerializer" in code
    # every method serde::Serializer requires is there (serde 1.x; i128 / u128 and collect_str have defaults)
    required = ["bool", "i8", "i16", "i32", "i64", "u8", "u16", "u32", "u64", "f32", "f64", "char", "str", "bytes", "none",
                "some", "unit", "unit_struct", "unit_variant", "newtype_struct", "newtype_variant", "seq", "tuple",
                "tuple_struct", "tuple_variant", "map", "struct", "struct_variant"]
    for m in required:
        assert re.search(r"fn serialize_%s\b" % m, code), m
    for assoc in ("Ok", "Error", "SerializeSeq", "SerializeTuple", "SerializeTupleStruct", "SerializeTupleVariant",
                  "SerializeMap", "SerializeStruct", "SerializeStructVariant"):
        assert re.search(r"type %s\s*=" % assoc, code), assoc
    render = strip_rust_comments(open(RENDER_RS).read())
    assert '#[cfg(feature = "hip")]' in render and "update_with_buffer" in render
    # the render goes through a context, in passes, the window presented from the library's callback after each
    for name in ("dray_context_create", "dray_context_set_scene", "dray_context_render_progressive", 'extern "C" fn show'):
        assert name in render, name
    sig = re.sub(r"\s+", " ", render)
    assert ("pub fn render( &mut self, world: &dyn Hittable, window: &mut Window, color_buffer: &mut Vec<u32>, "
            "raw_buffer: &mut Vec<Vec3>, count_buffer: &mut Vec<i32> ) -> Result<()>") in sig
    if not os.path.isdir(REFERENCE):
        pytest.skip("the reference's sources are not on this machine: names not cross-checked")
    # the type tags the walk matches are the names typetag serialises the reference's types under, and the fields it
    # reads are fields of those types
    tags = set(re.findall(r'^\s*"(\w+)"\s*=>', code, flags=re.M))
    assert tags == {"HittableList", "Sphere", "BoundingBox", "DefaultMaterial", "Transparent", "Lambertian", "Metal", "Dialectric"}
    ref = {f: open(os.path.join(REFERENCE, f)).read() for f in os.listdir(REFERENCE) if f.endswith(".rs")}
    everything = "\n".join(ref.values())
    for tag in tags:
        assert re.search(r"#\[typetag::serde\]\s*impl (Hittable|Material) for %s\b" % tag, everything), tag
    for struct, fields in (("Sphere", ["center", "radius", "mat"]), ("HittableList", ["objects"]), ("BoundingBox", ["axes"]),
                           ("Vec3", ["coords"]), ("Lambertian", ["albedo"]), ("Metal", ["albedo", "fuzz"]),
                           ("Dialectric", ["refraction_index"]), ("Interval", ["min", "max"])):
        body = re.search(r"pub struct %s\s*\{(.*?)\}" % struct, everything, flags=re.S).group(1)
        for f in fields:
            assert re.search(r"\b%s\s*:" % f, body), (struct, f)
            assert '"%s"' % f in code, (struct, f)
    # the render signature is the reference's
    ref_sig = re.sub(r"\s+", " ", re.search(r"pub fn render\((.*?)\)\s*->\s*Result<\(\)>", ref["camera.rs"], flags=re.S).group(1)).strip()
    got_sig = re.sub(r"\s+", " ", re.search(r"pub fn render\((.*?)\)\s*->\s*Result<\(\)>", render, flags=re.S).group(1)).strip()
    assert ref_sig == got_sig
