"""integration/apply_to_reference.sh run for real on a temporary copy of the reference (as far as an image
without rustc allows).  The reference never travels to the GPU box: the whole module skips when /root/reference is absent.

What a Rust compiler would reject and a text check can see is checked here: the script's exit status, brace balance of
the patched src/main.rs, the inserted statements' position against the statements that MOVE `scene`, `random_samples`
and `primitives` (E0382), identifiers whose definitions the patch deletes (E0425 / E0282), and that every `rtx_ffi::` /
`render_gpu::` item and every struct field the inserted code names exists with `pub` visibility."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SCRIPT = os.path.join(ROOT, "integration", "apply_to_reference.sh")

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "src", "main.rs")),
                                reason="the reference checkout is not on this machine")


def strip_rust(src):
    """comments, string and char literals out (what is left can be brace-counted and grepped for identifiers)"""
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r'"(?:\\.|[^"\\])*"', '""', src)
    return re.sub(r"'(?:\\.|[^'\\])'", "' '", src)


def function_body(lines, name):
    """(first, last) 0-based line numbers of `fn name` in a list of comment-free lines"""
    start = next(i for i, l in enumerate(lines) if re.search(r"\bfn %s\b" % name, l))
    depth, seen = 0, False
    for i in range(start, len(lines)):
        depth += lines[i].count("{") - lines[i].count("}")
        seen = seen or "{" in lines[i]
        if seen and depth == 0:
            return start, i
    raise AssertionError("fn %s does not close" % name)


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    dst = tmp_path_factory.mktemp("ref") / "Ray-Tracer-Rust"
    shutil.copytree(REF, dst, ignore=shutil.ignore_patterns(".git", "target"))
    r = subprocess.run(["bash", SCRIPT, str(dst), os.path.join(ROOT, "ray-tracer-rust_amd")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return dst


def test_script_applies_and_refuses_a_second_run(patched):
    for f in ("src/rtx_ffi.rs", "src/render_gpu.rs", "build.rs"):
        assert (patched / f).is_file(), f
    assert (patched / "Cargo.toml").read_text().count('build = "build.rs"') == 1
    again = subprocess.run(["bash", SCRIPT, str(patched), "x"], capture_output=True, text=True)
    assert again.returncode != 0 and "already patched" in again.stderr


def test_script_refuses_another_revision(tmp_path):
    dst = tmp_path / "r"
    shutil.copytree(REF, dst, ignore=shutil.ignore_patterns(".git", "target"))
    m = dst / "src" / "main.rs"
    m.write_text("// one more line\n" + m.read_text())
    before = m.read_text()
    r = subprocess.run(["bash", SCRIPT, str(dst), "x"], capture_output=True, text=True)
    assert r.returncode != 0 and "not the revision" in r.stderr
    assert m.read_text() == before and not (dst / "src" / "rtx_ffi.rs").exists()      # refused before touching anything


def test_patched_main_is_well_formed(patched):
    text = (patched / "src" / "main.rs").read_text()
    code = strip_rust(text)
    assert code.count("{") == code.count("}") and code.count("(") == code.count(")") and code.count("[") == code.count("]")
    assert len(re.findall(r"^mod rtx_ffi;$", text, re.M)) == 1 and len(re.findall(r"^mod render_gpu;$", text, re.M)) == 1
    assert len(re.findall(r"^static mut FLAT\b", text, re.M)) == 1
    lines = code.split("\n")

    # main(): the flatten lines are statements of main's own block, in front of the statement that moves `primitives`
    m0, m1 = function_body(lines, "main")
    flat = next(i for i in range(m0, m1) if "render_gpu::flatten(&primitives)" in lines[i])
    scene = next(i for i in range(m0, m1) if re.search(r"let scene = Scene\s*\{", lines[i]))
    bvh = next(i for i in range(m0, m1) if "BoundingVolumeHierarchy::new(primitives)" in lines[i])
    assert flat < scene < bvh
    depth = sum(l.count("{") - l.count("}") for l in lines[m0:flat])
    assert depth == 1, "the flatten statement sits inside a nested block or a struct literal (depth %d)" % depth
    assert "FLAT = Some(" in lines[flat + 1]

    # render(): nothing names a value after the statement that moved it
    r0, r1 = function_body(lines, "render")
    body = lines[r0:r1 + 1]
    for name, mover in (("scene", "Arc::new(scene)"), ("random_samples", "Arc::new(random_samples)"),
                        ("pixels", "Arc::new(pixels)")):
        at = next(i for i, l in enumerate(body) if mover in l)
        later = [l for l in body[at + 1:] if re.search(r"(?<![\w.])%s\b(?!_ptr)" % name, l)]
        assert not later, (name, later)
    # what the patch deleted is not named any more (the channel and the fan-out's clones)
    for gone in ("tx", "rx", "cur_scene", "cur_pixels", "cur_img", "cur_random_samples"):
        assert not re.search(r"\b%s\b" % gone, code), gone
    assert "mpsc::channel" not in code and "thread::spawn" not in code
    # and what the inserted code reads is defined in front of it, in render()
    call = next(i for i, l in enumerate(body) if "render_gpu::render_frame(" in l)
    for name in ("w", "h", "img", "scene_ptr", "random_samples_ptr"):
        assert any(re.search(r"let (mut )?%s\b" % name, l) for l in body[:call]), name
    # the image is saved after it was filled
    assert call < next(i for i, l in enumerate(body) if "img.save(" in l)


def test_inserted_code_names_only_what_exists(patched):
    main = strip_rust((patched / "src" / "main.rs").read_text())
    ffi = strip_rust((patched / "src" / "rtx_ffi.rs").read_text())
    gpu = strip_rust((patched / "src" / "render_gpu.rs").read_text())
    pub = lambda src: set(re.findall(r"pub (?:unsafe )?(?:fn|struct|const|static|type) (\w+)", src))
    for user in (main, gpu):
        used = set(re.findall(r"\brtx_ffi::(\w+)", user)) - {"self"}
        assert used and used <= pub(ffi), used - pub(ffi)
    braces = re.search(r"use rtx_ffi::\{([^}]*)\}", gpu).group(1)
    assert {n.strip() for n in braces.split(",")} - {"self"} <= pub(ffi)
    used = set(re.findall(r"\brender_gpu::(\w+)", main))
    assert used == {"FlatScene", "flatten", "render_frame"} and used <= pub(gpu)

    # arity of the one call main.rs makes into render_gpu, and of the ones render_gpu makes into the library
    def params(src, fn):
        sig = re.search(r"fn %s\s*\((.*?)\)\s*(?:->|\{|;)" % fn, src, re.S).group(1)
        return len([p for p in re.split(r",(?![^<]*>)", sig) if p.strip()])

    def args(src, call):
        i = src.index(call) + len(call)
        depth, n, j, any_ = 1, 0, i, False
        while depth:
            c = src[j]
            depth += c in "([{"
            depth -= c in ")]}"
            if c == "," and depth == 1:
                n += 1
            any_ = any_ or not c.isspace()
            j += 1
        return n + 1 if src[i:j - 1].strip() else 0

    assert args(main, "render_gpu::render_frame(") == params(gpu, "render_frame")
    assert args(main, "render_gpu::flatten(") == params(gpu, "flatten")
    for fn in ("rtx_scene_create", "rtx_render_frame", "rtx_scene_destroy", "rtx_device_count", "check"):
        assert args(gpu, "rtx_ffi::%s(" % fn) == params(ffi, fn), fn

    # RtxStats / RtxSceneDesc fields
    def fields(src, struct):
        return re.findall(r"pub (\w+):", re.search(r"pub struct %s \{(.*?)\n\}" % struct, src, re.S).group(1))

    assert set(re.findall(r"\bstats\.(\w+)", main)) <= set(fields(ffi, "RtxStats"))
    literal = re.search(r"let desc = RtxSceneDesc \{(.*?)\n    \};", gpu, re.S).group(1)
    assert re.findall(r"^\s*(\w+):", literal, re.M) == fields(ffi, "RtxSceneDesc")     # every field, once, in order


def test_reference_fields_the_binding_reads_are_public():
    """render_gpu.rs reads struct fields of the reference's own types; each must be `pub` there (private ones, like
    Triangle.e1/e2 or BVH.root, are why the primitives are flattened in main())"""
    gpu = strip_rust(open(os.path.join(ROOT, "integration", "render_gpu.rs")).read())
    where = {"t": ("src/tracer/primitives/triangle.rs", "Triangle"), "lt": ("src/tracer/primitives/triangle.rs", "Triangle"),
             "s": ("src/tracer/primitives/sphere.rs", "Sphere"), "camera": ("src/tracer/utils/camera.rs", "Camera")}
    seen = 0
    for var, (path, struct) in where.items():
        src = strip_rust(open(os.path.join(REF, path)).read())
        body = re.search(r"pub struct %s\s*\{(.*?)\}" % struct, src, re.S).group(1)
        public = set(re.findall(r"pub\s+(\w+)\s*:", body))
        used = set(re.findall(r"\b%s\.(\w+)\b(?!\()" % var, gpu))
        assert used and used <= public, (struct, used - public)
        seen += len(used)
    assert seen >= 12
    color = strip_rust(open(os.path.join(REF, "src/tracer/utils/color.rs")).read())
    assert set(re.findall(r"\.color\.(\w+)", gpu)) <= set(re.findall(r"pub\s+(\w+)\s*:", color))
