"""The OBJ loader beyond import_obj: every extension is opt-in, flags = 0 is import_obj itself, and on
the reference's own assets — which use none of the extensions — all flag sets give the same triangles."""
import importlib
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rtx():
    return importlib.import_module("ray-tracer-rust_amd")


def write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


@pytest.mark.parametrize("asset", ["bunny.obj", "big_bunny.obj"])
def test_reference_assets_are_read_identically(rtx, asset):
    path = os.path.join(ROOT, "models", asset)
    base = rtx.import_obj(path)
    t0, c0 = rtx.import_obj_ex(path, 0)
    assert np.array_equal(t0, base) and (c0 == 1.0).all()                 # Color::new(1,1,1), src/main.rs:146
    for flags in (rtx.OBJ_SLASHES, rtx.OBJ_RELATIVE, rtx.OBJ_POLYGONS, rtx.OBJ_SLASHES | rtx.OBJ_RELATIVE | rtx.OBJ_POLYGONS):
        t, c = rtx.import_obj_ex(path, flags)
        assert np.array_equal(t, base) and (c == 1.0).all()
    t, c = rtx.import_obj_ex(path, rtx.OBJ_ALL)
    assert np.array_equal(t, base)
    if asset == "big_bunny.obj":                                          # models/big_bunny.mtl: Kd 0.8 0.8 0.8
        assert np.allclose(c, 0.8) and os.path.exists(os.path.join(ROOT, "models", "big_bunny.mtl"))


def test_slashes_relative_indices_polygons(rtx, tmp_path):
    obj = write(tmp_path, "a.obj", """# a quad and a triangle
v 0 0 0
v 1 0 0
v 1 1 0
v\t0   1 0
vt 0 0
vn 0 0 1
f 1/1/1 2/1/1 3//1 4/1
v 2 2 2
f -1 -5 -4
""")
    t, c = rtx.import_obj_ex(obj, rtx.OBJ_ALL)
    want = np.array([[0, 0, 0, 1, 0, 0, 1, 1, 0], [0, 0, 0, 1, 1, 0, 0, 1, 0], [2, 2, 2, 0, 0, 0, 1, 0, 0]], np.float32)
    assert np.array_equal(t, want) and (c == 1.0).all()
    t, _ = rtx.import_obj_ex(obj, rtx.OBJ_SLASHES | rtx.OBJ_RELATIVE)     # without POLYGONS: the first three corners
    assert np.array_equal(t, want[[0, 2]])
    for flags, _what in ((rtx.OBJ_RELATIVE | rtx.OBJ_POLYGONS, "slash"), (rtx.OBJ_SLASHES | rtx.OBJ_POLYGONS, "negative")):
        with pytest.raises(rtx.RtxError) as e:
            rtx.import_obj_ex(obj, flags)
        assert e.value.code == rtx.ERR_IO
    with pytest.raises(rtx.RtxError):                                     # index out of range
        rtx.import_obj_ex(write(tmp_path, "b.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n"), rtx.OBJ_ALL)
    with pytest.raises(rtx.RtxError):
        rtx.import_obj_ex(write(tmp_path, "c.obj", "v 0 0 0\nv 1 0 0\nv 0 1 0\nf -1 -2 -4\n"), rtx.OBJ_ALL)
    with pytest.raises(rtx.RtxError) as e:
        rtx.import_obj_ex(obj, 16)
    assert e.value.code == rtx.ERR_BAD_ARG


def test_material_colours(rtx, tmp_path):
    write(tmp_path, "m.mtl", "newmtl red\nKa 0 0 0\nKd 1 0 0\nnewmtl grey\nKd 0.25 0.25 0.25\nnewmtl bare\n")
    obj = write(tmp_path, "m.obj", """mtllib m.mtl
v 0 0 0
v 1 0 0
v 0 1 0
f 1 2 3
usemtl red
f 1 2 3
usemtl grey
f 1 2 3
f 3 2 1
usemtl bare
f 1 2 3
usemtl nowhere
f 1 2 3
""")
    t, c = rtx.import_obj_ex(obj, rtx.OBJ_MATERIALS)
    assert len(t) == 6
    assert np.array_equal(c, np.array([[1, 1, 1], [1, 0, 0], [0.25] * 3, [0.25] * 3, [1, 1, 1], [1, 1, 1]], np.float32))
    _, c = rtx.import_obj_ex(obj, rtx.OBJ_SLASHES)                        # materials not asked for
    assert (c == 1.0).all()
    missing = write(tmp_path, "n.obj", "mtllib not_there.mtl\nusemtl red\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    _, c = rtx.import_obj_ex(missing, rtx.OBJ_ALL)
    assert (c == 1.0).all()
