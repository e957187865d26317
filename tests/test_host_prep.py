"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol
include/rtx.h declares, and the host-side scene preparation (no device work) agrees with the
oracle on everything that feeds pixel values.  No GPU compute calls here."""
import importlib
import os
import re
import subprocess

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rtx():
    return importlib.import_module("ray-tracer-rust_amd")


def model(name):
    return os.path.join(ROOT, "models", name)


def test_library_exports_every_declared_symbol(rtx):
    hdr = open(os.path.join(ROOT, "include", "rtx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rtxh?_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    out = subprocess.check_output(["nm", "-D", "--defined-only", rtx.rtx.LIB_PATH], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert declared <= exported, declared - exported
    assert declared == set(rtx.rtx._SIGS), declared ^ set(rtx.rtx._SIGS)     # the binding covers the whole header
    assert rtx.abi_version() == 3
    assert rtx.device_count() >= 0


def test_product_library_has_one_pipeline_and_reads_no_environment(rtx):
    """librtx.so must behave the same whatever the caller's environment holds: it imports no getenv, and its device
    code object carries the five kernels of the shipped pipeline only.  The earlier kernel forms and their
    RTX_VARIANT / RTX_LEAF_MAX / ... switches exist in librtx_ablation.so (make ablation) and nowhere else."""
    lib = os.path.join(ROOT, "ray-tracer-rust_amd", "librtx.so")
    undefined = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    blob = open(lib, "rb").read()
    names = lambda b: set(m.decode() for m in re.findall(rb"_ZN3rtx\d+([a-z0-9_]+_kernel)I?", b)
                          if not m.startswith(b"__device_stub__"))
    kernels = names(blob)
    assert kernels == {"reset_kernel", "probe_kernel", "count_classes_kernel", "order_tiles_kernel", "shade_tiles_kernel",
                       "reference_tiles_kernel"}, kernels
    abl = os.path.join(ROOT, "ray-tracer-rust_amd", "librtx_ablation.so")
    assert os.path.exists(abl), "make -C ray-tracer-rust_amd/csrc ablation"
    more = names(open(abl, "rb").read())
    assert kernels < more and "trace_shade_kernel" in more


def test_shipped_library_was_built_with_the_default_switches():
    """The kernel sources carry compile-time switches for A/B builds (tools/ab_build.sh puts those builds under
    gpurun_out/, never over the product).  librtx.so states what it was built with — rtx_build_switches_text, every
    switch and its value — and that text must equal what the sources give with NO -D at all, and name every switch the
    sources define: an experiment build in the product's place fails here."""
    csrc = os.path.join(ROOT, "ray-tracer-rust_amd", "csrc")
    blob = open(os.path.join(ROOT, "ray-tracer-rust_amd", "librtx.so"), "rb").read()
    shipped = set(m.decode() for m in re.findall(rb"rtx-build-switches:[ -~]*", blob))
    assert len(shipped) == 1, shipped
    shipped = shipped.pop()
    pre = subprocess.run(["/opt/rocm/bin/hipcc", "-E", "--cuda-host-only", "-std=c++17", "-I", csrc, os.path.join(csrc, "rtx_kernel.hip")],
                         capture_output=True, text=True, check=True).stdout
    decl = pre[pre.index("rtx_build_switches_text[]"):]
    default = "".join(re.findall(r'"([^"]*)"', decl[:decl.index(";")]))
    assert shipped == default, "librtx.so was not built with the default switches:\n%s\n%s" % (shipped, default)
    named = set(re.findall(r" (RTX_[A-Z0-9_]+)=", shipped))
    defined = set()
    for f in ("rtx_device.h", "rtx_traverse.hpp", "rtx_kernel.hip", "scene_prep.h", "rtx_ablation_kernels.hpp", "rtx_traverse_ablation.hpp"):
        src = open(os.path.join(csrc, f)).read()
        defined |= set(re.findall(r"^#ifndef (RTX_[A-Z0-9_]+)$", src, re.M))
        defined |= set(re.findall(r"^#\s*(?:if|elif)[^\n]*\b(RTX_EXPERIMENT_[A-Z0-9_]+)", src, re.M))
    assert defined <= named, "switches missing from rtx_build_switches_text: %s" % sorted(defined - named)
    assert " RTX_ABLATION=0" in shipped
    abl = open(os.path.join(ROOT, "ray-tracer-rust_amd", "librtx_ablation.so"), "rb").read()
    assert b" RTX_ABLATION=1" in abl


def test_no_device_means_error_not_fallback(rtx, samples_half):
    if rtx.device_count() > 0:
        pytest.skip("a GPU is present")
    with rtx.default_scene([model("bunny.obj")], 16, 16, samples_half[:64]) as s:
        with pytest.raises(rtx.RtxError) as e:
            s.render_rows()
        assert e.value.code == rtx.ERR_NO_DEVICE
        with pytest.raises(rtx.RtxError):
            s.render_frame((0,), 8)


def test_bad_arguments(rtx, samples_half):
    tris, rgb = rtx.default_primitives([model("bunny.obj")])
    with pytest.raises(rtx.RtxError) as e:
        rtx.Scene(0, 16, tris, rgb, samples_half[:8])
    assert e.value.code == rtx.ERR_BAD_ARG
    bad = tris.copy()
    bad[3, 4] = np.nan
    with pytest.raises(rtx.RtxError) as e:
        rtx.Scene(16, 16, bad, rgb, samples_half[:8], tie_rank=None)
    assert e.value.code == rtx.ERR_UNSUPPORTED
    with pytest.raises(rtx.RtxError) as e:
        rtx.import_obj("/nonexistent/file.obj")
    assert e.value.code == rtx.ERR_IO
    assert rtx.rtx._lib.rtx_strerror(-2).decode().startswith("no usable HIP device")


def test_helpers_match_oracle(rtx, orc):
    for name in ("bunny.obj", "big_bunny.obj"):
        assert np.array_equal(rtx.import_obj(model(name)), orc.import_obj(model(name)))
    assert np.array_equal(rtx.gen_samples(rtx.DEFAULT_SEED, 4096), orc.gen_samples(orc.SEED, 4096))
    assert rtx.DEFAULT_SEED == orc.SEED
    u, v, w = rtx.camera_new(rtx.DEFAULT_EYE, rtx.DEFAULT_LOOK_AT, rtx.DEFAULT_UP)
    import ctypes as C
    ou, ov, ow = (np.zeros(3, np.float32) for _ in range(3))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    orc.lib().orc_camera_new(fp(orc.f3(orc.EYE)), fp(orc.f3(orc.LOOK_AT)), fp(orc.f3(orc.UP)), fp(ou), fp(ov), fp(ow))
    assert np.array_equal(u, ou) and np.array_equal(v, ov) and np.array_equal(w, ow)
    t1, c1 = rtx.default_primitives([model("big_bunny.obj")])
    t2, c2 = orc.default_primitives(["big_bunny.obj"])
    assert np.array_equal(t1, t2) and np.array_equal(c1, c2)


def test_obj_import_edge_cases(rtx, tmp_path):
    p = tmp_path / "t.obj"
    # comments, unknown records, an empty last line and CRLF are ignored / tolerated (main.rs:128-145)
    p.write_text("# c\nmtllib x.mtl\no thing\nv 0 0 0\nv 1 0 0\r\nv 0 1 0\nv 0 0 1.5e0\nusemtl None\ns off\nf 1 2 3\nf 2 3 4\n\n")
    t = rtx.import_obj(str(p))
    assert t.tolist() == [[0, 0, 0, 1, 0, 0, 0, 1, 0], [1, 0, 0, 0, 1, 0, 0, 0, 1.5]]
    # faces may only use vertices already read (1-based into the list so far)
    p.write_text("v 0 0 0\nv 1 0 0\nf 1 2 3\nv 0 1 0\n")
    with pytest.raises(rtx.RtxError) as e:
        rtx.import_obj(str(p))
    assert e.value.code == rtx.ERR_IO
    p.write_text("v 0 0\n")            # tokens[3] missing: the reference panics, we report
    with pytest.raises(rtx.RtxError):
        rtx.import_obj(str(p))
    p.write_text("v 0  0 0\n")         # double space -> empty token -> parse failure in the reference
    with pytest.raises(rtx.RtxError):
        rtx.import_obj(str(p))
    p.write_text("f 1/1 2/2 3/3\n")    # slashes are not understood by the reference's parse::<usize>
    with pytest.raises(rtx.RtxError):
        rtx.import_obj(str(p))
    p.write_text("")
    assert rtx.import_obj(str(p)).shape == (0, 9)


def test_reference_leaf_rank_matches_oracle_tree(rtx, orc, samples_half):
    """rtxh_ref_leaf_rank restates BoundingVolumeHierarchy::new; the oracle restates it separately."""
    tris, rgb = orc.default_primitives(["big_bunny.obj"])
    order = orc.Scene(8, 8, tris, rgb, samples_half[:16]).leaf_order()
    rank = rtx.ref_leaf_rank(tris)
    assert sorted(rank.tolist()) == list(range(len(tris)))
    assert np.array_equal(np.argsort(rank), order)
    # the ground (pushed last, popped first) — bvh.rs:194
    small = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, -1, 2, 0, -1, 0, 2, -1],
                      [0, 0, -2, 4, 0, -2, 0, 4, -2], [0, 0, -3, 8, 0, -3, 0, 8, -3]], np.float32)
    assert np.argsort(rtx.ref_leaf_rank(small)).tolist() == [1, 0, 3, 2]      # hand-simulated in test_oracle_kat


def test_prepared_scene_matches_oracle(rtx, orc, samples_seeded):
    import ctypes as C
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    tris, rgb = rtx.default_primitives([model("big_bunny.obj")])
    with rtx.Scene(64, 64, tris, rgb, samples_seeded) as s:
        info = s.info()
        assert info["n_tris"] == 4969 and info["n_light_points"] == 100
        # light points = get_sample(T[i]) for i < 100 (main.rs:194-196)
        lp = s.light_points()
        lt = np.array(orc.LIGHT_TRI, np.float32)
        for i in (0, 1, 57, 99):
            o = np.zeros(3, np.float32)
            orc.lib().orc_triangle_get_sample(fp(lt[0:3]), fp(lt[3:6]), fp(lt[6:9]),
                                              float(samples_seeded[i, 0]), float(samples_seeded[i, 1]), fp(o))
            assert np.array_equal(lp[i], o)
        # normals = Triangle::new
        nrm = s.normals()
        for i in (0, 1, 2500, 4968):
            e1, e2, n = (np.zeros(3, np.float32) for _ in range(3))
            t = tris[i]
            orc.lib().orc_triangle_new(fp(t[0:3].copy()), fp(t[3:6].copy()), fp(t[6:9].copy()), fp(e1), fp(e2), fp(n))
            assert np.array_equal(nrm[i], n)
        assert nrm[4968].tolist() in ([0.0, 1.0, 0.0], [0.0, -1.0, 0.0], [-0.0, -1.0, -0.0], [-0.0, 1.0, -0.0])


def test_gamma_thresholds_reproduce_quantise_exactly(rtx, orc, samples_half):
    """byte(x) = #{b >= 1 : thr[b] <= x} must equal the oracle's (x.powf(1/2.2)*255) as u8 for every float
    tried: all floats within +-300 ulps of each threshold, a dense sweep, and the special values."""
    import ctypes as C
    with rtx.default_scene([model("bunny.obj")], 8, 8, samples_half[:64]) as s:
        thr = s.gamma_thresholds()
    assert thr[0] == -np.inf and (np.diff(thr[1:]) > 0).all() and thr[1] > 0 and thr[255] <= 1.0

    def device_rule(x):
        return np.searchsorted(thr[1:], x, side="right").astype(np.uint8)   # count of thresholds <= x

    def oracle_bytes(x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(len(x), np.uint8)
        tmp = (C.c_uint8 * 3)()
        c = np.zeros(3, np.float32)
        for i, v in enumerate(x):
            c[:] = v
            orc.lib().orc_color_to_rgb8(c.ctypes.data_as(C.POINTER(C.c_float)), tmp)
            out[i] = tmp[0]
        return out

    near = []
    for b in range(1, 256):
        bits = thr[b:b + 1].view(np.uint32)[0]
        near.append((np.arange(-300, 301, dtype=np.int64) + int(bits)).clip(0).astype(np.uint32).view(np.float32))
    near = np.concatenate(near)
    assert np.array_equal(device_rule(near), oracle_bytes(near))
    sweep = np.linspace(0, 1.25, 20001, dtype=np.float32)
    assert np.array_equal(device_rule(sweep), oracle_bytes(sweep))
    special = np.array([0.0, -0.0, 1.0, 2.0, np.inf, 1e-30, 5e-7], np.float32)
    assert np.array_equal(device_rule(special), oracle_bytes(special))
    # the kernel's binary search: b=0; for step in 128..1: if x >= thr[b+step]: b += step  (NaN -> 0)
    def kernel_rule(x):
        b = 0
        for step in (128, 64, 32, 16, 8, 4, 2, 1):
            if x >= thr[b + step]:
                b += step
        return b
    for x in np.concatenate([near[::97], special, np.array([np.nan], np.float32)]):
        exp = 0 if np.isnan(x) else int(device_rule(np.array([x]))[0])
        assert kernel_rule(x) == exp


def _check_stream(nodes, order, tris, n_tris, second_child=True):
    """Structural validity of the pre-order skip-linked stream (what the kernel's loop relies on)."""
    LEAF = 0x80000000
    n = len(nodes)
    f = nodes.view(np.float32)
    seen = np.zeros(n_tris, bool)
    assert sorted(order.tolist()) == list(range(n_tris))
    tmin = tris.reshape(-1, 3, 3).min(axis=1)
    tmax = tris.reshape(-1, 3, 3).max(axis=1)

    def walk(i):
        """returns (next index after subtree, bmin, bmax of everything below)"""
        info, link = int(nodes[i, 7]), int(nodes[i, 3])
        lo, hi = f[i, 0:3], f[i, 4:7]
        if info & LEAF:
            first = info & ~LEAF
            assert link >= 1 and first + link <= n_tris
            ids = order[first:first + link]
            assert not seen[ids].any()
            seen[ids] = True
            assert np.array_equal(lo, tmin[ids].min(axis=0)) and np.array_equal(hi, tmax[ids].max(axis=0))
            return i + 1, lo, hi
        assert i + 1 < link <= n
        nxt, lo1, hi1 = walk(i + 1)
        # an inner node of the library's stream names its second child (the shaft cut descends by it); the
        # reference-tree stream leaves the word zero
        assert info == (nxt if second_child else 0)
        nxt2, lo2, hi2 = walk(nxt)
        assert nxt2 == link
        assert np.array_equal(lo, np.minimum(lo1, lo2)) and np.array_equal(hi, np.maximum(hi1, hi2))
        return link, lo, hi

    import sys
    sys.setrecursionlimit(10000)
    end, _, _ = walk(0)
    assert end == n and seen.all()


@pytest.mark.parametrize("accel,leaf_max", [(0, 0), (0, 1), (0, 8), (1, 0)])
def test_traversal_stream_is_well_formed(rtx, samples_half, accel, leaf_max):
    tris, rgb = rtx.default_primitives([model("big_bunny.obj")])
    with rtx.Scene(32, 32, tris, rgb, samples_half[:64], accel=accel, leaf_max=leaf_max, tie_rank=None) as s:
        nodes, order = s.nodes()
        info = s.info()
        _check_stream(nodes, order, tris, len(tris))
        if accel == 1:
            assert info["n_nodes"] == 1 and info["max_leaf_tris"] == len(tris)
        else:
            assert info["max_leaf_tris"] <= (leaf_max or 4)
            assert info["n_nodes"] == 2 * info["n_leaves"] - 1


def test_primary_stream_is_the_same_tree_nearest_child_first(rtx, samples_half):
    """The primary rays walk a stream of their own (scene_prep.h: PreparedScene::primary_nodes): well formed, the same
    records as the shadow rays' stream (boxes and leaves, as a multiset), and at every inner node the child whose box centre
    lies nearer the eye comes first.  A brute-force scene has none: its one stream serves both."""
    tris, rgb = rtx.default_primitives([model("big_bunny.obj")])
    LEAF = 0x80000000
    with rtx.Scene(32, 32, tris, rgb, samples_half[:64]) as s:
        nodes, order = s.nodes()
        prim, own = s.primary_nodes()
        assert own and prim.shape == nodes.shape
        _check_stream(prim, order, tris, len(tris))
        # the same boxes and leaf contents, in another order (links and second-child words differ)
        key = lambda nd: sorted((r[0], r[1], r[2], r[4], r[5], r[6]) + ((r[7], r[3]) if r[7] & LEAF else (0, 0)) for r in nd.tolist())
        assert key(nodes) == key(prim)
        assert np.array_equal(prim[:2, [0, 1, 2, 4, 5, 6]], nodes[:2, [0, 1, 2, 4, 5, 6]])     # the root and the ground's leaf stay in front
        eye = np.array([0.0, 100.0, 200.0])           # main.rs:353 (the default scene's camera)
        f = prim.view(np.float32)
        centre = 0.5 * f[:, 0:3].astype(np.float64) + 0.5 * f[:, 4:7].astype(np.float64)
        d2 = ((centre - eye) ** 2).sum(axis=1)
        inner = [i for i in range(2, len(prim)) if not prim[i, 7] & LEAF]     # (records 0, 1: the root and the ground's leaf)
        assert len(inner) > 1000
        for i in inner:
            assert d2[i + 1] <= d2[int(prim[i, 7])]
    with rtx.Scene(32, 32, tris[:40], rgb[:40], samples_half[:64], accel=1) as s:
        nodes, _ = s.nodes()
        prim, own = s.primary_nodes()
        assert not own and np.array_equal(prim, nodes)


def test_stream_degenerate_inputs(rtx, samples_half):
    """single triangle; many coincident triangles (no centroid spread: split by list position)."""
    one = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32)
    with rtx.Scene(8, 8, one, np.ones((1, 3), np.float32), samples_half[:64]) as s:
        nodes, order = s.nodes()
        _check_stream(nodes, order, one, 1)
    same = np.repeat(one, 37, axis=0)
    with rtx.Scene(8, 8, same, np.ones((37, 3), np.float32), samples_half[:64]) as s:
        nodes, order = s.nodes()
        _check_stream(nodes, order, same, 37)
        assert s.info()["max_leaf_tris"] <= 4


def test_png_writer_roundtrip(rtx, tmp_path):
    rng = np.random.default_rng(3)
    for shape in ((5, 7, 3), (300, 411, 3)):      # second one spans several stored-deflate blocks
        img = rng.integers(0, 256, size=shape, dtype=np.uint8)
        p = str(tmp_path / "o.png")
        rtx.write_png(p, img)
        back = np.asarray(Image.open(p))
        assert back.shape == img.shape and np.array_equal(back, img)


def test_reference_tree_stream(rtx, orc, samples_half):
    """The stream of the reference's own tree (used for zero-component rays): 2n-1 records, well formed,
    its leaves in the oracle tree's left-to-right order, and ranks derived from it when tie_rank is NULL."""
    tris, rgb = rtx.default_primitives([model("big_bunny.obj")])
    osc = orc.Scene(8, 8, tris, rgb, samples_half[:16])
    with rtx.Scene(32, 32, tris, rgb, samples_half[:64]) as s:
        info = s.info()
        assert info["n_ref_nodes"] == 2 * len(tris) - 1 == osc.node_count()
        ref = s.ref_nodes()
        _, order = s.nodes()
        _check_stream(ref, order, tris, len(tris), second_child=False)
        leaves = ref[(ref[:, 7] & 0x80000000) != 0]
        assert (leaves[:, 3] == 1).all()
        assert np.array_equal(order[leaves[:, 7] & 0x7FFFFFFF], osc.leaf_order())
    with rtx.Scene(32, 32, tris, rgb, samples_half[:64], reference_tree=rtx.REFTREE_NEVER) as s:
        assert s.info()["n_ref_nodes"] == 0 and len(s.ref_nodes()) == 0
    with rtx.Scene(32, 32, tris[:50], rgb[:50], samples_half[:64], tie_rank=None) as s:
        assert s.info()["n_ref_nodes"] == 0          # tie_rank=None: index order, no reference tree


def test_synthetic_mesh_generator(rtx):
    """rtxh_synthetic_mesh against an independent numpy evaluation of the same recipe (SURVEY 8(d), configs[4])."""
    def splitmix_f32(seed, n):
        M = (1 << 64) - 1
        out, s = np.empty(n, np.float32), seed
        for i in range(n):
            s = (s + 0x9E3779B97F4A7C15) & M
            z = s
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
            z ^= z >> 31
            out[i] = np.float32(z >> 40) * np.float32(1.0 / 16777216.0)
        return out
    F = np.float32
    n = 400
    f = splitmix_f32(rtx.SYNTHETIC_SEED, 12 * n).reshape(n, 12)
    lo, hi = np.array([-92.4, 32.7, -60.5], F), np.array([59.7, 183.4, 57.6], F)
    c = lo + f[:, 0:3] * (hi - lo)
    v = np.tile(c, (1, 3)) + (F(2.0) * f[:, 3:12] - F(1.0))
    t = rtx.synthetic_mesh(n)
    assert np.array_equal(t, v.astype(F))
    assert (t.reshape(n, 3, 3).min(axis=(0, 1)) >= lo - 1).all() and (t.reshape(n, 3, 3).max(axis=(0, 1)) <= hi + 1).all()
    tris, rgb = rtx.synthetic_primitives(20000)
    assert tris.shape == (20001, 9) and tris[-1].tolist() == list(rtx.GROUND_TRI) and rgb[-1].tolist() == [0.5, 0.5, 0.5]
    with rtx.Scene(16, 16, tris, rgb, rtx.gen_samples(n_pairs=64), tie_rank=None) as s:
        nodes, order = s.nodes()
        _check_stream(nodes, order, tris, len(tris))


def test_global_triangles_sit_in_the_first_leaf(rtx, samples_half):
    """A triangle whose own box is about as large as the scene's (the ground of main()) is tested by every walk up
    front: stream = root, the leaf of those triangles (records 0 .. n_global-1), the tree proper."""
    LEAF = 0x80000000
    tris, rgb = rtx.default_primitives([model("big_bunny.obj")])
    with rtx.Scene(32, 32, tris, rgb, samples_half[:64], tie_rank=None) as s:
        info = s.info()
        nodes, order = s.nodes()
        assert info["n_global"] == 1 and int(order[0]) == len(tris) - 1            # the ground is the last primitive
        assert int(nodes[0, 7]) == 2 and int(nodes[0, 3]) == info["n_nodes"]        # root: inner (second child: the tree proper), skips to the end
        assert int(nodes[1, 7]) == LEAF | 0 and int(nodes[1, 3]) == 1               # its first child: the global leaf
        assert not int(nodes[2, 7]) & LEAF                                          # then the root of the tree proper
        _check_stream(nodes, order, tris, len(tris))
    mesh = tris[:-1]                                                                # no ground: nothing is global
    with rtx.Scene(32, 32, mesh, rgb[:-1], samples_half[:64], tie_rank=None) as s:
        assert s.info()["n_global"] == 0
    walls = np.array([[-9, -9, -5, 9, -9, -5, 0, 9, -5], [-9, -9, -6, 9, -9, -6, 0, 9, -6]], np.float32)
    with rtx.Scene(8, 8, walls, np.ones((2, 3), np.float32), samples_half[:64], tie_rank=None) as s:   # all global
        info = s.info()
        nodes, order = s.nodes()
        assert info["n_global"] == 2 and info["n_nodes"] == 1 and int(nodes[0, 3]) == 2


def test_scatter_tiles_places_every_share(rtx):
    """rtxh_scatter_tiles — the gather loop of rtx_render_frame, factored out: for any frame height, tile height and
    share count, scattering every share's packed rows rebuilds the frame, each row written exactly once."""
    rng = np.random.default_rng(11)
    for H, W, tile_rows, world in ((117, 5, 8, 3), (64, 3, 8, 8), (7, 2, 16, 2), (100, 4, 1, 7), (33, 1, 5, 1), (9, 2, 4, 6)):
        full = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        frame = np.full_like(full, 0xEE)
        written = np.zeros(H, np.int32)
        for rank in range(world):
            rows = [r for t in range(rank, (H + tile_rows - 1) // tile_rows, world)
                    for r in range(t * tile_rows, min(H, (t + 1) * tile_rows))]
            assert len(rows) == rtx.rtx.tiles_rows_of(H, rank, world, tile_rows)
            packed = full[rows] if rows else np.zeros((0, W, 3), np.uint8)
            rtx.scatter_tiles(frame, packed, rank, world, tile_rows)
            written[rows] += 1
        assert (written == 1).all() and np.array_equal(frame, full), (H, tile_rows, world)
    lib = rtx.rtx._lib
    buf = np.zeros((4, 4, 3), np.uint8)
    assert lib.rtxh_scatter_tiles(None, 4, 4, buf.ctypes.data, 0, 1, 8) == rtx.ERR_BAD_ARG
    assert lib.rtxh_scatter_tiles(buf.ctypes.data, 4, 4, buf.ctypes.data, 0, 0, 8) == rtx.ERR_BAD_ARG
    assert lib.rtxh_scatter_tiles(buf.ctypes.data, 4, 4, buf.ctypes.data, 0, 1, 0) == rtx.ERR_BAD_ARG


def test_render_frame_rejects_row_numbers_beyond_32_bits(rtx, samples_half):
    """n_devices x tile_rows is a 32-bit row stride inside the kernels: rtx_render_frame refuses what would wrap
    (before it looks at any device, so this runs on a CPU-only box)."""
    import ctypes as C
    with rtx.default_scene([model("bunny.obj")], 16, 16, samples_half[:64]) as s:
        out = np.zeros((16, 16, 3), np.uint8)
        devs = (C.c_int * 2)(0, 1)
        rc = rtx.rtx._lib.rtx_render_frame(s.handle, devs, 2, 0x80000000, out.ctypes.data, None)
        assert rc == rtx.ERR_BAD_ARG
        rc = rtx.rtx._lib.rtx_render_frame(s.handle, devs, 0, 8, out.ctypes.data, None)
        assert rc == rtx.ERR_BAD_ARG
