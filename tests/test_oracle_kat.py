"""Known-answer tests that pin the CPU oracle.

The reference has no tests, so every expected value here is
derived by hand from the cited reference lines (path:line under
/root/reference/) or recomputed independently in float64 inside the test.
"""
import ctypes as C
import math

import numpy as np

F = np.float32


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def v(*x):
    return np.array(x, dtype=F)


# ------------------------------------------------------------------ camera
def test_camera_basis_default_scene(orc):
    """camera.rs:17-35 with the literals of src/main.rs:353-356."""
    u, vv, w = v(0, 0, 0), v(0, 0, 0), v(0, 0, 0)
    orc.lib().orc_camera_new(_fp(v(*orc.EYE)), _fp(v(*orc.LOOK_AT)), _fp(v(*orc.UP)), _fp(u), _fp(vv), _fp(w))
    # independent float64 derivation
    e = np.array(orc.EYE, float) - np.array(orc.LOOK_AT, float)
    w64 = e / np.linalg.norm(e)
    u64 = np.cross([0, 1, 0], w64)
    u64 /= np.linalg.norm(u64)
    v64 = np.cross(u64, w64)
    v64 /= np.linalg.norm(v64)
    assert np.allclose(u, u64, atol=2e-7) and np.allclose(vv, v64, atol=2e-7) and np.allclose(w, w64, atol=2e-7)
    # SURVEY §8(a) A2 probe values; v points DOWN (image y grows downward)
    assert u.tolist() == [1.0, 0.0, 0.0]
    assert abs(vv[1] - (-0.9999995)) < 1e-6 and abs(vv[2] - 9.980035e-4) < 1e-9
    assert abs(w[1] - 9.9800341e-4) < 1e-9 and abs(w[2] - 0.99999946) < 1e-6
    assert vv[0] == 0.0 and w[0] == 0.0


# ------------------------------------------------------------------ triangle
UNIT = (v(0, 0, 0), v(1, 0, 0), v(0, 1, 0))


def tri_new(orc, v0, v1, v2):
    e1, e2, n = v(0, 0, 0), v(0, 0, 0), v(0, 0, 0)
    orc.lib().orc_triangle_new(_fp(v0), _fp(v1), _fp(v2), _fp(e1), _fp(e2), _fp(n))
    return e1, e2, n


def mt(orc, v0, e1, e2, o, d):
    t = C.c_float(float("nan"))
    some = orc.lib().orc_triangle_intersect(_fp(v0), _fp(e1), _fp(e2), _fp(o), _fp(d), C.byref(t))
    return (True, t.value) if some else (False, None)


def test_triangle_new(orc):
    """triangle.rs:22-34: e1 = v1-v0, e2 = v2-v0, normal = normalize(e1 x e2)."""
    e1, e2, n = tri_new(orc, *UNIT)
    assert e1.tolist() == [1, 0, 0] and e2.tolist() == [0, 1, 0] and n.tolist() == [0, 0, 1]
    e1, e2, n = tri_new(orc, v(1, 2, 3), v(4, 2, 3), v(1, 2, 7))  # e1=(3,0,0) e2=(0,0,4): cross=(0,-12,0)
    assert e1.tolist() == [3, 0, 0] and e2.tolist() == [0, 0, 4] and n.tolist() == [0, -1, 0]


def test_moller_trumbore_branches(orc):
    """triangle.rs:66-94, every branch, on the unit triangle (values derived by hand)."""
    v0 = UNIT[0]
    e1, e2, _ = tri_new(orc, *UNIT)
    down = v(0, 0, -1)
    assert mt(orc, v0, e1, e2, v(0.25, 0.25, 5), down) == (True, 5.0)       # interior
    assert mt(orc, v0, e1, e2, v(0.0, 0.5, 5), down) == (True, 5.0)         # u == 0 is inside (u < 0.0 false)  :80
    assert mt(orc, v0, e1, e2, v(0.5, 0.0, 5), down) == (True, 5.0)         # v == 0 is inside                   :86
    assert mt(orc, v0, e1, e2, v(0.5, 0.5, 5), down) == (True, 5.0)         # u+v == 1 is inside (u+v > 1 false) :86
    assert mt(orc, v0, e1, e2, v(1.0, 0.0, 5), down) == (True, 5.0)         # u == 1 is inside                   :80
    assert mt(orc, v0, e1, e2, v(0.75, 0.5, 5), down)[0] is False           # u+v = 1.25
    assert mt(orc, v0, e1, e2, v(-0.125, 0.5, 5), down)[0] is False         # u < 0
    assert mt(orc, v0, e1, e2, v(1.125, 0.0, 5), down)[0] is False          # u > 1
    assert mt(orc, v0, e1, e2, v(0.5, -0.125, 5), down)[0] is False         # v < 0
    assert mt(orc, v0, e1, e2, v(0.25, 0.25, 5), v(1, 0, 0))[0] is False    # parallel: det == 0                 :73
    # no sign test on t: a triangle behind the origin returns a negative distance      :92-93
    assert mt(orc, v0, e1, e2, v(0.25, 0.25, -5), down) == (True, -5.0)
    # two-sided (no back-face culling): det = -1
    assert mt(orc, v0, e1, e2, v(0.25, 0.25, -5), v(0, 0, 1)) == (True, 5.0)
    # absolute epsilon on det: e1 = (k,0,0) gives det = k for this ray
    assert mt(orc, v0, v(9e-6, 0, 0), e2, v(0, 0.25, 5), down)[0] is False  # |det| < 1e-5
    assert mt(orc, v0, v(-9e-6, 0, 0), e2, v(0, 0.25, 5), down)[0] is False
    some, t = mt(orc, v0, v(2e-5, 0, 0), e2, v(0, 0.25, 5), down)           # |det| > 1e-5 -> tested normally
    assert some and t == 5.0


def test_light_sample_keeps_reference_formula(orc):
    """triangle.rs:113-127 with c3 = v*sqrt(u): get_sample(.25,.25) on the light of main.rs:337-343.
    us = vs = .5, c1 = .5, c2 = .25, c3 = .125 (sum .875, not 1)."""
    lt = np.array(orc.LIGHT_TRI, dtype=F)
    out = v(0, 0, 0)
    orc.lib().orc_triangle_get_sample(_fp(lt[0:3]), _fp(lt[3:6]), _fp(lt[6:9]), 0.25, 0.25, _fp(out))
    assert out.tolist() == [-2.5, 262.5, -7.5]
    orc.lib().orc_triangle_get_sample(_fp(lt[0:3]), _fp(lt[3:6]), _fp(lt[6:9]), 1.0, 1.0, _fp(out))
    assert out.tolist() == [0.0, 300.0, 0.0]      # c1=0,c2=0,c3=1 -> v2
    orc.lib().orc_triangle_get_sample(_fp(lt[0:3]), _fp(lt[3:6]), _fp(lt[6:9]), 0.0, 0.7, _fp(out))
    assert out.tolist() == [-10.0, 300.0, -10.0]  # us=0 -> v0


# ------------------------------------------------------------------ bounding box
def slab(orc, bmin, bmax, o, d):
    t = C.c_float(float("nan"))
    af = C.c_int(0)
    some = orc.lib().orc_bbox_intersect(_fp(bmin), _fp(bmax), _fp(o), _fp(d), C.byref(t), C.byref(af))
    return (bool(some), t.value if some else None, af.value)


def test_slab_branches(orc):
    """bounding_box.rs:99-181 (hand-derived, incl. the IEEE corner cases the Rust code inherits)."""
    lo, hi = v(-1, -1, -1), v(1, 1, 1)
    assert slab(orc, lo, hi, v(0, 0, 0), v(0, 0, -1)) == (True, 0.0, 0)          # origin strictly inside -> Some(0) :104-108
    assert slab(orc, lo, hi, v(0, 0, 5), v(0, 0, -1)) == (True, 4.0, 0)          # x,y give (-inf,+inf); z: (1-5)/-1=4
    assert slab(orc, lo, hi, v(0, 0, 5), v(0, 0, 1))[0] is False                  # box behind: tmax=-4, tmax > 0 false
    assert slab(orc, lo, hi, v(3, 0, 5), v(0, 0, -1))[0] is False                 # x slab: tmax=-inf -> tzmin > tmax
    assert slab(orc, lo, hi, v(0, 0, -5), v(0, 0, 1)) == (True, 4.0, 0)          # positive-direction branch
    # -0.0 >= 0.0 is true, so the "positive" formulas divide by -0: tmin=+inf -> miss, although the ray hits
    assert slab(orc, lo, hi, v(0, 0, 5), v(-0.0, 0, -1))[0] is False
    # NaN in tymax (0/0) is ignored by the comparisons: a ray sliding exactly on the top face hits
    assert slab(orc, lo, hi, v(-5, 1, 0), v(1, 0, 0)) == (True, 4.0, 0)
    assert slab(orc, lo, hi, v(-5, 1.5, 0), v(1, 0, 0))[0] is False
    # oblique, all three axes active, negative-direction branches: o=(4,4,4), d=-(1,1,1)/sqrt3: t = 3*sqrt3
    d = v(-1, -1, -1) / F(math.sqrt(3))
    some, t, af = slab(orc, lo, hi, v(4, 4, 4), d)
    assert some and af == 0 and abs(t - 3 * math.sqrt(3)) < 1e-5


def test_slab_flat_ground_box(orc):
    """The ground triangle's AABB is flat (min.y == max.y == 0), main.rs:102-111 + triangle.rs:45-56."""
    g = np.array(orc.GROUND_TRI, dtype=F)
    bmin, bmax = v(0, 0, 0), v(0, 0, 0)
    orc.lib().orc_triangle_bbox(_fp(g[0:3]), _fp(g[3:6]), _fp(g[6:9]), _fp(bmin), _fp(bmax))
    assert bmin.tolist() == [-10000, 0, -10000] and bmax.tolist() == [10000, 0, 10000]
    assert slab(orc, bmin, bmax, v(0, 100, 200), v(0, -1, 0)) == (True, 100.0, 0)
    # a ray that starts exactly on the plane and goes up: tmin = tmax = 0, "tmax > 0" fails -> miss
    assert slab(orc, bmin, bmax, v(0, 0, 0), v(0, 1, 0))[0] is False
    # from just below the plane it passes (the leaf then rejects the hit through t < 1.0)
    assert slab(orc, bmin, bmax, v(0, -1e-3, 0), v(0, 1, 0))[0] is True


# ------------------------------------------------------------------ colour
def test_gamma_quantise(orc):
    """color.rs:10-13,28-33: (x.powf(1/2.2) * 255.0) as u8, truncating."""
    def q(x):
        out = (C.c_uint8 * 3)()
        c = v(x, x, x)
        orc.lib().orc_color_to_rgb8(_fp(c), out)
        assert out[0] == out[1] == out[2]
        return out[0]
    assert q(0.0) == 0 and q(1.0) == 255
    assert q(0.5) == 186      # 0.5^(1/2.2) = 0.72974 -> 186.08
    assert q(0.25) == 135     # 0.53252 -> 135.79 (truncated, not rounded to 136)
    assert q(1.5) == 255 and q(float("nan")) == 0   # Rust `as u8` saturates, NaN -> 0
    # monotone over a dense sweep, each value within 1e-3 of the float64 formula before truncation
    xs = np.linspace(0, 1, 4001, dtype=F)
    qs = np.array([q(float(x)) for x in xs])
    assert (np.diff(qs) >= 0).all()
    ref = np.floor(np.minimum(xs.astype(float) ** (1 / 2.2) * 255.0, 255.0))
    assert np.abs(qs - ref).max() <= 1 and (qs != ref).sum() <= 8


# ------------------------------------------------------------------ primary rays
def test_create_rays_index_is_px_major(orc):
    """main.rs:160-167: jitter index = (px*W + py + i) % len — px-major although the image is W wide."""
    W, H, n = 256, 128, 1000
    cam = [v(1, 0, 0), v(0, -1, 0), v(0, 0, 1)]
    eye = v(0, 0, 0)
    px, py = 3, 5
    k = (px * W + py) % n
    o, d0, d1 = v(0, 0, 0), v(0, 0, 0), v(0, 0, 0)
    T = np.zeros((n, 2), dtype=F)
    orc.lib().orc_create_ray(px, py, 0, W, H, _fp(eye), _fp(cam[0]), _fp(cam[1]), _fp(cam[2]), 288.0, _fp(T), n, _fp(o), _fp(d0))
    T[k] = (0.25, 0.75)
    orc.lib().orc_create_ray(px, py, 0, W, H, _fp(eye), _fp(cam[0]), _fp(cam[1]), _fp(cam[2]), 288.0, _fp(T), n, _fp(o), _fp(d1))
    assert not np.array_equal(d0, d1)
    a, b = px - W / 2 + 0.25, py - H / 2 + 0.75
    ref = np.array([a, -b, -288.0])
    ref /= np.linalg.norm(ref)
    assert np.allclose(d1, ref, atol=2e-7)
    T[k] = 0
    T[(py * W + px) % n] = (0.25, 0.75)          # the row-major index must NOT be the one used
    orc.lib().orc_create_ray(px, py, 0, W, H, _fp(eye), _fp(cam[0]), _fp(cam[1]), _fp(cam[2]), 288.0, _fp(T), n, _fp(o), _fp(d1))
    assert np.array_equal(d0, d1)


def test_sample_table_generator(orc):
    """splitmix64(seed) high-32 >> 8, * 2^-24; first outputs recomputed in Python ints."""
    def splitmix(seed, n):
        out, s, M = [], seed, (1 << 64) - 1
        for _ in range(n):
            s = (s + 0x9E3779B97F4A7C15) & M
            z = s
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
            z ^= z >> 31
            out.append(((z >> 32) >> 8) / 16777216.0)
        return out
    T = orc.gen_samples(orc.SEED, 64)
    assert T.shape == (64, 2)
    assert T.reshape(-1).tolist() == splitmix(orc.SEED, 128)
    assert (T >= 0).all() and (T < 1).all()
    big = orc.gen_samples(orc.SEED, 100000)
    assert abs(big.mean() - 0.5) < 5e-3


# ------------------------------------------------------------------ BVH build / traversal rules
def _tri_sized(s, z):
    # extent (s, s, 0) at depth z, facing +z; contains the point (0.1, 0.1)
    return [0, 0, z, s, 0, z, 0, s, z]


def _mini_scene(orc, tris, samples, **kw):
    tris = np.array(tris, dtype=F)
    rgb = np.ones((len(tris), 3), dtype=F)
    return orc.Scene(8, 8, tris, rgb, samples, **kw)


def test_bvh_build_order_small(orc, samples_half):
    """bvh.rs:173-226 on 4 primitives with extents 1,2,4,8 (hand-simulated):
    pop P3 -> closest P2 -> N1(l=P3,r=P2); pop P1 -> closest P0 -> N2(l=P1,r=P0);
    level 2: pop N2, closest N1 -> root(l=N2,r=N1).  Leaf order P1,P0,P3,P2."""
    s = _mini_scene(orc, [_tri_sized(1, 0), _tri_sized(2, -1), _tri_sized(4, -2), _tri_sized(8, -3)], samples_half[:16])
    assert s.node_count() == 7 and s.depth() == 3
    assert s.leaf_order().tolist() == [1, 0, 3, 2]


def test_bvh_odd_leftover_and_tie_rule(orc, samples_half):
    """Three coincident triangles: equal extents -> first minimum wins (strict <, bvh.rs:200), the
    leftover node is appended after the merged ones (:212-215), and on equal distances the RIGHT
    child is returned (:123-130).  Hand simulation: root(l=P1, r=N(l=P2, r=P0)) -> P0 wins."""
    t = _tri_sized(1, 0)
    s = _mini_scene(orc, [t, t, t], samples_half[:16])
    assert s.node_count() == 5 and s.leaf_order().tolist() == [1, 2, 0]
    h = s.closest_hit(v(0.1, 0.1, 5), v(0, 0, -1), orc.MODE_BVH)
    assert h.hit == 1 and h.t == 5.0 and h.tri == 0
    s2 = _mini_scene(orc, [t, t], samples_half[:16])   # root(l=P1, r=P0) -> P0
    assert s2.leaf_order().tolist() == [1, 0]
    assert s2.closest_hit(v(0.1, 0.1, 5), v(0, 0, -1), orc.MODE_BVH).tri == 0


def test_leaf_rejects_t_below_one(orc, samples_half):
    """bvh.rs:64-67: x < 1.0 -> None (also rejects hits behind the origin); t == 1.0 is kept;
    a nearer rejected triangle does not hide a farther valid one."""
    s = _mini_scene(orc, [_tri_sized(1, 0), _tri_sized(2, -3)], samples_half[:16])
    down = v(0, 0, -1)
    h = s.closest_hit(v(0.1, 0.1, 0.5), down)        # t=0.5 on tri 0 rejected, tri 1 at t=3.5
    assert h.hit == 1 and h.tri == 1 and h.t == 3.5
    h = s.closest_hit(v(0.1, 0.1, 1.0), down)        # t == 1.0 accepted
    assert h.hit == 1 and h.tri == 0 and h.t == 1.0
    h = s.closest_hit(v(0.1, 0.1, -3.5), down)       # both behind
    assert h.hit == 0
    h = s.closest_hit(v(0.1, 0.1, 5.0), down)        # closest of two
    assert h.hit == 1 and h.tri == 0 and h.t == 5.0 and list(h.p_hit) == [F(0.1), F(0.1), 0.0]


def test_obj_import_matches_file(orc):
    """main.rs:114-149 on the reference's own models: 2,503 v / 4,968 f each; first face checked by hand."""
    for name in ("bunny.obj", "big_bunny.obj"):
        t = orc.import_obj(orc.model_path(name))
        assert t.shape == (4968, 9)
    verts = []
    first_face = None
    with open(orc.model_path("big_bunny.obj")) as f:
        for line in f:
            tok = line.rstrip("\n").split(" ")
            if tok[0] == "v":
                verts.append([float(x) for x in tok[1:4]])
            elif tok[0] == "f" and first_face is None:
                first_face = [int(x) for x in tok[1:4]]
    exp = np.array([verts[i - 1] for i in first_face], dtype=F).reshape(9)
    assert np.array_equal(orc.import_obj(orc.model_path("big_bunny.obj"))[0], exp)
    tris, rgb = orc.default_primitives(["big_bunny.obj"])
    assert tris.shape == (4969, 9) and tris[-1].tolist() == list(orc.GROUND_TRI)   # ground LAST (main.rs:335)
    assert rgb[-1].tolist() == [0.5, 0.5, 0.5] and (rgb[:-1] == 1).all()
