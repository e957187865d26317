"""Cross-check the C oracle against the independent numpy-float32 restatement
(tests/np_ref.py), and the oracle's three closest-hit strategies against each
other.  CPU only.

Why the strategies must agree: the reference visits every BVH node whose slab
test passes (no pruning, bvh.rs:86-132) and a leaf counts only if its own AABB
passes first (:52).  IEEE subtraction and division are monotone, so a ray that
passes a leaf's slab test passes every enclosing box's: the faithful BVH result
equals "min over all triangles that pass their own AABB and Möller–Trumbore with
t >= 1" (LEAFBOX), whatever the tree's shape.  That identity is what lets the
GPU path use a different acceleration structure; these tests measure it.
"""
import numpy as np

import np_ref

F = np.float32


def _scene_args(orc):
    return dict(eye=orc.EYE, look_at=orc.LOOK_AT, up=orc.UP, distance=orc.DISTANCE, light9=orc.LIGHT_TRI)


def test_numpy_vs_oracle_full_image_const_table(orc, samples_half):
    """big_bunny 128x128 with T == 0.5: all 100 light samples coincide, so one shadow ray per hit
    pixel reproduces the whole shaded image in numpy.  Linear colours must be
    bit-identical; bytes within 1 (numpy's powf is not libm's)."""
    W = H = 128
    tris, rgb = orc.default_primitives(["big_bunny.obj"])
    s = orc.Scene(W, H, tris, rgb, samples_half)
    img, st, tri, lin = s.render_rows(mode=orc.MODE_BVH, want_tri=True, want_lin=True)
    py, px = np.mgrid[0:H, 0:W]
    # one light sample, then replicate the 100-step accumulation exactly: avg += c*lnd/100, 100 times
    lin1, ti = np_ref.render_pixels(px.ravel(), py.ravel(), W, H, tris, rgb, samples_half[:4], nb_light=1,
                                    **_scene_args(orc))
    # nb_light=1 divides by 1.0: lin1 = colour*lnd if lit else 0
    acc = np.zeros_like(lin1)
    contrib = lin1 / F(100.0)
    for _ in range(100):
        acc = acc + contrib
    acc = acc.reshape(H, W, 3)
    ti = ti.reshape(H, W)
    assert np.array_equal(np.where(ti < 0, 0xFFFFFFFF, ti).astype(np.uint32), tri)
    assert st["primary_hits"] == (ti >= 0).sum()
    assert np.array_equal(acc, lin), "linear colours differ: %d px" % (acc != lin).any(axis=2).sum()
    assert np.abs(np_ref.to_rgb8(acc).astype(int) - img.astype(int)).max() <= 1
    assert st["assert_tmin_gt_tmax"] == 0 and st["nonfinite_t"] == 0


def test_numpy_vs_oracle_seeded_pixels(orc, samples_seeded):
    """Seeded table, full 100-sample loop, on pixels chosen along the silhouette and the shadow edge."""
    W = H = 256
    tris, rgb = orc.default_primitives(["big_bunny.obj"])
    s = orc.Scene(W, H, tris, rgb, samples_seeded)
    rng = np.random.default_rng(7)
    # a coarse hit-class map from one cheap primary-only pass of the oracle
    s0 = orc.Scene(W, H, tris, rgb, samples_seeded, nb_light_sample=0)
    _, _, tri0 = s0.render_rows(mode=orc.MODE_BVH, want_tri=True)
    mesh = (tri0 != 0xFFFFFFFF) & (tri0 != len(tris) - 1)
    edge = mesh ^ np.roll(mesh, 1, axis=1)
    ys, xs = np.nonzero(edge)
    pick = rng.choice(len(ys), size=24, replace=False)
    pts = [(int(xs[i]), int(ys[i])) for i in pick]
    gy, gx = np.nonzero(tri0 == len(tris) - 1)
    pick = rng.choice(len(gy), size=24, replace=False)
    pts += [(int(gx[i]), int(gy[i])) for i in pick]
    pts += [(0, 0), (W - 1, H - 1), (128, 10)]
    px = np.array([p[0] for p in pts], np.uint32)
    py = np.array([p[1] for p in pts], np.uint32)
    lin_np, ti = np_ref.render_pixels(px, py, W, H, tris, rgb, samples_seeded, **_scene_args(orc))
    for k, (x, y) in enumerate(pts):
        lin_c = s.render_pixel(x, y, orc.MODE_BVH)
        assert np.array_equal(lin_c, lin_np[k]), (x, y, lin_c, lin_np[k])


def test_bvh_equals_leafbox_and_counts(orc, samples_seeded):
    """Faithful BVH == brute force gated by each leaf's own AABB, pixel for pixel, and they run the
    same number of triangle tests (identical candidate sets).  BRUTE (no gate) is compared too."""
    W = H = 96
    tris, rgb = orc.default_primitives(["big_bunny.obj"])
    s = orc.Scene(W, H, tris, rgb, samples_seeded)
    rows = (40, 24)   # rows 40..63: bunny body, silhouette, ground with shadow
    img_b, st_b, tri_b, lin_b = s.render_rows(*rows, mode=orc.MODE_BVH, want_tri=True, want_lin=True)
    img_l, st_l, tri_l, lin_l = s.render_rows(*rows, mode=orc.MODE_LEAFBOX, want_tri=True, want_lin=True)
    assert np.array_equal(tri_b, tri_l) and np.array_equal(lin_b, lin_l) and np.array_equal(img_b, img_l)
    assert st_b["tri_tests"] == st_l["tri_tests"]
    assert st_b["exact_ties"] == 0 and st_b["assert_tmin_gt_tmax"] == 0 and st_b["nonfinite_t"] == 0
    img_r, st_r, tri_r = s.render_rows(*rows, mode=orc.MODE_BRUTE, want_tri=True)
    assert np.array_equal(tri_b, tri_r)
    assert np.abs(img_b.astype(int) - img_r.astype(int)).max() <= 1


def test_primary_hit_mask_1080p_matches_survey_probe(orc, samples_half):
    """SURVEY §8(d) probe (independent numpy run by the surveyor, brute force, jitter 0.5):
    big_bunny 1080p -> 37,005 mesh px inside x 793..1056 / y 427..669, 999,919 ground px, first
    ground row 543; bunny.obj -> 0 mesh px, 1,022,304 ground px."""
    tris, rgb = orc.default_primitives(["big_bunny.obj"])
    s = orc.Scene(1920, 1080, tris, rgb, samples_half, nb_light_sample=0)
    _, st, tri = s.render_rows(mode=orc.MODE_BVH, want_tri=True)
    ground = tri == len(tris) - 1
    mesh = (tri != 0xFFFFFFFF) & ~ground
    assert mesh.sum() == 37005 and ground.sum() == 999919 and st["mesh_hits"] == 37005
    ys, xs = np.nonzero(mesh)
    # the probe's window is the projected mesh bbox +-2 px; every hit lies inside it
    assert 793 <= xs.min() and xs.max() <= 1056 and 427 <= ys.min() and ys.max() <= 669
    assert (xs.min(), xs.max(), ys.min(), ys.max()) == (796, 1053, 429, 665)
    assert np.nonzero(ground.any(axis=1))[0].min() == 543
    tris2, rgb2 = orc.default_primitives(["bunny.obj"])
    s2 = orc.Scene(1920, 1080, tris2, rgb2, samples_half, nb_light_sample=0)
    _, st2, tri2 = s2.render_rows(mode=orc.MODE_BVH, want_tri=True)
    assert st2["mesh_hits"] == 0 and (tri2 == len(tris2) - 1).sum() == 1022304


def test_tree_dependence_is_confined_to_negative_zero_directions(orc):
    """Where the faithful BVH and the leaf-gated brute force DISAGREE.  With a -0.0 direction component the slab
    test of an enclosing box divides by -0.0 and rejects through +-inf, while a flat leaf box at the origin's
    coordinate produces NaNs that the comparisons ignore — the reference's result then depends on its tree.
    This scene (integer lattice, eye on the lattice, all-zero jitter) produces such rays on the centre row;
    every pixel on which the two strategies differ must carry a -0.0 component, and rays with a +0.0
    component must agree (the GPU path relies on exactly this split, rtx_traverse.hpp)."""
    import ctypes as C
    rng = np.random.default_rng(1)
    v = rng.integers(-6, 7, size=(300, 3, 3)).astype(np.float32)
    v[..., 2] -= 14.0
    keep = np.linalg.norm(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]), axis=1) > 1e-3
    tris = v[keep].reshape(-1, 9)
    rgb = np.ones((len(tris), 3), np.float32)
    T = np.zeros((4096, 2), np.float32)
    kw = dict(eye=(0.0, 0.0, 0.0), look_at=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), distance=24.0,
              light_tri=(-2.0, 9.0, -3.0, 2.0, 9.0, -3.0, 0.0, 9.0, 1.0), nb_light_sample=0)
    W = H = 40
    s = orc.Scene(W, H, tris, rgb, T, **kw)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    cam = [np.zeros(3, np.float32) for _ in range(3)]
    orc.lib().orc_camera_new(fp(orc.f3(kw["eye"])), fp(orc.f3(kw["look_at"])), fp(orc.f3(kw["up"])), *map(fp, cam))
    n_neg = n_pos = n_diff = 0
    for py in range(H):
        for px in range(W):
            o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
            orc.lib().orc_create_ray(px, py, 0, W, H, fp(orc.f3(kw["eye"])), fp(cam[0]), fp(cam[1]), fp(cam[2]),
                                     24.0, fp(T), len(T), fp(o), fp(d))
            neg_zero = bool(((d == 0) & np.signbit(d)).any())
            pos_zero = bool(((d == 0) & ~np.signbit(d)).any())
            a = s.closest_hit(o, d, orc.MODE_BVH)
            b = s.closest_hit(o, d, orc.MODE_LEAFBOX)
            same = (a.hit, a.t if a.hit else 0.0) == (b.hit, b.t if b.hit else 0.0)
            n_neg += neg_zero
            n_pos += pos_zero and not neg_zero
            if not same:
                n_diff += 1
                assert neg_zero, (px, py, d)
    assert n_neg >= W // 2 - 1 and n_pos >= H // 2      # both kinds of ray are present in the scene
    print("rays with -0.0: %d, with only +0.0: %d, BVH != LEAFBOX on %d" % (n_neg, n_pos, n_diff))
