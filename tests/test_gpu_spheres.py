"""GPU parity for the Sphere arm of Primitive (src/tracer/primitives/sphere.rs) through the C ABI: byte-exact
against the oracle's faithful BVH over both arms.  The reference's main() never instantiates a sphere
(gen_random_spheres is dead code, src/main.rs:42-67), so these scenes are synthetic; parity unpinned like the
rest of the oracle."""
import importlib
import os

import numpy as np
import pytest

from test_host_spheres import mixed_scene

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def rtx():
    mod = importlib.import_module("ray-tracer-rust_amd")
    assert mod.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return mod


def same_image(img, ref, what):
    diff = np.abs(img.astype(np.int16) - ref.astype(np.int16))
    assert diff.max() == 0, "%s: %d differing bytes, max %d" % (what, int((diff != 0).sum()), int(diff.max()))


KAT = dict(eye=(0.0, 0.0, 0.0), look_at=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), distance=16.0,
           light_tri=(-1.0, 50.0, -1.0, 1.0, 50.0, -1.0, 0.0, 50.0, 1.0), nb_light_sample=16)


def kat_scene():
    tris = np.array([[-30.0, -30.0, -20.0, 30.0, -30.0, -20.0, 0.0, 30.0, -20.0],
                     [-40.0, -8.0, 5.0, 40.0, -8.0, 5.0, 0.0, -8.0, -60.0]], F)
    rgb = np.array([[1, 0, 0], [0.5, 0.5, 0.5]], F)
    spheres = np.array([[0.0, 0.0, -10.0, 2.0], [4.0, 1.0, -12.0, 3.0], [0.0, 0.0, -0.2, 0.5]], F)   # last: around the eye
    srgb = np.array([[0, 1, 0], [0, 0, 1], [1, 1, 0]], F)
    kinds = np.array([0, 1, 1, 0, 1], np.uint8)        # Vec order: tri0, sph0, sph1, tri1, sph2
    return tris, rgb, spheres, srgb, kinds


@pytest.mark.parametrize("accel", [0, 1])
def test_known_answer_scene(rtx, orc, samples_seeded, accel):
    """The scene of tests/test_oracle_sphere.py: two coloured spheres over a ground triangle in front of a wall, a
    third sphere around the eye (every hit on it has t < 1.0 -> rejected by the leaf rule, bvh.rs:64-67)."""
    tris, rgb, spheres, srgb, kinds = kat_scene()
    W = H = 32
    ref, ost, otri = orc.Scene(W, H, tris, rgb, samples_seeded, spheres=spheres, sphere_rgb=srgb, kinds=kinds,
                               **KAT).render_rows(mode=orc.MODE_BVH, want_tri=True)
    assert (otri == 1).sum() > 10 and (otri == 2).sum() > 10 and (otri == 4).sum() == 0
    with rtx.Scene(W, H, tris, rgb, samples_seeded, spheres=spheres, sphere_rgb=srgb, kinds=kinds, accel=accel,
                   **KAT) as s:
        img, st = s.render_rows(stats=True)
    assert st["primary_hits"] == ost["primary_hits"]
    same_image(img, ref, "sphere KAT scene accel=%d" % accel)
    ys, xs = np.nonzero(otri == 1)                      # the lit top of sphere 0 is green
    assert img[ys.min(), xs[ys.argmin()], 1] > 0 and img[ys.min(), xs[ys.argmin()], 0] == 0


@pytest.mark.parametrize("seed,table,nb_ray", [(21, "seed", 1), (22, "zero", 1), (23, "half", 1), (24, "seed", 2)])
def test_mixed_soups(rtx, orc, samples_seeded, samples_half, seed, table, nb_ray):
    """Random triangles and spheres interleaved in one Vec<Primitive>, axis-aligned camera at the origin.  The all-zero
    sample table puts exact zeros into the primary directions of the centre row and column (soft and hard direction
    classes: exact slab test, reference-tree re-render of the tile, sphere leaves included)."""
    tris, rgb, spheres, srgb, kinds = mixed_scene(np.random.default_rng(seed), 220, 60)
    T = {"zero": np.zeros((4096, 2), F), "seed": samples_seeded, "half": samples_half}[table]
    kw = dict(eye=(0.0, 0.0, 0.0), look_at=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), distance=24.0,
              light_tri=(-2.0, 9.0, -3.0, 2.0, 9.0, -3.0, 0.0, 9.0, 1.0), nb_light_sample=24, nb_ray=nb_ray)
    W = H = 40
    ref, ost, otri = orc.Scene(W, H, tris, rgb, T, spheres=spheres, sphere_rgb=srgb, kinds=kinds,
                               **kw).render_rows(mode=orc.MODE_BVH, want_tri=True)
    assert ost["nonfinite_t"] == 0
    hit_kinds = kinds[otri[otri != 0xFFFFFFFF]]
    assert (hit_kinds == 1).sum() > 50 and (hit_kinds == 0).sum() > 50
    with rtx.Scene(W, H, tris, rgb, T, spheres=spheres, sphere_rgb=srgb, kinds=kinds, **kw) as s:
        img, st = s.render_rows(stats=True)
        if table == "zero":
            assert st["redo_tiles"] > 0
    assert st["primary_hits"] == ost["primary_hits"]
    same_image(img, ref, "mixed soup seed %d" % seed)


def test_spheres_only_and_no_reference_tree(rtx, orc, samples_seeded):
    """A scene of spheres alone, rendered without the reference tree (RTX_REFTREE_NEVER): the checker is the
    oracle's leaf-gated brute force, equal to the faithful BVH where no ray has a hard direction component and
    no exact tie occurs (asserted)."""
    _, _, spheres, srgb, _ = mixed_scene(np.random.default_rng(31), 1, 120)
    none, none_rgb = np.zeros((0, 9), F), np.zeros((0, 3), F)
    kw = dict(eye=(0.3, 0.2, 0.0), look_at=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), distance=30.0,
              light_tri=(-2.0, 9.0, -3.0, 2.0, 9.0, -3.0, 0.0, 9.0, 1.0), nb_light_sample=16)
    W, H = 48, 40
    osc = orc.Scene(W, H, none, none_rgb, samples_seeded, spheres=spheres, sphere_rgb=srgb, **kw)
    ref, ost = osc.render_rows(mode=orc.MODE_BVH)
    ref_l, ost_l = osc.render_rows(mode=orc.MODE_LEAFBOX)
    assert np.array_equal(ref, ref_l) and ost["exact_ties"] == 0
    with rtx.Scene(W, H, none, none_rgb, samples_seeded, spheres=spheres, sphere_rgb=srgb, tie_rank=None, **kw) as s:
        assert s.info()["n_ref_nodes"] == 0
        img, st = s.render_rows(stats=True)
    assert st["primary_hits"] == ost["primary_hits"] > 100 and st["redo_tiles"] == 0
    same_image(img, ref, "spheres only")


def test_bunny_with_spheres_multi_tile_partition(rtx, orc, samples_seeded):
    """bunny.obj + ground + a few spheres standing on the ground, default camera, rendered as interleaved row tiles
    (the multi-GPU partition) and as whole rows: both equal the oracle."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tris, rgb = rtx.default_primitives([os.path.join(root, "models", "bunny.obj")])
    spheres = np.array([[-60.0, 20.0, -40.0, 20.0], [70.0, 15.0, 10.0, 15.0], [0.0, 130.0, -30.0, 12.0]], F)
    srgb = np.array([[0.9, 0.2, 0.2], [0.2, 0.9, 0.2], [0.2, 0.2, 0.9]], F)
    W, H = 96, 56
    ref, ost, otri = orc.Scene(W, H, tris, rgb, samples_seeded, spheres=spheres, sphere_rgb=srgb,
                               nb_light_sample=20).render_rows(mode=orc.MODE_BVH, want_tri=True)
    assert (otri >= len(tris)).sum() - (otri == 0xFFFFFFFF).sum() > 20          # sphere pixels
    with rtx.Scene(W, H, tris, rgb, samples_seeded, spheres=spheres, sphere_rgb=srgb, nb_light_sample=20) as s:
        img, st = s.render_rows(stats=True)
        frame = s.render_frame(devices=(0,), tile_rows=8)
    assert st["primary_hits"] == ost["primary_hits"]
    same_image(img, ref, "bunny + spheres")
    same_image(frame, ref, "bunny + spheres, tiled frame")
