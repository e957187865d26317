"""Host-side preparation of scenes holding both arms of Primitive (src/tracer/primitives/mod.rs:40-43): the
traversal stream keeps one arm per leaf, the reference-tree stream matches the oracle's tree, bad inputs are
refused.  No device needed."""
import importlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def rtx():
    return importlib.import_module("ray-tracer-rust_amd")


LEAF, SPHERE, INDEX = 0x80000000, 0x40000000, 0x3FFFFFFF
F = np.float32


def mixed_scene(rng, n_tris=150, n_spheres=40):
    c = rng.uniform(-6, 6, size=(n_tris, 1, 3)).astype(F)
    c[..., 2] -= 14.0
    tris = (c + rng.uniform(-2.0, 2.0, size=(n_tris, 3, 3)).astype(F)).reshape(-1, 9)
    rgb = rng.uniform(0.2, 1.0, size=(n_tris, 3)).astype(F)
    sc = rng.uniform(-6, 6, size=(n_spheres, 3)).astype(F)
    sc[:, 2] -= 14.0
    spheres = np.concatenate([sc, rng.uniform(0.2, 1.5, size=(n_spheres, 1)).astype(F)], axis=1)
    srgb = rng.uniform(0.2, 1.0, size=(n_spheres, 3)).astype(F)
    kinds = np.zeros(n_tris + n_spheres, np.uint8)
    kinds[rng.choice(n_tris + n_spheres, n_spheres, replace=False)] = 1
    return np.ascontiguousarray(tris), rgb, np.ascontiguousarray(spheres), srgb, kinds


def prim_boxes(tris, spheres, kinds):
    """Boxes in Vec order: triangle.rs:45-56 / sphere.rs:32-41 (f32 arithmetic)."""
    tb = np.concatenate([tris.reshape(-1, 3, 3).min(axis=1), tris.reshape(-1, 3, 3).max(axis=1)], axis=1)
    r = spheres[:, 3:4]
    sb = np.concatenate([spheres[:, :3] - r, spheres[:, :3] + r], axis=1).astype(F)
    out = np.zeros((len(kinds), 6), F)
    out[kinds == 0] = tb
    out[kinds == 1] = sb
    return out


def check_stream(nodes, order, boxes, kinds):
    n, n_prims = len(nodes), len(kinds)
    f = nodes.view(F)
    seen = np.zeros(n_prims, bool)
    leaves = []

    def walk(i):
        info, link = int(nodes[i, 7]), int(nodes[i, 3])
        lo, hi = f[i, 0:3], f[i, 4:7]
        if info & LEAF:
            first = info & INDEX
            assert link >= 1 and first + link <= n_prims
            ids = order[first:first + link]
            assert not seen[ids].any()
            seen[ids] = True
            assert (kinds[ids] == (1 if info & SPHERE else 0)).all(), "a leaf holds one arm only"
            assert np.array_equal(lo, boxes[ids, :3].min(axis=0)) and np.array_equal(hi, boxes[ids, 3:].max(axis=0))
            leaves.append(ids)
            return i + 1, lo, hi
        assert i + 1 < link <= n
        nxt, lo1, hi1 = walk(i + 1)
        assert info == nxt           # an inner node names its second child
        nxt2, lo2, hi2 = walk(nxt)
        assert nxt2 == link
        assert np.array_equal(lo, np.minimum(lo1, lo2)) and np.array_equal(hi, np.maximum(hi1, hi2))
        return link, lo, hi

    end, _, _ = walk(0)
    assert end == n and seen.all()
    return leaves


@pytest.mark.parametrize("accel,leaf_max", [(0, 0), (0, 1), (0, 16), (1, 0)])
def test_mixed_stream_keeps_one_arm_per_leaf(rtx, samples_half, accel, leaf_max):
    tris, rgb, spheres, srgb, kinds = mixed_scene(np.random.default_rng(7))
    boxes = prim_boxes(tris, spheres, kinds)
    with rtx.Scene(16, 16, tris, rgb, samples_half[:64], spheres=spheres, sphere_rgb=srgb, kinds=kinds, accel=accel,
                   leaf_max=leaf_max) as s:
        nodes, order = s.nodes()
        info = s.info()
        assert info["n_tris"] == len(kinds) and sorted(order.tolist()) == list(range(len(kinds)))
        leaves = check_stream(nodes, order, boxes, kinds)
        if accel == 1:
            assert info["n_nodes"] == 3 and len(leaves) == 2      # a root over one leaf per arm
        # shade rows: a sphere's row holds its origin
        nrm = s.normals()
        assert np.array_equal(nrm[kinds == 1], spheres[:, :3])
        assert np.allclose(np.linalg.norm(nrm[kinds == 0], axis=1), 1.0, atol=1e-5)


def test_default_order_is_triangles_then_spheres(rtx, samples_half):
    tris, rgb, spheres, srgb, _ = mixed_scene(np.random.default_rng(8), 20, 5)
    kinds = np.array([0] * 20 + [1] * 5, np.uint8)
    with rtx.Scene(8, 8, tris, rgb, samples_half[:64], spheres=spheres, sphere_rgb=srgb) as s:
        nodes, order = s.nodes()
        check_stream(nodes, order, prim_boxes(tris, spheres, kinds), kinds)
        assert np.array_equal(s.normals()[20:], spheres[:, :3])


def test_spheres_only_scene(rtx, samples_half):
    _, _, spheres, srgb, _ = mixed_scene(np.random.default_rng(9), 1, 33)
    kinds = np.ones(33, np.uint8)
    none = np.zeros((0, 9), F)
    with rtx.Scene(8, 8, none, np.zeros((0, 3), F), samples_half[:64], spheres=spheres, sphere_rgb=srgb) as s:
        nodes, order = s.nodes()
        check_stream(nodes, order, prim_boxes(none, spheres, kinds), kinds)
        assert all(int(x) & SPHERE for x in nodes[:, 7] if int(x) & LEAF)


def test_reference_tree_over_both_arms_matches_oracle(rtx, orc, samples_half):
    """The reference clusters by the primitives' boxes, whatever the arm (bvh.rs:173-226): the library's stream of
    that tree has the oracle's left-to-right leaf order, and its leaves carry the arm flag."""
    tris, rgb, spheres, srgb, kinds = mixed_scene(np.random.default_rng(10), 90, 25)
    osc = orc.Scene(8, 8, tris, rgb, samples_half[:64], spheres=spheres, sphere_rgb=srgb, kinds=kinds)
    with rtx.Scene(8, 8, tris, rgb, samples_half[:64], spheres=spheres, sphere_rgb=srgb, kinds=kinds) as s:
        ref = s.ref_nodes()
        _, order = s.nodes()
        assert len(ref) == 2 * len(kinds) - 1 == osc.node_count()
        leaf_rows = [r for r in ref if int(r[7]) & LEAF]
        ids = np.array([order[int(r[7]) & INDEX] for r in leaf_rows])
        assert np.array_equal(ids, osc.leaf_order())
        assert all(bool(int(r[7]) & SPHERE) == bool(kinds[i]) for r, i in zip(leaf_rows, ids))
        assert all(int(r[3]) == 1 for r in leaf_rows)


def test_bad_sphere_inputs_are_refused(rtx, samples_half):
    tris, rgb, spheres, srgb, kinds = mixed_scene(np.random.default_rng(11), 10, 4)
    bad_kinds = kinds.copy()
    bad_kinds[np.nonzero(kinds == 0)[0][0]] = 1                       # five spheres announced, four given
    with pytest.raises(rtx.RtxError) as e:
        rtx.Scene(8, 8, tris, rgb, samples_half[:64], spheres=spheres, sphere_rgb=srgb, kinds=bad_kinds)
    assert e.value.code == rtx.ERR_BAD_ARG
    nan = spheres.copy()
    nan[2, 3] = np.nan
    with pytest.raises(rtx.RtxError) as e:
        rtx.Scene(8, 8, tris, rgb, samples_half[:64], spheres=nan, sphere_rgb=srgb, kinds=kinds)
    assert e.value.code == rtx.ERR_UNSUPPORTED
    with pytest.raises(ValueError):
        rtx.Scene(8, 8, tris, rgb, samples_half[:64], spheres=spheres, sphere_rgb=srgb, kinds=kinds[:-1])
