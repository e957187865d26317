"""GPU tests of the launch structure (probe_kernel -> count/order -> shade_tiles_kernel -> reference_tiles_kernel):
the corners where the passes hand work to each other.  Results are still checked against the oracle."""
import importlib
import os

import numpy as np
import pytest

from test_host_spheres import mixed_scene

pytestmark = pytest.mark.gpu
F = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rtx():
    mod = importlib.import_module("ray-tracer-rust_amd")
    assert mod.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return mod


AXIS = dict(eye=(0.0, 0.0, 0.0), look_at=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), distance=24.0,
            light_tri=(-2.0, 9.0, -3.0, 2.0, 9.0, -3.0, 0.0, 9.0, 1.0))


def soup(seed, n=260):
    rng = np.random.default_rng(seed)
    v = rng.integers(-6, 7, size=(n, 3, 3)).astype(F)
    v[..., 2] -= 14.0
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    v = v[np.linalg.norm(np.cross(e1, e2), axis=1) > 1e-3]
    return v.reshape(-1, 9), rng.uniform(0.2, 1.0, size=(len(v), 3)).astype(F)


@pytest.mark.parametrize("nb_ray,nb_light", [(2, 12), (3, 5)])
def test_several_primary_rays_with_tiles_queued_for_the_reference_walk(rtx, orc, nb_ray, nb_light):
    """nb_ray > 1 and an all-zero sample table: every primary ray of the centre row/column has a zero direction
    component, so the scheduling pass queues those tiles at r = 0 and must keep them queued (once) at r = 1, 2; the
    running sums of the other tiles travel through HBM between the passes; hit counts are the oracle's."""
    tris, rgb = soup(41)
    T = np.zeros((4096, 2), F)
    W = H = 40
    ref, ost = orc.Scene(W, H, tris, rgb, T, nb_ray=nb_ray, nb_light_sample=nb_light, **AXIS).render_rows(mode=orc.MODE_BVH)
    with rtx.Scene(W, H, tris, rgb, T, nb_ray=nb_ray, nb_light_sample=nb_light, **AXIS) as s:
        img, st = s.render_rows(stats=True)
        again = s.render_rows()                               # the workspace is reused by the next launch
    assert st["redo_tiles"] > 0 and st["redo_tiles"] <= (W // 8) * (H // 8)       # queued once, not once per ray
    assert st["primary_rays"] == nb_ray * W * H and st["primary_hits"] == ost["primary_hits"]
    assert np.array_equal(img, ref) and np.array_equal(again, ref)


def test_light_batches_and_partial_tiles(rtx, orc, samples_seeded):
    """More light samples than one LDS batch holds (128) and a frame whose edge tiles are partial in both directions."""
    tris, rgb = soup(42, 120)
    W, H = 37, 21
    kw = dict(AXIS, nb_light_sample=150)
    ref, ost = orc.Scene(W, H, tris, rgb, samples_seeded, **kw).render_rows(mode=orc.MODE_BVH)
    with rtx.Scene(W, H, tris, rgb, samples_seeded, **kw) as s:
        img, st = s.render_rows(stats=True)
        rows = np.concatenate([s.render_rows(0, 5), s.render_rows(5, 16)])          # launches that start off the tile grid
    assert st["primary_hits"] == ost["primary_hits"]
    assert np.array_equal(img, ref) and np.array_equal(rows, ref)


def test_no_light_samples_and_empty_scene_view(rtx, orc, samples_seeded):
    """nb_light_sample = 0 (every hit pixel stays black: the scheduling pass has nothing to probe) and a camera that
    sees nothing (every tile is finished by the scheduling pass, the shading pass has no tile to pull)."""
    tris, rgb = soup(43, 60)
    W = H = 24
    with rtx.Scene(W, H, tris, rgb, samples_seeded, **dict(AXIS, nb_light_sample=0)) as s:
        img, st = s.render_rows(stats=True)
    assert st["primary_hits"] > 0 and st["shadow_rays"] == 0 and not img.any()
    away = dict(AXIS, look_at=(0.0, 0.0, 1.0), nb_light_sample=8)
    ref, ost = orc.Scene(W, H, tris, rgb, samples_seeded, **away).render_rows(mode=orc.MODE_BVH)
    with rtx.Scene(W, H, tris, rgb, samples_seeded, **away) as s:
        img, st = s.render_rows(stats=True)
    assert ost["primary_hits"] == 0 == st["primary_hits"] and np.array_equal(img, ref)


def test_mixed_arms_several_primary_rays_zero_table(rtx, orc):
    tris, rgb, spheres, srgb, kinds = mixed_scene(np.random.default_rng(44), 150, 40)
    T = np.zeros((4096, 2), F)
    kw = dict(AXIS, nb_ray=2, nb_light_sample=10)
    W = H = 32
    ref, ost = orc.Scene(W, H, tris, rgb, T, spheres=spheres, sphere_rgb=srgb, kinds=kinds, **kw).render_rows(mode=orc.MODE_BVH)
    with rtx.Scene(W, H, tris, rgb, T, spheres=spheres, sphere_rgb=srgb, kinds=kinds, **kw) as s:
        img, st = s.render_rows(stats=True)
    assert st["primary_hits"] == ost["primary_hits"] and st["redo_tiles"] > 0
    assert np.array_equal(img, ref)


def test_launch_timings(rtx, samples_seeded):
    """rtx_launch_timings: one (scheduling, shading) pair per launch, oldest first, bounded by the ring."""
    tris, rgb = soup(45)
    with rtx.Scene(64, 64, tris, rgb, samples_seeded, **dict(AXIS, nb_light_sample=16)) as s:
        a, b = s.launch_timings()
        assert len(a) == 0 and len(b) == 0
        for _ in range(3):
            s.render_rows()
        sched, shade = s.launch_timings()
        assert len(sched) == len(shade) == 3
        assert (shade > 0).all() and (sched >= 0).all() and (sched + shade < 1000.0).all()
        assert len(s.launch_timings(max_launches=2)[0]) == 2
        for _ in range(70):
            s.render_rows(0, 8)
        assert len(s.launch_timings(max_launches=100)[0]) == 64          # RTX_TIMING_RING


def test_every_tile_of_a_4096_square_frame_is_rendered_exactly_once(rtx, orc, samples_seeded):
    """BASELINE configs[3] at full size (262,144 tiles: 256 workgroups of the count / order kernels): the frame is
    rendered into two buffers pre-filled with different bytes — a tile the cost-ordered schedule skipped would keep
    the filler, a tile rendered twice could not be told, so the launch's own hit count is checked against the
    number of non-black pixels too — and 131 single rows spread over the whole height are compared with the oracle."""
    import os
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    W = H = 4096
    frames = []
    with rtx.default_scene([os.path.join(root, "models", "big_bunny.obj")], W, H, samples_seeded) as s:
        nbytes = s.tiles_bytes(0, 1, 8)
        assert nbytes == W * H * 3
        ctr = torch.zeros(8, dtype=torch.int64, device="cuda:0")
        for filler in (0xAA, 0x55):
            buf = torch.full((nbytes,), filler, dtype=torch.uint8, device="cuda:0")
            s.render_tiles_device(0, 0, 1, 8, buf.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream,
                                  ctr.data_ptr() if filler == 0xAA else None)
            torch.cuda.synchronize()
            frames.append(buf.cpu().numpy().reshape(H, W, 3))
        hits = int(ctr[0])
    assert np.array_equal(frames[0], frames[1])
    # sky pixels are exactly black; so are hit pixels in full shadow, a minority
    lit = int((frames[0].reshape(-1, 3).max(axis=1) > 0).sum())
    assert 0.5 * hits < lit <= hits
    # 128 single rows against the oracle's faithful BVH, in bit-reversed order over the frame's height with the low bits
    # varied (so that neither a band of the frame nor a residue of the row number is left out): about 10 s of oracle time
    osc = orc.default_scene(["big_bunny.obj"], W, H, samples_seeded)
    rows = sorted({(int("{:012b}".format(k)[::-1], 2) + 13 * k) % H for k in range(128)} | {2000, 2100, 4095})
    assert len(rows) >= 128
    bad = []
    for row in rows:
        ref, _ = osc.render_rows(row, 1, mode=orc.MODE_BVH)
        if not np.array_equal(frames[0][row:row + 1], ref):
            bad.append((row, int((frames[0][row:row + 1] != ref).any(axis=2).sum())))
    assert not bad, "rows that differ from the oracle (row, pixels): %s" % bad[:8]


def test_scenes_whose_triangles_are_all_or_partly_global(rtx, orc, samples_seeded):
    """Triangles as large as the scene are tested up front, outside the tree (host: n_global): a scene of nothing but
    such triangles (one-leaf stream), and a soup over a floor that spans it (one global triangle beside the tree)."""
    kw = dict(AXIS, nb_light_sample=12)
    W = H = 32
    walls = np.array([[-9, -9, -5, 9, -9, -5, 0, 9, -5], [-9, -9, -6, 9, -9, -6, 0, 9, -6]], F)
    tris, rgb = soup(46, 150)
    floor = np.array([[-40, -7, 10, 40, -7, 10, 0, -7, -60]], F)
    for name, t, c, want in (("walls", walls, np.array([[1, 0.5, 0.25], [0.2, 0.9, 0.4]], F), 2),
                             ("soup on a floor", np.concatenate([tris, floor]),
                              np.concatenate([rgb, np.array([[0.5, 0.5, 0.5]], F)]), 1)):
        ref, ost = orc.Scene(W, H, t, c, samples_seeded, **kw).render_rows(mode=orc.MODE_BVH)
        with rtx.Scene(W, H, t, c, samples_seeded, **kw) as s:
            assert s.info()["n_global"] == want, name
            img, st = s.render_rows(stats=True)
        assert st["primary_hits"] == ost["primary_hits"] > 0 and np.array_equal(img, ref), name


def _floor_scene(seed):
    """A big tilted floor (a global triangle) with a soup hovering 0.3 .. 3 units over it, seen from above; the light
    is below the floor (shadow rays cross it at t around 1: the `t < 1.0` rule of bvh.rs:64 decides), near the horizon
    (grazing rays over the floor) or overhead."""
    rng = np.random.default_rng(seed)
    tilt = rng.uniform(-0.15, 0.15, size=2)
    def height(x, z):
        return tilt[0] * x + tilt[1] * z
    corners = np.array([[-150.0, -120.0], [160.0, -110.0], [5.0, 170.0]]) + rng.uniform(-5, 5, size=(3, 2))
    floor = np.array([[cx, height(cx, cz), cz] for cx, cz in corners], F).reshape(1, 9)
    n = 90
    c = rng.uniform(-10, 10, size=(n, 2))
    lift = rng.choice([0.3, 0.6, 0.9, 1.0, 1.1, 1.5, 3.0], size=n) * rng.uniform(0.97, 1.03, size=n)
    centre = np.stack([c[:, 0], height(c[:, 0], c[:, 1]) + lift, c[:, 1]], axis=1)
    soup = (centre[:, None, :] + rng.uniform(-0.8, 0.8, size=(n, 3, 3)) * np.array([1.0, 0.15, 1.0])).astype(F)
    e1, e2 = soup[:, 1] - soup[:, 0], soup[:, 2] - soup[:, 0]
    soup = soup[np.linalg.norm(np.cross(e1, e2), axis=1) > 1e-3].reshape(-1, 9)
    tris = np.concatenate([soup, floor]).astype(F)
    rgb = rng.uniform(0.2, 1.0, size=(len(tris), 3)).astype(F)
    where = seed % 3
    ly = (-25.0, 1.2, 40.0)[where]
    lx = (3.0, 60.0, -4.0)[where]
    light = (lx - 2.0, ly, -3.0, lx + 2.0, ly, -3.0, lx, ly + (0.5 if where == 1 else 0.0), 2.0)
    cam = dict(eye=(1.0, 30.0, 22.0), look_at=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), distance=70.0, light_tri=light)
    return tris, rgb, cam


@pytest.mark.parametrize("seed", range(9))
def test_plane_shortcut_of_global_triangles(rtx, orc, samples_seeded, seed):
    """rtx_traverse.hpp: plane_rules_out skips Triangle::intersect on a global triangle when the plane alone shows that
    the answer is None; here the answers sit on both sides of every one of its conditions."""
    tris, rgb, cam = _floor_scene(seed)
    W = H = 56
    kw = dict(cam, nb_light_sample=12)
    ref, ost = orc.Scene(W, H, tris, rgb, samples_seeded, **kw).render_rows(mode=orc.MODE_BVH)
    with rtx.Scene(W, H, tris, rgb, samples_seeded, **kw) as s:
        assert s.info()["n_global"] == 1
        img, st = s.render_rows(stats=True)
    assert st["primary_hits"] == ost["primary_hits"] > 0.5 * W * H
    assert np.array_equal(img, ref)


_CUT_VS_WHOLE_CHILD = r"""
import importlib, sys, hashlib, json
import numpy as np
sys.path.insert(0, %(root)r)
rtx = importlib.import_module('ray-tracer-rust_amd')
F = np.float32
T = rtx.gen_samples()[:8192]

def soup(seed, n, centre, extent, size):
    g = np.random.default_rng(seed)
    c = g.uniform(-extent, extent, (n, 1, 3)) + np.asarray(centre, F)
    t = (c + g.uniform(-size, size, (n, 3, 3))).astype(F).reshape(n, 9)
    return t, g.uniform(0.2, 1.0, (n, 3)).astype(F)

out = {}
# 1. the light INSIDE the mesh: the light's box overlaps every tile's hit box, shafts run in every direction
t, c = soup(1, 3000, (0, 0, 0), 40, 2.5)
out['light inside'] = dict(tris=t, rgb=c, eye=(0, 10, 150), look_at=(0, 0, 0), distance=90.0,
                           light_tri=np.array([-3, 1, -3, 3, 1, -3, 0, -2, 3], F), nb_light_sample=24)
# 2. coordinates around one million: the margins of the shaft test are relative to the scene's magnitude
t, c = soup(2, 2500, (1.0e6, 2.0e6, -1.5e6), 60, 4.0)
floor = np.array([[1.0e6 - 500, 2.0e6 - 70, -1.5e6 + 500, 1.0e6 + 500, 2.0e6 - 70, -1.5e6 + 500, 1.0e6, 2.0e6 - 70, -1.5e6 - 800]], F)
out['coordinates of a million'] = dict(tris=np.concatenate([t, floor]), rgb=np.concatenate([c, np.array([[0.5, 0.5, 0.5]], F)]),
                                        eye=(1.0e6, 2.0e6 + 30, -1.5e6 + 260), look_at=(1.0e6, 2.0e6 - 20, -1.5e6), distance=100.0,
                                        light_tri=np.array([1.0e6 - 15, 2.0e6 + 200, -1.5e6 - 10, 1.0e6 + 15, 2.0e6 + 200, -1.5e6 - 10, 1.0e6, 2.0e6 + 200, -1.5e6 + 12], F),
                                        nb_light_sample=20)
# 3. spheres among the triangles, and two primary rays per pixel
t, c = soup(3, 1500, (0, 30, 0), 50, 3.0)
g = np.random.default_rng(33)
sph = np.concatenate([g.uniform(-50, 50, (200, 3)) + np.array([0, 30, 0]), g.uniform(0.5, 4.0, (200, 1))], axis=1).astype(F)
floor = np.array([[-400, -25, 300, 400, -25, 300, 0, -25, -600]], F)
out['spheres, two primary rays'] = dict(tris=np.concatenate([t, floor]), rgb=np.concatenate([c, np.array([[0.5, 0.5, 0.5]], F)]),
                                         spheres=sph, sphere_rgb=g.uniform(0.2, 1.0, (200, 3)).astype(F),
                                         eye=(0, 60, 220), look_at=(0, 20, 0), distance=110.0, nb_ray=2, nb_light_sample=16)
res = {}
for name, kw in out.items():
    tris, rgb = kw.pop('tris'), kw.pop('rgb')
    with rtx.Scene(160, 120, tris, rgb, T, **kw) as s:
        info = s.info()
        img, st = s.render_rows(stats=True)
    res[name] = dict(sha=hashlib.sha1(img.tobytes()).hexdigest(), hits=int(st['primary_hits']), rays=int(st['rays']),
                     lit=int((img.max(axis=2) > 0).sum()), nodes=int(info['n_nodes']), node_visits=int(st['wave_node_visits']))
print('RESULT', json.dumps(res))
"""


def test_cut_and_whole_stream_walks_give_the_same_bytes():
    """The per-tile shaft cut decides which subtrees a tile's shadow rays may see: a cut that is not
    a superset would show as missing shadows.  The same scenes are rendered by librtx_ablation.so twice, each in a child
    process — with the cuts, and with RTX_CUT_MAX_NODES=0, which makes every chunk walk the whole stream — and must give
    the same bytes: the light inside the mesh (its box overlaps the tiles' hit boxes), coordinates around 10^6 (the
    margins are relative), spheres with two primary rays per pixel."""
    import json
    import subprocess
    import sys
    code = _CUT_VS_WHOLE_CHILD % dict(root=ROOT)
    results = []
    for cut_max in (None, "0"):
        env = dict(os.environ, RTX_PY_ABLATION="1")
        env.pop("RTX_CUT_MAX_NODES", None)
        if cut_max is not None:
            env["RTX_CUT_MAX_NODES"] = cut_max
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        assert line, out.stderr[-1200:]
        results.append(json.loads(line[0][7:]))
    cut, whole = results
    assert set(cut) == set(whole) and len(cut) == 3
    for name in cut:
        a, b = cut[name], whole[name]
        assert a["hits"] == b["hits"] > 0 and a["rays"] == b["rays"] and a["lit"] > 0, (name, a, b)
        assert a["sha"] == b["sha"], (name, a, b)
        assert a["node_visits"] != b["node_visits"], "the two renders of %r walked the same records: one form ran twice" % name
