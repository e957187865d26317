"""Randomised parity: scenes with random cameras, lights, primitive mixes, frame sizes, ray counts — everything the
shorter-than-IEEE sequences, the cost-ordered schedule and the up-front leaves see differently from the fixed goldens.
Each scene is rendered by the oracle (faithful BVH) and through the C ABI; bytes and hit counts must be equal.
RTX_STRESS_SCENES scales it up for a soak (default 16 scenes, a few seconds)."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def rtx():
    mod = importlib.import_module("ray-tracer-rust_amd")
    assert mod.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    return mod


def random_scene(rng):
    n_tris = int(rng.integers(1, 400))
    scale = float(10.0 ** rng.uniform(-1.0, 3.0))               # scene extent from 0.1 to 1000 units
    centre = rng.uniform(-1.0, 1.0, 3) * scale * float(rng.choice([0.0, 1.0, 30.0]))
    c = rng.uniform(-1.0, 1.0, size=(n_tris, 1, 3)) * scale
    size = scale * float(10.0 ** rng.uniform(-2.0, 0.0))
    v = (centre + c + rng.uniform(-1.0, 1.0, size=(n_tris, 3, 3)) * size).astype(F)
    tris = v.reshape(-1, 9)
    rgb = rng.uniform(0.0, 1.0, size=(n_tris, 3)).astype(F)
    if rng.random() < 0.5:                                      # grey scene: the one-channel accumulation path
        rgb[:] = rgb[:, :1]
    if rng.random() < 0.6:                                      # a floor spanning the scene: a "global" triangle
        y = float(centre[1] - 1.2 * scale)
        floor = np.array([[centre[0] - 40 * scale, y, centre[2] - 40 * scale, centre[0] + 40 * scale, y,
                           centre[2] - 40 * scale, centre[0], y, centre[2] + 40 * scale]], F)
        tris = np.concatenate([tris, floor])
        rgb = np.concatenate([rgb, np.array([[0.5, 0.5, 0.5]], F)])
    n_sph = int(rng.integers(0, 30)) if rng.random() < 0.4 else 0
    spheres = np.concatenate([centre + rng.uniform(-1.0, 1.0, size=(n_sph, 3)) * scale,
                              rng.uniform(0.02, 0.3, size=(n_sph, 1)) * scale], axis=1).astype(F)
    srgb = rng.uniform(0.0, 1.0, size=(n_sph, 3)).astype(F)
    kinds = np.zeros(len(tris) + n_sph, np.uint8)
    kinds[rng.choice(len(kinds), n_sph, replace=False)] = 1
    direction = rng.normal(size=3)
    eye = centre + direction / np.linalg.norm(direction) * scale * rng.uniform(1.5, 4.0)
    light_c = centre + np.array([rng.uniform(-1, 1), rng.uniform(1.5, 4.0), rng.uniform(-1, 1)]) * scale
    light = (light_c + rng.uniform(-0.1, 0.1, size=(3, 3)) * scale).astype(F).reshape(9)
    W, H = int(rng.integers(9, 70)), int(rng.integers(9, 70))
    kw = dict(eye=tuple(float(x) for x in eye.astype(F)), look_at=tuple(float(x) for x in centre.astype(F)),
              up=(0.0, 1.0, 0.0), distance=float(rng.uniform(0.6, 2.5) * max(W, H)),
              light_tri=tuple(float(x) for x in light), nb_ray=int(rng.choice([1, 1, 1, 2])),
              nb_light_sample=int(rng.choice([1, 7, 32, 100, 130])))
    return W, H, tris, rgb, spheres, srgb, kinds, kw


def test_random_scenes_match_the_oracle(rtx, orc, samples_seeded):
    n_scenes = int(os.environ.get("RTX_STRESS_SCENES", "16"))
    rng = np.random.default_rng(20261004)
    checked = hits = 0
    for k in range(n_scenes):
        W, H, tris, rgb, spheres, srgb, kinds, kw = random_scene(rng)
        extra = dict(spheres=spheres, sphere_rgb=srgb, kinds=kinds) if len(spheres) else {}
        ref, ost = orc.Scene(W, H, tris, rgb, samples_seeded, **extra, **kw).render_rows(mode=orc.MODE_BVH)
        if ost["nonfinite_t"]:
            continue                                            # outside the parity contract
        with rtx.Scene(W, H, tris, rgb, samples_seeded, **extra, **kw) as s:
            img, st = s.render_rows(stats=True)
        assert st["primary_hits"] == ost["primary_hits"], "scene %d: %r" % (k, kw)
        assert np.array_equal(img, ref), "scene %d: %d bytes differ (%dx%d, %d tris, %d spheres, %r)" % (
            k, int((img != ref).sum()), W, H, len(tris), len(spheres), kw)
        checked += 1
        hits += ost["primary_hits"]
    assert checked >= n_scenes * 3 // 4 and hits > 100 * checked
