#!/usr/bin/env python3
"""Development tool: compare GPU and oracle pixel by pixel on one random-soup scene."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib as orc
rtx = importlib.import_module("ray-tracer-rust_amd")

def soup(rng, n, lattice):
    if lattice:
        v = rng.integers(-6, 7, size=(n, 3, 3)).astype(np.float32); v[..., 2] -= 14.0
    else:
        c = rng.uniform(-6, 6, size=(n, 1, 3)).astype(np.float32); c[..., 2] -= 14.0
        v = c + rng.uniform(-2.5, 2.5, size=(n, 3, 3)).astype(np.float32)
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    keep = np.linalg.norm(np.cross(e1, e2), axis=1) > 1e-3
    v = v[keep]
    return v.reshape(-1, 9), rng.uniform(0.2, 1.0, size=(len(v), 3)).astype(np.float32)

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tris, rgb = soup(np.random.default_rng(seed), 300, True)
T = np.zeros((4096, 2), np.float32)
kw = dict(eye=(0.0, 0.0, 0.0), look_at=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), distance=24.0,
          light_tri=(-2.0, 9.0, -3.0, 2.0, 9.0, -3.0, 0.0, 9.0, 1.0), nb_light_sample=nl)
W = H = 40
osc = orc.Scene(W, H, tris, rgb, T, **kw)
ref, ost, otri = osc.render_rows(mode=orc.MODE_BVH, want_tri=True)
_, _, ltri = osc.render_rows(mode=orc.MODE_LEAFBOX, want_tri=True)
for reft in (rtx.REFTREE_AUTO, rtx.REFTREE_NEVER):
    with rtx.Scene(W, H, tris, rgb, T, reference_tree=reft, **kw) as s:
        img, st = s.render_rows(stats=True)
    ghit = img.any(axis=2) if nl else None
    print("reftree", reft, "gpu hits", st["primary_hits"], "oracle", ost["primary_hits"], "redo tiles", st["redo_tiles"])
if nl:
    ohit = otri != 0xFFFFFFFF
    d = np.argwhere((img != ref).any(axis=2))
    print("differing pixels (y,x):", d.tolist()[:40])
    for y, x in d[:12]:
        print((y, x), "gpu", img[y, x], "ref", ref[y, x], "otri", otri[y, x], "leafbox tri", ltri[y, x])
print("bvh vs leafbox primary differ at", np.argwhere(otri != ltri).tolist()[:20])
