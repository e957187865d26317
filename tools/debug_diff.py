#!/usr/bin/env python3
"""Development tool: where does a GPU render differ from the oracle?  python tools/debug_diff.py W H [obj]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib
rtx = importlib.import_module("ray-tracer-rust_amd")
W, H = int(sys.argv[1]), int(sys.argv[2])
obj = sys.argv[3] if len(sys.argv) > 3 else "big_bunny.obj"
T = rtx.gen_samples()
with rtx.default_scene([os.path.join(ROOT, "models", obj)], W, H, T) as s:
    img, st = s.render_rows(stats=True)
ref, ost = orclib.default_scene([obj], W, H, T).render_rows(mode=orclib.MODE_BVH)
d = np.abs(img.astype(int) - ref.astype(int)).max(axis=2)
ys, xs = np.nonzero(d)
print("hits", st["primary_hits"], ost["primary_hits"], "diff px", len(ys), "max", d.max())
for y, x in list(zip(ys, xs))[:40]:
    print("  px", x, "py", y, "tile", x // 8, y // 8, "gpu", img[y, x], "ref", ref[y, x])
if len(ys):
    tiles = sorted(set((x // 8, y // 8) for y, x in zip(ys, xs)))
    print(len(tiles), "tiles:", tiles[:40])
    print("brighter on GPU:", int((img.astype(int).sum(axis=2) > ref.astype(int).sum(axis=2)).sum()),
          "darker:", int((img.astype(int).sum(axis=2) < ref.astype(int).sum(axis=2)).sum()))
