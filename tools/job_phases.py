#!/usr/bin/env python3
"""Development tool: where a job of a tile with an empty cut spends its time.  Needs a library built with
EXTRA=-DRTX_EXPERIMENT_PHASES=1.      python tools/job_phases.py [c3|c4|c2]"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtx = importlib.import_module("ray-tracer-rust_amd")
wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
W, H = {"c3": (1920, 1080), "c2": (1920, 1080), "c4": (4096, 4096)}[wl]
scene = rtx.default_scene([os.path.join(ROOT, "models", "bunny.obj" if wl == "c2" else "big_bunny.obj")], W, H, rtx.gen_samples())
scene.upload(0)
nb = scene.tiles_bytes(0, 1, 8)
out = torch.zeros(nb, dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    scene.render_tiles_device(0, 0, 1, 8, out.data_ptr(), nb, st, None)
torch.cuda.synchronize()
td = scene.tile_descs(0)
t = td[:6, 3].astype(np.float64)
n = t[5]
names = ["descriptor + hit records + pixel slots -> LDS (to the barrier)", "grey check + light points -> LDS (to the barrier)",
         "shadow rays of all chunks (to the barrier)", "ordered sums (one wavefront; to the barrier)", "quantise + store (to the end barrier)"]
print(json.dumps({"workload": wl, "jobs_with_empty_cut": int(n)}))
for k in range(5):
    print("%-70s %7.2f us per job" % (names[k], t[k] / n / 100.0))
print("%-70s %7.2f us per job" % ("sum (claim to end barrier)", t[:5].sum() / n / 100.0))
