#!/usr/bin/env python3
"""Development tool: how long probe_kernel's wavefronts take, per tile.  Needs a library built with
EXTRA=-DRTX_EXPERIMENT_PROBE_PHASES=1 (a tile's descriptor then carries the duration of its primary walk and of what
follows it — hit records, bounds, the cut's descent — in 10 ns ticks).    python tools/probe_phases.py [c3|c4] [shares]"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtx = importlib.import_module("ray-tracer-rust_amd")
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
shares = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H = (4096, 4096) if wl == "c4" else (1920, 1080)
scene = rtx.default_scene([os.path.join(ROOT, "models", "big_bunny.obj")], W, H, rtx.gen_samples())
scene.upload(0)
nb = scene.tiles_bytes(0, shares, 8)
out = torch.zeros(nb, dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    scene.render_tiles_device(0, 0, shares, 8, out.data_ptr(), nb, st, None)
torch.cuda.synchronize()
sched, shade = scene.launch_timings(0, 3)
td = scene.tile_descs(0)
walk = (td[:, 3] & 0xFFFF) / 100.0
rest = (td[:, 3] >> 16) / 100.0
n_hit, n_cut = td[:, 1], (td[:, 2] >> 8) & 0xFF
print(json.dumps({"workload": wl, "shares": shares, "tiles": int(len(td)), "sched_ms": float(sched[-1])}))
for name, m in (("no hit", n_hit == 0), ("hit, empty cut", (n_hit > 0) & (n_cut == 0)), ("hit, cut 1..7", (n_cut > 0) & (n_cut < 8)), ("hit, cut 8..16", n_cut >= 8)):
    if m.any():
        print("%-16s %6d tiles   primary walk: mean %6.2f us, max %6.2f us    hit records + bounds + cut: mean %6.2f us, max %6.2f us" % (
            name, int(m.sum()), walk[m].mean(), walk[m].max(), rest[m].mean(), rest[m].max()))
tot = walk + rest
print("longest wavefronts (us):", np.sort(tot)[-8:].round(1).tolist(), " 99th percentile %.1f, median %.1f" % (np.percentile(tot, 99), np.median(tot)))
