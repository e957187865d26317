#!/usr/bin/env python3
"""Development tool: per-tile cost of the shading pass.  Needs a library built with EXTRA=-DRTX_EXPERIMENT_TIMELINE=1
(every job adds its duration to its tile's descriptor); prints how the frame's workgroup-time splits over the tiles by
the size of their cut.      python tools/tile_timeline.py [c3|c4|c5|c2] [resident workgroups] [shares]
(resident workgroups: 256 CUs x 3 of the cut form's 6 waves per SIMD, x 4 of the whole-tree form's 8; default by workload)"""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtx = importlib.import_module("ray-tracer-rust_amd")
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
W, H = {"c3": (1920, 1080), "c2": (1920, 1080), "c4": (4096, 4096), "c5": (4096, 4096)}[wl]
WGS = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if wl == "c5" else 768)
SHARES = int(sys.argv[3]) if len(sys.argv) > 3 else 1     # time share 0 of this many (what one GPU of N renders)
T = rtx.gen_samples()
if wl == "c5":
    tris, rgb = rtx.synthetic_primitives(1000000)
    scene = rtx.Scene(W, H, tris, rgb, T)
else:
    scene = rtx.default_scene([os.path.join(ROOT, "models", "bunny.obj" if wl == "c2" else "big_bunny.obj")], W, H, T)
scene.upload(0)
nb = scene.tiles_bytes(0, SHARES, 8)
out = torch.zeros(nb, dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    scene.render_tiles_device(0, 0, SHARES, 8, out.data_ptr(), nb, st, None)
torch.cuda.synchronize()
sched, shade = scene.launch_timings(0, 3)
td = scene.tile_descs(0)
cls, n_hit, flags, ticks = td[:, 0], td[:, 1], td[:, 2], td[:, 3].astype(np.float64)
n_cut = (flags >> 8) & 0xFF
us = ticks / 100.0
work = n_hit > 0
print(json.dumps({"workload": wl, "tiles": int(len(td)), "tiles_with_hits": int(work.sum()), "shade_ms": float(shade[-1]),
                  "sched_ms": float(sched[-1]), "sum_job_us": float(us.sum()), "wg_time_available_us": float(shade[-1] * 1e3 * WGS),
                  "busy_fraction": float(us.sum() / (shade[-1] * 1e3 * WGS))}))
edges = [0, 1, 2, 4, 8, 16, 32, 64, 65]
for a, b in zip(edges[:-1], edges[1:]):
    m = work & (n_cut >= a) & (n_cut < b)
    if m.any():
        print("cut %2d..%2d: %6d tiles  %8.0f us total (%4.1f %%)  mean %7.2f us  max %7.2f us  mean hits %4.1f" % (
            a, b - 1, int(m.sum()), us[m].sum(), 100 * us[m].sum() / us.sum(), us[m].mean(), us[m].max(), n_hit[m].mean()))
# by ray numbering (bit 0 of the flags: sample-major = all 64 hit pixels on one primitive) and by duration
for name, m0 in (("sample-major (one surface)", work & ((flags & 1) != 0)), ("pixel-major (mixed)", work & ((flags & 1) == 0))):
    if not m0.any():
        continue
    print("%s: %d tiles, %.0f us total (%.1f %%), mean %.1f us" % (name, int(m0.sum()), us[m0].sum(), 100 * us[m0].sum() / us.sum(), us[m0].mean()))
    for a, b in ((0, 30), (30, 60), (60, 120), (120, 250), (250, 500), (500, 1000), (1000, 1e9)):
        m = m0 & (us >= a) & (us < b)
        if m.any():
            print("    %5.0f..%5.0f us: %7d tiles  %9.0f us (%4.1f %% of the frame)" % (a, min(b, 99999), int(m.sum()), us[m].sum(), 100 * us[m].sum() / us.sum()))
# the estimate against the measurement: per cost class (what the order and the splitting go by) the tiles' measured durations
print("cost class: tiles, measured us mean / min / max, of which pixel-major")
for k in sorted(set(cls[work].tolist())):
    m = work & (cls == k)
    print("    class %2d: %6d tiles  %8.1f / %7.1f / %7.1f us   pixel-major %5d (mean %7.1f us)" % (
        k, int(m.sum()), us[m].mean(), us[m].min(), us[m].max(), int((m & ((flags & 1) == 0)).sum()),
        us[m & ((flags & 1) == 0)].mean() if (m & ((flags & 1) == 0)).any() else 0.0))
top = np.argsort(-us)[:8]
tx = (W + 7) // 8
def tile_xy(t):   # the library numbers tiles by 8 x 8 blocks (rtx_kernel.hip: tile_xy)
    blocks_x, b, j = (tx + 7) // 8, t >> 6, t & 63
    return (b % blocks_x) * 8 + (j & 7), (b // blocks_x) * 8 + (j >> 3)
print("costliest tiles:", [tile_xy(int(t)) + (int(n_cut[t]), int(n_hit[t]), round(float(us[t]), 1)) for t in top])
