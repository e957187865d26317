#!/usr/bin/env python3
"""Development tool: frame time against RtxSceneDesc.leaf_max (a run-time field of the product library).
    python tools/leaf_sweep.py [c3|c4|c5] [leaf sizes ...]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
rtx = importlib.import_module("ray-tracer-rust_amd")
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
sizes = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 4, 6, 8]
W, H = (4096, 4096) if wl in ("c4", "c5") else (1920, 1080)
T = rtx.gen_samples()
st = torch.cuda.current_stream().cuda_stream
for leaf in sizes:
    if wl == "c5":
        tris, rgb = rtx.synthetic_primitives(1000000)
        scene = rtx.Scene(W, H, tris, rgb, T, leaf_max=leaf)
    else:
        scene = rtx.default_scene([os.path.join(ROOT, "models", "big_bunny.obj")], W, H, T, leaf_max=leaf)
    scene.upload(0)
    nb = scene.tiles_bytes(0, 1, 8)
    out = torch.zeros(nb, dtype=torch.uint8, device="cuda:0")
    ctr = torch.zeros(8, dtype=torch.int64, device="cuda:0")
    scene.render_tiles_device(0, 0, 1, 8, out.data_ptr(), nb, st, ctr.data_ptr())
    for _ in range(23 if wl != "c5" else 6):
        scene.render_tiles_device(0, 0, 1, 8, out.data_ptr(), nb, st, None)
    torch.cuda.synchronize()
    sched, shade = scene.launch_timings(0, 20 if wl != "c5" else 5)
    c = ctr.cpu().tolist()
    print(json.dumps({"workload": wl, "leaf_max": leaf, "nodes": scene.info()["n_nodes"], "sched_ms": round(float(sched.mean()), 4),
                      "shade_ms": round(float(shade.mean()), 4), "node_fetches": c[3], "prim_fetches": c[4]}), flush=True)
    scene.close()
