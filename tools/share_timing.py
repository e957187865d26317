#!/usr/bin/env python3
"""Development tool: what one rank of an N-GPU run has to do, timed on ONE device.

    python tools/share_timing.py [c3|c4] [steps] [N,N,...]

For N in 1, 2, 4, 8 and every rank r < N it launches rank r's share of the frame (row tiles t = r, r+N, ... of 8 rows,
the partition bench.py uses) `steps` times back to back and reports the device time per launch (HIP events on the
launch's stream) and the host wall time per launch.  max over ranks = the frame time an N-GPU run cannot beat; the
ratio to N = 1 is the strong-scaling ceiling of the partition itself (tile imbalance + the launch's fixed cost), before
any multi-process effect.  One JSON line per N.
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    W, H = (4096, 4096) if wl == "c4" else (1920, 1080)
    rtx = importlib.import_module("ray-tracer-rust_amd")
    scene = rtx.default_scene([os.path.join(ROOT, "models", "big_bunny.obj")], W, H, rtx.gen_samples())
    scene.upload(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    base = None
    worlds = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4, 8]
    for world in worlds:
        per_rank = []
        for rank in range(world):
            nbytes = scene.tiles_bytes(rank, world, 8)
            out = torch.zeros(max(nbytes, 16), dtype=torch.uint8, device=dev)
            for _ in range(3):
                scene.render_tiles_device(0, rank, world, 8, out.data_ptr(), nbytes, stream.cuda_stream, None)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            for _ in range(steps):
                scene.render_tiles_device(0, rank, world, 8, out.data_ptr(), nbytes, stream.cuda_stream, None)
            e1.record(stream)
            enqueue = (time.perf_counter() - t0) / steps * 1e3
            torch.cuda.synchronize(dev)
            wall = (time.perf_counter() - t0) / steps * 1e3
            sched, shade = scene.launch_timings(0, min(steps, 64))
            per_rank.append(dict(ms=e0.elapsed_time(e1) / steps, wall_ms=wall, enqueue_ms=enqueue,
                                 sched_ms=float(sched.mean()), shade_ms=float(shade.mean())))
        worst = max(p["ms"] for p in per_rank)
        if base is None:
            base = worst
        print(json.dumps({
            "workload": wl, "n": world, "frame_ms_max_over_ranks": round(worst, 4),
            "min_over_ranks": round(min(p["ms"] for p in per_rank), 4),
            "speedup_ceiling": round(base / worst, 3), "efficiency_ceiling": round(base / worst / world, 3),
            "sched_ms": round(max(p["sched_ms"] for p in per_rank), 4),
            "shade_ms": round(max(p["shade_ms"] for p in per_rank), 4),
            "host_enqueue_ms_per_launch": round(max(p["enqueue_ms"] for p in per_rank), 4),
            "wall_ms_per_launch": round(max(p["wall_ms"] for p in per_rank), 4)}), flush=True)
    scene.close()


if __name__ == "__main__":
    main()
