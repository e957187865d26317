#!/usr/bin/env python3
"""Development tool (not part of the product or the tests; needs librtx_ablation.so and RTX_PY_ABLATION=1): time kernel variants and build options
on one GPU, each in its own process because RTX_VARIANT / RTX_LEAF_MAX are read once per process.

    python tools/sweep.py variants            # RTX_VARIANT 0..7 on big_bunny 1080p
    python tools/sweep.py leaf                # leaf size / SAH box cost sweep
    python tools/sweep.py waveprof [out.npy]  # per-tile work + residency of the default variant

Every child prints one JSON line; the image hash shows that all variants produce the same bytes.
"""
import hashlib
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(width, height, reps):
    import numpy as np
    rtx = importlib.import_module("ray-tracer-rust_amd")
    s = rtx.default_scene([os.path.join(ROOT, "models", "big_bunny.obj")], width, height, rtx.gen_samples())
    img, st = s.render_rows(stats=True)
    import time
    # plain (un-counted) kernel: wall time of render_rows without stats, includes the 6 MB D2H copy
    s.render_rows()
    t0 = time.perf_counter()
    for _ in range(reps):
        s.render_rows()
    wall = (time.perf_counter() - t0) / reps * 1e3
    info = s.info()
    print(json.dumps({
        "variant": os.environ.get("RTX_VARIANT", "default"), "leaf_max": os.environ.get("RTX_LEAF_MAX", "default"),
        "box_cost": os.environ.get("RTX_SAH_BOX_COST", "default"),
        "counted_kernel_ms": round(st["kernel_ms"], 3), "plain_wall_ms": round(wall, 3),
        "mrays_per_s_wall": round(st["rays"] / wall / 1e3, 1),
        "node_visits": st["wave_node_visits"], "tri_visits": st["wave_tri_visits"],
        "box_tests": st["box_tests"], "tri_tests": st["tri_tests"], "n_nodes": info["n_nodes"],
        "max_leaf": info["max_leaf_tris"], "sha1": hashlib.sha1(img.tobytes()).hexdigest()[:12]}), flush=True)


def run_child(env_extra, width=1920, height=1080, reps=5):
    env = dict(os.environ, RTX_PY_ABLATION="1")   # the variants and build knobs live in librtx_ablation.so
    env.update(env_extra)
    out = subprocess.run([sys.executable, __file__, "child", str(width), str(height), str(reps)], env=env,
                         capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("FAILED", env_extra, out.stderr[-2000:])
        return None
    print(line[-1], flush=True)
    return json.loads(line[-1])


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "variants"
    if mode == "child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    elif mode == "variants":
        for v in range(8):
            run_child({"RTX_VARIANT": str(v)})
    elif mode == "leaf":
        variant = sys.argv[2] if len(sys.argv) > 2 else "3"
        for leaf in (1, 2, 3, 4, 8):
            for cost in ("0.125", "0.25", "0.5", "1.0"):
                if leaf == 1 and cost != "0.25":
                    continue     # the SAH termination cannot matter when every leaf holds one triangle
                run_child({"RTX_LEAF_MAX": str(leaf), "RTX_SAH_BOX_COST": cost, "RTX_VARIANT": variant})
    elif mode == "waveprof":
        import numpy as np
        rtx = importlib.import_module("ray-tracer-rust_amd")
        s = rtx.default_scene([os.path.join(ROOT, "models", "big_bunny.obj")], 1920, 1080, rtx.gen_samples())
        s.render_rows()
        prof = s.wave_profile()
        if len(sys.argv) > 2:
            np.save(sys.argv[2], prof)
        work = prof[..., 0] + prof[..., 1]
        dur = (prof[..., 3] - prof[..., 2]).astype(np.float64) / 100.0     # microseconds
        t0 = prof[..., 2][prof[..., 2] > 0].min()
        end = (prof[..., 3].astype(np.float64) - t0) / 100.0
        start = (prof[..., 2].astype(np.float64) - t0) / 100.0
        print("tiles", prof.shape[:2], "kernel span us", end.max())
        print("steps per tile: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (
            work.mean(), *np.percentile(work, [50, 90, 99]), work.max()))
        print("tile residency us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (
            dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
        busy = work > 0
        print("us per step (tiles with work): p10 %.3f p50 %.3f p90 %.3f" % tuple(
            np.percentile(dur[busy] / work[busy], [10, 50, 90])))
        print("sum of residency (wave-us) %.0f  -> average resident waves %.0f" % (dur.sum(), dur.sum() / end.max()))
        for q in (0.5, 0.8, 0.9, 0.95, 0.99, 1.0):
            print("  %3.0f%% of tiles finished by %.0f us; started by %.0f us" % (
                q * 100, np.quantile(end, q), np.quantile(start, q)))
        ph = prof[..., 4:7].astype(np.float64) / 100.0                      # us: primary, shadow (slowest wave), accumulate
        tot = ph.sum(axis=(0, 1))
        print("phase time summed over tiles (us): primary %.0f  shadow %.0f  accumulate %.0f  | residency %.0f" % (
            tot[0], tot[1], tot[2], dur.sum()))
        heavy = dur > np.percentile(dur, 90)
        th = ph[heavy].sum(axis=0)
        print("heaviest 10%% of tiles: primary %.1f%%  shadow %.1f%%  accumulate %.1f%% of their residency" % tuple(
            100 * th / dur[heavy].sum()))
        shadow_tiles = ph[..., 1] > 0
        print("tiles with shadow work %d: median us primary %.1f shadow %.1f accumulate %.1f" % (
            shadow_tiles.sum(), *np.median(ph[shadow_tiles], axis=0)))
        rows = work.sum(axis=1)
        print("work by tile row (top 10):", np.argsort(rows)[::-1][:10].tolist())


if __name__ == "__main__":
    main()
