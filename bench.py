#!/usr/bin/env python3
"""bench.py — headline benchmark: Mrays/s + frame ms, big_bunny.obj @ 1920x1080 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU)

A step = one whole frame of the hot path (primary rays + 100 shadow rays per hit pixel + gamma/RGB8
store).  With N ranks the frame is split into row tiles dealt round-robin (tile t -> rank t % N,
SURVEY.md §8(e)); there is no data-path collective, RCCL is used only for the barrier and for the
max-over-ranks of the elapsed time.  Total work is fixed as N grows: "scaling": "strong".

Inputs (triangles, BVH stream, sample table) are resident in HBM before the timed region; the
output stays in HBM (torch tensor).  torch is plumbing here: device memory, stream, events,
torch.distributed.  Every ray is traced by the HIP kernels behind the C ABI (include/rtx.h).

One JSON line on stdout (rank 0).  Besides the contract fields:
  roofline       the roof that BINDS the workload, as SURVEY 8(d) asks: FP32 VALU for the OBJ configurations (their
                 records, 0.6 MB, live in L2 and the scalar cache), HBM for the synthetic 1M-triangle mesh (89 MB of
                 records) — a copy of roofline_valu resp. roofline_hbm, which are both always printed.  roofline_hbm: the
                 dominant kernel (shade_tiles_kernel, the shading pass of a launch) timed by the library's HIP events
                 around that pass on the launch's stream (rtx_launch_timings), algorithmic bytes as in DESIGN.md section
                 5; `traffic` = HBM bytes of that kernel from the PMC passes of an EARLIER profiled run (traffic_source
                 says which), not of this run.
  per_rank       every rank's own device times per launch (kernel, scheduling pass, shading pass): SURVEY 8(e).
  cpu_baseline   the CPU oracle in faithful-BVH mode (= the reference's src/tracer algorithm, "port") timed on this
                 box's host cores on a bounded sample of the same frame, with the reference's thread policy
                 (num_cpus - 1, src/main.rs:269) applied to the cores this job may use.
  parity         the rows the cpu_baseline leg rendered, compared byte for byte with the frame the LAST TIMED STEP
                 left in HBM: the throughput number belongs to the right image.
  frame_ms_incl_d2h   the same frame through rtx_render_rows into pinned host memory (the reference's timed region
                 ends with the pixels in host memory, src/main.rs:291-295,305); never `value`.
  host           what the CPU numbers were measured on.
  scaling_config BASELINE configs[3] (big_bunny.obj 4096x4096, the configuration of the scaling curve), the same
                 partition and timing, fewer steps.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (first: its bundled HIP runtime must be the one in the process)
import numpy as np  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[]; the metric is quoted on configs[2]
    "c2": dict(desc="configs[1]: bunny.obj 1920x1080", objs=["bunny.obj"], width=1920, height=1080),
    "c3": dict(desc="configs[2]: big_bunny.obj 1920x1080", objs=["big_bunny.obj"], width=1920, height=1080),
    "c4": dict(desc="configs[3]: big_bunny.obj 4096x4096", objs=["big_bunny.obj"], width=4096, height=4096),
    "c1b": dict(desc="big_bunny.obj 256x256 (test size)", objs=["big_bunny.obj"], width=256, height=256),
    # configs[4]: the reference's O(n^2) BVH build is infeasible at 10^6 primitives: its cpu_baseline is the
    # oracle's leaf-gated brute force on a fixed 64x64 centre crop
    "c5": dict(desc="configs[4]: synthetic 1M-triangle random mesh 4096x4096", objs=[], synthetic=1000000,
               width=4096, height=4096),
    "c5s": dict(desc="synthetic 100k-triangle random mesh 1024x1024 (reduced configs[4])", objs=[], synthetic=100000,
                width=1024, height=1024),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3       # FP32 vector peak = 1024 SIMD-32 x 32 lanes x 2 flop (FMA) x 2.4 GHz: reached with plain v_fma_f32
FLOP_PER_TRI_TEST = 46         # full Möller–Trumbore path incl. the division
FLOP_PER_BOX_TEST = 29         # conservative box test: 6 fma (12), 6 min/max, max3+min3 (4), 2 compares, the rest of the vote
BYTES_PER_HIT_REC = 48         # p_hit, normal, colour: written by probe_kernel, read once by shade_tiles_kernel
BYTES_PER_TRI_REC = 36         # v0,e1,e2 consumed per test
BYTES_PER_BOX_REC = 24         # bmin,bmax
BYTES_PER_PIXEL_IO = 11        # 8 B of the sample table + 3 B framebuffer
N_SIMD, N_CU, PEAK_GHZ = 1024, 256, 2.4


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--tile-rows", type=int, default=8)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = the reference's policy: usable cores - 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scaling-config", action="store_true", help="skip the configs[3] line inside the JSON")
    ap.add_argument("--save-png", default="")
    # rehearsal of the N > 1 path on a one-GPU box: every rank on cuda:0, gloo for the barrier and the reductions
    # (RCCL refuses two ranks on one device).  Not a measurement: the ranks share the device.
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--ranks-share-device-0", action="store_true")
    return ap.parse_args()


def host_info():
    """nproc, the cores this job may actually use (affinity and the cgroup's CPU quota), CPU model."""
    nproc = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = nproc
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            a, b = f.read().split()
            if a != "max":
                quota = float(a) / float(b)
    except (OSError, ValueError):
        pass
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    usable = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    return {"nproc": nproc, "affinity": affinity, "cgroup_cpu_quota": quota, "usable_cores": usable, "cpu_model": model}


def cpu_baseline(wl, samples_np, seconds, threads, host, rtx):
    """Oracle on a bounded sample of the same frame.  OBJ workloads: faithful BVH (= the reference's algorithm) on
    2-row bands spread over the frame; returns the rows it rendered so that the caller can compare them with the GPU
    frame.  Synthetic workloads: leaf-gated brute force on a fixed 64x64 centre crop."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    if threads <= 0:
        threads = max(1, host["usable_cores"] - 1)          # src/main.rs:269: num_cpus::get() - 1 worker threads
    W, H = wl["width"], wl["height"]
    rendered = []                                            # (x0, y0, image block)
    rays, secs = 0, 0.0
    if wl.get("synthetic"):
        tris, rgb = rtx.synthetic_primitives(wl["synthetic"])
        sc = orclib.Scene(W, H, tris, rgb, samples_np, build_bvh=False)
        side = 64
        x0, y0 = W // 2 - side // 2, H // 2 - side // 2
        done = 0
        for r in range(side):                                # a row of the crop at a time, until the budget is spent
            t0 = time.perf_counter()
            img, st = sc.render_window(x0, y0 + r, side, 1, mode=orclib.MODE_LEAFBOX, nthreads=threads)
            secs += time.perf_counter() - t0
            rays += st["primary_rays"] + st["shadow_rays"]
            rendered.append((x0, y0 + r, img))
            done += 1
            if secs >= seconds:
                break
        sample = ("%d of 64 rows of the fixed 64x64 centre crop (x %d.., y %d..), oracle leaf-gated brute force (the "
                  "reference's O(n^2) tree cannot be built at this size; equal to its result for every ray without a "
                  "-0.0 direction component), %d threads, %.1f s, %d rays; an EXTRAPOLATION of what the reference would "
                  "do if it could build its tree" % (done, x0, y0, threads, secs, rays))
        kind_note = "port-brute"
    else:
        sc = orclib.default_scene(wl["objs"], W, H, samples_np)
        band = 2
        n_bands = H // band
        bits = max(1, (n_bands - 1).bit_length())
        # bit-reversed band order: wherever the time budget stops it, the sample is spread evenly over the frame
        order = sorted(range(n_bands), key=lambda b: int(format(b, "0%db" % bits)[::-1], 2))
        rows = 0
        for b in order:
            t0 = time.perf_counter()
            img, st = sc.render_rows(b * band, band, mode=orclib.MODE_BVH, nthreads=threads)
            secs += time.perf_counter() - t0
            rays += st["primary_rays"] + st["shadow_rays"]
            rendered.append((0, b * band, img))
            rows += band
            if secs >= seconds:
                break
        sample = ("%d of %d rows of the same frame (2-row bands in bit-reversed order, evenly spread), oracle "
                  "faithful-BVH mode, %d threads, %.1f s, %d rays" % (rows, H, threads, secs, rays))
        kind_note = "port"
    sc.close()
    return {
        "value": float("%.6g" % (rays / secs / 1e6)), "unit": "Mrays/s", "cores": host["usable_cores"], "threads": threads,
        "kind": "port", "mode": kind_note, "thread_policy": "usable cores - 1 (src/main.rs:269: num_cpus - 1)",
        "sample": sample,
    }, rendered


def reduce_counters(counters, world, via_host=False):
    """Sum the per-rank kernel counters (each rank counted only its own row tiles)."""
    if via_host:
        counters = counters.cpu()
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
    return counters.cpu().tolist()


def reduce_times(values, world, device):
    """MAX over ranks of the timed region and of the per-launch device times, and every rank's own values (SURVEY 8(e):
    per-device kernel time, to expose imbalance): -> (max list, [per-rank list, ...])."""
    tt = torch.tensor(list(values), dtype=torch.float64, device=device)
    if world > 1:
        import torch.distributed as dist
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        per_rank = [t.cpu().tolist() for t in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    else:
        per_rank = [tt.cpu().tolist()]
    return tt.cpu().tolist(), per_rank


class Run:
    """One workload on this rank's share of the frame: scene in HBM, counted launch, timed launches."""

    def __init__(self, rtx, wl, samples, rank, world, local_rank, dev, tile_rows):
        self.rtx, self.wl, self.rank, self.world, self.local_rank, self.dev, self.tile_rows = rtx, wl, rank, world, local_rank, dev, tile_rows
        W, H = wl["width"], wl["height"]
        if wl.get("synthetic"):
            tris, rgb = rtx.synthetic_primitives(wl["synthetic"])
            self.scene = rtx.Scene(W, H, tris, rgb, samples)
            self.asset = "synthetic mesh (splitmix64 seed %d)" % rtx.SYNTHETIC_SEED
        else:
            self.scene = rtx.default_scene([os.path.join(ROOT, "models", o) for o in wl["objs"]], W, H, samples)
            self.asset = "reference asset models/%s" % wl["objs"][0]
        self.info = self.scene.info()
        self.scene.upload(local_rank)                      # inputs resident in HBM before timing
        self.nbytes = self.scene.tiles_bytes(rank, world, tile_rows)
        self.out = torch.zeros(max(self.nbytes, 16), dtype=torch.uint8, device=dev)
        self.counters = torch.zeros(8, dtype=torch.int64, device=dev)
        self.stream = torch.cuda.current_stream(dev)

    def step(self, count=False):
        self.scene.render_tiles_device(self.local_rank, self.rank, self.world, self.tile_rows, self.out.data_ptr(), self.nbytes,
                                       self.stream.cuda_stream, self.counters.data_ptr() if count else None)

    def barrier(self):
        torch.cuda.synchronize(self.dev)
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(self.dev)

    def counted(self, via_host):
        self.step(count=True)                               # one counted launch, outside the timed region
        torch.cuda.synchronize(self.dev)
        return reduce_counters(self.counters, self.world, via_host)

    def timed(self, steps, warmup, via_host):
        for _ in range(warmup):
            self.step()
        self.barrier()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for e0, e1 in evs:
            e0.record(self.stream)
            self.step()
            e1.record(self.stream)
        self.barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / max(1, steps)   # this rank's launches
        # the launch is two passes; the library brackets them with HIP events on the launch's stream
        # (rtx_launch_timings): these are the timed region's own launches, newest last
        sched_t, shade_t = self.scene.launch_timings(self.local_rank, min(steps, 64))
        sched_ms = float(sched_t.mean()) if len(sched_t) else 0.0
        shade_ms = float(shade_t.mean()) if len(shade_t) else kernel_ms
        return reduce_times((elapsed, kernel_ms / 1e3, sched_ms / 1e3, shade_ms / 1e3), self.world,
                            torch.device("cpu") if via_host else self.dev)

    def frame(self):
        """This rank's share of the frame, as the last launch left it in HBM, scattered into a whole frame."""
        W, H = self.wl["width"], self.wl["height"]
        frame = np.zeros((H, W, 3), np.uint8)
        self.rtx.scatter_tiles(frame, self.out[:self.nbytes].cpu().numpy().reshape(-1, W, 3), self.rank, self.world, self.tile_rows)
        return frame

    def incl_d2h_ms(self, steps):
        """Whole frame through rtx_render_rows into pinned host memory: launch + D2H + synchronise, per frame."""
        W, H = self.wl["width"], self.wl["height"]
        pinned = torch.empty(H * W * 3, dtype=torch.uint8, pin_memory=True)
        for _ in range(2):
            self.scene.render_rows(0, H, device=self.local_rank, out_ptr=pinned.data_ptr())
        t0 = time.perf_counter()
        for _ in range(steps):
            self.scene.render_rows(0, H, device=self.local_rank, out_ptr=pinned.data_ptr())
        return (time.perf_counter() - t0) / steps * 1e3, pinned.numpy().reshape(H, W, 3)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.ranks_share_device_0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    via_host = args.backend == "gloo"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if via_host:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    rtx = importlib.import_module("ray-tracer-rust_amd")
    wl = WORKLOADS[args.workload]
    W, H = wl["width"], wl["height"]
    samples = rtx.gen_samples()
    host = host_info()
    run = Run(rtx, wl, samples, rank, world, local_rank, dev, args.tile_rows)
    info = run.info

    c = run.counted(via_host)
    primary_hits, box_tests, tri_tests, node_visits, tri_visits = c[0], c[1], c[2], c[3], c[4]
    sched_node_visits, sched_tri_visits = c[6], c[7]       # the scheduling pass's (primary rays') share of the fetches
    primary_rays = W * H * rtx.NB_RAY
    r_total = primary_rays + rtx.NB_LIGHT_SAMPLE * primary_hits

    (elapsed, kernel_s, sched_s, shade_s), per_rank_times = run.timed(args.steps, args.warmup, via_host)
    ms_per_step = elapsed / args.steps * 1e3
    gpu_frame = run.frame() if world == 1 else None         # what the last timed step left in HBM

    if args.save_png and world == 1:
        rtx.write_png(args.save_png, gpu_frame)

    incl_d2h = None
    if world == 1:
        d2h_ms, host_frame = run.incl_d2h_ms(max(3, min(args.steps, 10)))
        incl_d2h = {"frame_ms_incl_d2h": round(d2h_ms, 4),
                    "what": "rtx_render_rows of the whole frame into pinned host memory: launch + %d-byte D2H + "
                            "synchronise, wall clock per frame" % (W * H * 3),
                    "same_bytes_as_timed_frame": bool(np.array_equal(host_frame, gpu_frame))}

    # BASELINE configs[3] — the configuration of the scaling curve — inside the same line
    scaling = None
    if args.workload == "c3" and not args.no_scaling_config:
        wl4 = WORKLOADS["c4"]
        run4 = Run(rtx, wl4, samples, rank, world, local_rank, dev, args.tile_rows)
        c4 = run4.counted(via_host)
        steps4 = max(3, args.steps // 2)
        (e4, k4, s4, h4), per_rank4 = run4.timed(steps4, min(args.warmup, 2), via_host)
        r4 = wl4["width"] * wl4["height"] * rtx.NB_RAY + rtx.NB_LIGHT_SAMPLE * c4[0]
        ms4 = e4 / steps4 * 1e3
        scaling = {"workload": wl4["desc"], "n_gpus": world, "steps": steps4, "ms_per_step": round(ms4, 4),
                   "value": round(r4 / (ms4 / 1e3) / 1e6, 3), "unit": "Mrays/s", "rays_per_frame": r4,
                   "schedule_ms": round(s4 * 1e3, 4), "shade_ms": round(h4 * 1e3, 4),
                   "per_rank": [{"rank": i, "kernel_ms": round(t[1] * 1e3, 4), "schedule_ms": round(t[2] * 1e3, 4),
                                 "shade_ms": round(t[3] * 1e3, 4)} for i, t in enumerate(per_rank4)]}
        run4.scene.close()

    if rank == 0:
        mrays = r_total / (ms_per_step / 1e3) / 1e6
        # roofline of the dominant kernel = shade_tiles_kernel (the shading pass).  Per launch = per rank; counts are
        # whole-frame sums, so divide by world.  Its algorithmic bytes: the records its walks consume, the hit record
        # of every hit pixel (written by the scheduling pass, read once), the framebuffer.
        launch_alg_bytes = (tri_visits * BYTES_PER_TRI_REC + node_visits * BYTES_PER_BOX_REC + BYTES_PER_PIXEL_IO * W * H) / world
        alg_bytes = ((tri_visits - sched_tri_visits) * BYTES_PER_TRI_REC + (node_visits - sched_node_visits) * BYTES_PER_BOX_REC +
                     BYTES_PER_HIT_REC * primary_hits + 3 * W * H) / world
        ach_gbs = alg_bytes / shade_s / 1e9
        flops = (tri_tests * FLOP_PER_TRI_TEST + box_tests * FLOP_PER_BOX_TEST) / world
        ach_tf = flops / kernel_s / 1e12
        # SURVEY §8(d) brute-force-equivalent HBM-level bytes (G = 256 rays share a staged record)
        b_alg_brute = (-(-r_total // 256)) * info["n_tris"] * 36 + 11 * W * H
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
                tr = tj.get(args.workload, {})
                # of the dominant kernel when the profile separates the kernels of a launch, else of the launch
                traffic = tr.get("hbm_bytes_shade_kernel_n%d" % world) or tr.get("hbm_bytes_per_launch_n%d" % world)
                if traffic is not None:
                    traffic_source = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an EARLIER run of this workload (%s), not of this run" % (
                        tr.get("profiled") or tj.get("profiled") or "profiles/traffic.json")
        pmc = None      # SQ counters of an earlier profiled run of the same workload (profiles/pmc_summary.json)
        ppath = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if os.path.exists(ppath) and world == 1:
            with open(ppath) as f:
                pmc = json.load(f).get(args.workload)
        line = {
            "metric": "Mrays/sec + frame ms, big_bunny.obj @ 1920x1080" if args.workload == "c3"
                      else "Mrays/sec + frame ms, " + wl["desc"],
            "value": round(mrays, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32",
            "data": "%s + seeded sample table (splitmix64 seed %d); no dataset download" % (run.asset, rtx.DEFAULT_SEED),
            "config": {"workload": wl["desc"] + ", 1 spp, 100 light samples, default scene of src/main.rs:327-358",
                       "width": W, "height": H, "n_tris": info["n_tris"], "tile_rows": args.tile_rows,
                       "partition": "row tiles, tile t -> rank t %% %d, no collective" % world +
                                    (" (REHEARSAL: all ranks on one device)" if args.ranks_share_device_0 and world > 1 else ""),
                       "accel": "sah-bvh leaf<=%d, %d nodes" % (info["max_leaf_tris"], info["n_nodes"])},
            "frame_ms": round(ms_per_step, 4),
            "rays_per_frame": r_total, "primary_hits": primary_hits,
            "primary_mrays_per_s": round(primary_rays / (ms_per_step / 1e3) / 1e6, 3),
            "roofline": None,        # the roof that binds this workload: filled in below
            "roofline_hbm": {"bound": "hbm", "achieved": round(ach_gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(ach_gbs / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_source,
                             "kernel": "shade_tiles_kernel", "kernel_ms": round(shade_s * 1e3, 4),
                             "algorithmic_bytes_per_launch": int(alg_bytes),
                             "launch": {"passes": "probe_kernel + order_tiles_kernel (scheduling), shade_tiles_kernel + "
                                                  "reference_tiles_kernel (shading)",
                                        "schedule_ms": round(sched_s * 1e3, 4), "shade_ms": round(shade_s * 1e3, 4),
                                        "launch_ms": round(kernel_s * 1e3, 4),
                                        "survey_8d_bytes": int(launch_alg_bytes),
                                        "survey_8d_gbs": round(launch_alg_bytes / kernel_s / 1e9, 3)},
                             "note": "scene records %.2f MB (L2-resident when < 4 MB); survey_b_alg_* = SURVEY 8(d) "
                                     "brute-force-equivalent bytes" % ((info["node_bytes"] + info["tri_bytes"]) / 1e6),
                             "redo_tiles": c[5],
                             "survey_b_alg_bytes": int(b_alg_brute),
                             "survey_b_alg_frac": round(b_alg_brute / world / kernel_s / 1e9 / HBM_PEAK_GBS, 4)},
            "roofline_valu": {"bound": "valu", "achieved": round(ach_tf, 3), "peak": VALU_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": round(ach_tf / VALU_PEAK_TFLOPS, 5), "traffic": traffic,
                              "traffic_source": traffic_source,
                              "kernel": "the launch (probe_kernel + shade_tiles_kernel: the counters do not separate their tests)",
                              "kernel_ms": round(kernel_s * 1e3, 4),
                              "algorithmic_flop_per_launch": int(flops),
                              # MI355X_MICROARCH.md: SIMD-32, a wave64 v_fma_f32 issues in 2 cycles; 157.3 TFLOP/s is reached
                              # with plain FMAs.  Pixel arithmetic here may not fuse (bit parity with the reference), so
                              # one flop per lane-slot: 78.6 Tflop/s is the ceiling of this instruction mix
                              "peak_unfused": VALU_PEAK_TFLOPS / 2,
                              "frac_unfused": round(ach_tf / (VALU_PEAK_TFLOPS / 2), 5),
                              "pmc": pmc,
                              "box_tests": box_tests, "tri_tests": tri_tests,
                              "wave_node_visits": node_visits, "wave_tri_visits": tri_visits},
            "per_rank": [{"rank": i, "kernel_ms": round(t[1] * 1e3, 4), "schedule_ms": round(t[2] * 1e3, 4),
                          "shade_ms": round(t[3] * 1e3, 4)} for i, t in enumerate(per_rank_times)],
            "host": host,
        }
        # `roofline` := whichever roof binds — FP32 VALU for the OBJ configurations (their records are
        # L2-resident: 0.6 MB), HBM for the synthetic 1M-triangle mesh (89 MB of records) —, the other always beside it
        binding = "roofline_hbm" if wl.get("synthetic") else "roofline_valu"
        line["roofline"] = dict(line[binding], other="roofline_valu" if binding == "roofline_hbm" else "roofline_hbm")
        # (the ab tools read the two passes' times from the line's roofline block whichever roof it is)
        line["roofline"]["launch"] = line["roofline_hbm"]["launch"]
        if incl_d2h:
            line.update(incl_d2h)
        if scaling:
            line["scaling_config"] = scaling
        if world == 1 and not args.no_cpu_baseline:
            base, rendered = cpu_baseline(wl, samples, args.cpu_seconds, args.cpu_threads, host, rtx)
            base["gpu_over_cpu"] = round(mrays / base["value"], 1)
            line["cpu_baseline"] = base
            # the oracle's pixels against the frame of the last timed step
            px_diff, max_abs, n_px = 0, 0, 0
            for x0, y0, img in rendered:
                got = gpu_frame[y0:y0 + img.shape[0], x0:x0 + img.shape[1]]
                d = np.abs(got.astype(np.int32) - img.astype(np.int32))
                px_diff += int((d.max(axis=2) > 0).sum())
                max_abs = max(max_abs, int(d.max()))
                n_px += img.shape[0] * img.shape[1]
            line["parity"] = {"rows" if not wl.get("synthetic") else "crop_rows": len(rendered) * (1 if wl.get("synthetic") else 2),
                              "pixels": n_px, "max_abs_diff": max_abs, "px_diff": px_diff,
                              "checked": "oracle pixels of the cpu_baseline sample vs the frame the last timed step left in HBM"}
        else:
            line["cpu_baseline"] = None
            line["parity"] = None
        print(json.dumps(line), flush=True)

    run.scene.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
