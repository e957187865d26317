#!/usr/bin/env python3
"""Development tool: turn the CSVs of tools/collect_profiles.sh into the two small summaries bench.py reads,
profiles/pmc_summary.json and profiles/traffic.json, and copy the evidence into profiles/<round>/.

    python tools/summarise_profiles.py gpurun_out/prof_j c3 profiles/r01 j

Counters are averaged over the dispatches of the uncounted kernels (template argument COUNT = false) of the
profiled run; one launch = reset_kernel + probe_kernel + count_classes_kernel + order_tiles_kernel + shade_tiles_kernel + reference_tiles_kernel.
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950
(MI355X_MICROARCH.md), so the x2 figure is used as the upper bound, WRITE_SIZE as it is.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

KERNELS = {"reset_kernel": "reset", "probe_kernel<false": "probe", "order_tiles_kernel": "order", "count_classes_kernel": "count", "shade_tiles_kernel<false": "shade",
           "reference_tiles_kernel<false": "redo", "trace_shade_kernel<false": "fused"}
N_SIMD = 1024               # 256 CUs x 4
N_CU = 256
VALU_CYCLES = 2.0           # cycles a wave64 vector instruction occupies a SIMD-32 (MI355X_MICROARCH.md)
N_XCD = 8                   # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of the dispatch = value / 8


def which(name):
    for k, v in KERNELS.items():
        if ("rtx::" + k) in name:
            return v
    return None


def load_counters(path):
    """-> {kernel: {counter: mean value per dispatch}}, {kernel: mean duration ns}"""
    acc, dur = {}, {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = which(r["Kernel_Name"])
            if not k:
                continue
            acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            dur.setdefault(k, {})[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    mean = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
    mdur = {k: sum(d.values()) / len(d) for k, d in dur.items()}
    return mean, mdur


def main():
    src, wl, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(dst, exist_ok=True)
    counters, durations = {}, {}
    paths = sorted(glob.glob(os.path.join(src, "pmc*", "%s_counter_collection.csv" % wl))) or \
        sorted(glob.glob(os.path.join(src, "pmc*", "*counter_collection.csv")))
    for i, path in enumerate(paths):
        c, d = load_counters(path)
        for k, v in c.items():
            counters.setdefault(k, {}).update(v)
        for k, v in d.items():
            durations.setdefault(k, []).append(v)
        group = "_".join(sorted({n for v in c.values() for n in v}))[:60].lower()
        shutil.copy(path, os.path.join(dst, "%s_pmc_%s_%s.csv" % (tag, wl, group)))
    stats = glob.glob(os.path.join(src, "stats", "%s_kernel_stats.csv" % wl)) or glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, wl)))
    bench = os.path.join(src, "bench_%s.json" % wl)
    if os.path.exists(bench):
        shutil.copy(bench, os.path.join(dst, "%s_bench_%s.json" % (tag, wl)))

    summary = {}
    for k, c in counters.items():
        if "SQ_INSTS_VALU" not in c:
            continue
        ms = sum(durations[k]) / len(durations[k]) / 1e6
        e = {"kernel_ms_under_pmc": round(ms, 4),
             "sq_insts_valu": c.get("SQ_INSTS_VALU"), "sq_insts_salu": c.get("SQ_INSTS_SALU"),
             "sq_insts_smem": c.get("SQ_INSTS_SMEM"), "sq_insts_lds": c.get("SQ_INSTS_LDS"),
             "sq_waves": c.get("SQ_WAVES"),
             "sq_wait_any": c.get("SQ_WAIT_ANY"), "sq_wait_inst_any": c.get("SQ_WAIT_INST_ANY"),
             "sq_wave_cycles": c.get("SQ_WAVE_CYCLES")}
        if c.get("SQC_DCACHE_REQ"):
            e["sqc_dcache_hit_rate"] = round(c.get("SQC_DCACHE_HITS", 0.0) / c["SQC_DCACHE_REQ"], 4)
        if c.get("SQ_INSTS_VALU"):
            # Machine model (MI355X_MICROARCH.md): a SIMD is 32 lanes wide, a wave64 vector instruction occupies it for
            # 2 cycles (quarter-rate ones — v_rcp/v_sqrt/... — for 8: not separable from these counters, so the busy
            # figures below are LOWER bounds by the few per cent of such instructions); the scalar unit issues one
            # instruction per cycle per compute unit.
            if c.get("SQ_ACTIVE_INST_SCA"):
                e["sq_active_inst_sca"] = c["SQ_ACTIVE_INST_SCA"]
            e["valu_busy_at_2p4_ghz"] = round(VALU_CYCLES * c["SQ_INSTS_VALU"] / (ms * 1e-3 * 2.4e9 * N_SIMD), 4)
            if c.get("SQ_INSTS_SALU") is not None:
                e["salu_per_cu_per_cycle_at_2p4_ghz"] = round((c["SQ_INSTS_SALU"] + (c.get("SQ_INSTS_SMEM") or 0.0)) /
                                                              (ms * 1e-3 * 2.4e9 * N_CU), 4)
            if ms < 0.05:
                # a dispatch of a few microseconds: its wall time is mostly launch overhead and GRBM_GUI_ACTIVE / 8 over it
                # gives "clocks" of 5-7 GHz (MI355X_MICROARCH.md: the quotient reads high below ~0.3 ms); no clock, no
                # busy figures for such a kernel — the raw counts above are what there is
                e["note"] = "dispatch under 50 us: no derived clock or busy figures"
                for key in ("valu_busy_at_2p4_ghz", "salu_per_cu_per_cycle_at_2p4_ghz"):
                    e.pop(key, None)
            elif c.get("GRBM_GUI_ACTIVE"):
                # GRBM_GUI_ACTIVE / 8 = shader-clock cycles of the dispatch (the GRBM pass ran separately)
                cycles = c["GRBM_GUI_ACTIVE"] / N_XCD
                e["gpu_cycles"] = cycles
                e["shader_clock_ghz"] = round(cycles / (ms * 1e6), 3)
                e["valu_busy"] = round(VALU_CYCLES * c["SQ_INSTS_VALU"] / (cycles * N_SIMD), 4)
                if c.get("SQ_INSTS_SALU") is not None:
                    e["salu_per_cu_per_cycle"] = round((c["SQ_INSTS_SALU"] + (c.get("SQ_INSTS_SMEM") or 0.0)) / (cycles * N_CU), 4)
                    e["salu_per_valu"] = round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 4)
            if c.get("SQ_WAVE_CYCLES"):
                # of the wavefronts' resident time: parked on a wait (s_waitcnt, barrier) / ready but not issued
                for name, key in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active_inst_any", "SQ_ACTIVE_INST_ANY")):
                    if c.get(key) is not None:
                        e["frac_" + name] = round(c[key] / c["SQ_WAVE_CYCLES"], 4)   # both count in units of 4 cycles
        summary[k] = e
    pmc_path = os.path.join(root, "profiles", "pmc_summary.json")
    allp = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
    allp["_comment"] = ("SQ counters per dispatch of the uncounted kernels (rocprofv3 --pmc, one MI355X, averaged over the "
                        "profiled launches; sources profiles/<round>/<tag>_pmc_*.csv; tools/summarise_profiles.py). "
                        "valu_busy = SQ_INSTS_VALU x 2 cycles (SIMD-32: a wave64 instruction occupies the SIMD for 2) / "
                        "(shader cycles of the dispatch x 1024 SIMDs); salu_per_cu_per_cycle = (SQ_INSTS_SALU + SQ_INSTS_SMEM) / "
                        "(cycles x 256 CUs): the scalar unit issues one per cycle per CU; frac_* = share of the wavefronts' "
                        "resident cycles.  *_at_2p4_ghz: against the peak clock when no GRBM pass is there (lower bounds).")
    allp[wl] = {"source": "%s/%s_pmc_%s_*.csv" % (os.path.relpath(dst, root), tag, wl), "kernels": summary}
    json.dump(allp, open(pmc_path, "w"), indent=1)

    traffic_path = os.path.join(root, "profiles", "traffic.json")
    allt = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
    fetch = {k: c.get("FETCH_SIZE") for k, c in counters.items() if c.get("FETCH_SIZE") is not None}
    write = {k: c.get("WRITE_SIZE") for k, c in counters.items() if c.get("WRITE_SIZE") is not None}
    if fetch and write:
        launch = ("probe", "count", "order", "shade", "redo") if "shade" in fetch else ("fused", "redo")
        f_kib = sum(fetch.get(k, 0.0) for k in launch)
        w_kib = sum(write.get(k, 0.0) for k in launch)
        allt["_comment"] = ("HBM bytes per launch (all kernels of one launch) from separate rocprofv3 --pmc FETCH_SIZE / "
                            "--pmc WRITE_SIZE runs of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`; KiB; "
                            "FETCH_SIZE x2 (gfx950 under-reports wide reads by 2x: upper bound), WRITE_SIZE as reported.")
        allt[wl] = {"fetch_size_kib_raw": {k: round(v, 1) for k, v in fetch.items()},
                    "write_size_kib_raw": {k: round(v, 1) for k, v in write.items()},
                    "hbm_bytes_per_launch_n1": int((2.0 * f_kib + w_kib) * 1024),
                    "hbm_bytes_shade_kernel_n1": int((2.0 * fetch.get("shade", 0.0) + write.get("shade", 0.0)) * 1024),
                    "source": "%s/%s_pmc_%s_{fetch_size,write_size}.csv" % (os.path.relpath(dst, root), tag, wl),
                    "profiled": "commit %s, profiles/%s" % (
                        subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?",
                        os.path.relpath(dst, os.path.join(root, "profiles")))}
        json.dump(allt, open(traffic_path, "w"), indent=1)
    print(json.dumps({"pmc": summary, "traffic": allt.get(wl)}, indent=1))


if __name__ == "__main__":
    main()
