#!/bin/bash
# Development tool: A/B of PREBUILT libraries build_ab/librtx_<i>.so (i = 0, 1, ...) on one GPU box, interleaved: bench
# lines of WORKLOADS, and with SHARES="c3:8 c4:8" one rank's share of an N-way frame (tools/share_timing.py).  The libraries are loaded through RTX_PY_LIB; the product's file is not touched.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
WORKLOADS="${WORKLOADS:-c3 c4}"
SHARES="${SHARES:-}"
ROUNDS="${ROUNDS:-2}"
STEPS="${STEPS:-30}"
n=$(ls "$ROOT"/build_ab/librtx_[0-9]*.so | wc -l)
for round in $(seq 1 "$ROUNDS"); do
    for wl in $WORKLOADS; do
        for i in $(seq 0 $((n - 1))); do
            ms=$(RTX_PY_LIB="$ROOT/build_ab/librtx_$i.so" timeout -k 10 300 python "$ROOT/bench.py" --workload "$wl" --steps "$STEPS" --warmup 3 --no-cpu-baseline --no-scaling-config 2>/dev/null | grep '^{' | python -c 'import sys, json; d = json.loads(sys.stdin.read()); l = d["roofline"]["launch"]; print(d["ms_per_step"], "sched", l["schedule_ms"], "shade", l["shade_ms"])')
            echo "round $round $wl lib $i: $ms ms"
        done
    done
    for sh in $SHARES; do
        for i in $(seq 0 $((n - 1))); do
            out=$(RTX_PY_LIB="$ROOT/build_ab/librtx_$i.so" timeout -k 10 300 python "$ROOT/tools/share_timing.py" "${sh%%:*}" "$STEPS" "${sh##*:}" 2>/dev/null | grep '^{' | python -c 'import sys, json; d = json.loads(sys.stdin.read()); print(d["frame_ms_max_over_ranks"], "sched", d["sched_ms"], "shade", d["shade_ms"])')
            echo "round $round share $sh lib $i: $out ms"
        done
    done
done
