#!/bin/bash
# Development tool (not part of the product): A/B two or more builds of librtx.so on ONE GPU box, so that box-to-box
# clock differences cancel.  Each argument is the EXTRA= flag string of a build ("" = default build); every build is
# benched ROUNDS times, interleaved.
#   tools/ab_build.sh "" "-DRTX_SCALAR_RAY_NUMBERING=0"
#   WORKLOADS="c3 c2" ROUNDS=3 tools/ab_build.sh "" "-DRTX_SHADE_WAVES_PER_SIMD=6"
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/ray-tracer-rust_amd/csrc"
WORKLOADS="${WORKLOADS:-c3 c2 c4}"
ROUNDS="${ROUNDS:-3}"
STEPS="${STEPS:-30}"
mkdir -p "$ROOT/gpurun_out/ab"
i=0
for flags in "$@"; do
    make -s -C "$CSRC" clean; make -s -j4 -C "$CSRC" all EXTRA="$flags" OUT="$ROOT/gpurun_out/ab/librtx_$i.so" > "$ROOT/gpurun_out/ab/build_$i.log" 2>&1 || { echo "build $i failed"; tail -5 "$ROOT/gpurun_out/ab/build_$i.log"; exit 1; }
    i=$((i + 1))
done
n=$i
for round in $(seq 1 "$ROUNDS"); do
    for wl in $WORKLOADS; do
        for i in $(seq 0 $((n - 1))); do
            cp "$ROOT/gpurun_out/ab/librtx_$i.so" "$ROOT/ray-tracer-rust_amd/librtx.so"
            ms=$(timeout -k 10 300 python "$ROOT/bench.py" --workload "$wl" --steps "$STEPS" --warmup 3 --no-cpu-baseline --no-scaling-config 2>/dev/null | grep '^{' | python -c 'import sys, json; d = json.loads(sys.stdin.read()); l = d["roofline"]["launch"]; print(d["ms_per_step"], "sched", l["schedule_ms"], "shade", l["shade_ms"])')
            echo "round $round $wl build $i: $ms ms"
        done
    done
done
rm -f "$ROOT"/gpurun_out/ab/librtx_*.so
