#!/bin/bash
# Development tool: the record fetches of one counted frame for several builds of librtx.so (EXTRA= flag strings), on one
# box.      WORKLOADS="c3 c5" tools/count_builds.sh "-DRTX_EXPERIMENT_PAIR=1" "-DRTX_EXPERIMENT_PAIR=2"
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/ray-tracer-rust_amd/csrc"
WORKLOADS="${WORKLOADS:-c3 c5}"
mkdir -p "$ROOT/gpurun_out/ab"
i=0
for flags in "$@"; do
    make -s -C "$CSRC" clean; make -s -j4 -C "$CSRC" all EXTRA="$flags" OUT="$ROOT/gpurun_out/ab/librtx_$i.so" > "$ROOT/gpurun_out/ab/build_$i.log" 2>&1 || { echo "build $i failed"; tail -5 "$ROOT/gpurun_out/ab/build_$i.log"; exit 1; }
    i=$((i + 1))
done
n=$i
for wl in $WORKLOADS; do
    for i in $(seq 0 $((n - 1))); do
        cp "$ROOT/gpurun_out/ab/librtx_$i.so" "$ROOT/ray-tracer-rust_amd/librtx.so"
        out=$(timeout -k 10 300 python "$ROOT/bench.py" --workload "$wl" --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-config 2>/dev/null | grep '^{' | python -c 'import sys, json; d = json.loads(sys.stdin.read()); c = d["roofline_valu"]; print({k: c[k] for k in ("box_tests", "tri_tests", "wave_node_visits", "wave_tri_visits")}, d["ms_per_step"])')
        echo "$wl build $i: $out"
    done
done
rm -f "$ROOT"/gpurun_out/ab/librtx_*.so
