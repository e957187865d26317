#!/bin/bash
# Development tool: the record fetches of one counted frame for several builds of librtx.so (EXTRA= flag strings), on one
# box.  Builds live under gpurun_out/ab/ and are loaded through RTX_PY_LIB (the product's files are not touched).
#   WORKLOADS="c3 c5" tools/count_builds.sh "" "-DRTX_HOME_FIRST=0"
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/ray-tracer-rust_amd/csrc"
WORKLOADS="${WORKLOADS:-c3 c5}"
AB="$ROOT/gpurun_out/ab"
mkdir -p "$AB"
i=0
for flags in "$@"; do
    rm -rf "$AB/obj_$i"
    make -s -j4 -C "$CSRC" all EXTRA="$flags" OBJ="$AB/obj_$i" OUT="$AB/librtx_$i.so" > "$AB/build_$i.log" 2>&1 || { echo "build $i failed"; tail -5 "$AB/build_$i.log"; exit 1; }
    i=$((i + 1))
done
n=$i
for wl in $WORKLOADS; do
    for i in $(seq 0 $((n - 1))); do
        out=$(RTX_PY_LIB="$AB/librtx_$i.so" timeout -k 10 300 python "$ROOT/bench.py" --workload "$wl" --steps 5 --warmup 2 --no-cpu-baseline --no-scaling-config 2> "$AB/run_$i.err" | grep '^{' | python -c 'import sys, json; d = json.loads(sys.stdin.read()); c = d["roofline_valu"]; print({k: c[k] for k in ("box_tests", "tri_tests", "wave_node_visits", "wave_tri_visits")}, d["ms_per_step"])')
        echo "$wl build $i: $out"
    done
done
rm -rf "$AB"/obj_* "$AB"/librtx_*.so
