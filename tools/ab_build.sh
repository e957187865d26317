#!/bin/bash
# Development tool (not part of the product): A/B two or more builds of librtx.so on ONE GPU box, so that box-to-box
# clock differences cancel.  Each argument is the EXTRA= flag string of a build ("" = default build); every build is
# benched ROUNDS times, interleaved.  Every variant has its own object directory and library under gpurun_out/ab/ and is
# loaded through RTX_PY_LIB: the product's obj/ and librtx.so are never touched.  Each build's compiler output and each
# bench run's stderr are kept (gpurun_out/ab/build_<i>.log, run_<i>.err): a variant that prints no line has said why.
#   tools/ab_build.sh "" "-DRTX_SCALAR_RAY_NUMBERING=0"
#   WORKLOADS="c3 c2" ROUNDS=3 tools/ab_build.sh "" "-DRTX_SHADE_WAVES_PER_SIMD=6"
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/ray-tracer-rust_amd/csrc"
WORKLOADS="${WORKLOADS:-c3 c2 c4}"
ROUNDS="${ROUNDS:-3}"
STEPS="${STEPS:-30}"
AB="$ROOT/gpurun_out/ab"
mkdir -p "$AB"
i=0
for flags in "$@"; do
    rm -rf "$AB/obj_$i"
    make -s -j4 -C "$CSRC" all EXTRA="$flags" OBJ="$AB/obj_$i" OUT="$AB/librtx_$i.so" > "$AB/build_$i.log" 2>&1 || { echo "build $i failed"; tail -5 "$AB/build_$i.log"; exit 1; }
    echo "build $i: EXTRA='$flags'"
    i=$((i + 1))
done
n=$i
for round in $(seq 1 "$ROUNDS"); do
    for wl in $WORKLOADS; do
        for i in $(seq 0 $((n - 1))); do
            line=$(RTX_PY_LIB="$AB/librtx_$i.so" timeout -k 10 300 python "$ROOT/bench.py" --workload "$wl" --steps "$STEPS" --warmup 3 --no-cpu-baseline --no-scaling-config 2> "$AB/run_$i.err" | grep '^{')
            rc=$?
            if [ -z "$line" ]; then
                echo "round $round $wl build $i: NO BENCH LINE (status $rc); stderr:"; tail -5 "$AB/run_$i.err"
                continue
            fi
            ms=$(echo "$line" | python -c 'import sys, json; d = json.loads(sys.stdin.read()); l = d["roofline"]["launch"]; print(d["ms_per_step"], "sched", l["schedule_ms"], "shade", l["shade_ms"])')
            echo "round $round $wl build $i: $ms ms"
        done
    done
done
rm -rf "$AB"/obj_* "$AB"/librtx_*.so
