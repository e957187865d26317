#!/bin/bash
# usage: tools/_tl_run.sh <workload>   (timeline experiment build + run, with a watchdog that prints where python hangs)
cd "$(dirname "$0")/.."
rm -rf gpurun_out/ab/obj_tl && make -s -j4 -C ray-tracer-rust_amd/csrc all EXTRA="${TL_FLAGS:--DRTX_EXPERIMENT_TIMELINE=1}" OBJ="$PWD/gpurun_out/ab/obj_tl" OUT="$PWD/gpurun_out/ab/librtx_tl.so" > gpurun_out/tl_build.log 2>&1
export RTX_PY_LIB="$PWD/gpurun_out/ab/librtx_tl.so"
for wl in "$@"; do
timeout -k 5 150 python -u -c "
import faulthandler, sys; faulthandler.dump_traceback_later(90, exit=True)
sys.argv=['tools/tile_timeline.py','$wl'] + '${TL_ARGS:-}'.split()
__file__='tools/tile_timeline.py'
exec(open('tools/tile_timeline.py').read())
" > gpurun_out/r2_timeline_$wl.txt 2>&1
tail -30 gpurun_out/r2_timeline_$wl.txt
done
