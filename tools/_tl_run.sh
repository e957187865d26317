#!/bin/bash
# usage: tools/_tl_run.sh <workload>   (timeline experiment build + run, with a watchdog that prints where python hangs)
cd "$(dirname "$0")/.."
make -s -C ray-tracer-rust_amd/csrc clean && make -s -j4 -C ray-tracer-rust_amd/csrc all EXTRA="${TL_FLAGS:--DRTX_EXPERIMENT_TIMELINE=1}" > gpurun_out/r2_tl_build.log 2>&1
for wl in "$@"; do
timeout -k 5 150 python -u -c "
import faulthandler, sys; faulthandler.dump_traceback_later(90, exit=True)
sys.argv=['tools/tile_timeline.py','$wl'] + '${TL_ARGS:-}'.split()
__file__='tools/tile_timeline.py'
exec(open('tools/tile_timeline.py').read())
" > gpurun_out/r2_timeline_$wl.txt 2>&1
tail -30 gpurun_out/r2_timeline_$wl.txt
done
