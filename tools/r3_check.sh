#!/bin/bash
# round-3 one-off: suite, default bench line, 2-rank rehearsal on one device, div_denom range check, RTX_MAX_CUT at its limits
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3_p_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r3_p_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/r3_p_bench.json 2> gpurun_out/r3_p_bench.err; cut -c1-400 gpurun_out/r3_p_bench.json
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --ranks-share-device-0 --no-cpu-baseline > gpurun_out/r3_p_bench_2ranks.json 2> gpurun_out/r3_p_bench_2ranks.err; python -c "
import json; d=[json.loads(l) for l in open('gpurun_out/r3_p_bench_2ranks.json') if l.startswith('{')][0]; print('2 ranks:', d['ms_per_step'], d['per_rank'], d['scaling_config']['per_rank'])"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/div_denom_check.hip -o /tmp/div_denom_check && timeout -k 10 600 /tmp/div_denom_check > gpurun_out/r3_p_div_denom_check.log 2>&1; tail -2 gpurun_out/r3_p_div_denom_check.log
mkdir -p gpurun_out/ab
for cut in 1 51; do
  rm -rf gpurun_out/ab/obj_cut; make -s -j4 -C ray-tracer-rust_amd/csrc all EXTRA="-DRTX_MAX_CUT=$cut" OBJ=$PWD/gpurun_out/ab/obj_cut OUT=$PWD/gpurun_out/ab/librtx_cut$cut.so > gpurun_out/ab/build_cut$cut.log 2>&1
  echo "RTX_MAX_CUT=$cut: $(RTX_PY_LIB=$PWD/gpurun_out/ab/librtx_cut$cut.so timeout -k 10 300 python tools/debug_diff.py 640 360 2>&1 | head -1)"
done > gpurun_out/r3_p_max_cut_limits.log 2>&1; cat gpurun_out/r3_p_max_cut_limits.log
rm -rf gpurun_out/ab/obj_cut gpurun_out/ab/librtx_cut*.so
