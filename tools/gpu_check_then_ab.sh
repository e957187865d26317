#!/bin/bash
# Development tool: the GPU parity suite, then (unless the suite was killed at its time limit: a hung kernel must not be
# followed by more GPU work) an interleaved A/B of builds.   tools/gpu_check_then_ab.sh <tag> <ab_build.sh arguments...>
cd "$(dirname "$0")/.."
tag="$1"; shift
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "gpurun_out/${tag}_tests.log" 2>&1
rc=$?
tail -3 "gpurun_out/${tag}_tests.log"
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite killed at its limit (status $rc): no further GPU step"; exit $rc; fi
[ $# -gt 0 ] && tools/ab_build.sh "$@" > "gpurun_out/${tag}_ab.log" 2>&1
cat "gpurun_out/${tag}_ab.log" 2>/dev/null | tail -40
exit $rc
