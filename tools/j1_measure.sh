#!/bin/bash
# Development tool: what BASELINE.json's north_star design costs next to the shipped one (csrc/rtx_j1_ablation.hpp), all
# through librtx_ablation.so so that the builds compared differ in RTX_J1 only.   tools/j1_measure.sh
cd "$(dirname "$0")/.."
run() {   # workload, mode, steps
  RTX_PY_ABLATION=1 RTX_J1=$2 timeout -k 10 600 python bench.py --workload $1 --steps $3 --warmup 2 --no-cpu-baseline --no-scaling-config 2>/dev/null | grep '^{' | python -c 'import sys, json; d = json.loads(sys.stdin.read()); l = d["roofline"]["launch"]; c = d["roofline_valu"]; print(d["ms_per_step"], "ms: scheduling pass", l["schedule_ms"], "shading pass", l["shade_ms"], "| box records", c["wave_node_visits"], "primitive records", c["wave_tri_visits"])'
}
for round in 1 2; do
  for mode in 0 1; do echo "round $round c3 RTX_J1=$mode: $(run c3 $mode 20)"; done
  for mode in 0 1; do echo "round $round c5 RTX_J1=$mode: $(run c5 $mode 5)"; done
  for mode in 0 2 3; do echo "round $round c3 RTX_J1=$mode (scheduling pass = primary rays): $(run c3 $mode 5)"; done
  for mode in 0 2 3; do echo "round $round c1b RTX_J1=$mode (scheduling pass = primary rays): $(run c1b $mode 10)"; done
done
