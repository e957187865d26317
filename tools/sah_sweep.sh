#!/bin/bash
# Development tool: frame time against the SAH's box cost and the leaf cap, through librtx_ablation.so (which reads
# RTX_SAH_BOX_COST / RTX_LEAF_MAX from the environment; the product takes neither).   WORKLOADS="c5 c3" tools/sah_sweep.sh
cd "$(dirname "$0")/.."
for wl in ${WORKLOADS:-c5 c3}; do
  for cost in ${COSTS:-1.0 0.5 0.3 0.2}; do
    for leaf in ${LEAVES:-4 2}; do
      out=$(RTX_PY_ABLATION=1 RTX_SAH_BOX_COST=$cost RTX_LEAF_MAX=$leaf timeout -k 10 300 python bench.py --workload $wl --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-scaling-config 2>/dev/null | grep '^{' | python -c 'import sys, json; d = json.loads(sys.stdin.read()); l = d["roofline"]["launch"]; c = d["roofline_valu"]; print(d["ms_per_step"], "shade", l["shade_ms"], "nodes", c["wave_node_visits"], "tris", c["wave_tri_visits"])')
      echo "$wl box_cost $cost leaf_max $leaf: $out"
    done
  done
done
