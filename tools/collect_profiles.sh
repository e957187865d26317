#!/bin/bash
# Development tool: collects, on a GPU box, the evidence behind DESIGN.md section 5 for one workload into
# gpurun_out/prof_<tag>/ — the bench line, the rocprofv3 kernel statistics of the same command, and the PMC
# passes (one counter group per run, --pmc only with --kernel-trace, as the pool requires).
#   tools/collect_profiles.sh <tag> [workload]        e.g.  tools/collect_profiles.sh j c3
# tools/summarise_profiles.py turns the CSVs into profiles/pmc_summary.json and profiles/traffic.json.
set -u
TAG="${1:?tag}"
WL="${2:-c3}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

python3 "$ROOT/bench.py" --workload "$WL" --steps 30 --warmup 3 > "$OUT/bench_$WL.json" 2> "$OUT/bench_$WL.err" || { echo "bench failed"; tail -5 "$OUT/bench_$WL.err"; exit 1; }
echo "bench: $(cut -c1-200 "$OUT/bench_$WL.json")"

timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o "$WL" -- \
    python3 "$ROOT/bench.py" --workload "$WL" --steps 20 --warmup 3 --no-cpu-baseline --no-scaling-config > "$OUT/stats_$WL.log" 2>&1 || { echo "kernel-trace failed"; exit 1; }
echo "kernel stats done"

i=0
for group in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_SMEM" \
             "GRBM_GUI_ACTIVE SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    # shellcheck disable=SC2086
    timeout -k 10 400 rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/pmc$i" -o "$WL" -- \
        python3 "$ROOT/bench.py" --workload "$WL" --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-config > "$OUT/pmc$i.log" 2>&1 || { echo "pmc group $i failed"; tail -3 "$OUT/pmc$i.log"; exit 1; }
    echo "pmc group $i done: $group"
    i=$((i + 1))
done
# LDS bank conflicts of the same command: extra LDS-array cycles against all LDS-array cycles
# (MI355X_MICROARCH.md, LDS); not fatal when the counters are not there
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/pmc_lds" -o "$WL" -- \
    python3 "$ROOT/bench.py" --workload "$WL" --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-config > "$OUT/pmc_lds.log" 2>&1 && echo "pmc LDS group done" || echo "pmc LDS group failed (see pmc_lds.log)"
# keep only what the summary needs (the traces of the counted warm-up launches are large)
find "$OUT" -name "*_agent_info.csv" -delete
ls -la "$OUT" "$OUT"/pmc0 | head -30
