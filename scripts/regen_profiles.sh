#!/bin/bash
# Regenerates every side line quoted in DESIGN.md §9 / profiles/r02_*_bench.json.  Run on the GPU box, from
# the repo root:   scripts/regen_profiles.sh [part]      part = a | b | c (each fits one gpurun call), default: all
# Outputs go to gpurun_out/$ROUND/; copy what you want judged into profiles/ (scripts/collect_profiles.py does).
set -e
ROUND=${ROUND:-r03}
cd "$(dirname "$0")/.."
out=gpurun_out/$ROUND
mkdir -p $out
part=${1:-all}
run() { # name, bench args...
  name=$1; shift
  echo "== $name: bench.py $*"
  python bench.py "$@" > $out/${name}_bench.json 2> $out/${name}_bench.log || { tail -5 $out/${name}_bench.log; return 1; }
  python - "$out/${name}_bench.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
r = j["roofline"]
print(f"   value {j['value']:.0f} {j['unit']}  recall {j.get('recall_at_10')}  ms/step {j['ms_per_step']:.3f}  {r['bound']} {r['achieved']:.1f} {r['unit']} = {r['frac']*100:.1f} %")
PY
}
if [ $part = a ] || [ $part = all ]; then
  run hnsw10m --headline-only                   # the headline line (BASELINE metric config)
  run hnsw1m --workload hnsw1m --ef 128         # BASELINE configs[1]: 1M x 768, M=32, ef=128
  run hnsw10m_ef128 --headline-only --ef 128 --no-cpu-baseline --no-latency
  run recompute10m --workload recompute10m      # configs[2], exhaustive (the reference's algorithm)
fi
if [ $part = b ] || [ $part = all ]; then
  run recompute10m_graph --workload recompute10m_graph                       # configs[2] on a graph, throughput batch
  run recompute10m_graph_batch64 --workload recompute10m_graph --batch 64 --steps 200 --warmup 20 --no-cpu-baseline   # ... at its stated batch
  run hnsw100m_shard8_one_gpu --workload hnsw100m_shard8 --no-latency         # the per-GPU slice of configs[3]
  run scan10m --workload scan10m
fi
if [ $part = c ] || [ $part = all ]; then
  run vamana10m1536 --workload vamana10m1536    # configs[4] search leg (DiskANN R=64, 1536-d)
  run vamana10m1536_r32 --workload vamana10m1536_r32 --no-cpu-baseline        # ... at the reference's default degree (graph_degree 32, cli/build.rs:79)
  run vamana10m1536_r32_hybrid --workload vamana10m1536_r32 --hybrid --no-latency --cpu-queries 1024   # configs[4] as a whole
fi
if [ $part = d ] || [ $part = all ]; then
  run hnsw10m_clusters65536 --headline-only --clusters 65536 --no-cpu-baseline --no-latency   # every query of a launch in (almost) its own cluster: bounds the cache contribution
  run hnsw10m_sigma015 --headline-only --sigma 0.15 --no-cpu-baseline --no-latency              # SURVEY 8d's sigma
fi
