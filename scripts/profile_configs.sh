#!/bin/bash
# rocprofv3 kernel stats of the compact legs the default bench run reports under `other_configs` that profile_side.sh does not cover
# (hnsw1m at ef = 128, vamana10m1536_r32 with the hybrid leg), + FETCH_SIZE / WRITE_SIZE of the Vamana traversal.  GPU box, repo root.
set -e
ROUND=${ROUND:-r03}
R=$(cd "$(dirname "$0")/.." && pwd)
out=$R/gpurun_out/$ROUND
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_hnsw1m -o $ROUND -- python3 $R/bench.py --workload hnsw1m --ef 128 --steps 10 --warmup 2 --no-cpu-baseline --no-latency > $out/prof_hnsw1m.json 2> $out/prof_hnsw1m.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_vamana_r32_hybrid -o $ROUND -- python3 $R/bench.py --workload vamana10m1536_r32 --hybrid --steps 10 --warmup 2 --no-cpu-baseline --no-latency > $out/prof_vamana_r32_hybrid.json 2> $out/prof_vamana_r32_hybrid.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-include-regex "beam_search_kernel" --output-format csv -d $out/prof_vam_$c -o $ROUND -- python3 $R/bench.py --workload vamana10m1536_r32 --ef 80 --steps 4 --warmup 1 --no-cpu-baseline --no-latency > $out/prof_vam_$c.json 2> $out/prof_vam_$c.log || echo "pmc pass $c failed"
done
for dd in $out/prof_hnsw1m $out/prof_vamana_r32_hybrid $out/prof_vam_FETCH_SIZE $out/prof_vam_WRITE_SIZE; do
  find $dd -type f ! -name ${ROUND}_kernel_stats.csv ! -name ${ROUND}_counter_collection.csv -delete
done
ls $out | grep "prof_hnsw1m\|prof_vam"
