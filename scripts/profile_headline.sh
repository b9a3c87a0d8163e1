#!/bin/bash
# rocprofv3 evidence for the headline bench line (run on the GPU box from the repo root): kernel-trace + stats of the default bench
# command, then SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE, L2 hit/miss) as MI355X_MICROARCH.md §HBM prescribes.
set -e
ROUND=${ROUND:-r03}
R=$(cd "$(dirname "$0")/.." && pwd)
out=$R/gpurun_out/$ROUND
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_stats -o $ROUND -- python3 $R/bench.py --headline-only --no-cpu-baseline --no-latency > $out/prof_stats.json 2> $out/prof_stats.log
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-include-regex "beam_search_kernel" --output-format csv -d $out/prof_$tag -o $ROUND -- python3 $R/bench.py --headline-only --no-cpu-baseline --no-latency --steps 4 --warmup 1 > $out/prof_$tag.json 2> $out/prof_$tag.log || echo "pmc pass $c failed"
done
rocprofv3 -L 2>/dev/null | grep -i -E "mall|dram|EA0_RDREQ" | head -40 > $out/counters_mem.txt || true
# keep what scripts/collect_profiles.py reads (gpurun merges at most 64 MiB back): kernel stats, counter collections, and of the
# kernel trace only the query kernels' dispatches
for dd in $out/prof_*/; do
  [ -f $dd/${ROUND}_kernel_trace.csv ] && { head -1 $dd/${ROUND}_kernel_trace.csv > $dd/t.csv; grep -E "beam_search|fused_fstat" $dd/${ROUND}_kernel_trace.csv >> $dd/t.csv || true; mv $dd/t.csv $dd/${ROUND}_kernel_trace.csv; }
  find $dd -type f ! -name ${ROUND}_kernel_stats.csv ! -name ${ROUND}_kernel_trace.csv ! -name ${ROUND}_counter_collection.csv -delete
done
ls $out
