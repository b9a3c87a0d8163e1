#!/bin/bash
# rocprofv3 kernel stats for the side workloads whose kernels DESIGN.md prices (run on the GPU box from the repo root)
set -e
ROUND=${ROUND:-r03}
R=$(cd "$(dirname "$0")/.." && pwd)
out=$R/gpurun_out/$ROUND
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wl in recompute10m_graph recompute10m; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$wl -o $ROUND -- python3 $R/bench.py --workload $wl --no-cpu-baseline --no-latency > $out/prof_$wl.json 2> $out/prof_$wl.log
done
for c in FETCH_SIZE WRITE_SIZE; do # HBM-side traffic of the recompute-on traversal (520-B rows: whole 128-B lines are what moves)
  rocprofv3 --pmc $c --kernel-include-regex "beam_search_feat256_kernel" --output-format csv -d $out/prof_rg_$c -o $ROUND -- python3 $R/bench.py --workload recompute10m_graph --no-cpu-baseline --no-latency --steps 4 --warmup 1 > $out/prof_rg_$c.json 2> $out/prof_rg_$c.log || echo "pmc pass $c (recompute10m_graph) failed"
done
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --kernel-include-regex "beam_search_kernel" --output-format csv -d $out/prof_rdreq -o $ROUND -- python3 $R/bench.py --headline-only --no-cpu-baseline --no-latency --steps 4 --warmup 1 > $out/prof_rdreq.json 2> $out/prof_rdreq.log || echo "rdreq pass failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "fused_fstat_kernel" --output-format csv -d $out/prof_mfma -o $ROUND -- python3 $R/bench.py --workload recompute10m --no-cpu-baseline --steps 4 --warmup 1 > $out/prof_mfma.json 2> $out/prof_mfma.log || echo "mfma pass failed"
# keep what scripts/collect_profiles.py reads (gpurun merges at most 64 MiB back): kernel stats, counter collections, and of the
# kernel trace only the query kernels' dispatches
for dd in $out/prof_*/; do
  [ -f $dd/${ROUND}_kernel_trace.csv ] && { head -1 $dd/${ROUND}_kernel_trace.csv > $dd/t.csv; grep -E "beam_search|fused_fstat" $dd/${ROUND}_kernel_trace.csv >> $dd/t.csv || true; mv $dd/t.csv $dd/${ROUND}_kernel_trace.csv; }
  find $dd -type f ! -name ${ROUND}_kernel_stats.csv ! -name ${ROUND}_kernel_trace.csv ! -name ${ROUND}_counter_collection.csv -delete
done
ls $out | head -50
