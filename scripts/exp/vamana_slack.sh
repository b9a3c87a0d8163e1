#!/bin/bash
# Vamana back-edge slack (build.hip: P pending back-edges per node before a full list is re-pruned; 0 = strict): build time / recall / QPS
cd "$(dirname "$0")/../.."
wl=${1:-vamana1m1536}; shift
show() { python -c "import json,sys; j=json.loads(sys.stdin.read()); print('   build %.1f s  ef %s  recall %.4f  %.0f q/s  %.1f %% HBM' % (j['config']['index_build_s'], j['config']['ef_search'], j['recall_at_10'], j['value'], 100*j['roofline']['frac']))"; }
for P in "$@"; do
  echo "== $wl, $P pending back-edges per node, ef 72"
  LEANN_VAMANA_PENDING=$P python bench.py --workload $wl --ef 72 --recall-queries 2000 --no-cpu-baseline --no-latency 2>/dev/null | show
done
