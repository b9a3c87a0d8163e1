#!/bin/bash
# recompute-on search, rows of 256 features: four rows per wave instruction (in-tree) against one row per wave load (LEANN_DEBUG_NO_FEAT256=1)
cd "$(dirname "$0")/../.."
show() { python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('   %.0f q/s  recall %.4f  ef %s  %.1f %%  kernel %.3f ms' % (j['value'], j['recall_at_10'], j['config']['ef_search'], 100*r['frac'], r['kernel_avg_ms']))"; }
for rep in 1 2; do
  echo "== one row per wave load (rep $rep)"; LEANN_DEBUG_NO_FEAT256=1 python bench.py --workload recompute10m_graph --ef 52 --no-cpu-baseline --no-latency 2>/dev/null | show
  echo "== four rows per wave instruction (rep $rep)"; python bench.py --workload recompute10m_graph --ef 52 --no-cpu-baseline --no-latency 2>/dev/null | show
done
