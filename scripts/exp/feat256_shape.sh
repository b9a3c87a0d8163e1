#!/bin/bash
# recompute-on search over 256-feature rows, four rows per wave instruction: groups in flight per wave (G), workgroups per CU (OCC),
# projected query in LDS or in registers (QLDS)
cd "$(dirname "$0")/../.."
show() { python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('   %.0f q/s  recall %.4f  ef %s  %.1f %%  kernel %.3f ms' % (j['value'], j['recall_at_10'], j['config']['ef_search'], 100*r['frac'], r['kernel_avg_ms']))"; }
echo "== in-tree"; python bench.py --workload recompute10m_graph --ef 52 --no-cpu-baseline --no-latency 2>/dev/null | show
for v in "$@"; do
  echo "== $v"
  scripts/variant.sh "$v" python bench.py --workload recompute10m_graph --ef 52 --no-cpu-baseline --no-latency 2>/dev/null | show
done
