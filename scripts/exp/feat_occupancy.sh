#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "-DLEANN_FEAT_R1=6 -DLEANN_FEAT_OCC=7" "-DLEANN_FEAT_R1=8 -DLEANN_FEAT_OCC=7" "-DLEANN_FEAT_R1=6 -DLEANN_FEAT_OCC=6" "-DLEANN_FEAT_R1=5 -DLEANN_FEAT_OCC=8"; do
  echo "== $cfg"
  scripts/variant.sh "$cfg" python bench.py --workload recompute10m_graph --no-cpu-baseline --no-latency 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['recall_at_10'], j['roofline']['frac'], j['roofline']['kernel_avg_ms'])"
done
