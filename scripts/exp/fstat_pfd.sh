#!/bin/bash
# fused_fstat_kernel: distance (k-steps) between a feature fragment's last MFMA and its refill load
cd "$(dirname "$0")/../.."
for d in 0 2 4; do
  echo "== LEANN_FSTAT_PFD=$d"
  scripts/variant.sh "-DLEANN_FSTAT_PFD=$d" python bench.py --workload recompute10m --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('  ', j['value'], j['recall_at_10'], r['frac'], r['fused_encode_score_ms'])"
done
