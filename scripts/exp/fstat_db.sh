#!/bin/bash
# fused_fstat_kernel: second feature-register set refilled inside the first weight visit (LEANN_FSTAT_DB=1) vs the in-tree refill in
# the last score visit.  Parity tests run against the variant first.
cd "$(dirname "$0")/../.."
show() { python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('  ', j['value'], j['recall_at_10'], r['frac'], r['fused_encode_score_ms'])"; }
scripts/variant.sh "-DLEANN_FSTAT_DB=1" python -m pytest tests/test_gpu_recompute.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for rep in 1 2; do
  echo "== in-tree (rep $rep)"; python bench.py --workload recompute10m --no-cpu-baseline 2>/dev/null | show
  echo "== LEANN_FSTAT_DB=1 (rep $rep)"; LEANN_LIB=$PWD/gpurun_out/variant/libleann_hip_variant.so python bench.py --workload recompute10m --no-cpu-baseline 2>/dev/null | show
done
