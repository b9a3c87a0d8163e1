#!/bin/bash
# same-box A/B of single-query latency (leann_backend_search from host memory): build/ab/libleann_head.so vs the in-tree library
cd "$(dirname "$0")/../.."
show() { python -c "import json,sys; j=json.loads(sys.stdin.read()); q=j['single_query']; print('   p50 %.4f ms  p99 %.4f ms  mean %.4f  (ef %s; batch value %.0f q/s)' % (q['p50_ms'], q['p99_ms'], q['mean_ms'], q['ef'], j['value']))"; }
for wl in "$@"; do
  for rep in 1 2; do
    echo "== $wl, other revision (rep $rep)"; LEANN_LIB=$PWD/build/ab/libleann_head.so python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | show
    echo "== $wl, in-tree (rep $rep)"; python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | show
  done
done
