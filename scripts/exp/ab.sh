#!/bin/bash
# same-box A/B of two libraries: build/ab/libleann_head.so (a build of another revision, cross-compiled beforehand) vs the in-tree one
cd "$(dirname "$0")/../.."
show() { python -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print('   %.0f q/s  recall %.4f  ef %s  %.1f %%  kernel %.3f ms' % (j['value'], j['recall_at_10'], j['config']['ef_search'], 100*r['frac'], r['kernel_avg_ms']))"; }
for wl in "$@"; do
  for rep in 1 2; do
    echo "== $wl, other revision (rep $rep)"; LEANN_LIB=$PWD/build/ab/libleann_head.so python bench.py --workload $wl --ef 56 --no-cpu-baseline --no-latency 2>/dev/null | show
    echo "== $wl, in-tree (rep $rep)"; python bench.py --workload $wl --ef 56 --no-cpu-baseline --no-latency 2>/dev/null | show
  done
done
