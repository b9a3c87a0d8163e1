#!/bin/bash
# latency form of the hop loop: touching the adjacency lists of the evaluated rows (L2 hit for the next candidate's list) on / off
cd "$(dirname "$0")/../.."
show() { python -c "import json,sys; j=json.loads(sys.stdin.read()); l=j.get('single_query',{}); print('   %.0f q/s  single query p50 %.3f ms p99 %.3f ms  ms/step %.3f' % (j['value'], l.get('p50_ms',0), l.get('p99_ms',0), j['ms_per_step']))"; }
echo "== hnsw10m, touch ON (in-tree)"; python bench.py --ef 56 --no-cpu-baseline 2>/dev/null | show
echo "== hnsw10m, touch OFF"; scripts/variant.sh "-DLEANN_NO_ADJ_TOUCH" python bench.py --ef 56 --no-cpu-baseline 2>/dev/null | show
echo "== recompute10m_graph batch 64, touch ON"; python bench.py --workload recompute10m_graph --ef 56 --batch 64 --steps 200 --warmup 20 --no-cpu-baseline --no-latency 2>/dev/null | show
echo "== recompute10m_graph batch 64, touch OFF"; LEANN_LIB=$PWD/gpurun_out/variant/libleann_hip_variant.so python bench.py --workload recompute10m_graph --ef 56 --batch 64 --steps 200 --warmup 20 --no-cpu-baseline --no-latency 2>/dev/null | show
