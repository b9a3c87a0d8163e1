#!/bin/bash
# rocprofv3 kernel stats for the side workloads whose kernels DESIGN.md prices (run on the GPU box from the repo root)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
out=$R/gpurun_out/r02
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wl in recompute10m_graph recompute10m; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$wl -o r02 -- python3 $R/bench.py --workload $wl --no-cpu-baseline --no-latency > $out/prof_$wl.json 2> $out/prof_$wl.log
done
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --kernel-include-regex "beam_search_kernel" --output-format csv -d $out/prof_rdreq -o r02 -- python3 $R/bench.py --no-cpu-baseline --no-latency --steps 4 --warmup 1 > $out/prof_rdreq.json 2> $out/prof_rdreq.log || echo "rdreq pass failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "fused_fstat_kernel" --output-format csv -d $out/prof_mfma -o r02 -- python3 $R/bench.py --workload recompute10m --no-cpu-baseline --steps 4 --warmup 1 > $out/prof_mfma.json 2> $out/prof_mfma.log || echo "mfma pass failed"
ls $out | head -50
