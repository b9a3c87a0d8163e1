#!/bin/bash
# Diagnostic build with in-kernel s_memtime stamps (never shipped: outputs go to gpurun_out/, timings of this
# build are not quotable — read the SHARES).  Usage: scripts/stamps.sh  (on the GPU box)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/stamps
for f in api build gen scan recompute; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -DLEANN_STAMPS -c leann-rs_amd/csrc/$f.hip -o gpurun_out/stamps/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/stamps/libleann_hip_stamps.so gpurun_out/stamps/*.o
if [ "$1" = "fstat" ]; then shift; LEANN_LIB=$PWD/gpurun_out/stamps/libleann_hip_stamps.so python scripts/stamps_fstat.py "$@"; else LEANN_LIB=$PWD/gpurun_out/stamps/libleann_hip_stamps.so python scripts/stamps.py "$@"; fi
