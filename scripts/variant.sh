#!/bin/bash
# Builds a variant library with extra -D flags into gpurun_out/variant/ and runs a command with LEANN_LIB pointing at it.
# Usage: scripts/variant.sh "-DLEANN_NT_ROWS" python bench.py --no-cpu-baseline
set -e
cd "$(dirname "$0")/.."
FLAGS="$1"; shift
mkdir -p gpurun_out/variant
rm -f gpurun_out/variant/*.o
for src in leann-rs_amd/csrc/*.hip; do
  f=$(basename "$src" .hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt $FLAGS -c "$src" -o gpurun_out/variant/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/variant/libleann_hip_variant.so gpurun_out/variant/*.o
LEANN_LIB=$PWD/gpurun_out/variant/libleann_hip_variant.so "$@"
