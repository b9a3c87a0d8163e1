#!/bin/bash
# cross-compiles the library of another revision (default HEAD) into build/ab/libleann_head.so for scripts/exp/ab.sh
set -e
cd "$(dirname "$0")/../.."
rev=${1:-HEAD}
tmp=$(mktemp -d)
git archive "$rev" leann-rs_amd/csrc include | tar -x -C "$tmp"
pids=()
for src in "$tmp"/leann-rs_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -c "$src" -o "${src%.hip}.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
mkdir -p build/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libleann_head.so "$tmp"/leann-rs_amd/csrc/*.o
rm -rf "$tmp"
echo "build/ab/libleann_head.so = $(git rev-parse --short "$rev")"
