#!/bin/bash
# Diagnostic build with in-kernel s_memtime stamps (-DLEANN_STAMPS); prints where a hop / a unit spends its cycles.
#   scripts/stamps.sh [rows] [ef] [feat]          traversal kernel (phases B / C / D per hop)
#   scripts/stamps.sh fstat [args]                fused recompute kernel
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "fstat" ]; then shift; exec scripts/variant.sh "-DLEANN_STAMPS" python scripts/stamps_fstat.py "$@"; fi
exec scripts/variant.sh "-DLEANN_STAMPS" python scripts/stamps.py "$@"
