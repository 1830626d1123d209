#!/bin/bash
# round 4: tap-minor K order of the fused conv kernel against the tap-major order (AMX_LN_TAP_MAJOR=1), same box, alternating
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for rep in 1 2 3; do
  for mode in 0 1; do
    echo "== AMX_LN_TAP_MAJOR=$mode (run $rep)"
    AMX_LN_TAP_MAJOR=$mode python3 tools/geometry_sweep.py f16x3 32:10 2>&1 | grep "ms/step"
  done
done
for mode in 0 1; do
  echo "== bf16 AMX_LN_TAP_MAJOR=$mode"
  AMX_LN_TAP_MAJOR=$mode python3 tools/geometry_sweep.py bf16 32:10 2>&1 | grep "ms/step"
done
