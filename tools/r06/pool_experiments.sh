#!/bin/bash
# where does pool_wgrad_kernel's time go: variants built on the box (timing only; results are wrong for 1 and 2)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for v in ${POOL_VARIANTS:-1 2 0}; do
  TFGX_EXTRA_HIPCC_FLAGS="-DTFGX_POOL_EXPERIMENT=$v" python -c "
import os
from tf_geometric_amd import _build
os.utime(os.path.join(_build.CSRC, 'tfgx_poolgrad.hip'))
_build.build(verbose=False)" 2>&1 | grep -v "warning\|~\|\^\||" | tail -2
  echo "== experiment $v"
  bash tools/r06/profile_layers.sh maxpool 2>&1 | grep "pool_wgrad_kernel" | cut -c1-150
done
