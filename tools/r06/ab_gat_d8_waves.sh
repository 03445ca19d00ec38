#!/bin/bash
# same box: the library as built (TFGX_GAT_D8_WAVES=4), then attn rebuilt with 5, then 4 again
set -e
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_gat_d8_waves.jsonl; : > $out
python tools/r06/time_gat.py waves4_a >> $out
TFGX_EXTRA_HIPCC_FLAGS="-DTFGX_GAT_D8_WAVES=5" python -c "
import os
from tf_geometric_amd import _build
os.utime(os.path.join(_build.CSRC, 'tfgx_attn.hip'))
_build.build(verbose=False)" 2>&1 | grep -v warning | tail -2
python tools/r06/time_gat.py waves5 >> $out
python -c "
import os
from tf_geometric_amd import _build
os.utime(os.path.join(_build.CSRC, 'tfgx_attn.hip'))
_build.build(verbose=False)" 2>&1 | grep -v warning | tail -2
python tools/r06/time_gat.py waves4_b >> $out
cat $out
