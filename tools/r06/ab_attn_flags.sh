#!/bin/bash
# same box: the library as built, then tfgx_attn.hip rebuilt with the given flags, then as built again — tools/r06/time_gat.py each time
#   gpurun -- 'bash tools/r06/ab_attn_flags.sh "-DTFGX_GAT_QG_WAVES=5" qg_waves5'
set -e
cd "$GRAFT_REPO_ROOT"
flags="$1"; tag="${2:-variant}"; src="${3:-tfgx_attn.hip}"
out=gpurun_out/r06_ab_${tag}.jsonl; : > $out
rebuild() {
TFGX_EXTRA_HIPCC_FLAGS="$1" python -c "
import os
from tf_geometric_amd import _build
os.utime(os.path.join(_build.CSRC, '$src'))
_build.build(verbose=False)" 2>&1 | grep -v warning | tail -2
}
T="${TFGX_AB_SCRIPT:-tools/r06/time_gat.py}"
python $T base_a >> $out
rebuild "$flags"
python $T "$tag" >> $out
rebuild ""
python $T base_b >> $out
python - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["tag"], {k: round(v, 3) for k, v in d.items() if k != "tag"})
PY
