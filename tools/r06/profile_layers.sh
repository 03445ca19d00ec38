#!/bin/bash
# kernel traces of the layer forward + backward runs bench.py times (C3 A = 8 / 64, C4 max-pool) -> gpurun_out/r06_trace_<which>.md
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for which in "$@"; do
  d=/tmp/prof_$which; rm -rf $d
  rocprofv3 --kernel-trace --stats -d $d -o r -- python tools/r06/profile_layer_fwd_bwd.py $which 5 > /tmp/prof_$which.log 2>&1
  db=$(find $d -name "*.db" -printf "%s %p\n" | sort -nr | head -1 | cut -d" " -f2-)
  python tools/rocpd_summary.py $db > gpurun_out/r06_trace_$which.md
  head -32 gpurun_out/r06_trace_$which.md | cut -c1-200
done
# launch sequence of the last step(s): TFGX_SEQ=<n dispatches> bash tools/r06/profile_layers.sh maxpool
if [ -n "${TFGX_SEQ:-}" ]; then
  for which in "$@"; do
    db=$(find /tmp/prof_$which -name "*.db" -printf "%s %p\n" | sort -nr | head -1 | cut -d" " -f2-)
    python tools/rocpd_sequence.py $db $TFGX_SEQ > gpurun_out/r06_sequence_$which.md
  done
fi
