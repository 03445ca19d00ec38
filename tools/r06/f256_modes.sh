#!/bin/bash
# F = 256 is bimodal from process to process: N plain processes (the distribution), then N processes under rocprofv3 --kernel-trace --pmc
# (one counter set per process; kernel durations and counters of the same process side by side) -> gpurun_out/r06_f256_modes.md
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
out=gpurun_out/r06_f256_modes.md; : > $out
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > gpurun_out/r06_counters_avail.txt
echo "## plain processes" >> $out
for i in 1 2 3 4 5 6; do python tools/r06/f256_child.py 256 2>/dev/null | tail -1 >> $out; done
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  for i in 1 2 3; do
    d=/tmp/pmc_f256_$RANDOM; rm -rf $d
    (cd /tmp && rocprofv3 --kernel-trace --pmc $set -d $d -o r -- python $GRAFT_REPO_ROOT/tools/r06/f256_child.py 256 2>/dev/null | tail -1) > /tmp/child.json
    db=$(find $d -name "*.db" -printf "%s %p\n" | sort -nr | head -1 | cut -d" " -f2-)
    echo "## pmc: $set  (process $i)" >> $out
    cat /tmp/child.json >> $out
    python tools/rocpd_summary.py $db 2>/dev/null | grep "seg_reduce_kernel" | cut -c1-220 >> $out
    rm -rf $d
  done
done
cat $out | cut -c1-230
