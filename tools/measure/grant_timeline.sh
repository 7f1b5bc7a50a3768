#!/bin/bash
# Kernel timelines of single grant calls and multi-grant calls (tools/probe/tti_probe under rocprofv3 --kernel-trace; tools/measure/call_timeline.py)
# usage (GPU box): bash tools/measure/grant_timeline.sh [out dir]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-$R/gpurun_out/tl}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 5,1 5,4 4,4 0,1 3,4; do
  export TTI_PROBE_ONLY=$c
  rocprofv3 --kernel-trace -d $O/t_$c -o t -- $R/tools/probe/tti_probe > $O/run_$c.txt 2>&1 < /dev/null || exit 1
  echo "=== TTI_PROBE_ONLY=$c" >> $O/timeline.txt
  cat $O/run_$c.txt >> $O/timeline.txt
  python3 $R/tools/measure/call_timeline.py $O/t_$c 30 1 >> $O/timeline.txt
done
