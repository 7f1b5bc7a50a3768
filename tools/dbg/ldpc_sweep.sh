#!/bin/bash
# LDPC bench leg under the development knobs (words per workgroup, resident slots): one line per setting
out=${1:-gpurun_out/r2b/ldpc_sweep.txt}
mkdir -p $(dirname $out); : > $out
for cfg in "4 0" "1 0" "1 1024" "1 1536" "2 0" "2 768" "3 0"; do
  set -- $cfg
  export LDPC_PCPB=$1; if [ "$2" != "0" ]; then export LDPC_SLOTS=$2; else unset LDPC_SLOTS; fi
  r=$(timeout -k 10 200 python bench.py --steps 3 --warmup 1 --only ldpc --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']['ldpc']; print(e['roofline']['avg_launch_ms'], e['ldpc_kernel_mbit_per_s'])") || exit 1
  echo "pcpb=$1 slots=$2 : $r" | tee -a $out
done
