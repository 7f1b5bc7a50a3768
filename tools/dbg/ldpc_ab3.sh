#!/bin/bash
for pc in 0 1 4; do
  if [ $pc = 0 ]; then unset LDPC_PCPB; else export LDPC_PCPB=$pc; fi
  echo "pcpb=$pc ab: $(timeout -k 10 120 python tools/dbg/ldpc_ab.py 16384 384 2>&1 | tail -1)"
  echo "pcpb=$pc bench: $(timeout -k 10 300 python bench.py --steps 3 --warmup 1 --only ldpc --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']['ldpc']; print(e['value'], e['roofline']['avg_launch_ms'], e['ldpc_kernel_mbit_per_s'])")"
done
