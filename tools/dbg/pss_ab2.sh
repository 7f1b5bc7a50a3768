#!/bin/bash
out=gpurun_out/r2b/pss_ab2.txt; mkdir -p gpurun_out/r2b; : > $out
for v in pair wave block; do
  r=$(SRSRAN_HIP_PSS_VARIANT=$v timeout -k 10 200 python bench.py --steps 3 --warmup 1 --only cellsearch --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']['cellsearch']; print(e['ms_per_step'], e['captures_per_s'], e['results_correct'])") || exit 1
  echo "$v : $r" | tee -a $out
done
