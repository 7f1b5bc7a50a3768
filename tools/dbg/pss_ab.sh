#!/bin/bash
# cell-search bench leg (timing only) with each library given: libsrsran_phy_hip first, then the ab_* variants named
out=gpurun_out/r2b/pss_ab.txt; mkdir -p gpurun_out/r2b; : > $out
for lib in libsrsran_phy_hip "$@"; do
  r=$(SRSRAN_HIP_LIB=$PWD/srslte_amd/lib/$lib.so timeout -k 10 200 python bench.py --steps 3 --warmup 1 --only cellsearch --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']['cellsearch']; print(e['ms_per_step'], e['captures_per_s'], e['results_correct'])") || exit 1
  echo "$lib : $r" | tee -a $out
done
