#!/bin/bash
# cell-search correlation kernel: the product ("wave": block spectrum parked once, read back twice) against "recompute" (no parked spectra): bench figure,
# SQ counters and HBM-side bytes of each (separate PMC passes).  Run from the repository root on the GPU box; output: gpurun_out/round4/pss_wall.txt
export TMPDIR=/tmp
OUT=gpurun_out/round4
mkdir -p $OUT
C="python bench.py --steps 3 --warmup 1 --only cellsearch --no-cpu"
: > $OUT/pss_wall.txt
for V in wave recompute; do
  export SRSRAN_HIP_PSS_VARIANT=$V
  echo "==== variant $V" >> $OUT/pss_wall.txt
  $C 2> /dev/null | python -c "import sys,json; e=json.loads(sys.stdin.read().strip().splitlines()[-1])['extra']['cellsearch']; print('bench: %.3f ms per 256 captures, %.0f captures/s, results correct: %s' % (e['ms_per_step'], e['captures_per_s'], e['results_correct']))" >> $OUT/pss_wall.txt &&
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/v_s -o p -- $C > /dev/null 2> $OUT/v.err &&
  rocprofv3 --pmc FETCH_SIZE -d $OUT/v_f -o p -- $C > /dev/null 2> $OUT/v.err &&
  rocprofv3 --pmc WRITE_SIZE -d $OUT/v_w -o p -- $C > /dev/null 2> $OUT/v.err &&
  python tools/rocpd_summary.py $OUT/v_s $OUT/v_f $OUT/v_w | grep -E "pss_wave_kernel" >> $OUT/pss_wall.txt
  rm -rf $OUT/v_s $OUT/v_f $OUT/v_w
done
unset SRSRAN_HIP_PSS_VARIANT
cut -c1-170 $OUT/pss_wall.txt
