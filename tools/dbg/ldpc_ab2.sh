#!/bin/bash
out=gpurun_out/r2b/ldpc_ab2.txt; mkdir -p gpurun_out/r2b; : > $out
for z in 384 320 256 208 176 144; do
  for pc in 0 1 2 3 4; do
    if [ $pc = 0 ]; then unset LDPC_PCPB; else export LDPC_PCPB=$pc; fi
    r=$(timeout -k 10 120 python tools/dbg/ldpc_ab.py 16384 $z 2>&1 | tail -1) || exit 1
    echo "Z=$z pcpb=$pc : $r" | tee -a $out
  done
done
