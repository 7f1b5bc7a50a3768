#!/bin/bash
out=gpurun_out/r2b/ldpc_ab.txt; mkdir -p gpurun_out/r2b; : > $out
for lib in libsrsran_phy_hip "$@"; do
  for pc in 4 1; do
    r=$(SRSRAN_HIP_LIB=$PWD/srslte_amd/lib/$lib.so LDPC_PCPB=$pc timeout -k 10 120 python tools/dbg/ldpc_ab.py 2>&1 | tail -1) || exit 1
    echo "$lib pcpb=$pc : $r" | tee -a $out
  done
done
