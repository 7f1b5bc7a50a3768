#!/bin/bash
# LDPC (BG1 Z=384, 20 iterations, 16,384 words) with the product library and the variant libraries named, resident and all-slot launches
for lib in libsrsran_phy_hip "$@"; do
  for sl in 1280 0; do
    if [ $sl = 0 ]; then unset LDPC_SLOTS; else export LDPC_SLOTS=$sl; fi
    echo "$lib slots=$sl : $(SRSRAN_HIP_LIB=$PWD/srslte_amd/lib/$lib.so timeout -k 10 120 python tools/dbg/ldpc_ab.py 2>&1 | tail -1)"
  done
done
