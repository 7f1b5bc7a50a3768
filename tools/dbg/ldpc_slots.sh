#!/bin/bash
# LDPC (BG1 Z=384, 20 iterations, 16,384 words): workgroups launched (LDPC_SLOTS) against the 1280 that are resident at a time
for sl in ${@:-0 2560 3840 5120}; do
  if [ $sl = 0 ]; then unset LDPC_SLOTS; else export LDPC_SLOTS=$sl; fi
  echo "slots=$sl : $(timeout -k 10 120 python tools/dbg/ldpc_ab.py 2>&1 | tail -1)"
done
