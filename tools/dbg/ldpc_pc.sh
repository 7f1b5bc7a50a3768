#!/bin/bash
for pc in 1 2 4; do echo "pcpb=$pc : $(LDPC_PCPB=$pc timeout -k 10 120 python tools/dbg/ldpc_ab.py 2>&1 | tail -1)"; done
for z in 320 256 208 144; do echo "Z=$z : $(timeout -k 10 120 python tools/dbg/ldpc_ab.py 16384 $z 2>&1 | tail -1)"; done
