#!/bin/bash
# the three LDPC decoders (int8 packed, int16, float) on 16,384 words of BG1 Z=384 at 20 iterations (tools/dbg/ldpc_time.py)
timeout -k 10 300 python tools/dbg/ldpc_time.py 2>&1 | tail -3
LDPC_PACKED=0 timeout -k 10 300 python tools/dbg/ldpc_time.py 2>&1 | head -1 | sed 's/^/plain int8 kernel: /'
