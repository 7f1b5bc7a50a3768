#!/bin/bash
# cell-search bench leg, 8 steps: ms per step and captures/s (optionally: SRSRAN_HIP_PSS_VARIANT / SRSRAN_HIP_LIB in the environment)
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --only cellsearch --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']['cellsearch']; print(e['ms_per_step'], e['captures_per_s'], e['results_correct'])"
