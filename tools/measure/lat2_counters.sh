#!/bin/bash
# SQ counters of the latency kernels (one wave / two waves per block) on tools/measure/lat_time.py K = 6144: how much of a wave's time is instruction
# issue, how much waiting.  usage (GPU box): bash tools/measure/lat2_counters.sh [out dir]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-$R/gpurun_out/lat2_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/p1 -o p -- python3 $R/tools/measure/lat_time.py 6144 0 > /dev/null 2> $O/p1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/p2 -o p -- python3 $R/tools/measure/lat_time.py 6144 0 > /dev/null 2> $O/p2.err
rocprofv3 --kernel-trace --stats -d $O/t -o t -- python3 $R/tools/measure/lat_time.py 6144 0 > /dev/null 2> $O/t.err
python3 $R/tools/rocpd_summary.py $O/p1 $O/p2 $O/t | grep -E "tdec_lat_kernel|^##|kernel-trace|pmc:" > $O/summary.txt
rm -rf $O/p1 $O/p2 $O/t
