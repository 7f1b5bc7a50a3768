#!/bin/bash
# counters of the cell-search kernels (separate passes, as the MI355X guide prescribes)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2b/pss_pmc; mkdir -p $OUT
B="python3 $R/bench.py --steps 2 --warmup 1 --only cellsearch --no-cpu"
cd $R
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/sq -o p -- $B > /dev/null 2> $OUT/sq.err &&
rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $OUT/sq2 -o p -- $B > /dev/null 2> $OUT/sq2.err &&
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $OUT/sq3 -o p -- $B > /dev/null 2> $OUT/sq3.err &&
rocprofv3 --kernel-trace --stats -d $OUT/tr -o p -- $B > /dev/null 2> $OUT/tr.err
python tools/rocpd_summary.py $OUT/sq $OUT/sq2 $OUT/sq3 > $OUT/../pss_pmc.txt 2>&1
find $OUT/tr -name "*kernel_stats*" | head -1 | xargs -I{} cp {} $OUT/../pss_kernel_stats.csv
rm -rf $OUT
