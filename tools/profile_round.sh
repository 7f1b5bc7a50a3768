#!/bin/bash
# Round measurement pass on the GPU box (run through gpurun from the repository root):
#   full GPU test suite, bench.py, the secondary benches, rocprofv3 kernel trace of bench.py and separate PMC passes.
# Outputs go to gpurun_out/round/; the summaries to keep are copied into profiles/ by hand afterwards.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/round
mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err && tail -1 $OUT/bench.json | cut -c1-400 &&
python tools/bench_nr.py > $OUT/bench_nr.json 2> $OUT/bench_nr.err &&
python tools/bench_sync.py > $OUT/bench_sync.json 2> $OUT/bench_sync.err &&
python tools/bench_sch.py > $OUT/bench_sch.json 2> $OUT/bench_sch.err &&
python tools/bench_pusch_rx.py > $OUT/bench_pusch_rx.json 2> $OUT/bench_pusch_rx.err &&
python tools/bench_nr_rx.py > $OUT/bench_nr_rx.json 2> $OUT/bench_nr_rx.err &&
python tools/bench_uplink.py > $OUT/bench_uplink.json 2> $OUT/bench_uplink.err &&
python tools/bench_sch_nr.py > $OUT/bench_sch_nr.json 2> $OUT/bench_sch_nr.err &&
python tools/dbg/dft_time.py 1200 900 600 300 144 72 12 > $OUT/dft_time.txt 2> $OUT/dft_time.err &&
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/trace.err &&
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $OUT/pmc_fetch.err &&
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $OUT/pmc_write.err &&
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/pmc_sq -o p -- python bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $OUT/pmc_sq.err &&
( rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE -d $OUT/pmc_lds -o p -- python bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $OUT/pmc_lds.err &&
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE -d $OUT/pmc_lds_nr -o p -- python tools/bench_nr.py --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_lds_nr.err &&
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE -d $OUT/pmc_lds_sync -o p -- python tools/bench_sync.py > /dev/null 2> $OUT/pmc_lds_sync.err &&
  python tools/rocpd_summary.py $OUT/pmc_lds $OUT/pmc_lds_nr $OUT/pmc_lds_sync > $OUT/pmc_lds.txt ) || echo "LDS counter pass failed (see pmc_lds*.err)"
python tools/rocpd_summary.py $OUT/trace > $OUT/kernel_stats.txt &&
python tools/rocpd_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq > $OUT/pmc.txt &&
rocprofv3 --kernel-trace -d $OUT/trace_nr -o nr -- python tools/bench_nr.py > /dev/null 2> $OUT/trace_nr.err &&
rocprofv3 --kernel-trace -d $OUT/trace_sync -o sync -- python tools/bench_sync.py > /dev/null 2> $OUT/trace_sync.err &&
python tools/rocpd_summary.py $OUT/trace_nr $OUT/trace_sync > $OUT/kernel_stats_nr_sync.txt &&
rocprofv3 --kernel-trace -d $OUT/trace_pusch -o p -- python tools/bench_pusch_rx.py > /dev/null 2> $OUT/trace_pusch.err &&
rocprofv3 --kernel-trace -d $OUT/trace_nrrx -o p -- python tools/bench_nr_rx.py > /dev/null 2> $OUT/trace_nrrx.err &&
rocprofv3 --kernel-trace -d $OUT/trace_uplink -o p -- python tools/bench_uplink.py > /dev/null 2> $OUT/trace_uplink.err &&
python tools/rocpd_summary.py $OUT/trace_pusch $OUT/trace_nrrx $OUT/trace_uplink > $OUT/kernel_stats_chains.txt &&
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_nr -o p -- python tools/bench_nr.py --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_fetch_nr.err &&
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_nr -o p -- python tools/bench_nr.py --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_write_nr.err &&
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_pusch -o p -- python tools/bench_pusch_rx.py --steps 2 > /dev/null 2> $OUT/pmc_fetch_pusch.err &&
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_pusch -o p -- python tools/bench_pusch_rx.py --steps 2 > /dev/null 2> $OUT/pmc_write_pusch.err &&
python tools/rocpd_summary.py $OUT/pmc_fetch_nr $OUT/pmc_write_nr $OUT/pmc_fetch_pusch $OUT/pmc_write_pusch > $OUT/pmc_chains.txt
echo "profile pass rc=$?"
# the rocpd databases are large (gpurun copies back at most 64 MiB): keep the text summaries only
rm -rf $OUT/trace $OUT/trace_* $OUT/pmc_fetch* $OUT/pmc_write* $OUT/pmc_sq $OUT/pmc_lds $OUT/pmc_lds_nr $OUT/pmc_lds_sync 2>/dev/null
du -sh $OUT
