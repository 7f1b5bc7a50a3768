#!/bin/bash
# Round measurement pass on the GPU box (run through gpurun from the repository root):
#   GPU test suite, bench.py (headline + extra legs), the chain benches, the handle-API bench, rocprofv3 kernel trace of bench.py,
#   separate PMC passes (traffic, SQ, LDS), the turbo launch-shape variants and the PSS kernels with their counters, and the lane-mapping probe.
# Outputs go to gpurun_out/round/; the summaries to keep are copied into profiles/ by hand afterwards.
set -o pipefail
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/round4}
mkdir -p $OUT
B="python bench.py --steps 2 --warmup 1 --extra-steps 2 --no-cpu"
if [ "$SKIP_PYTEST" != "1" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
fi
python bench.py > $OUT/bench.json 2> $OUT/bench.err && tail -1 $OUT/bench.json | cut -c1-300 &&
python tools/bench_handle.py > $OUT/bench_handle.json 2> $OUT/bench_handle.err &&
python tools/bench_tti.py --out $OUT/tti.json > /dev/null 2> $OUT/tti.err &&
( rocprofv3 --kernel-trace --stats -d $OUT/trace_tti -o t -- python tools/bench_tti.py --calls 100 --snrs 6.0 --ntb 1,64 > /dev/null 2> $OUT/trace_tti.err; python tools/rocpd_summary.py $OUT/trace_tti > $OUT/tti_kernel_stats.txt; rm -rf $OUT/trace_tti ) &&
python tools/seam_bench.py > $OUT/seam_time.json 2> $OUT/seam_time.err &&
python tools/bench_ref_programs.py > $OUT/ref_programs.json 2> $OUT/ref_programs.err &&
( for m in cold init "warmup 3"; do ./tools/probe/warm_probe $m; done ) > $OUT/warm_probe.txt 2>&1 &&
./tools/probe/tti_probe > $OUT/tti_probe.txt 2>&1 &&
python tools/measure/ldpc_small.py > $OUT/ldpc_small.txt 2> /dev/null &&
python tools/measure/pss_ab.py > $OUT/pss_ab.txt 2> /dev/null &&
python tools/measure/es_time.py 1 > $OUT/es_time.txt 2> /dev/null && python tools/measure/es_time.py 64 >> $OUT/es_time.txt 2> /dev/null &&
( python tools/measure/lat_time.py 6144 0; python tools/measure/lat_time.py 5824 1; python tools/measure/lat_time.py 1024 1; python tools/measure/lat_time.py 6144 0 8 ) 2> /dev/null | grep "K=" > $OUT/lat_time.txt &&
python tools/measure/gen_time.py 2> /dev/null | grep "K=" > $OUT/gen_time.txt &&
( bash tools/measure/grant_timeline.sh $PWD/$OUT/tl > /dev/null 2>&1; grep -v "simple_timer\|generateRocpd\|tool.cpp" $OUT/tl/timeline.txt > $OUT/grant_timeline.txt; rm -rf $OUT/tl ) &&
python tools/bench_sch.py > $OUT/bench_sch.json 2> $OUT/bench_sch.err &&
python tools/bench_pusch_rx.py > $OUT/bench_pusch_rx.json 2> $OUT/bench_pusch_rx.err &&
python tools/bench_nr_rx.py > $OUT/bench_nr_rx.json 2> $OUT/bench_nr_rx.err &&
python tools/bench_sch_nr.py > $OUT/bench_sch_nr.json 2> $OUT/bench_sch_nr.err &&
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python bench.py --steps 5 --warmup 1 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/trace.err &&
python tools/rocpd_summary.py $OUT/trace > $OUT/kernel_stats.txt &&
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- $B > /dev/null 2> $OUT/pmc_fetch.err &&
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- $B > /dev/null 2> $OUT/pmc_write.err &&
python tools/pmc_to_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/traffic.json &&
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/pmc_sq -o p -- $B > /dev/null 2> $OUT/pmc_sq.err &&
python tools/rocpd_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq > $OUT/pmc.txt &&
( rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE -d $OUT/pmc_lds -o p -- $B > /dev/null 2> $OUT/pmc_lds.err &&
  python tools/rocpd_summary.py $OUT/pmc_lds > $OUT/pmc_lds.txt ) || echo "LDS counter pass failed (see pmc_lds.err)"
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_lds
if [ "$SKIP_VARIANTS" != "1" ]; then
# ---- turbo launch-shape variants (DESIGN.md par. 3.2): time, traffic, VALU and wait counters of each
T="python bench.py --steps 3 --warmup 1 --no-extras --no-cpu"
: > $OUT/turbo_variants.txt
export SRSRAN_HIP_LIB=$PWD/tools/probe/lib/libsrsran_phy_hip_variants.so  # the measured-and-rejected kernels live in a library of their own
for V in product waves1 persistent; do
  export SRSRAN_HIP_TDEC_VARIANT=$V
  echo "==== variant $V" >> $OUT/turbo_variants.txt
  $T 2> /dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench: %.1f Mbit/s, %.3f ms per step, turbo kernel %.3f ms' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms']))" >> $OUT/turbo_variants.txt &&
  rocprofv3 --pmc FETCH_SIZE -d $OUT/v_f -o p -- $T > /dev/null 2> $OUT/v.err &&
  rocprofv3 --pmc WRITE_SIZE -d $OUT/v_w -o p -- $T > /dev/null 2> $OUT/v.err &&
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/v_s -o p -- $T > /dev/null 2> $OUT/v.err &&
  python tools/rocpd_summary.py $OUT/v_f $OUT/v_w $OUT/v_s | grep -E "tdec_win" >> $OUT/turbo_variants.txt
  rm -rf $OUT/v_f $OUT/v_w $OUT/v_s
done
unset SRSRAN_HIP_TDEC_VARIANT SRSRAN_HIP_LIB
# ---- the 8-bit decoders (what srsenb / srsue run): time against the 16-bit one and the reference's, traffic and VALU counters
python tools/measure/turbo8_time.py > $OUT/turbo8_time.txt 2> /dev/null
: > $OUT/pmc_turbo8.txt
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  rocprofv3 --pmc $C -d $OUT/t8 -o p -- python tools/measure/turbo8_time.py > /dev/null 2> $OUT/v.err &&
  python tools/rocpd_summary.py $OUT/t8 | grep -E "tdec_win" >> $OUT/pmc_turbo8.txt
  rm -rf $OUT/t8
done
# ---- PSS correlation kernels (DESIGN.md par. 3.4): the product (one wave per block), two waves per block, round 1's workgroup per block
C="python bench.py --steps 3 --warmup 1 --only cellsearch --no-cpu"
: > $OUT/pss_variants.txt
export SRSRAN_HIP_LIB=$PWD/tools/probe/lib/libsrsran_phy_hip_variants.so
for V in wave recompute pair block; do
  export SRSRAN_HIP_PSS_VARIANT=$V
  echo "==== variant $V" >> $OUT/pss_variants.txt
  $C 2> /dev/null | python -c "import sys,json; e=json.loads(sys.stdin.read().strip().splitlines()[-1])['extra']['cellsearch']; print('bench: %.3f ms per 256 captures, %.0f captures/s, results correct: %s' % (e['ms_per_step'], e['captures_per_s'], e['results_correct']))" >> $OUT/pss_variants.txt &&
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/v_s -o p -- $C > /dev/null 2> $OUT/v.err &&
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/v_l -o p -- $C > /dev/null 2> $OUT/v.err &&
  rocprofv3 --pmc FETCH_SIZE -d $OUT/v_f -o p -- $C > /dev/null 2> $OUT/v.err &&
  rocprofv3 --pmc WRITE_SIZE -d $OUT/v_w -o p -- $C > /dev/null 2> $OUT/v.err &&
  python tools/rocpd_summary.py $OUT/v_s $OUT/v_l $OUT/v_f $OUT/v_w | grep -E "pss_(wave|pair|block)_kernel" >> $OUT/pss_variants.txt
  rm -rf $OUT/v_s $OUT/v_l $OUT/v_f $OUT/v_w
done
unset SRSRAN_HIP_PSS_VARIANT SRSRAN_HIP_LIB
fi
( cd tools/probe && hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o roundtrip_probe roundtrip_probe.hip -lpthread 2> /dev/null; ./roundtrip_probe 0 ) > $OUT/roundtrip_probe.txt 2>&1
( cd tools/probe && gcc -O2 -I../../include seam_threads.c -o seam_threads -L../../srslte_amd/lib -lsrsran_phy_hip -Wl,-rpath,'$ORIGIN/../../srslte_amd/lib' -lpthread -lm 2> /dev/null
  echo "== the library's default (it asks for 8 hardware queues)"; ./seam_threads; echo "== GPU_MAX_HW_QUEUES=4 (the runtime's own default)"; GPU_MAX_HW_QUEUES=4 ./seam_threads ) > $OUT/seam_threads.txt 2>&1
python tools/measure/enc_time.py > $OUT/enc_time.txt 2> /dev/null
( cd tools/probe && hipcc --offload-arch=gfx950 -O3 -o acs_layout_probe acs_layout_probe.hip 2> /dev/null; ./acs_layout_probe ) > $OUT/acs_layout_probe.txt 2>&1
cat $OUT/acs_layout_probe.txt
echo "profile pass rc=$?"
du -sh $OUT
