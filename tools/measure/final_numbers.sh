#!/bin/bash
# The numbers DESIGN.md / INTEGRATION.md quote for the latency paths, in one short pass (GPU box, from the repository root): bench.py, the reference's own
# programs as stopwatch, the TTI probe, the latency kernels' tables, the seam and handle benches.  usage: bash tools/measure/final_numbers.sh [out dir]
O=${1:-gpurun_out/final}
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err && tail -c 400 $O/bench.json &&
python tools/bench_ref_programs.py > $O/ref_programs.json 2> $O/ref_programs.err &&
./tools/probe/tti_probe > $O/tti_probe.txt 2>&1 &&
( python tools/measure/es_time.py 1; python tools/measure/es_time.py 64 ) 2> /dev/null | grep "n_tb" > $O/es_time.txt &&
( python tools/measure/lat_time.py 6144 0; python tools/measure/lat_time.py 5824 1; python tools/measure/lat_time.py 1024 1; python tools/measure/lat_time.py 2048 1 8; python tools/measure/lat_time.py 6144 0 8 ) 2> /dev/null | grep "K=" > $O/lat_time.txt &&
python tools/measure/gen_time.py 2> /dev/null | grep "K=" > $O/gen_time.txt &&
python tools/seam_bench.py > $O/seam_time.json 2> $O/seam_time.err &&
python tools/bench_handle.py > $O/bench_handle.json 2> $O/bench_handle.err &&
python tools/bench_tti.py --out $O/tti.json > /dev/null 2> $O/tti.err &&
( for m in cold init "warmup 3"; do ./tools/probe/warm_probe $m; done ) > $O/warm_probe.txt 2>&1
echo "final numbers rc=$?"
