#!/bin/bash
out=gpurun_out/r2b/noslp_ab.txt; mkdir -p gpurun_out/r2b; : > $out
for lib in libsrsran_phy_hip ab_noslp; do
  export SRSRAN_HIP_LIB=$PWD/srslte_amd/lib/$lib.so
  r=$(timeout -k 10 300 python bench.py --steps 5 --warmup 2 --only ldpc,uplink --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lte_ofdm', d.get('ofdm_msamples_per_s'), 'nr_ofdm', d['extra']['ldpc']['ofdm_msamples_per_s'], 'uplink', d['extra']['uplink']['value'])") || exit 1
  echo "$lib : $r" | tee -a $out
  r=$(timeout -k 10 200 python tools/dbg/dft_time.py 2>&1 | tail -4 | tr '\n' ' ')
  echo "$lib dft: $r" | tee -a $out
  r=$(timeout -k 10 200 python tools/bench_pusch_rx.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pusch_rx', d['value'], d.get('ms_per_step'))")
  echo "$lib : $r" | tee -a $out
done
