import sys, json
sys.path.insert(0,'.'); sys.path.insert(0,'./tools')
import os

import torch, bench, bench_legs as L
import srslte_amd as S
S.capi.check(S.lib().srsran_hip_set_device(0), "dev")
torch.cuda.set_device(0)
ctx = bench.Ctx(0, 1, 0, torch.device("cuda",0), torch.device("cuda",0), None, torch.cuda.current_stream().cuda_stream)
for sf in (46, 92, 184):
    e = L.leg_uplink(ctx, steps=4, warmup=1, want_cpu=False, sf=sf)
    print(sf, round(e["value"]), round(e["ms_per_step"],3), e["tb_crc_ok"], e["stage_ms"], flush=True)
    torch.cuda.empty_cache()
