"""The N>1 path: units shard across ranks with no data-path collective; one config broadcast at start.
Exercised with world_size-2 gloo on the CPU (the partition + broadcast logic of srslte_amd.sharding)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from srslte_amd import sharding
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
cfg = sharding.broadcast_config({"n_prb": 100, "n_fft": 2048, "k_cb": 6144, "nit": 8} if dist.get_rank() == 0 else None)
lo, hi = sharding.shard_range(1001, dist.get_rank(), dist.get_world_size())
t = sharding.max_over_ranks(float(dist.get_rank() + 1))
print(json.dumps({"rank": dist.get_rank(), "cfg": cfg, "lo": lo, "hi": hi, "t": t}), flush=True)
dist.destroy_process_group()
"""


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    res = sorted((json.loads(o.strip().splitlines()[-1]) for o in outs), key=lambda d: d["rank"])
    assert res[0]["cfg"] == res[1]["cfg"] == {"n_prb": 100, "n_fft": 2048, "k_cb": 6144, "nit": 8}
    assert (res[0]["lo"], res[0]["hi"], res[1]["lo"], res[1]["hi"]) == (0, 501, 501, 1001)
    assert res[0]["t"] == res[1]["t"] == 2.0


def test_shard_range_partitions_exactly():
    from srslte_amd import sharding

    for n in (0, 1, 7, 64, 1001):
        for w in (1, 2, 3, 8):
            parts = [sharding.shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in parts) - min(h - l for l, h in parts) <= 1
