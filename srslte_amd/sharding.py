"""Multi-GPU plumbing: one process per GPU, independent units (subframes, code blocks, captures) are
split into contiguous ranges per rank; the only collective is the broadcast of the cell/decoder
configuration at start-up (RCCL on GPUs, gloo in the CPU tests).  No data-path reduction exists."""
import json

import torch
import torch.distributed as dist


def shard_range(n_units, rank, world):
    """contiguous block [lo, hi) of rank `rank`; sizes differ by at most one"""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_config(cfg, device=None):
    """rank 0 passes a dict of ints, the others None; every rank returns the dict"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return cfg
    dev = device if device is not None else torch.device("cpu")
    if dist.get_rank() == 0:
        blob = json.dumps(cfg, sort_keys=True).encode()
        n = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    else:
        n = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.broadcast(n, src=0)
    buf = torch.zeros(int(n.item()), dtype=torch.uint8, device=dev)
    if dist.get_rank() == 0:
        buf = torch.tensor(list(blob), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=0)
    return json.loads(bytes(buf.cpu().tolist()).decode())


def max_over_ranks(x, device=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=device if device is not None else torch.device("cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
