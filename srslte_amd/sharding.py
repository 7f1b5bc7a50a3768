"""Multi-GPU plumbing: one process per GPU, independent units (subframes, code blocks, captures) are
split into contiguous ranges per rank; the only collective is the broadcast of the cell/decoder
configuration at start-up (RCCL on GPUs, gloo in the CPU tests).  No data-path reduction exists."""
import datetime
import json
import os

import torch
import torch.distributed as dist


def shard_range(n_units, rank, world):
    """contiguous block [lo, hi) of rank `rank`; sizes differ by at most one"""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_config(cfg, device=None):
    """rank 0 passes a dict of ints, the others None; every rank returns the dict"""
    if not (dist.is_available() and dist.is_initialized()):
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
    if not (dist.is_available() and dist.is_initialized()):
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device=device if device is not None else torch.device("cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_store(rank, world, timeout):
    """The rendezvous store of the job, under a prefix of this job's own.

    Launched by bench.py's own launcher (or by hand): rank 0 hosts a TCPStore on MASTER_ADDR:MASTER_PORT.  Launched by
    `python -m torch.distributed.run` (what the driver does for N > 1): the elastic agent already listens on MASTER_PORT and says so with
    TORCHELASTIC_USE_AGENT_STORE=True -- every rank, 0 included, is then a CLIENT of the agent's store (a second server on that port is
    EADDRINUSE).  Same rule as torch's own env:// rendezvous (torch/distributed/rendezvous.py: _create_c10d_store)."""
    host, port = os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"])
    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True":
        store = dist.TCPStore(host, port, world, is_master=False, timeout=timeout)
    else:
        store = dist.TCPStore(host, port, world, is_master=(rank == 0), timeout=timeout, wait_for_workers=True, multi_tenant=True)
    # the agent's store outlives a restart of the workers: keep this job's keys apart from an earlier attempt's
    run = "%s/%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "run"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
    return dist.PrefixStore("srslte_amd/" + run, store)


def init_collectives(rank, world, device=None, prefer="nccl", timeout_s=120):
    """Creates the job's process group -- also at world size 1, so that a 1-GPU run executes the very RCCL bootstrap an N-GPU run needs.

    `prefer="nccl"` (= RCCL on ROCm; `device` is this rank's GPU): every rank tries to build the RCCL communicator and run one
    all-reduce, then publishes the outcome in the rendezvous store, and the ranks read each other's outcome BEFORE anyone decides what
    to do next.  All succeeded -> RCCL.  Anything else -> every rank tears its attempt down and the group is rebuilt on gloo over the
    SAME store (no second port, no rank left behind on the other backend); the label says which ranks failed and why.  A rank that
    hangs inside the RCCL bootstrap because a peer never arrived ends in the process group's timeout, i.e. in a failed run.
    `prefer="gloo"`: gloo directly (CPU rehearsals, ranks sharing one GPU).

    Returns (label, payload_device): the device collective payloads must live on."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:  # a lone worker that nobody launched: any free port will do
        import socket

        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        s.close()
    to = datetime.timedelta(seconds=timeout_s)
    store = job_store(rank, world, to)
    cpu = torch.device("cpu")
    if prefer != "nccl":
        dist.init_process_group("gloo", store=dist.PrefixStore("gloo", store), rank=rank, world_size=world, timeout=to)
        return "gloo", cpu
    err = ""
    try:
        kw = {"device_id": device} if device is not None and device.type == "cuda" else {}
        dist.init_process_group("nccl", store=dist.PrefixStore("rccl", store), rank=rank, world_size=world, timeout=to, **kw)
        t = torch.ones(1, device=device)
        dist.all_reduce(t)  # the first collective builds the communicator: fail here, not in the timed region
        torch.cuda.synchronize()
        if int(t.item()) != world:
            err = "all_reduce over %d ranks returned %s" % (world, t.item())
    except Exception as e:  # noqa: BLE001 -- whatever it is, the other ranks must hear about it
        err = "%s: %s" % (type(e).__name__, str(e).replace("\n", " ")[:160])
    store.set("rccl_outcome_%d" % rank, err if err else "ok")
    outcomes = [store.get("rccl_outcome_%d" % r).decode() for r in range(world)]  # blocks until every rank has published
    bad = [r for r, o in enumerate(outcomes) if o != "ok"]
    if not bad:
        return "nccl (RCCL)", device
    if dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    dist.init_process_group("gloo", store=dist.PrefixStore("gloo", store), rank=rank, world_size=world, timeout=to)
    return "gloo (RCCL failed on ranks %s: %s)" % (bad, outcomes[bad[0]]), cpu


def ranks_seen(local_rank, device=None, payload_device=None):
    """one record per rank -- rank, local rank, host, the GPU it is bound to (PCI bus id / uuid) -- gathered over the job's process group:
    a line that lists N ranks on N different devices is the evidence that N GPUs took part"""
    import socket

    me = {"rank": dist.get_rank() if dist.is_initialized() else 0, "local_rank": local_rank, "host": socket.gethostname(), "device": None}
    if device is not None and device.type == "cuda":
        p = torch.cuda.get_device_properties(device)
        me["device"] = {"index": device.index, "name": p.name, "uuid": str(getattr(p, "uuid", "")),
                        "pci": "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", 0), getattr(p, "pci_device_id", 0))}
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [me]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, me)
    return out
