"""Multi-GPU layout: series are sharded by id across ranks, one process per GPU, NO data-path
collective (each (series_id, dim_id) model is independent -- the reference gets the same
independence from Spark's hash partitioning, /root/reference/src/jobs/prophet_modeler.py:139-141).
torch.distributed is used only for the start/stop barrier, the max-over-ranks step time and the
gather of per-rank counts/results (backend "nccl" = RCCL on GPUs, "gloo" in CPU tests)."""
import os

import numpy as np


def env_rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), \
        int(os.environ.get('LOCAL_RANK', '0'))


def shard_bounds(n_items, rank, world):
    """Contiguous block partition [lo, hi) of n_items series over `world` ranks (sizes differ by
    at most one).  Contiguous keeps each rank's panel a dense [N_r][T] slab."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_of(series_index, n_items, world):
    """Inverse of shard_bounds: which rank owns series_index."""
    base, rem = divmod(int(n_items), int(world))
    cut = rem * (base + 1)
    if series_index < cut:
        return series_index // (base + 1)
    return rem + (series_index - cut) // max(base, 1)


def init_process_group(backend=None):
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # TSF_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses two ranks per device) --
            # how the multi-rank path of bench.py is exercised on a 1-GPU box
            backend = os.environ.get('TSF_DIST_BACKEND') or None
        if backend is None:
            import torch
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (step time)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if dist.get_backend() == 'gloo':
        device = 'cpu'
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if dist.get_backend() == 'gloo':
        device = 'cpu'
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_rows(local_rows, device=None):
    """Concatenate per-rank [n_r][C] float64 arrays on every rank, in rank order (results of a
    sharded fit; rank order == series order because shards are contiguous)."""
    import torch
    import torch.distributed as dist
    local_rows = np.ascontiguousarray(local_rows, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return local_rows
    world = dist.get_world_size()
    dev = device or 'cpu'
    n = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    C = local_rows.shape[1]
    mx = max(counts)
    pad = torch.zeros((mx, C), dtype=torch.float64, device=dev)
    pad[:local_rows.shape[0]] = torch.from_numpy(local_rows).to(dev)
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return np.concatenate([b[:c].cpu().numpy() for b, c in zip(bufs, counts)], axis=0)
