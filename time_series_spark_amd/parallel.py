"""Multi-GPU layout: series are sharded by id across ranks, one process per GPU, NO data-path
collective (each (series_id, dim_id) model is independent -- the reference gets the same
independence from Spark's hash partitioning, /root/reference/src/jobs/prophet_modeler.py:139-141).
Series i lives on rank i mod world (shard_indices).  torch.distributed is used only for the start/stop
barrier, the max-over-ranks step time and the gather of per-rank counts / results back into series order
(backend "nccl" = RCCL on GPUs, "gloo" in CPU tests)."""
import os

import numpy as np


def env_rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), \
        int(os.environ.get('LOCAL_RANK', '0'))


def shard_indices(n_items, rank, world):
    """The series of rank `rank`: i with i mod world == rank (what bench.py --gpus N and
    forecaster.fit_aligned(devices=...) deal out).  Interleaved, not contiguous: evaluation counts vary
    3-40 x per series and neighbours of a panel tend to be alike, so i mod world evens the ranks out
    (SURVEY 8e); the price is that a rank's panel is a strided gather of the caller's, made once."""
    return np.arange(int(rank), int(n_items), int(world), dtype=np.int64)


def shard_rank(series_index, world):
    """Inverse of shard_indices: the rank that owns series_index."""
    return int(series_index) % int(world)


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


def gather_rows(local_rows, n_items=None, device=None):
    """Rows of a sharded result ([n_r][C] float64 per rank, rank r holding the series
    shard_indices(n_items, r, world) in that order) gathered on every rank and put back in SERIES order.
    n_items None: plain concatenation in rank order (per-rank summaries)."""
    import torch
    import torch.distributed as dist
    local_rows = np.ascontiguousarray(local_rows, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return local_rows
    if dist.get_backend() == 'gloo':
        device = 'cpu'
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
    parts = [b[:c].cpu().numpy() for b, c in zip(bufs, counts)]
    if n_items is None:
        return np.concatenate(parts, axis=0)
    out = np.zeros((int(n_items), C))
    for r, part in enumerate(parts):
        idx = shard_indices(n_items, r, world)
        if len(idx) != part.shape[0]:
            raise ValueError('rank %d holds %d rows, its shard has %d series' % (r, part.shape[0], len(idx)))
        out[idx] = part
    return out
