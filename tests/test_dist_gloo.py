"""World-size-2 CPU test of the multi-GPU plumbing (gloo): the i mod world shard-by-id partition,
max-over-ranks step time, gather of per-rank results back into series order.  The data path itself has
no collective (series are independent), so there is nothing else to exercise."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    os.environ.update({'RANK': str(rank), 'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank),
                       'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port)})
    import torch.distributed as dist
    from time_series_spark_amd import parallel
    r, w, _ = parallel.init_process_group(backend='gloo')
    mine = parallel.shard_indices(n_items, r, w)
    # stand-in for the per-rank fit: row i -> [i, 2i]
    local = np.stack([mine.astype(np.float64), 2.0 * mine], axis=1)
    parallel.barrier()
    t = parallel.max_over_ranks(1.0 + r)
    tot = parallel.sum_over_ranks(len(mine))
    allrows = parallel.gather_rows(local, n_items)
    q.put((r, list(mine), t, tot, allrows))
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    world, n_items = 2, 11
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [[0, 2, 4, 6, 8, 10], [1, 3, 5, 7, 9]]
    for r in res:
        assert r[2] == 2.0                       # max over ranks of (1 + rank)
        assert r[3] == n_items
        assert np.array_equal(r[4][:, 0], np.arange(n_items)) and np.array_equal(r[4][:, 1], 2.0 * np.arange(n_items))


def _fit_worker(rank, world, port, q):
    """One rank of the sharded fit as bench.py --gpus N and forecaster.fit_aligned(devices=...) lay it out:
    series i belongs to rank i mod world; every rank fits its share (here with the CPU oracle standing in
    for the device -- the product has no CPU path), no collective touches the data, the results are gathered
    and put back in series order."""
    os.environ.update({'RANK': str(rank), 'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank),
                       'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port)})
    import torch.distributed as dist
    from oracle import canon_lib as cl
    from time_series_spark_amd import parallel, synth
    from tests import helpers
    r, w, _ = parallel.init_process_group(backend='gloo')
    spec = helpers.make_case('short_90')[0]
    ds, y = synth.make_panel(7, 90, 'linear', seed=5)
    csp = helpers.oracle_spec(spec)
    mine = parallel.shard_indices(len(y), r, w)
    rows = []
    for n in mine:
        o = cl.fit(csp, ds, y[n])
        rows.append(np.concatenate([[n, o['n_eval'], o['f']], o['theta']]))
    parallel.barrier()
    allrows = parallel.gather_rows(np.array(rows), len(y))
    q.put((r, allrows))
    dist.destroy_process_group()


def test_two_rank_sharded_fit_equals_the_single_process_fit():
    from oracle import canon_lib as cl
    from time_series_spark_amd import synth
    from tests import helpers
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fit_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    spec = helpers.make_case('short_90')[0]
    ds, y = synth.make_panel(7, 90, 'linear', seed=5)
    csp = helpers.oracle_spec(spec)
    for _r, rows in res:                      # every rank holds the gathered result
        assert rows.shape[0] == len(y)
        # gather_rows puts the rows back in series order (rank r holds the series i with i mod world == r)
        assert list(rows[:, 0].astype(int)) == list(range(len(y)))
        back = rows
        for n in range(len(y)):
            o = cl.fit(csp, ds, y[n])
            assert back[n, 1] == o['n_eval'] and back[n, 2] == o['f']
            assert np.array_equal(back[n, 3:], o['theta'])
