"""Two panels on two HIP streams (one tsf_ctx each): how much of a launch's tail -- the last long
series of panel A keeping a few wavefronts busy -- is filled by panel B.  The device entry points of
the C-ABI are asynchronous on the stream they are given, so a caller with more than one batch gets
this by construction.  One JSON line per configuration.

  python tools/bench_overlap.py cfg2 ref10k
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench_configs as bc  # noqa: E402
from time_series_spark_amd.device import DeviceForecaster  # noqa: E402


def run(name, reps=3):
    import torch
    dev = torch.device('cuda', 0)
    desc, spec, ds_np, y_np, floor, cap, extra, exf, bps = bc.build(name)
    N, T = y_np.shape
    to = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    # panel B: the same series in reverse order (another arrival order of the long ones)
    panels = []
    for rev in (False, True):
        sl = slice(None, None, -1) if rev else slice(None)
        f = DeviceForecaster(spec, 0)
        panels.append((f, to(ds_np), to(y_np[sl]), f.alloc_fit_output(N),
                       to(None if floor is None else floor[sl]), to(None if cap is None else cap[sl]), to(extra)))
    # TSF_OVERLAP_PRIO=1: the second stream with high priority (its own hardware queue class)
    prio = -1 if os.environ.get('TSF_OVERLAP_PRIO') else 0
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=prio)]

    def fit(i, stream):
        f, ds, y, out, fl, cp, ex = panels[i]
        with torch.cuda.stream(stream):
            f.fit_aligned(ds, y, out, floor=fl, cap=cp, extra=ex)

    def timed(fn):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    fit(0, streams[0]); fit(1, streams[1]); torch.cuda.synchronize()      # warm-up
    one = timed(lambda: fit(0, streams[0]))
    seq = timed(lambda: (fit(0, streams[0]), fit(1, streams[0])))
    par = timed(lambda: (fit(0, streams[0]), fit(1, streams[1])))
    a, b = panels[0][3], panels[1][3]
    same = bool(torch.equal(a.theta, torch.flip(b.theta, dims=[0])))
    print(json.dumps({'config': name, 'workload': desc, 'series_per_panel': N,
                      'one_panel_ms': 1e3 * one, 'two_panels_one_stream_ms': 1e3 * seq,
                      'two_panels_two_streams_ms': 1e3 * par,
                      'series_per_s_two_streams': 2 * N / par, 'series_per_s_one_stream': 2 * N / seq,
                      'results_identical_across_orders': same}), flush=True)


if __name__ == '__main__':
    for nm in (sys.argv[1:] or ['cfg2', 'ref10k']):
        run(nm)
