"""dev tool: what tsf_set_cost_hints buys on the configurations of tools/bench_configs.py -- each panel fitted as is,
then again with the evaluation counts of that fit as hints (fit-path kernel time from the library's HIP events).

  python tools/hints_probe.py cfg2 ref10k cfg4
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench_configs as bc  # noqa: E402
from time_series_spark_amd.device import DeviceForecaster  # noqa: E402

dev = torch.device('cuda', 0)
for name in sys.argv[1:]:
    desc, spec, ds_np, y_np, floor, cap, extra, exf, bps = bc.build(name)
    N, T = y_np.shape
    to = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ds, y, fl, cp, ex = to(ds_np), to(y_np), to(floor), to(cap), to(extra)
    f = DeviceForecaster(spec, 0)
    out = f.alloc_fit_output(N)
    f.fit_aligned(ds, y, out, floor=fl, cap=cp, extra=ex)
    torch.cuda.synchronize()
    ne = out.n_eval.cpu().numpy().astype(np.int32)
    ref = out.theta.clone()
    res = {}
    for tag in ('as_is', 'hinted'):
        f.set_profiling(True)
        for _ in range(3):
            if tag == 'hinted':
                f.set_cost_hints(ne)
            f.fit_aligned(ds, y, out, floor=fl, cap=cp, extra=ex)
        torch.cuda.synchronize()
        res[tag] = [round(float(v), 3) for v in f.profile_read()]
        f.set_profiling(False)
    print(json.dumps({'config': name, 'series': N, 'fit_kernel_ms': res, 'max_evals': int(ne.max()), 'mean_evals': float(ne.mean()),
                      'same_bits': bool(torch.equal(ref, out.theta))}), flush=True)
