"""dev probe: how much of the cfg2 launch is the tail?  Fits the bench panel as is, then again with the
series ordered longest-first (n_eval of the first fit: knowledge a real caller does not have) and
shortest-first.  The longest-first time is the launch without its tail."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from time_series_spark_amd import synth  # noqa: E402


def main():
    N = int(os.environ.get('N', 10000))
    dev = torch.device('cuda', 0)
    spec = bench.cfg2_spec()
    ds_np, y_np = synth.make_panel(N, bench.T_POINTS, 'linear', seed=751)
    ds = torch.from_numpy(ds_np).to(dev)
    f = bench.DeviceForecaster(spec, 0)

    def run(yy):
        y = torch.from_numpy(np.ascontiguousarray(yy)).to(dev)
        o = f.alloc_fit_output(y.shape[0])
        f.fit_aligned(ds, y, o)
        torch.cuda.synchronize()
        f.set_profiling(True)
        for _ in range(4):
            f.fit_aligned(ds, y, o)
        torch.cuda.synchronize()
        k = f.profile_read()
        f.set_profiling(False)
        return o, k

    o, k = run(y_np)
    ne = o.n_eval.cpu().numpy()
    out = {'N': N, 'as_is': k, 'mean_evals': float(ne.mean()), 'max_evals': int(ne.max())}
    order = np.argsort(-ne, kind='stable')
    _, out['longest_first'] = run(y_np[order])
    _, out['shortest_first'] = run(y_np[order[::-1]])
    print(json.dumps(out))


if __name__ == '__main__':
    main()
