"""dev (round 6): a residual-form launch against the point at which its fits are handed to the cooperative kernel
(tsf_spec.coop_after: -1 = the tail rule -- when no more than ~2 fits per CU are still running; n >= 0 = every fit that
has used n evaluations): BASELINE-shaped panels of the reference's model."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import bench_configs as bc
from time_series_spark_amd import forecaster as fc, synth
from time_series_spark_amd.device import DeviceForecaster

H = 90
for name in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['ref10k', 'cfg4', 'cfg1']):
    desc, spec, ds_np, y_np, floor, cap, extra, exf, bps = bc.build(name)
    dev = torch.device('cuda', 0)
    N, T = y_np.shape
    to = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ds, y, fl, cp, ex = to(ds_np), to(y_np), to(floor), to(cap), to(extra)
    base = None
    for after in [int(v) for v in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['-1', '300', '600', '1000', '1500', '2500', '4000'])]:
        sp = fc.ModelSpec.from_dict(dict(spec.to_dict(), lbfgs=dict(spec.lbfgs, coop_after=after)))
        f = DeviceForecaster(sp, 0)
        out = f.alloc_fit_output(N)
        f.fit_aligned(ds, y, out, floor=fl, cap=cp, extra=ex)
        torch.cuda.synchronize()
        f.set_profiling(True)
        for _ in range(2):
            f.fit_aligned(ds, y, out, floor=fl, cap=cp, extra=ex)
        torch.cuda.synchronize()
        kms = f.profile_read()
        ne = out.n_eval.cpu().numpy().astype(np.int64)
        th = out.theta.cpu().numpy()
        if base is None:
            base = th
        print('%-8s coop_after %5d: fit %8.2f ms  (max evals %d, over the hand-over point: %d series, their evaluations beyond it %.2f M)  bits %s'
              % (name, after, float(np.mean(kms)), ne.max(), int((ne > after).sum()) if after >= 0 else -1,
                 float(np.maximum(ne - after, 0).sum()) / 1e6 if after >= 0 else -1, 'same' if np.array_equal(th, base) else 'DIFFER'), flush=True)
        del f, out
