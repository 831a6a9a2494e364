"""dev (round 6): a residual-form launch with workgroups of the cooperative kernel resident BESIDE the one-wave kernel
(option coop_live = number of workgroups, coop_live_after = evaluations a fit has spent when it hands itself over):
BASELINE-shaped panels of the reference's model.  Prints the launch time, how many fits were handed over, and whether a
bit moved."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
import bench_configs as bc
from time_series_spark_amd import forecaster as fc, synth
from time_series_spark_amd.device import DeviceForecaster

names = sys.argv[1].split(',') if len(sys.argv) > 1 else ['ref10k', 'cfg4', 'ref100k']
lives = [int(v) for v in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['0', '8', '16'])]
afters = [int(v) for v in (sys.argv[3].split(',') if len(sys.argv) > 3 else ['2500'])]
for name in names:
    desc, spec, ds_np, y_np, floor, cap, extra, exf, bps = bc.build(name)
    dev = torch.device('cuda', 0)
    N, T = y_np.shape
    to = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ds, y, fl, cp, ex = to(ds_np), to(y_np), to(floor), to(cap), to(extra)
    base = None
    for live in lives:
        for after in (afters if live else afters[:1]):
            f = DeviceForecaster(spec, 0)
            f.ctx.set_option('coop_live', live)
            f.ctx.set_option('coop_live_after', after)
            out = f.alloc_fit_output(N)
            f.fit_aligned(ds, y, out, floor=fl, cap=cp, extra=ex)
            torch.cuda.synchronize()
            f.set_profiling(True)
            for _ in range(3):
                f.fit_aligned(ds, y, out, floor=fl, cap=cp, extra=ex)
            torch.cuda.synchronize()
            kms = f.profile_read()
            ne = out.n_eval.cpu().numpy().astype(np.int64)
            th = out.theta.cpu().numpy()
            if base is None:
                base = th
                top = np.argsort(-ne)[:4]
                print('%-8s longest fits: %s' % (name, ', '.join('series %d: %d evaluations' % (int(i), int(ne[i])) for i in top)), flush=True)
            print('%-8s coop_live %3d after %5d: fit %8.2f ms (%s)  max evals %d, fits beyond the point %d  bits %s'
                  % (name, live, after, float(np.mean(kms)), ' '.join('%.1f' % v for v in kms), ne.max(), int((ne > after).sum()),
                     'same' if np.array_equal(th, base, equal_nan=True) else 'DIFFER'), flush=True)
            del f, out
