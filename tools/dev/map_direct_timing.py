"""dev (round 6): converge = MAP computed directly (map_quad_kernel) against the Stan-rule fit and against the continuation
(map_kernel, option map_direct = 0) on the BASELINE shapes of linear / additive models: fit-path kernel time per launch."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from time_series_spark_amd import _lib, forecaster as fc, synth
from time_series_spark_amd.device import DeviceForecaster

dev = torch.device('cuda', 0)
for name, N, T, dt in (('cfg2', 10000, 730, np.float64), ('cfg3_full', 100000, 1095, np.float64), ('cfg5', 1000000, 90, np.float32)):
    if len(sys.argv) > 1 and name not in sys.argv[1].split(','):
        continue
    ds_np, y_np = synth.make_panel(N, T, 'linear', seed=751, dtype=dt)
    seas = fc.ModelSpec.auto_seasonalities(ds_np, yearly=True) if T >= 365 else fc.ModelSpec.auto_seasonalities(ds_np)
    ds, y = torch.from_numpy(ds_np).to(dev), torch.from_numpy(y_np).to(dev)
    res = {}
    for tag, conv, direct in (('stan', _lib.CONVERGE_STAN, -1), ('map_direct', _lib.CONVERGE_MAP, -1), ('map_continuation', _lib.CONVERGE_MAP, 0)):
        if tag == 'map_continuation' and N > 100000:
            continue
        f = DeviceForecaster(fc.ModelSpec(growth='linear', seasonalities=seas, converge=conv), 0)
        f.ctx.set_option('map_direct', direct)
        out = f.alloc_fit_output(N)
        f.fit_aligned(ds, y, out)
        torch.cuda.synchronize()
        f.set_profiling(True)
        for _ in range(3):
            f.fit_aligned(ds, y, out)
        torch.cuda.synchronize()
        kms = f.profile_read()
        ne = out.n_eval.cpu().numpy().astype(np.int64)
        st = out.status.cpu().numpy()
        res[tag] = {'fit_path_ms': [round(float(v), 3) for v in kms], 'series_per_s': N / (min(kms) * 1e-3), 'mean_n_eval': float(ne.mean()), 'max_n_eval': int(ne.max()),
                    'mean_n_iter': float(out.n_iter.cpu().numpy().mean()),
                    'status_counts': {str(int(k)): int(v) for k, v in zip(*np.unique(st, return_counts=True))}}
        if tag == 'map_direct':
            fv = out.fval.cpu().numpy()
        if tag == 'map_continuation':
            res[tag]['max_abs_fval_difference_to_direct'] = float(np.nanmax(np.abs(out.fval.cpu().numpy() - fv)))
        del f, out
    print(json.dumps({'config': name, 'N': N, 'T': T, **res}), flush=True)
