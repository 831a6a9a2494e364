"""dev tool (run under rocprofv3): the cfg2 panel fitted once, then `tsf_predict_dev` (point forecast +
the reference's int / clamp step) a few times and `tsf_predict_intervals` on a 2 000-series slice --
the dispatches profiles/r03_predict/ and profiles/r03_intervals/ summarise."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from time_series_spark_amd import forecaster as fc, synth  # noqa: E402
from time_series_spark_amd.device import DeviceForecaster  # noqa: E402

N, T, H = 10000, 730, 90
spec = fc.ModelSpec(growth='linear', seasonalities=[{'name': 'yearly', 'period': 365.25, 'fourier_order': 10},
                                                    {'name': 'weekly', 'period': 7, 'fourier_order': 3}])
ds_np, y_np = synth.make_panel(N, T, 'linear', seed=751)
fut_np = ds_np[-1] + synth.DAY_NS * np.arange(1, H + 1)
dev = torch.device('cuda', 0)
ds, y, fut = (torch.from_numpy(a).to(dev) for a in (ds_np, y_np, fut_np))
f = DeviceForecaster(spec, 0)
out = f.alloc_fit_output(N)
yhat = torch.zeros((N, H), dtype=torch.float64, device=dev)
yint = torch.zeros((N, H), dtype=torch.int32, device=dev)
f.fit_aligned(ds, y, out)
for _ in range(int(os.environ.get('PREDICT_REPS', '5'))):
    f.predict(out, fut, yhat, yint)
torch.cuda.synchronize()
if os.environ.get('WITH_INTERVALS', '1') == '1':
    th = out.theta.cpu().numpy()
    r = fc.predict_intervals(spec, th[:2000], out.y_scale.cpu().numpy()[:2000], out.grid_numpy(), fut_np, seed=7)
    print('intervals', r[1].shape, float(np.nanmean(r[2] - r[1])))
print('predict ok', float(yhat.mean().item()))
