"""dev tool: where the host-pointer entry point (tsf_fit_aligned with host buffers) spends its time on the bench
panel, next to the PCIe floor of the same bytes.  TSF_HOST_TIMING=1 makes the library print its phases.

  TSF_HOST_TIMING=1 python tools/host_entry_probe.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from time_series_spark_amd import forecaster as fc, synth  # noqa: E402

N, T, H = 10000, 730, 90
ds, y = synth.make_panel(N, T, 'linear', seed=751)
spec = fc.ModelSpec(growth='linear', seasonalities=[{'name': 'yearly', 'period': 365.25, 'fourier_order': 10},
                                                    {'name': 'weekly', 'period': 7, 'fourier_order': 3}])
fut = ds[-1] + synth.DAY_NS * np.arange(1, H + 1)
dev = torch.device('cuda', 0)
# PCIe floor: the same 58 MB, pageable and pinned
yt = torch.from_numpy(y)
yp = yt.pin_memory()
for name, src in (('pageable', yt), ('pinned', yp)):
    d = src.to(dev); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        d = src.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print('H2D %s %.1f MB: %.3f ms (%.1f GB/s)' % (name, y.nbytes / 1e6, 1e3 * dt, y.nbytes / dt / 1e9), flush=True)
for rep in range(4):
    t0 = time.perf_counter()
    r = fc.fit_aligned(spec, ds, y)
    t1 = time.perf_counter()
    fc.predict(spec, r.theta, r.y_scale, r.grid, fut)
    t2 = time.perf_counter()
    print('call %d: fit_aligned (host pointers) %.3f ms, predict %.3f ms' % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1)), flush=True)
