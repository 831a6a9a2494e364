import sys, ctypes, numpy as np
sys.path.insert(0,'/root/repo')
from tests import helpers
from time_series_spark_amd import _lib, forecaster as fc
def used():
    ctx=fc.get_context(); v=ctypes.c_int32(-1); ctx.check(_lib.load().tsf_last_fit_route(ctx.handle, ctypes.byref(v))); return v.value
for N in (6, 48, 3000):
    spec, ds, y, floor, cap, extra, fut, exf = helpers.make_case('cfg4_holidays', N=N, seed=5)
    r=fc.fit_aligned(spec, ds, y, floor=floor, cap=cap, extra=extra)
    print(N, 'route', used(), 'K', spec.K, 'ok', int((r.status>0).sum()), 'rk', spec.lbfgs.get('residual_kernel'))
