import sys, os, time, ctypes, numpy as np
sys.path.insert(0,'/root/repo')
from time_series_spark_amd import _lib, forecaster as fc, synth
YEARLY = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
WEEKLY = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
T=1095; N=20000
ds=synth.daily_grid(T); ex,names=synth.holiday_matrix(ds,10)
_,y=synth.make_panel(N,T,'logistic',seed=751,holidays=ex)
spec=fc.ModelSpec(growth='logistic',seasonality_mode='multiplicative',seasonalities=[YEARLY,WEEKLY],extra=[{'name':n} for n in names])
cap=y.max(axis=1)*1.1
ctx=fc.get_context(); L=_lib.load(); ms=ctypes.c_float(0)
for rep in range(2):
    ctx.check(L.tsf_set_profiling(ctx.handle,1))
    t0=time.time(); r=fc.fit_aligned(spec,ds,y,floor=np.zeros(N),cap=cap,extra=ex); dt=time.time()-t0
    ctx.check(L.tsf_last_fit_kernel_ms(ctx.handle,ctypes.byref(ms)))
print('T',T,'N',N,'sparse' if os.environ.get('TSF_SPARSE_EXTRA')!='0' else 'dense','fit kernel ms %.1f'%ms.value,'mean evals %.1f max %d'%(r.n_eval.mean(), r.n_eval.max()), 'ok', int((r.status>0).sum()))
