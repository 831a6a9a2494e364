"""dev: converge = MAP on the GPU against oracle/true_map.py (tools/true_map_solve.py) on a few series per BASELINE shape."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import true_map_solve as tms
from time_series_spark_amd import _lib, forecaster as fc, synth

H = 90
for kind, n in [a.split(':') for a in (sys.argv[1:] or ['cfg2:32', 'ref:16', 'cfg5:32', 'cfg4:8'])]:
    n = int(n)
    ds, y, cap, kw, hol = tms.panel(kind, n)
    seas = {'cfg5': [tms.WEEKLY]}.get(kind, [tms.YEARLY, tms.WEEKLY])
    extra, ex, exf = None, None, None
    fut = ds[-1] + synth.DAY_NS * np.arange(1, H + 1)
    if hol is not None:
        allm, names = synth.holiday_matrix(np.concatenate([ds, fut]), 10)
        ex, exf = np.ascontiguousarray(allm[:, :len(ds)]), np.ascontiguousarray(allm[:, len(ds):])
        extra = [{'name': nm} for nm in names]
    mk = lambda **o: fc.ModelSpec(growth=kw['growth'], seasonality_mode=kw['seasonality_mode'], seasonalities=seas, extra=extra, **o)
    fl = np.zeros(n)
    capv = cap if kw['growth'] == 'logistic' else None
    t0 = time.time(); rs = fc.fit_aligned(mk(), ds, y, floor=fl, cap=capv, extra=ex); t_stan = time.time() - t0
    t0 = time.time(); rm = fc.fit_aligned(mk(converge=_lib.CONVERGE_MAP), ds, y, floor=fl, cap=capv, extra=ex); t_map = time.time() - t0
    out = os.path.join(tempfile.gettempdir(), 'tm_%s.npz' % kind)
    t0 = time.time(); subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'true_map_solve.py'), kind, str(n), out]); t_cpu = time.time() - t0
    z = np.load(out)
    pred = lambda th, r: fc.predict(mk(), th, r.y_scale, r.grid, fut, floor=fl, cap=capv, extra_future=exf)
    ym, yg, ys = pred(z['theta_map'], rm), pred(rm.theta, rm), pred(rs.theta, rs)
    rel = lambda a: np.max(np.abs(a - ym) / np.abs(ym), axis=1)
    rg, r0 = rel(yg), rel(ys)
    print('%s n=%d: stan fit %.1f ms (evals mean %.0f), map fit %.1f ms (evals mean %.0f max %d, iters mean %.0f); status %s' % (
        kind, n, 1e3 * t_stan, rs.n_eval.mean(), 1e3 * t_map, rm.n_eval.mean(), rm.n_eval.max(), (rm.n_iter - rs.n_iter).mean(),
        dict(zip(*np.unique(rm.status, return_counts=True)))))
    print('   forecast max-over-horizon rel err vs TRUE MAP: stan-rule median %.2e max %.2e | MAP mode median %.2e p99 %.2e max %.2e ; kkt(true map) max %.1e; cpu solver %.1f s; f_gpu - f_map max %.2e min %.2e'
          % (np.median(r0), r0.max(), np.median(rg), np.quantile(rg, 0.99), rg.max(), z['kkt'].max(), t_cpu, (rm.fval - z['f_map']).max(), (rm.fval - z['f_map']).min()), flush=True)
