"""Dev tool: bitwise comparison of the HIP path with the canonical oracle at every level.
Run on the GPU box: `python tools/gpu_debug.py > gpurun_out/debug.log`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from time_series_spark_amd import forecaster as fc, synth
from oracle import canon_lib as cl

def bits(a): return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)
def nbitdiff(a, b): return int(np.sum(bits(a) != bits(b)))

def selftests():
    rng = np.random.default_rng(0)
    n = 200000
    a = rng.normal(0, 1, n) * np.exp(rng.uniform(-30, 30, n)); b = rng.normal(0, 1, n) * np.exp(rng.uniform(-30, 30, n))
    print('div   mismatches', nbitdiff(fc.selftest_math(0, a, b), a / b), 'of', n)
    print('sqrt  mismatches', nbitdiff(fc.selftest_math(1, np.abs(a)), np.sqrt(np.abs(a))), 'of', n)
    L = cl.lib()
    x = rng.uniform(-60, 60, 20000)
    print('exp   mismatches', nbitdiff(fc.selftest_math(2, x), [L.cn_det_exp(v) for v in x]))
    x = np.exp(rng.uniform(-30, 30, 20000))
    print('log   mismatches', nbitdiff(fc.selftest_math(3, x), [L.cn_det_log(v) for v in x]))
    import ctypes
    x = rng.uniform(-5000, 5000, 20000); s = ctypes.c_double(); c = ctypes.c_double(); S = []; C = []
    for v in x:
        L.cn_det_sincos(v, ctypes.byref(s), ctypes.byref(c)); S.append(s.value); C.append(c.value)
    print('sin   mismatches', nbitdiff(fc.selftest_math(4, x), S), ' cos', nbitdiff(fc.selftest_math(5, x), C))
    # fma(a,b,a) reference via exact rational arithmetic is slow; use math.fma if present, else skip
    try:
        import math
        ref = [math.fma(u, v, u) for u, v in zip(a[:20000], b[:20000])]
        print('fma   mismatches', nbitdiff(fc.selftest_math(6, a[:20000], b[:20000]), ref))
    except AttributeError:
        print('fma   (math.fma unavailable, skipped)')

def cl_spec(spec, **opt):
    seas = [(s['period'], s['fourier_order'], s.get('mode', spec.seasonality_mode), s.get('prior_scale', 10.0)) for s in spec.seasonalities]
    ex = [(e.get('mode', spec.seasonality_mode), e.get('prior_scale', 10.0)) for e in spec.extra]
    o = dict(spec.lbfgs); o.update(opt)
    return cl.make_spec(growth=spec.growth, n_changepoints=spec.n_changepoints, changepoint_range=spec.changepoint_range,
                        changepoint_prior_scale=spec.changepoint_prior_scale, seasonalities=seas, extra=ex, **o)

def run_case(name, growth, mode, N, T, yearly=True, max_iters=(1, 2, 5, 20, None)):
    print('=' * 100); print('CASE', name, growth, mode, 'N', N, 'T', T)
    ds, y = synth.make_panel(N, T, 'linear' if growth == 'linear' else 'logistic', seed=11)
    seas = []
    if yearly: seas.append({'name': 'yearly', 'period': 365.25, 'fourier_order': 10})
    seas.append({'name': 'weekly', 'period': 7, 'fourier_order': 3})
    cap = y.max(axis=1) * 1.1; floor = np.zeros(N)
    rng = np.random.default_rng(5)
    for mi in max_iters:
        opt = {} if mi is None else {'max_iter': mi}
        spec = fc.ModelSpec(growth=growth, seasonality_mode=mode, seasonalities=seas, **opt)
        csp = cl_spec(spec)
        if mi == max_iters[0]:
            X, t, grid = fc.design(spec, ds)
            des = cl.design(csp, ds, y[0], 0.0, cap[0])
            print(' design X mismatches', nbitdiff(X, des['X']), 'of', X.size, ' t', nbitdiff(t, des['t']), ' tchange', nbitdiff(grid['t_change'][0][:des['info'].S], des['t_change']), 'S', grid['S'][0], des['info'].S)
            stride = spec.theta_stride
            th = np.zeros((N, stride))
            for n in range(N):
                d = cl.design(csp, ds, y[n], 0.0, cap[n])
                th[n, 0] = d['k0']; th[n, 1] = d['m0']
            th += rng.normal(0, 0.01, th.shape)
            f, g = fc.eval_aligned(spec, ds, y, th, floor=floor, cap=cap)
            bad = 0
            for n in range(N):
                fo, go, rc = cl.eval_at(csp, ds, y[n], th[n], 0.0, cap[n])
                df = nbitdiff(f[n], fo); dg = nbitdiff(g[n], go)
                if df or dg:
                    bad += 1
                    if bad <= 2:
                        w = np.where(bits(g[n]) != bits(go))[0]
                        print('  eval series', n, 'f', f[n], fo, 'gdiff idx', w[:10], 'rel', np.max(np.abs(g[n] - go) / (1e-300 + np.abs(go))))
            print(' eval mismatching series', bad, 'of', N)
        t0 = time.time(); r = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap); dt = time.time() - t0
        nb = 0; worst = 0.0; itd = 0
        for n in range(N):
            o = cl.fit(csp, ds, y[n], 0.0, cap[n])
            d = nbitdiff(r.theta[n], o['theta'])
            if d or r.n_iter[n] != o['n_iter'] or r.status[n] != o['status'] or r.n_eval[n] != o['n_eval']:
                nb += 1
                worst = max(worst, np.max(np.abs(r.theta[n] - o['theta'])))
                if nb <= 3:
                    print('   series', n, 'gpu it/ev/st', r.n_iter[n], r.n_eval[n], r.status[n], 'oracle', o['n_iter'], o['n_eval'], o['status'], 'f', r.fval[n], o['f'], 'maxabs dtheta', np.max(np.abs(r.theta[n] - o['theta'])))
        print(' fit max_iter', mi, ': series differing', nb, 'of', N, 'worst |dtheta|', worst, ' gpu time %.3fs' % dt, 'mean iters', r.n_iter.mean(), 'mean evals', r.n_eval.mean(), 'status', np.unique(r.status, return_counts=True))
    # predict
    fut = ds[-1] + synth.DAY_NS * np.arange(1, 91)
    yh, yi = fc.predict(spec, r.theta, r.y_scale, r.grid, fut, floor=floor, cap=cap, want_int=True)
    nb = 0; worst = 0
    for n in range(N):
        o = cl.fit(csp, ds, y[n], 0.0, cap[n])
        yo, _ = cl.predict(csp, o, fut, 0.0, cap[n])
        nb += nbitdiff(yh[n], yo); worst = max(worst, np.max(np.abs(yh[n] - yo) / np.abs(yo)))
    print(' predict bit mismatches', nb, 'of', yh.size, 'worst rel', worst, ' int ok', bool(np.all(yi == np.maximum(np.trunc(yh), 0).astype(np.int32))))

if __name__ == '__main__':
    selftests()
    N = int(os.environ.get('DBG_N', '16'))
    run_case('cfg2-like', 'linear', 'additive', N, 730)
    run_case('ref-like', 'logistic', 'multiplicative', N, 730)
    run_case('lin-mult', 'linear', 'multiplicative', N, 365, yearly=False)
    run_case('short', 'linear', 'additive', N, 90, yearly=False, max_iters=(5, None))
    run_case('log-add', 'logistic', 'additive', N, 730, max_iters=(5, None))
