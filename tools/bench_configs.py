"""Times the hot path on the other BASELINE.json configurations (bench.py is the headline
cfg2 line the driver reads; this tool gives the table in DESIGN.md section 7).  Same rules:
inputs resident in HBM, step = fit + 90-step predict, fit-path kernel time from the library's
HIP events.  One JSON line per configuration.

  python tools/bench_configs.py cfg1 cfg3 cfg4 cfg5 cfg2_resid
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from time_series_spark_amd import forecaster as fc, synth  # noqa: E402
from time_series_spark_amd.device import DeviceForecaster  # noqa: E402

YEARLY = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
WEEKLY = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
H = 90


def build(name):
    """-> (description, spec, ds, y, floor, cap, extra, extra_future, bytes_per_series)"""
    if name.endswith(('_mfma', '_wave', '_coop')) and not name.startswith(('cap', 'long')):
        # the same configuration with the residual kernel forced (residual_kernel = MFMA / WAVE / COOP;
        # the default AUTO is the one-wave kernel with the cooperative tail)
        from time_series_spark_amd import _lib
        out = list(build(name[:-5]))
        d = out[1].to_dict()
        rk = {'_mfma': _lib.RK_MFMA, '_wave': _lib.RK_WAVE, '_coop': _lib.RK_COOP}[name[-5:]]
        d['lbfgs'] = dict(d['lbfgs'], residual_kernel=rk)
        out[1] = fc.ModelSpec.from_dict(d)
        out[0] += {'_mfma': ' [matrix-core residual kernel]', '_wave': ' [one-wave residual kernel, no cooperative tail]',
                   '_coop': ' [cooperative kernel from the first evaluation]'}[name[-5:]]
        return tuple(out)
    if name in ('cfg2', 'cfg2_resid', 'cfg2x4', 'cfg2x16'):
        # cfg2x4 / cfg2x16: the cfg2 model on 40 000 / 160 000 series (throughput, not the longest series)
        N, T = {'cfg2x4': 40000, 'cfg2x16': 160000}.get(name, 10000), 730
        lb = {'eval_form': 1} if name.endswith('resid') else {}
        spec = fc.ModelSpec(growth='linear', seasonalities=[YEARLY, WEEKLY], **lb)
        ds, y = synth.make_panel(N, T, 'linear', seed=751)
        return ('%d x 730 linear additive yearly+weekly' % N + (' (residual form forced)' if lb else ''),
                spec, ds, y, None, None, None, None, T * 8 + 54 * 8 + H * 8)
    if name in ('ref10k', 'ref100k'):   # the reference's own model settings on an aligned panel
        N, T = (10000 if name == 'ref10k' else 100000), 730
        ds, y = synth.make_panel(N, T, 'logistic', seed=751)
        spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[YEARLY, WEEKLY])
        return ('%d x 730 logistic + multiplicative yearly+weekly (reference settings, aligned)' % N,
                spec, ds, y, np.zeros(N), y.max(axis=1) * 1.1, None, None, T * 8 + 54 * 8 + H * 8 + 8)
    if name.startswith('cap'):
        # throughput without stragglers: the reference's settings, max_iter capped.  cap<N>_<wave|mfma>
        parts = name.split('_')
        N, T = int(parts[0][3:]), 730
        from time_series_spark_amd import _lib
        rk = {'wave': _lib.RK_WAVE, 'mfma': _lib.RK_MFMA}[parts[1]]
        ds, y = synth.make_panel(N, T, 'logistic', seed=751)
        spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[YEARLY, WEEKLY],
                            residual_kernel=rk, max_iter=150)
        return ('%d x 730 reference settings, max_iter 150, residual kernel %s' % (N, parts[1]),
                spec, ds, y, np.zeros(N), y.max(axis=1) * 1.1, None, None, T * 8 + 54 * 8 + H * 8 + 8)
    if name.startswith('long'):
        # long histories (the reference's example config is 15-minute data): which residual kernel?
        # long<T>_<wave|mfma>[_<N>], e.g. long8760_mfma_2048
        parts = name.split('_')
        T = int(parts[0][4:])
        N = int(parts[2]) if len(parts) > 2 else 2048
        from time_series_spark_amd import _lib
        rk = {'wave': _lib.RK_WAVE, 'mfma': _lib.RK_MFMA}[parts[1]]
        ds, y = synth.make_panel(N, T, 'logistic', seed=751)
        spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[YEARLY, WEEKLY],
                            residual_kernel=rk)
        return ('%d x %d logistic + multiplicative yearly+weekly, residual kernel %s' % (N, T, parts[1]),
                spec, ds, y, np.zeros(N), y.max(axis=1) * 1.1, None, None, T * 8 + 54 * 8 + H * 8 + 8)
    if name in ('lin_hol', 'lin_hol_resid'):   # linear + additive with 30 holiday columns: P = 84, two-slot kernels
        N, T = 10000, 730
        ds = synth.daily_grid(T)
        fut = ds[-1] + synth.DAY_NS * np.arange(1, H + 1)
        allm, names = synth.holiday_matrix(np.concatenate([ds, fut]), 10)
        extra, exf = np.ascontiguousarray(allm[:, :T]), np.ascontiguousarray(allm[:, T:])
        ds, y = synth.make_panel(N, T, 'linear', seed=751, holidays=extra)
        lb = {'eval_form': 1} if name.endswith('resid') else {}
        spec = fc.ModelSpec(growth='linear', seasonalities=[YEARLY, WEEKLY], extra=[{'name': n} for n in names], **lb)
        return ('10000 x 730 linear additive + 10 holidays x [-1,+1]' + (' (residual form forced)' if lb else ''),
                spec, ds, y, None, None, extra, exf, T * 8 + (3 + 25 + spec.K) * 8 + H * 8)
    if name == 'cfg1':
        N, T = 100, 365
        ds, y = synth.make_panel(N, T, 'logistic', seed=751)
        seas = fc.ModelSpec.auto_seasonalities(ds, seasonality_mode='multiplicative')
        spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=seas)
        return ('100 x 365, reference settings (logistic, floor 0, cap 1.1 max y, multiplicative, auto seasonalities)',
                spec, ds, y, np.zeros(N), y.max(axis=1) * 1.1, None, None, T * 8 + (3 + 25 + spec.K) * 8 + H * 8)
    if name in ('cfg3', 'cfg3_full'):
        # cfg3: one GPU's share of 100 000 series over 8 GPUs; cfg3_full: BASELINE's whole panel on one GPU (876 MB of y)
        N, T = (12500 if name == 'cfg3' else 100000), 1095
        ds, y = synth.make_panel(N, T, 'linear', seed=751)
        seas = fc.ModelSpec.auto_seasonalities(ds)
        spec = fc.ModelSpec(growth='linear', seasonalities=seas)
        return ('%d x 1095 (%s cfg3) linear additive, yearly on by auto' % (N, '1/8 of' if name == 'cfg3' else 'all of'),
                spec, ds, y, None, None, None, None, T * 8 + (3 + 25 + spec.K) * 8 + H * 8)
    if name == 'cfg4':
        N, T = 50000, 730
        ds = synth.daily_grid(T)
        fut = ds[-1] + synth.DAY_NS * np.arange(1, H + 1)
        allm, names = synth.holiday_matrix(np.concatenate([ds, fut]), 10)
        extra, exf = np.ascontiguousarray(allm[:, :T]), np.ascontiguousarray(allm[:, T:])
        ds, y = synth.make_panel(N, T, 'logistic', seed=751, holidays=extra)
        spec = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative',
                            seasonalities=[YEARLY, WEEKLY], extra=[{'name': n} for n in names])
        return ('50000 x 730 logistic + floor, multiplicative, 25 changepoints, 10 holidays x window [-1,+1]',
                spec, ds, y, np.zeros(N), y.max(axis=1) * 1.1, extra, exf,
                T * 8 + (3 + 25 + spec.K) * 8 + H * 8 + 8)
    if name in ('cfg5', 'cfg5_newton', 'cfg5_newton_1m'):
        # cfg5_newton: the optimiser fbprophet itself picks for T = 90 (100x the work per series:
        # 100 000 series instead of 1 000 000 so that the run stays short)
        N, T = (100000 if name == 'cfg5_newton' else 1000000), 90      # cfg5_newton_1m: BASELINE's own size
        ds, y = synth.make_panel(N, T, 'linear', seed=751, dtype=np.float32)
        seas = fc.ModelSpec.auto_seasonalities(ds)
        if name != 'cfg5':
            from time_series_spark_amd import _lib
            spec = fc.ModelSpec(growth='linear', seasonalities=seas, algorithm=_lib.ALGO_NEWTON)
            return ('%d x 90 fp32 y, linear additive, weekly only, Stan Newton (fbprophet\'s choice for T<100)' % N,
                    spec, ds, y, None, None, None, None, T * 4 + (3 + 25 + spec.K) * 4 + H * 4)
        spec = fc.ModelSpec(growth='linear', seasonalities=seas)
        return ('1000000 x 90 fp32 y, linear additive, weekly only (L-BFGS; fbprophet would use Newton for T<100)',
                spec, ds, y, None, None, None, None, T * 4 + (3 + 25 + spec.K) * 4 + H * 4)
    raise SystemExit('unknown config ' + name)


def run(name, steps=2, warmup=1):
    import torch
    desc, spec, ds_np, y_np, floor, cap, extra, exf, bps = build(name)
    dev = torch.device('cuda', 0)
    N, T = y_np.shape
    fut_np = ds_np[-1] + synth.DAY_NS * np.arange(1, H + 1)
    to = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    ds, y, fut, fl, cp, ex, exfd = to(ds_np), to(y_np), to(fut_np), to(floor), to(cap), to(extra), to(exf)
    f = DeviceForecaster(spec, 0)
    out = f.alloc_fit_output(N)
    yhat = torch.zeros((N, H), dtype=torch.float64, device=dev)

    def step():
        f.fit_aligned(ds, y, out, floor=fl, cap=cp, extra=ex)
        f.predict(out, fut, yhat, None, floor=fl, cap=cp, extra_future=exfd)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    f.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kms = f.profile_read()
    n_eval = out.n_eval.cpu().numpy().astype(np.int64)
    status = out.status.cpu().numpy()
    fit_ms = float(np.mean(kms))
    res = {'config': name, 'workload': desc, 'series': N, 'points': T, 'K': spec.K,
           'P': 3 + spec.n_changepoints + spec.K, 'series_per_s': N / dt, 'ms_per_step': 1e3 * dt,
           'fit_kernel_ms': fit_ms, 'mean_evals': float(n_eval.mean()), 'max_evals': int(n_eval.max()),
           'evals_per_s': float(n_eval.sum()) / (fit_ms * 1e-3),
           'algorithmic_bytes_per_series': bps, 'hbm_GBps_algorithmic': bps * N / (fit_ms * 1e-3) / 1e9,
           'status_counts': {str(int(k)): int(v) for k, v in zip(*np.unique(status, return_counts=True))},
           'finite_forecasts': bool(torch.isfinite(yhat[status > 0]).all().item())}
    print(json.dumps(res), flush=True)


if __name__ == '__main__':
    for nm in (sys.argv[1:] or ['cfg1', 'cfg3', 'cfg4', 'cfg5', 'cfg2_resid']):
        run(nm)
