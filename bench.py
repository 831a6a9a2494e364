#!/usr/bin/env python
"""bench.py -- series fitted/sec on the BASELINE.json headline workload.

Workload ("cfg2"): 10 000 synthetic daily series x 730 points per GPU,
linear trend with 25 changepoints + weekly (order 3) and yearly (order 10, forced on: 730 daily
points span 729 d < fbprophet's 730 d auto threshold) additive Fourier seasonality, MAP fit by
Stan-style L-BFGS (Stan's default tolerances), then a 90-step forecast.  A "step" is ONE PASS
of that fit + predict over the whole panel with the panel already resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Multi-GPU: series are sharded by id, one rank per GPU, every rank fits its own 10 000-series
panel (weak scaling), no data-path collective; torch.distributed (RCCL) only brackets the
timed region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from time_series_spark_amd import forecaster as fc, parallel, synth  # noqa: E402
from time_series_spark_amd.device import DeviceForecaster  # noqa: E402

N_SERIES = int(os.environ.get('BENCH_N', '10000'))     # per GPU
T_POINTS = int(os.environ.get('BENCH_T', '730'))
HORIZON = int(os.environ.get('BENCH_H', '90'))
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
FP64_PEAK_TFLOPS = 78.6         # MI355X fp64 vector peak (spec)


def cfg2_spec():
    return fc.ModelSpec(growth='linear', seasonality_mode='additive', n_changepoints=25,
                        seasonalities=[{'name': 'yearly', 'period': 365.25, 'fourier_order': 10},
                                       {'name': 'weekly', 'period': 7, 'fourier_order': 3}])


def oracle_spec(spec):
    from oracle import canon_lib as cl
    seas = [(s['period'], s['fourier_order'], s.get('mode', spec.seasonality_mode),
             s.get('prior_scale', spec.seasonality_prior_scale)) for s in spec.seasonalities]
    # cfg2 is linear growth + additive columns on an aligned panel: the product evaluates the
    # data term in quadratic (Gram) form there unless eval_form forces the residual form; the
    # checker follows the same choice (oracle eval_mode)
    return cl.make_spec(growth=spec.growth, n_changepoints=spec.n_changepoints,
                        changepoint_range=spec.changepoint_range,
                        changepoint_prior_scale=spec.changepoint_prior_scale, seasonalities=seas,
                        eval_mode=int(spec.lbfgs.get('eval_form', 0) != 1))


def cpu_baseline(spec, ds, y, fut, budget_s=12.0, yhat_gpu=None):
    """The CPU oracle (oracle/prophet_canon.c: same model, same Stan L-BFGS, plain C) timed on
    this box's host cores on a bounded sample of the same panel: one series per task on a
    thread pool of os.cpu_count() threads (ctypes releases the GIL) -- the shape of the
    reference's Spark local[*] (one Python worker per core).  Reported, not the target."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import canon_lib as cl
    cl.lib()
    csp = oracle_spec(spec)
    cores = os.cpu_count() or 1

    def one(n):
        r = cl.fit(csp, ds, y[n])
        cl.predict(csp, r, fut)
        return r['n_eval']

    t0 = time.perf_counter()
    one(0)
    per = max(time.perf_counter() - t0, 1e-4)
    sample = int(min(y.shape[0], max(cores * 2, budget_s * cores / per)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        evals = list(ex.map(one, range(sample)))
    dt = time.perf_counter() - t0
    worst = None
    if yhat_gpu is not None:        # checker use of the same oracle: GPU forecasts vs oracle
        worst = 0.0
        for n in range(len(yhat_gpu)):
            yo, _ = cl.predict(csp, cl.fit(csp, ds, y[n]), fut)
            worst = max(worst, float(np.max(np.abs(yhat_gpu[n] - yo) / np.abs(yo))))
    return {'value': sample / dt, 'unit': 'series/s', 'cores': cores, 'kind': 'port',
            'parity_max_rel_err': worst,
            'sample': '%d of %d series of the same panel (fit + %d-step forecast), '
                      'oracle/prophet_canon.c on %d threads, %.1f s wall'
                      % (sample, y.shape[0], len(fut), cores, dt),
            'mean_evals': float(np.mean(evals))}


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_latest.json, written by tools/pmc_summary.py from separate FETCH_SIZE and
    WRITE_SIZE runs of this same command): 2 x FETCH_SIZE (the gfx950 correction of
    MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests as 64 B) + WRITE_SIZE, both in KiB.
    bench.py cannot collect counters itself (they need rocprofv3 around the process)."""
    path = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
    try:
        with open(path) as fh:
            d = json.load(fh)
        k = d['kernels'][kernel]
        if d.get('series_per_launch') != N_SERIES or d.get('points') != T_POINTS:
            return None, 'profiles/pmc_latest.json is for another workload size'
        return (2.0 * k['FETCH_SIZE_KiB'] + k['WRITE_SIZE_KiB']) * 1024.0, d.get('source', path)
    except Exception as e:       # no profile committed for this kernel
        return None, 'unavailable: %s' % e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    rank, world, local = parallel.init_process_group()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback in the product path)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    spec = cfg2_spec()
    ds_np, y_np = synth.make_panel(N_SERIES, T_POINTS, 'linear', seed=751 + rank)
    fut_np = ds_np[-1] + synth.DAY_NS * np.arange(1, HORIZON + 1)
    ds = torch.from_numpy(ds_np).to(dev)
    y = torch.from_numpy(y_np).to(dev)
    fut = torch.from_numpy(fut_np).to(dev)
    f = DeviceForecaster(spec, local)
    out = f.alloc_fit_output(N_SERIES)
    yhat = torch.zeros((N_SERIES, HORIZON), dtype=torch.float64, device=dev)
    yint = torch.zeros((N_SERIES, HORIZON), dtype=torch.int32, device=dev)

    def step():
        f.fit_aligned(ds, y, out)
        f.predict(out, fut, yhat, yint)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    f.set_profiling(True)
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    parallel.barrier()
    dt = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt, dev if world > 1 else None)
    kernel_ms = f.profile_read()
    f.set_profiling(False)

    if rank != 0:
        return
    # measured HBM ceiling on this device beside the 8 TB/s spec figure: device-to-device copy of
    # 1 GiB (reads + writes counted), outside the timed region (torch only moves memory here)
    try:
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
    except Exception:
        copy_gbps = None
    n_eval = out.n_eval.cpu().numpy().astype(np.int64)
    n_iter = out.n_iter.cpu().numpy()
    status = out.status.cpu().numpy()
    P = 3 + spec.n_changepoints + spec.K
    bytes_per_series = T_POINTS * 8 + P * 8 + HORIZON * 8
    fit_ms = float(np.mean(kernel_ms)) if kernel_ms else float('nan')
    achieved = bytes_per_series * N_SERIES / (fit_ms * 1e-3) / 1e9
    quad = spec.lbfgs.get('eval_form', 0) != 1      # cfg2 is linear + additive + aligned
    kernel = 'fit_quad_kernel' if quad else 'fit_kernel'
    traffic, traffic_src = pmc_traffic(kernel)
    # residual form: every evaluation is a pass over the T x K design values (BASELINE.md
    # section 4); quadratic form: a P x P mat-vec (2 P^2 flops) plus O(P) -- the residual-form
    # passes it still needs (initial point + re-centring, ~4 % of the evaluations) are not
    # counted separately by the kernel, so only the lower bound is quoted for it
    if quad:
        flops_per_eval = 2 * P * P + 16 * P
    else:
        flops_per_eval = 4 * T_POINTS * spec.K + 20 * T_POINTS + 6 * spec.n_changepoints
    tflops = float(n_eval.sum()) * flops_per_eval / (fit_ms * 1e-3) / 1e12
    res = {
        'metric': 'series_fitted_per_sec', 'value': world * N_SERIES * args.steps / dt,
        'unit': 'series/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'cfg2: %d series x %d daily points per GPU, linear trend + 25 '
                               'changepoints, weekly(3)+yearly(10) additive Fourier, MAP L-BFGS '
                               '(Stan default tolerances) + %d-step forecast'
                               % (N_SERIES, T_POINTS, HORIZON),
                   'series_per_gpu': N_SERIES, 'points': T_POINTS, 'horizon': HORIZON,
                   'K': spec.K, 'S': spec.n_changepoints, 'P': P,
                   'eval_form': 'quadratic (Gram) form, re-centred' if quad else 'residual form',
                   'parallelism': 'shard-by-id x%d' % world},
        'roofline': {'bound': 'hbm', 'kernel': kernel, 'achieved': achieved,
                     'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS,
                     'peak_measured_device_copy': copy_gbps,
                     'traffic': traffic, 'traffic_source': traffic_src,
                     'algorithmic_bytes_per_launch': bytes_per_series * N_SERIES,
                     'kernel_ms_avg': fit_ms, 'launches_timed': len(kernel_ms),
                     'note': 'HBM sees each series once in and once out; the fit itself is a '
                             'dependent-latency-bound fp64 L-BFGS loop on registers/LDS '
                             '(SURVEY 8d, DESIGN.md section 5), so the HBM fraction is small by '
                             'construction; evaluation rate and fp64 figure alongside',
                     'evaluations_per_s': float(n_eval.sum()) / (fit_ms * 1e-3),
                     'flops_per_evaluation_algorithmic': flops_per_eval,
                     'fp64_tflops_algorithmic': tflops,
                     'fp64_frac_of_vector_peak': tflops / FP64_PEAK_TFLOPS},
        'optimizer': {'mean_iters': float(n_iter.mean()), 'mean_evals': float(n_eval.mean()),
                      'total_evals': int(n_eval.sum()),
                      'evals_p50_p90_p99_max': [float(v) for v in np.percentile(n_eval, [50, 90, 99, 100])],
                      'status_counts': {str(int(k)): int(v) for k, v in
                                        zip(*np.unique(status, return_counts=True))}},
    }
    # host-pointer entry point (what a DataFrame caller uses): the panel crosses PCIe, device
    # buffers are allocated per call; never part of `value`
    if world == 1:
        t0 = time.perf_counter()
        rh = fc.fit_aligned(spec, ds_np, y_np)
        fc.predict(spec, rh.theta, rh.y_scale, rh.grid, fut_np)
        res['host_pointer_entry'] = {'ms_per_call': 1e3 * (time.perf_counter() - t0),
                                     'note': 'tsf_fit_aligned + tsf_predict with host buffers: H2D of the '
                                             '%.0f MB panel, per-call hipMalloc, D2H of the results'
                                             % (y_np.nbytes / 1e6)}
    # cpu_baseline leg (rank 0, N=1 only): the CPU oracle timed on the host cores, and -- the
    # same leg, the oracle as checker -- the GPU forecasts of the sampled series compared with it
    if world == 1 and not args.no_cpu_baseline:
        try:
            res['cpu_baseline'] = cpu_baseline(spec, ds_np, y_np, fut_np, yhat_gpu=yhat[:4].cpu().numpy())
            res['forecast_max_rel_err_vs_oracle'] = res['cpu_baseline'].pop('parity_max_rel_err')
        except Exception as e:
            res['cpu_baseline'] = {'value': None, 'unit': 'series/s', 'cores': os.cpu_count(),
                                   'kind': 'port', 'sample': 'failed: %s' % e}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
