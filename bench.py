#!/usr/bin/env python
"""bench.py -- series fitted/sec on the BASELINE.json headline workload.

Workload ("cfg2"): 10 000 synthetic daily series x 730 points per GPU,
linear trend with 25 changepoints + weekly (order 3) and yearly (order 10, forced on: 730 daily
points span 729 d < fbprophet's 730 d auto threshold) additive Fourier seasonality, MAP fit by
Stan-style L-BFGS (Stan's default tolerances), then a 90-step forecast.  A "step" is ONE PASS
of that fit + predict over the whole panel with the panel already resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Multi-GPU: series are sharded by id, one rank per GPU, no data-path collective;
torch.distributed (RCCL) only brackets the timed regions and gathers the evaluation counts.
`value` is the BASELINE metric read literally for every N: ONE 10 000-series panel, split over the
ranks for N > 1 (series i on rank i mod N: parallel.shard_indices) -- STRONG scaling.  A launch cannot
end before its longest fit, so this panel stops scaling early (`strong_scaling_expectation` states the
ceiling next to the measurement).  Beside it, timed the same way:
  * `weak_scaling` (N > 1): every rank its own 10 000-series panel;
  * `cfg3_sharded` (every N): BASELINE config 3 -- 100 000 series x 1 095 points as 8 blocks of 12 500,
    block b on rank b mod N -- the configuration that has enough work per GPU to scale;
  * `other_baseline_configs` (N = 1, rank 0): BASELINE configs 1 and 4 and the reference's own model settings on
    10 000 x 730 -- the residual-form kernels, one warm-up + one or two timed steps each (tools/bench_configs.py has
    every configuration at full length).
Rank 0 prints one JSON line.
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
        yo, _ = cl.predict(csp, r, fut)
        return r['n_eval'], yo

    t0 = time.perf_counter()
    one(0)
    per = max(time.perf_counter() - t0, 1e-4)
    sample = int(min(y.shape[0], max(cores * 2, budget_s * cores / per)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        done = list(ex.map(one, range(sample)))
    dt = time.perf_counter() - t0
    evals = [d[0] for d in done]
    worst, checked = None, 0
    if yhat_gpu is not None:        # checker use of the same oracle: GPU forecasts vs oracle
        checked = min(len(yhat_gpu), sample)
        worst = max(float(np.max(np.abs(yhat_gpu[n] - done[n][1]) / np.abs(done[n][1]))) for n in range(checked))
    return {'value': sample / dt, 'unit': 'series/s', 'cores': cores, 'kind': 'port',
            'parity_max_rel_err': worst, 'parity_series_checked': checked,
            'sample': '%d of %d series of the same panel (fit + %d-step forecast), '
                      'oracle/prophet_canon.c on %d threads, %.1f s wall'
                      % (sample, y.shape[0], len(fut), cores, dt),
            'mean_evals': float(np.mean(evals))}


def cpu_baseline_python(budget_series=4096, timeout_s=240):
    """A CPU baseline SHAPED like the reference's path (Spark local[*]: one Python worker per core, one
    series per UDF call; /root/reference/tests/unit/prophet_modeler_test.py:20): oracle/py_baseline.py --
    per series a pandas frame -> the literal ProphetOracle(...).fit -> make_future_dataframe -> predict, under
    multiprocessing.Pool(os.cpu_count()), on a bounded sample (>= 512 series) of the same panel.  Its own
    process (subprocess: nothing forks the process that holds the HIP context).  Labelled "restated python --
    not fbprophet+Stan": real fbprophet + Stan's autodiff + pystan's marshalling would be slower still."""
    import subprocess
    cores = os.cpu_count() or 1
    series = int(min(N_SERIES, max(512, min(budget_series, 16 * cores))))
    cmd = [sys.executable, '-m', 'oracle.py_baseline', '--series', str(series), '--procs', str(cores),
           '--points', str(T_POINTS), '--horizon', str(HORIZON), '--total', str(N_SERIES)]
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout_s)
    if out.returncode != 0:
        raise RuntimeError('oracle.py_baseline failed: %s' % out.stderr[-400:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def vs_true_map(n_cfg2=48, n_ref=16, timeout_s=420):
    """Distance of stopped fits to the TRUE MAP (round-4 review, item 3): tools/true_map_report.py on the first series
    of this panel and of the reference-model panel, in its own process (CPU only: the canonical oracle is the GPU's
    arithmetic bit for bit, and nothing may fork the process that holds the HIP context).  The full-size run
    (256 + 64 series) is committed under profiles/r05_true_map/report.json."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'true_map_report.py'), str(n_cfg2), str(n_ref)],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout_s)
    if out.returncode != 0:
        raise RuntimeError('tools/true_map_report.py failed: %s' % out.stderr[-400:])
    d = json.loads(out.stdout)
    d['full_size_run'] = 'profiles/r05_true_map/report.json (256 cfg2 + 64 reference-model series)'
    return d


def kernel_sources_digest():
    """sha256 (first 16 hex digits) over the HIP sources + the C-ABI header, in name order."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, 'time_series_spark_amd', 'csrc', '*')))
    files.append(os.path.join(ROOT, 'include', 'tsf.h'))
    files.append(os.path.join(ROOT, 'include', 'tsf_dev.h'))
    for f in files:
        if f.endswith(('.h', '.hip', '.inc', '.cpp')):
            h.update(os.path.basename(f).encode())
            with open(f, 'rb') as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def parity_context(f, spec, ds, y, fut, yhat_quad, n=256):
    """What "identical to the oracle" does and does not mean: Stan's L-BFGS at
    Stan's tolerances stops far from the optimum, so the forecast is a chaotic function of rounding.
    Two measurements on the first n series of the panel, both on the GPU: (a) the same series
    fitted in the residual evaluation form (Stan's own order of operations) instead of the
    quadratic form, (b) the quadratic form with ONE input value per series moved by one ulp.
    Reported as {median, p90} over series of the per-series median relative forecast difference:
    the level at which ANY implementation -- including another build of Stan -- reproduces the
    reference's output."""
    import torch
    from time_series_spark_amd import _lib
    n = min(n, y.shape[0])
    out = {}
    for key, sp, yy in (
            ('forecast_rel_err_quadratic_vs_residual',
             fc.ModelSpec.from_dict(dict(spec.to_dict(), lbfgs=dict(spec.lbfgs, eval_form=_lib.EVAL_RESIDUAL))), y[:n]),
            ('forecast_rel_err_one_ulp_perturbation', spec, None)):
        if yy is None:
            yy = y[:n].clone()
            mid = yy.shape[1] // 2
            yy[:, mid] = torch.nextafter(yy[:, mid], torch.full_like(yy[:, mid], float('inf')))
        g = DeviceForecaster(sp, f.device_index)
        o = g.alloc_fit_output(n)
        yh = torch.zeros((n, len(fut)), dtype=torch.float64, device=y.device)
        g.fit_aligned(ds, yy.contiguous(), o)
        g.predict(o, fut, yh, None)
        rel = (torch.abs(yh - yhat_quad[:n]) / torch.abs(yhat_quad[:n])).median(dim=1).values.cpu().numpy()
        out[key] = {'median': float(np.median(rel)), 'p90': float(np.quantile(rel, 0.9)), 'series': int(n)}
    return out


def map_mode_leg(f, spec, ds, y, fut, out_stan, yhat_stan, n_true=48):
    """tsf_spec.converge = MAP on the headline panel (round 6): the maximum a posteriori estimate of the model instead of
    the point Stan's tests stop at.  For this model (linear growth, additive seasonality, aligned panel) it is computed
    DIRECTLY (map_quad_kernel: sigma in closed form and the L1-regularised quadratic programme by an active-set method, in
    turn; a dozen Cholesky solves per series, no L-BFGS trajectory); `as_a_continuation` is the general route every other
    model takes (map_kernel: the Stan-rule fit carried on by an orthant-wise L-BFGS), run here on the same panel.  With
    their COST beside the Stan-rule step, and, on the first n_true series, the distance of the fits' 90-day forecasts to an
    independent solver's optimum (oracle/true_map.py through tools/true_map_solve.py, in processes of its own: CPU only)."""
    import subprocess
    import tempfile
    import torch
    from time_series_spark_amd import _lib
    sp = fc.ModelSpec.from_dict(dict(spec.to_dict(), lbfgs=dict(spec.lbfgs, converge=_lib.CONVERGE_MAP)))
    g = DeviceForecaster(sp, f.device_index)
    n = y.shape[0]
    o = g.alloc_fit_output(n)
    yh = torch.zeros((n, len(fut)), dtype=torch.float64, device=y.device)

    def step():
        g.fit_aligned(ds, y, o)
        g.predict(o, fut, yh, None)
    def timed(reps):
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    ne0 = out_stan.n_eval.cpu().numpy().astype(np.int64)
    cont = None
    try:
        g.ctx.set_option('map_direct', 0)
        dtc = timed(1)
        nec = o.n_eval.cpu().numpy().astype(np.int64)
        stc = o.status.cpu().numpy()
        cont = {'what': 'option map_direct = 0: the Stan-rule fit carried on to the estimate by map_kernel (the route of every model '
                        'that is not linear / additive on an aligned panel)', 'ms_per_step': 1e3 * dtc, 'series_per_s': n / dtc,
                'mean_evals': float(nec.mean()), 'max_evals': int(nec.max()),
                'status_counts': {str(int(k)): int(v) for k, v in zip(*np.unique(stc, return_counts=True))}}
        fval_c = o.fval.clone()
    except Exception as e:
        cont = {'error': str(e)}
    g.ctx.set_option('map_direct', -1)
    dt_ = timed(3)
    ne = o.n_eval.cpu().numpy().astype(np.int64)
    st = o.status.cpu().numpy()
    res = {'what': 'the same panel with tsf_spec.converge = MAP: the maximum a posteriori estimate computed directly '
                   '(map_quad_kernel, tsf_map_quad.h) + forecast; never `value`',
           'ms_per_step': 1e3 * dt_, 'series_per_s': n / dt_,
           'mean_evals_stan_rule': float(ne0.mean()), 'mean_rounds': float(o.n_iter.cpu().numpy().mean()),
           'mean_cholesky_solves': float(ne.mean()) - 1.0, 'max_cholesky_solves': int(ne.max()) - 1,
           'status_counts': {str(int(k)): int(v) for k, v in zip(*np.unique(st, return_counts=True))},
           'objective_gain_median': float(np.median((out_stan.fval - o.fval).cpu().numpy())),
           'as_a_continuation': cont}
    if cont and 'error' not in cont:
        cont['max_abs_objective_difference_to_direct'] = float(torch.max(torch.abs(fval_c - o.fval)).item())
    moved = (torch.abs(yh - yhat_stan) / torch.abs(yhat_stan)).median(dim=1).values.cpu().numpy()
    res['forecast_rel_change_vs_stan_rule'] = {'median': float(np.median(moved)), 'p90': float(np.quantile(moved, 0.9))}
    try:
        n_true = min(n_true, n)
        path = os.path.join(tempfile.gettempdir(), 'tsf_bench_true_map_%d.npz' % os.getpid())
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
        np.savez(path + '.panel.npz', ds=ds.cpu().numpy(), y=y[:n_true].cpu().numpy())
        subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'true_map_solve.py'), 'cfg2', str(n_true), path,
                        '--panel', path + '.panel.npz'], cwd=ROOT, env=env, check=True, capture_output=True, timeout=300)
        z = np.load(path)
        os.unlink(path)
        os.unlink(path + '.panel.npz')
        ys = o.y_scale.cpu().numpy()[:n_true]
        th_map = o.theta.cpu().numpy()[:n_true]
        rs = fc.fit_aligned(spec, ds.cpu().numpy(), y[:n_true].cpu().numpy())      # (host entry: its grid record for predict)
        fut_np = fut.cpu().numpy()
        y_true = fc.predict(spec, z['theta_map'], ys, rs.grid, fut_np)
        y_map = fc.predict(spec, th_map, ys, rs.grid, fut_np)
        y_st = fc.predict(spec, rs.theta, rs.y_scale, rs.grid, fut_np)
        for key, yy in (('map_mode', y_map), ('stan_rule', y_st)):
            rel = np.max(np.abs(yy - y_true) / np.abs(y_true), axis=1)
            res['forecast_max_rel_err_over_horizon_vs_true_map_' + key] = {
                'median': float(np.median(rel)), 'p90': float(np.quantile(rel, 0.9)), 'max': float(rel.max()), 'series': int(n_true)}
        res['true_map_solver'] = 'oracle/true_map.py (delta split, scipy L-BFGS-B with bounds), kkt max %.1e' % float(z['kkt'].max())
    except Exception as e:
        res['vs_true_map_error'] = str(e)
    return res


def quad_waves_per_cu(n_series, n_cu):
    """Wave slots per CU of the route an aligned linear/additive panel of n_series takes (tsf_quad_launch.h /
    tsf_inst_quad.hip): the register-M kernel (8) up to 3 series per its wave slot, the 16-wave pooled kernel
    from 8 series per its wave slot on, else the 12-wave kernel."""
    if n_series <= 3 * 8 * n_cu:
        return 8
    if n_series >= 8 * 16 * n_cu:
        return 16
    return 12


def simulate_strong_scaling(n_eval, fit_ms, n_cu, waves_per_cu=12):
    """What `bench.py --gpus G` (ONE panel split i mod G) should show, from THIS run's per-series
    evaluation counts: every GPU is a queue of n_cu x waves_per_cu wave slots that takes its series in
    index order, a series holds its slot for n_eval x tau, and tau is calibrated so that G = 1
    reproduces the measured fit-path kernel time.  A model (one evaluation time for busy and idle
    phases), labelled as such: a first real multi-GPU run has something to be compared with."""
    import heapq
    n_eval = np.asarray(n_eval, dtype=np.float64)
    slots = int(n_cu) * waves_per_cu

    def makespan(ev):
        if len(ev) <= slots:
            return float(ev.max())
        h = [0.0] * slots
        heapq.heapify(h)
        end = 0.0
        for e in ev:
            t = heapq.heappop(h) + e
            end = max(end, t)
            heapq.heappush(h, t)
        return end
    base = makespan(n_eval)
    tau_ms = fit_ms / base
    out = {'model': 'greedy queue over %d wave slots per GPU, tau = %.3f us per evaluation (calibrated on G = 1)' % (slots, 1e3 * tau_ms),
           'label': 'simulated', 'gpus': {}}
    for g in (1, 2, 4, 8):
        ms = max(makespan(n_eval[r::g]) for r in range(g)) * tau_ms
        out['gpus'][str(g)] = {'fit_kernel_ms': ms, 'series_per_s_kernel_only': len(n_eval) / (ms * 1e-3)}
    out['longest_series_ms'] = float(n_eval.max()) * tau_ms
    return out


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
        # counters are only as current as the kernels they were collected on: the profile records a
        # digest of the kernel sources (tools/pmc_summary.py) and a stale one is refused
        if d.get('kernel_sources_sha16') != kernel_sources_digest():
            return None, ('profiles/pmc_latest.json was collected on other kernel sources (%s, now %s): '
                          're-run tools/gpu_round.sh' % (d.get('kernel_sources_sha16'), kernel_sources_digest()))
        return (2.0 * k['FETCH_SIZE_KiB'] + k['WRITE_SIZE_KiB']) * 1024.0, d.get('source', path)
    except Exception as e:       # no profile committed for this kernel
        return None, 'unavailable: %s' % e


def pmc_step_traffic():
    """HBM bytes of ONE STEP -- every kernel of the fit + forecast (setup_grid, setup_series, gram_build, fit_quad,
    future_design, predict), each launched once per step -- from the same committed PMC passes: sum over kernels of
    2 x FETCH_SIZE + WRITE_SIZE (the factor 2 calibrated for every load shape of this library:
    profiles/r05_fetch_calib).  roofline.traffic is the DOMINANT kernel's share of this."""
    path = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
    try:
        with open(path) as fh:
            d = json.load(fh)
        if d.get('kernel_sources_sha16') != kernel_sources_digest() or d.get('series_per_launch') != N_SERIES:
            return None
        per = {k: (2.0 * v.get('FETCH_SIZE_KiB', 0.0) + v.get('WRITE_SIZE_KiB', 0.0)) * 1024.0 for k, v in d.get('step_kernels', {}).items()}
        return {'bytes': float(sum(per.values())), 'per_kernel_bytes': per} if per else None
    except Exception:
        return None


def pmc_valu(kernel, kernel_ms, total_evals, n_cu):
    """The bound that actually binds this kernel (not HBM): vector-instruction issue.  From the
    same committed PMC passes (SQ_INSTS_VALU per launch): wave-instructions per evaluation, and the
    fraction of the launch's issue slots they fill -- one fp64 wave-instruction occupies a SIMD for 4
    cycles (16 lanes per cycle), 4 SIMDs per CU, at the MI355X peak engine clock of 2.4 GHz
    (MI355X_MICROARCH.md), over the kernel time THIS run measured.  None when the profile is stale."""
    path = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
    try:
        with open(path) as fh:
            d = json.load(fh)
        if d.get('kernel_sources_sha16') != kernel_sources_digest() or d.get('series_per_launch') != N_SERIES:
            return None
        insts = d['kernels'][kernel]['SQ_INSTS_VALU']
        slots = n_cu * 4 * (kernel_ms * 1e-3 * 2.4e9) / 4.0
        return {'valu_wave_insts_per_launch': insts, 'valu_wave_insts_per_evaluation': insts / float(total_evals),
                'valu_issue_frac': insts / slots,
                'note': 'SQ_INSTS_VALU of the committed rocprofv3 pass / (SIMDs x kernel cycles / 4 cycles per fp64 wave-instruction at 2.4 GHz)'}
    except Exception:
        return None


def other_baseline_configs(dev, local):
    """BASELINE configs 1 and 4 and the reference's own model settings (logistic growth, multiplicative seasonality:
    prophet_modeler.py:56-65) on 10 000 x 730: the residual-form kernels, inputs resident in HBM, step = fit + 90-step
    forecast as in the headline leg; fit-path kernel time from the library's HIP events."""
    import torch
    yearly = {'name': 'yearly', 'period': 365.25, 'fourier_order': 10}
    weekly = {'name': 'weekly', 'period': 7, 'fourier_order': 3}
    out = {}

    def leg(name, desc, spec, ds_np, y_np, cap, extra, exf, steps):
        N = y_np.shape[0]
        fut_np = ds_np[-1] + synth.DAY_NS * np.arange(1, HORIZON + 1)
        to = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        ds, y, fut, fl, cp, ex, exfd = to(ds_np), to(y_np), to(fut_np), to(np.zeros(N)), to(cap), to(extra), to(exf)
        f = DeviceForecaster(spec, local)
        o = f.alloc_fit_output(N)
        yh = torch.zeros((N, HORIZON), dtype=torch.float64, device=dev)

        def step():
            f.fit_aligned(ds, y, o, floor=fl, cap=cp, extra=ex)
            f.predict(o, fut, yh, None, floor=fl, cap=cp, extra_future=exfd)

        step()
        torch.cuda.synchronize()
        f.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - t0) / steps
        kms = f.profile_read()
        f.set_profiling(False)
        ne = o.n_eval.cpu().numpy().astype(np.int64)
        st = o.status.cpu().numpy()
        fk = float(np.mean(kms)) if kms else None
        out[name] = {'workload': desc, 'series': N, 'series_per_s': N / dt_, 'ms_per_step': 1e3 * dt_,
                     'fit_kernel_ms': fk, 'mean_evals': float(ne.mean()),
                     'max_evals': int(ne.max()), 'fitted': int((st > 0).sum()),
                     # a launch cannot end before its longest fit: Stan's iteration limit (10 000, status 40) makes a few
                     # series in 10 000 run 30 000 evaluations, and they are the launch time of a 10 000-series panel --
                     # the evaluation rate is the figure that describes the kernel
                     'evaluations_per_s': None if not fk else float(ne.sum()) / (fk * 1e-3),
                     # how much of the launch is its ONE longest fit (the dice: which series reaches Stan's iteration
                     # limit moves with the last bit): longest fit's evaluations x the time per evaluation such a fit
                     # takes alone on the cooperative kernel (5.05 us, profiles/r05_coop) over the launch
                     'longest_fit_ms_over_launch_ms': None if not fk else min(1.0, float(ne.max()) * 5.05e-3 / fk),
                     'status_counts': {str(int(k)): int(v) for k, v in zip(*np.unique(st, return_counts=True))},
                     'finite_forecasts': bool(torch.isfinite(yh[o.status > 0]).all().item())}

    ref = fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[yearly, weekly])
    ds1, y1 = synth.make_panel(100, 365, 'logistic', seed=751)
    leg('cfg1', 'BASELINE config 1: 100 x 365, the reference settings (logistic, floor 0, cap 1.1 max y, multiplicative, '
        'auto seasonalities)',
        fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative',
                     seasonalities=fc.ModelSpec.auto_seasonalities(ds1, seasonality_mode='multiplicative')),
        ds1, y1, y1.max(axis=1) * 1.1, None, None, 3)
    dsr, yr = synth.make_panel(10000, T_POINTS, 'logistic', seed=751)
    leg('reference_settings_10k', '10 000 x 730, the reference settings (logistic, multiplicative, yearly + weekly)',
        ref, dsr, yr, yr.max(axis=1) * 1.1, None, None, 2)
    ds4 = synth.daily_grid(T_POINTS)
    fut4 = ds4[-1] + synth.DAY_NS * np.arange(1, HORIZON + 1)
    allm, names = synth.holiday_matrix(np.concatenate([ds4, fut4]), 10)
    ex4, exf4 = np.ascontiguousarray(allm[:, :T_POINTS]), np.ascontiguousarray(allm[:, T_POINTS:])
    _, y4 = synth.make_panel(50000, T_POINTS, 'logistic', seed=751, holidays=ex4)
    leg('cfg4', 'BASELINE config 4: 50 000 x 730, logistic + floor, multiplicative, 10 holidays x window [-1, +1] '
        '(30 indicator columns, P = 84: the sparse-column kernel, DESIGN.md 5g)',
        fc.ModelSpec(growth='logistic', seasonality_mode='multiplicative', seasonalities=[yearly, weekly],
                     extra=[{'name': n} for n in names]), ds4, y4, y4.max(axis=1) * 1.1, ex4, exf4, 1)
    try:
        out['irregular_reference_model'] = irregular_leg(ref)
        out['lattice_reference_model'] = irregular_leg(ref, lattice=True)
    except Exception as e:
        out['irregular_reference_model'] = {'error': str(e)}
    return out


def boundary_legs():
    """The path at the boundary the reference keeps (north_star: "DataFrame-in/DataFrame-out ... fitted end-to-end";
    prophet_modeler.py:41-79, :102-143; prophet_scorer.py:147-165) -- host IO, packing, H2D, fit, blobs, parquet, predict,
    CSV included, so NEVER `value`:
      files_to_files      the two jobs as their drivers run them (tools/e2e_bench.py): Hive-partitioned CSV ->
                          ProphetModeler.model -> model parquet -> ProphetScorer.score -> forecast CSV, cfg2 and the
                          reference's own model, 10 000 x 730; median of 3 passes after a warm-up pass in this process
      dataframe_boundary  model_panel(config)(pdf) -> model frame -> forecast_panel(config)(models) -> forecast frame on a
                          pandas frame of the reference's schema (int32 ids, datetime64 ds, int32 y), cfg2."""
    import pandas as pd
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import e2e_bench
    from time_series_spark_amd.jobs import prophet_modeler as pm, prophet_scorer as ps
    out = {}
    for kind in ('cfg2', 'reference'):
        try:
            r = e2e_bench.run(N_SERIES, T_POINTS, kind, passes=3)
            out['files_to_files_' + kind] = {
                'series_per_s': r['series_per_s_files_to_files'], 'series_per_s_best_pass': r['series_per_s_best_pass'],
                'total_s': r['total_s'], 'modeler_s': r['modeler_s'], 'scorer_s': r['scorer_s'], 'passes': r['passes'],
                'models': r['models'], 'forecast_rows': r['forecast_rows'], 'chunks': r['chunks'],
                'workload': '%d partition directories x %d rows of CSV -> models (parquet) -> %d-day forecasts (CSV), %s'
                            % (N_SERIES, T_POINTS, HORIZON, 'cfg2 model' if kind == 'cfg2' else
                               "the reference's model (logistic, multiplicative: prophet_modeler.py:65)")}
        except Exception as e:
            out['files_to_files_' + kind] = {'error': str(e)}
    try:
        ds, y = synth.make_panel(N_SERIES, T_POINTS, 'linear', seed=2)
        pdf = pd.DataFrame({'series_id': np.repeat(np.arange(N_SERIES, dtype=np.int32), T_POINTS),
                            'dim_id': np.ones(N_SERIES * T_POINTS, dtype=np.int32),
                            'ds': np.tile(ds.astype('datetime64[ns]'), N_SERIES), 'y': y.reshape(-1).astype(np.int32)})
        cfg = {'model': {'floor': 0, 'cap_multiplier': 1.1,
                         'prophet': {'growth': 'linear', 'seasonality_mode': 'additive', 'yearly_seasonality': True}},
               'forecast': {'periods': HORIZON, 'frequency': 'D'}}
        devnull, keep = open(os.devnull, 'w'), sys.stdout
        sys.stdout = devnull
        try:
            tm, tf = [], []
            for _ in range(4):
                t0 = time.perf_counter()
                models = pm.model_panel(cfg)(pdf)
                t1 = time.perf_counter()
                fdf = ps.forecast_panel(cfg)(models)
                t2 = time.perf_counter()
                tm.append(t1 - t0)
                tf.append(t2 - t1)
        finally:
            sys.stdout = keep
        m, f_ = float(np.median(tm[1:])), float(np.median(tf[1:]))
        out['dataframe_boundary'] = {'series_per_s': N_SERIES / (m + f_), 'model_panel_s': m, 'forecast_panel_s': f_,
                                     'models': int(len(models)), 'forecast_rows': int(len(fdf)),
                                     'workload': 'cfg2: a %d-row pandas frame [series_id, dim_id, ds, y] -> model frame -> '
                                                 'forecast frame [series_id, dim_id, ds, yhat]' % len(pdf)}
    except Exception as e:
        out['dataframe_boundary'] = {'error': str(e)}
    return out


def cfg5_legs(dev, local):
    """BASELINE config 5 (1 000 000 series x 90 points, fp32 y, weekly seasonality only: the retail-SKU shape) in the
    driver-run line (round-5 review): under L-BFGS at full size, and under the optimiser fbprophet itself takes below
    100 rows (Stan's Newton: `'Newton' if T < 100`), at full size too since round 6 took that launch from 15 to
    10.5 s (one timed step, no warm-up step of its own: the L-BFGS leg before it has warmed the context, and a 100 000-series
    sample is not representative -- one series of the first 100 000 runs 605 000 evaluations and ends a short launch alone)."""
    import torch
    from time_series_spark_amd import _lib
    out = {}
    T5 = 90
    ds5, y5 = synth.make_panel(1000000, T5, 'linear', seed=751, dtype=np.float32)
    seas = fc.ModelSpec.auto_seasonalities(ds5)
    fut_np = ds5[-1] + synth.DAY_NS * np.arange(1, HORIZON + 1)
    dsd, futd = torch.from_numpy(ds5).to(dev), torch.from_numpy(fut_np).to(dev)
    for name, n, spec, steps in (
            ('cfg5', 1000000, fc.ModelSpec(growth='linear', seasonalities=seas), 2),
            ('cfg5_newton', 1000000, fc.ModelSpec(growth='linear', seasonalities=seas, algorithm=_lib.ALGO_NEWTON), 1)):
        try:
            yd = torch.from_numpy(np.ascontiguousarray(y5[:n])).to(dev)
            f = DeviceForecaster(spec, local)
            o = f.alloc_fit_output(n)
            yh = torch.zeros((n, HORIZON), dtype=torch.float64, device=dev)

            def step():
                f.fit_aligned(dsd, yd, o)
                f.predict(o, futd, yh, None)
            if name == 'cfg5':
                step()
            torch.cuda.synchronize()
            f.set_profiling(True)
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            dt_ = (time.perf_counter() - t0) / steps
            kms = f.profile_read()
            f.set_profiling(False)
            ne = o.n_eval.cpu().numpy().astype(np.int64)
            st = o.status.cpu().numpy()
            P5 = 3 + spec.n_changepoints + spec.K
            alg = float(n) * (T5 * 4 + P5 * 4 + HORIZON * 4)            # 856 B per series
            fk = float(np.mean(kms)) if kms else None
            out[name] = {'workload': ('BASELINE config 5: %d x 90, fp32 y, linear + weekly(3), ' % n) +
                                     ('Stan L-BFGS' if name == 'cfg5' else
                                      "Stan's Newton = fbprophet's own choice below 100 rows; one timed step, no warm-up step"),
                         'series': n, 'series_per_s': n / dt_, 'ms_per_step': 1e3 * dt_, 'fit_kernel_ms': fk,
                         'mean_evals': float(ne.mean()), 'max_evals': int(ne.max()), 'fitted': int((st > 0).sum()),
                         'status_counts': {str(int(k)): int(v) for k, v in zip(*np.unique(st, return_counts=True))},
                         'algorithmic_bytes': alg, 'algorithmic_GBps': None if not fk else alg / (fk * 1e-3) / 1e9}
            del yd, o, yh, f
        except Exception as e:
            out[name] = {'error': str(e)}
    return out


def ranks_seen(dev, local):
    """Who took part (round-5 review: make the first real 8-GPU run self-checking): world size and backend as
    torch.distributed reports them, and every rank's device -- name, UUID, PCI bus id -- gathered on rank 0, so that the
    driver can see that RCCL saw N ranks on N DISTINCT GPUs."""
    import torch
    import torch.distributed as dist
    p = torch.cuda.get_device_properties(local)
    mine = {'rank': int(os.environ.get('RANK', '0')), 'local_rank': int(os.environ.get('LOCAL_RANK', '0')),
            'device_index': int(local), 'name': p.name, 'uuid': str(getattr(p, 'uuid', '')),
            'pci_bus_id': int(getattr(p, 'pci_bus_id', -1)), 'visible_devices': torch.cuda.device_count(),
            'pid': os.getpid(), 'host': os.uname().nodename}
    if dist.is_available() and dist.is_initialized():
        allr = [None] * dist.get_world_size()
        dist.all_gather_object(allr, mine)
        backend, world = dist.get_backend(), dist.get_world_size()
    else:
        allr, backend, world = [mine], None, 1
    ids = {(r['host'], r['uuid'] or r['pci_bus_id'], r['device_index'] if not r['uuid'] else 0) for r in allr}
    return {'world_size': world, 'backend': backend, 'ranks': allr, 'distinct_gpus': len(ids)}


def irregular_leg(spec, N=10000, lattice=False):
    """lattice = True (round 6): the fixture's shape taken literally -- every series at its OWN subset of the slots of one time
    lattice (the fixture: Thu-Sun at 11:15 and 21:45; here 600..730 of 730 days, tools/bench_irregular.py's third panel): a
    row is 22 bytes (t, y, segment word, lattice point), the base pairs are the lattice points' from one shared table.
    lattice = False: the reference's own call (prophet_modeler.py:65: logistic growth, multiplicative seasonality) on the reference's
    own DATA SHAPE (its fixture: every (series_id, dim_id) at its own irregular timestamps): N series of 600..730 rows,
    no two sharing a timestamp vector, none on a lattice (tools/bench_irregular.py's panel).  Host-pointer ragged entry
    point; fit-path kernel time from the library's events.  Algorithmic bytes as SURVEY 8d defines them (ds + y in,
    theta out, once per series); counter traffic from the committed PMC pass of tools/bench_irregular.py."""
    import ctypes
    from time_series_spark_amd import _lib
    T = 730
    rng = np.random.default_rng(12 if lattice else 11)
    lens = rng.integers(600, T + 1, N)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ds, y = synth.make_panel(N, T, 'logistic', seed=751)
    if lattice:
        keep = [np.sort(rng.choice(T, size=c, replace=False)) for c in lens]
        dsr = np.concatenate([ds[k] for k in keep])
    else:
        keep = [np.arange(c) for c in lens]
        dsr = np.concatenate([ds[:c] + rng.integers(-6 * 3600, 6 * 3600, c) * 1_000_000_000 for c in lens])
    yr = np.concatenate([y[i][k] for i, k in enumerate(keep)])
    cap = np.array([y[i][k].max() * 1.1 for i, k in enumerate(keep)])
    ctx = fc.get_context()
    L = _lib.load()
    ms = ctypes.c_float(0.0)
    kms = []
    for rep in range(2):
        ctx.check(L.tsf_set_profiling(ctx.handle, 1))
        r = fc.fit_ragged(spec, off, dsr, yr, floor=np.zeros(N), cap=cap)
        ctx.check(L.tsf_last_fit_kernel_ms(ctx.handle, ctypes.byref(ms)))
        kms.append(float(ms.value))
    ctx.check(L.tsf_set_profiling(ctx.handle, 0))
    k = min(kms) * 1e-3
    P = 3 + spec.n_changepoints + spec.K
    alg = float(np.sum(lens * 16 + P * 8))
    row_bytes = float(np.sum(np.ceil(lens / 64) * 64 * ((4 if lattice else 2 * 2 * 8) + 8 + 8 + 2) * r.n_eval))
    res = {'workload': ('%d series at their own 600..730 of the 730 slots of one daily lattice (the reference\'s fixture: own subsets of '
                        'fixed time slots), ' if lattice else
                        '%d series of 600..730 rows, each at its own irregular timestamps on NO lattice (the fixture\'s shape with a '
                        'jitter of seconds: the worst case), ') % N +
                       'logistic growth + multiplicative yearly(10) + weekly(3): prophet_modeler.py:65',
           'series': N, 'rows': int(lens.sum()), 'fit_kernel_ms': kms, 'series_per_s_kernel': N / k,
           'mean_evals': float(r.n_eval.mean()), 'max_evals': int(r.n_eval.max()),
           'evaluations_per_s': float(r.n_eval.sum()) / k, 'fitted': int((r.status > 0).sum()),
           'status_counts': {str(int(a)): int(b) for a, b in zip(*np.unique(r.status, return_counts=True))},
           'algorithmic_bytes': alg, 'algorithmic_GBps': alg / k / 1e9, 'roofline_frac_of_8TBps': alg / k / 1e9 / HBM_PEAK_GBPS,
           'bytes_read_by_the_evaluations': row_bytes, 'bytes_read_by_the_evaluations_over_algorithmic': row_bytes / alg,
           'kernel': ('fit_kernel<28, logistic, multiplicative, lattice, HARM yearly 10 + weekly 3> (rows of 22 bytes, base pairs of '
                      'the lattice points from one shared table, harmonics in registers) + cooperative tail') if lattice else
                     ('fit_kernel<28, logistic, multiplicative, HARM yearly 10 + weekly 3> (base pairs per row, harmonics in '
                      'registers) + cooperative tail')}
    try:
        with open(os.path.join(ROOT, 'profiles', 'irregular_pmc_latest.json')) as fh:
            d = json.load(fh)
        if d.get('kernel_sources_sha16') == kernel_sources_digest():
            res['traffic'] = 2.0 * d['lattice_fit_kernel_FETCH_SIZE_KiB' if lattice else 'fit_kernel_FETCH_SIZE_KiB'] * 1024.0
            res['traffic_over_algorithmic'] = res['traffic'] / alg
            res['traffic_source'] = d.get('source')
        else:
            res['traffic'] = None
            res['traffic_source'] = 'profiles/irregular_pmc_latest.json was collected on other kernel sources'
    except Exception as e:
        res['traffic'] = None
        res['traffic_source'] = 'unavailable: %s' % e
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-cfg3', action='store_true', help='skip the cfg3_sharded leg (100 000 x 1 095 over the ranks)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the other_baseline_configs leg (cfg1, cfg4, reference settings)')
    ap.add_argument('--no-boundary', action='store_true', help='skip the files_to_files / dataframe_boundary legs')
    ap.add_argument('--timed-only', action='store_true',
                    help='only warm-up + the K timed steps (for rocprofv3 runs: every dispatch of the fit '
                         'kernel is then a full-panel launch, so per-kernel means are per launch)')
    args = ap.parse_args()

    import torch
    rank, world, local = parallel.init_process_group()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback in the product path)')
    local = local % torch.cuda.device_count()      # one visible device per rank (ROCR_VISIBLE_DEVICES) or all
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    spec = cfg2_spec()
    # N = 1: the BASELINE.json metric as stated -- one 10 000 x 730 panel on one GPU.
    # N > 1: the SAME quantity ("series fitted/sec on a 10k x 730-pt panel at 1/2/4/8 MI355X"): ONE
    # 10 000-series panel split over the ranks, series i on rank i mod N (strong scaling; no
    # collective on the data path), is `value`; the weak-scaling run (every rank its own 10 000-series
    # panel) is timed the same way afterwards and reported beside it.
    ds_np, y_full = synth.make_panel(N_SERIES, T_POINTS, 'linear', seed=751)
    y_np = np.ascontiguousarray(y_full[parallel.shard_indices(N_SERIES, rank, world)])
    n_local = y_np.shape[0]
    fut_np = ds_np[-1] + synth.DAY_NS * np.arange(1, HORIZON + 1)
    ds = torch.from_numpy(ds_np).to(dev)
    y = torch.from_numpy(y_np).to(dev)
    fut = torch.from_numpy(fut_np).to(dev)
    f = DeviceForecaster(spec, local)
    n_cu = torch.cuda.get_device_properties(local).multi_processor_count

    def timed_leg(yy, f=f, ds=ds, fut=fut):
        """W warm-up + K timed steps (fit + 90-step forecast of the panel `yy`, resident in HBM),
        bracketed by barrier + synchronize on both sides, MAX over ranks."""
        n = yy.shape[0]
        o = f.alloc_fit_output(n)
        yh = torch.zeros((n, HORIZON), dtype=torch.float64, device=dev)
        yi = torch.zeros((n, HORIZON), dtype=torch.int32, device=dev)

        def step():
            f.fit_aligned(ds, yy, o)
            f.predict(o, fut, yh, yi)

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
        dt_ = time.perf_counter() - t0
        dt_ = parallel.max_over_ranks(dt_, dev if world > 1 else None)
        kms = f.profile_read()
        f.set_profiling(False)
        return dt_, kms, o, yh

    dt, kernel_ms, out, yhat = timed_leg(y)
    seen = ranks_seen(dev, local)        # (a collective: every rank)

    weak = None
    if world > 1:
        _, yw_np = synth.make_panel(N_SERIES, T_POINTS, 'linear', seed=751 + rank)
        dtw, kmsw, _, _ = timed_leg(torch.from_numpy(yw_np).to(dev))
        weak = {'value': world * N_SERIES * args.steps / dtw, 'unit': 'series/s', 'ms_per_step': 1e3 * dtw / args.steps,
                'series_per_gpu': N_SERIES, 'series_total': world * N_SERIES, 'scaling': 'weak',
                'fit_kernel_ms_rank0': float(np.mean(kmsw)) if kmsw else None,
                'note': 'every rank fits its own %d-series panel (per-GPU work fixed)' % N_SERIES}

    # BASELINE config 3 (100 000 x 1 095, "sharded across 8 MI355X"): the configuration with enough work per
    # GPU to scale -- 8 blocks of 12 500 series (block b = make_panel(12 500, 1 095, seed 3000 + b)), block b on
    # rank b mod N, no collective; same step (fit + 90-step forecast), same bracketing, MAX over ranks
    cfg3 = None
    if not args.timed_only and not args.no_cfg3 and 8 % world == 0:
        try:
            C3_BLOCKS, C3_N, C3_T = 8, int(os.environ.get('BENCH_C3_BLOCK', '12500')), 1095
            mine = [b for b in range(C3_BLOCKS) if b % world == rank]
            ds3_np = synth.daily_grid(C3_T)
            y3 = torch.empty((len(mine) * C3_N, C3_T), dtype=torch.float64, device=dev)
            for k, b in enumerate(mine):
                y3[k * C3_N:(k + 1) * C3_N] = torch.from_numpy(synth.make_panel(C3_N, C3_T, 'linear', seed=3000 + b)[1]).to(dev)
            spec3 = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds3_np))
            f3 = DeviceForecaster(spec3, local)
            ds3 = torch.from_numpy(ds3_np).to(dev)
            fut3 = torch.from_numpy(ds3_np[-1] + synth.DAY_NS * np.arange(1, HORIZON + 1)).to(dev)
            c3_ready = True
        except Exception as e:
            c3_ready, cfg3 = False, {'error': 'setup on rank %d: %s' % (rank, e)}
        # every rank says whether ITS setup worked before any collective of the leg (round-4 advice: a rank that
        # failed before all_gather left the others hanging in it); the leg runs on all ranks or on none
        if parallel.sum_over_ranks(0.0 if c3_ready else 1.0, dev if world > 1 else None) > 0:
            cfg3 = cfg3 if cfg3 is not None else {'error': 'setup failed on another rank'}
            c3_ready = False
        try:
            if not c3_ready:
                raise RuntimeError(cfg3['error'])
            dt3, kms3, o3, _ = timed_leg(y3, f3, ds3, fut3)
            ev3 = o3.n_eval.cpu().numpy().astype(np.float64)
            # per-rank summary rows gathered on every rank: [series, total evaluations, longest fit, fit-path kernel ms]
            rows3 = parallel.gather_rows(np.array([[len(ev3), ev3.sum(), ev3.max(), float(np.mean(kms3)) if kms3 else np.nan]]), None, dev)
            total3 = C3_BLOCKS * C3_N
            cfg3 = {'value': total3 * args.steps / dt3, 'unit': 'series/s', 'ms_per_step': 1e3 * dt3 / args.steps,
                    'scaling': 'strong', 'series_total': total3, 'series_per_gpu': len(mine) * C3_N, 'points': C3_T,
                    'K': spec3.K, 'P': 3 + spec3.n_changepoints + spec3.K,
                    'workload': 'BASELINE config 3: %d series x %d daily points (8 blocks of %d, block b on rank b mod %d), '
                                'linear trend + 25 changepoints, weekly(3)+yearly(10) additive (fbprophet auto rule: span 1 094 d), '
                                'MAP L-BFGS + %d-step forecast' % (total3, C3_T, C3_N, world, HORIZON),
                    'fit_kernel_ms_per_rank': [float(v) for v in rows3[:, 3]],
                    'evaluations_total': float(rows3[:, 1].sum()), 'evaluations_longest_fit': float(rows3[:, 2].max()),
                    'waves_per_cu': quad_waves_per_cu(len(mine) * C3_N, n_cu)}
            del y3, f3, o3
        except Exception as e:
            cfg3 = {'error': str(e)}

    # the evaluation counts of the whole cfg2 panel on every rank, in series order (rank r holds the series i mod world == r)
    n_eval_all = parallel.gather_rows(out.n_eval.cpu().numpy().astype(np.float64).reshape(-1, 1), N_SERIES, dev)[:, 0]
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
    achieved = bytes_per_series * n_local / (fit_ms * 1e-3) / 1e9
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
        'metric': 'series_fitted_per_sec', 'value': N_SERIES * args.steps / dt,
        'unit': 'series/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
        # `value` is the BASELINE metric read literally at every N: ONE 10 000-series panel, split over the ranks for
        # N > 1 -- total work fixed as N grows = STRONG scaling, also at N = 1 (rounds 1-3 reported per-GPU panels for
        # N > 1, i.e. weak scaling: those numbers are `weak_scaling.value` now)
        'scaling': 'strong',
        'value_scaling': 'strong: one %d-series panel whatever N (weak-scaling figure beside it in weak_scaling for N > 1)' % N_SERIES,
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'ranks_seen': seen,
        'config': {'workload': 'cfg2: ONE panel of %d series x %d daily points%s, linear trend + 25 '
                               'changepoints, weekly(3)+yearly(10) additive Fourier, MAP L-BFGS '
                               '(Stan default tolerances) + %d-step forecast'
                               % (N_SERIES, T_POINTS,
                                  '' if world == 1 else ' split over %d GPUs (series i on rank i mod %d, no collective)' % (world, world),
                                  HORIZON),
                   'series_total': N_SERIES, 'series_per_gpu': n_local, 'points': T_POINTS, 'horizon': HORIZON,
                   'K': spec.K, 'S': spec.n_changepoints, 'P': P,
                   'eval_form': 'quadratic (Gram) form, re-centred' if quad else 'residual form',
                   'parallelism': 'shard-by-id x%d' % world},
        'roofline': {'bound': 'hbm', 'kernel': kernel, 'achieved': achieved,
                     'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS,
                     'peak_measured_device_copy': copy_gbps,
                     'traffic': traffic, 'traffic_kernel': kernel + ' (the dominant kernel only; step_traffic = all kernels of the step)',
                     'traffic_source': traffic_src,
                     'fetch_size_factor': '2 x FETCH_SIZE: calibrated on 4 GiB streamed once with each load shape of the library '
                                          '(8 B, 16 B, 2 B per lane, strided design rows): counter / bytes = 0.5000 for all four '
                                          '(profiles/r05_fetch_calib/calibration.json)',
                     'algorithmic_bytes_per_launch': bytes_per_series * n_local,
                     'kernel_ms_avg': fit_ms, 'launches_timed': len(kernel_ms),
                     'note': 'HBM sees each series once in and once out; the fit itself is a '
                             'vector-issue- and latency-bound fp64 L-BFGS loop on registers/LDS '
                             '(SURVEY 8d, DESIGN.md section 5), so the HBM fraction is small by '
                             'construction; the weights of a residual pass stay in registers (round 3), '
                             'so the measured traffic is ~1.2 x the algorithmic bytes; '
                             'evaluation rate and fp64 figure alongside',
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
    if world == 1:
        v = pmc_valu(kernel, fit_ms, int(n_eval.sum()), n_cu)
        if v is not None:
            res['roofline'].update(v)
        stt = pmc_step_traffic()
        if stt is not None:
            res['roofline']['step_traffic'] = stt['bytes']
            res['roofline']['step_traffic_per_kernel'] = stt['per_kernel_bytes']
            res['roofline']['step_traffic_over_algorithmic'] = stt['bytes'] / float(bytes_per_series * n_local)
    if weak is not None:
        res['weak_scaling'] = weak
    if cfg3 is not None:
        res['cfg3_sharded'] = cfg3
    if world == 1 and not args.no_other_configs and not args.timed_only:
        res['other_baseline_configs'] = other_baseline_configs(dev, local)
        try:
            res['other_baseline_configs'].update(cfg5_legs(dev, local))
        except Exception as e:
            res['other_baseline_configs']['cfg5'] = {'error': str(e)}
    if world == 1 and not args.no_boundary and not args.timed_only:
        res['boundary'] = boundary_legs()
    # What the strong-scaled legs SHOULD show, stated next to what they do show: a launch cannot end before
    # its longest fit, and a rank's queue cannot drain faster than its wave slots allow.  tau = time per
    # evaluation of one wave, calibrated on THIS run (rank 0's fit-path kernel time / the queue model's
    # makespan for rank 0's own share); the model then gives every N.  `ceiling` = N -> infinity.
    try:
        wpc = quad_waves_per_cu(n_local, n_cu)
        sim = simulate_strong_scaling(n_eval_all[parallel.shard_indices(N_SERIES, 0, world)], fit_ms, n_cu, wpc)
        tau_ms = sim['longest_series_ms'] / float(n_eval.max())
        exp = {'label': 'simulated (queue model, one evaluation time), NOT measured', 'tau_us_per_evaluation': 1e3 * tau_ms,
               'cfg2': {'longest_fit_evaluations': float(n_eval_all.max()),
                        'ceiling_ms_fit_kernel': float(n_eval_all.max()) * tau_ms,
                        'ceiling_series_per_s_kernel_only': N_SERIES / (float(n_eval_all.max()) * tau_ms * 1e-3),
                        'gpus': {}}}
        for G in (1, 2, 4, 8):
            # per-G route: the kernel a share of N_SERIES / G takes has its own number of wave slots
            import heapq
            w_g = quad_waves_per_cu((N_SERIES + G - 1) // G, n_cu)
            worst = 0.0
            for r in range(G):
                ev = n_eval_all[parallel.shard_indices(N_SERIES, r, G)]
                slots = n_cu * w_g
                if len(ev) <= slots:
                    worst = max(worst, float(ev.max()))
                else:
                    h = [0.0] * slots
                    for e in ev:
                        heapq.heappush(h, heapq.heappop(h) + e)
                    worst = max(worst, max(h))
            exp['cfg2']['gpus'][str(G)] = {'fit_kernel_ms': worst * tau_ms, 'speedup_vs_1': None}
        base = exp['cfg2']['gpus']['1']['fit_kernel_ms']
        for G in exp['cfg2']['gpus']:
            exp['cfg2']['gpus'][G]['speedup_vs_1'] = base / exp['cfg2']['gpus'][G]['fit_kernel_ms']
        if cfg3 is not None and 'error' not in cfg3:
            # cfg3: enough series per GPU that the queue, not the longest fit, sets the time: work / slots, with the
            # longest fit as the floor
            ev_tot, ev_max = cfg3['evaluations_total'], cfg3['evaluations_longest_fit']
            tau3 = None
            k0 = cfg3['fit_kernel_ms_per_rank'][0]
            slots0 = n_cu * cfg3['waves_per_cu']
            if k0 == k0:
                tau3 = k0 / max(ev_tot / world / slots0, ev_max)
            exp['cfg3'] = {'tau_us_per_evaluation': None if tau3 is None else 1e3 * tau3,
                           'longest_fit_evaluations': ev_max, 'gpus': {}}
            if tau3 is not None:
                for G in (1, 2, 4, 8):
                    w_g = quad_waves_per_cu(cfg3['series_total'] // G, n_cu)
                    # a fourth wave per SIMD slows every wave by ~23 %: tau scales with the route
                    rel = {8: 0.82, 12: 1.0, 16: 1.23}[w_g] / {8: 0.82, 12: 1.0, 16: 1.23}[cfg3['waves_per_cu']]
                    ms = max(ev_tot / G / (n_cu * w_g), ev_max) * tau3 * rel
                    exp['cfg3']['gpus'][str(G)] = {'fit_kernel_ms': ms}
                b3 = exp['cfg3']['gpus']['1']['fit_kernel_ms']
                for G in exp['cfg3']['gpus']:
                    exp['cfg3']['gpus'][G]['speedup_vs_1'] = b3 / exp['cfg3']['gpus'][G]['fit_kernel_ms']
        res['strong_scaling_expectation'] = exp
    except Exception as e:
        res['strong_scaling_expectation'] = {'error': str(e)}
    if world == 1 and not args.timed_only:
        # The strong-scaled run, rank by rank, on THIS GPU: `--gpus G` gives rank r the series r, r + G, ... of
        # the one panel and there is no collective on the data path, so a G-GPU step lasts as long as its slowest
        # rank -- and each rank's share can be timed here, one after the other (same step: fit + forecast).
        # Not a multi-GPU measurement (no second device, no launch skew between processes), labelled so.
        try:
            rr = {}
            for G in (2, 4, 8):
                per_rank = []
                for r in range(G):
                    ys = torch.from_numpy(np.ascontiguousarray(y_full[parallel.shard_indices(N_SERIES, r, G)])).to(dev)
                    o_ = f.alloc_fit_output(ys.shape[0])
                    yh_ = torch.zeros((ys.shape[0], HORIZON), dtype=torch.float64, device=dev)
                    yi_ = torch.zeros((ys.shape[0], HORIZON), dtype=torch.int32, device=dev)
                    f.fit_aligned(ds, ys, o_); f.predict(o_, fut, yh_, yi_)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        f.fit_aligned(ds, ys, o_); f.predict(o_, fut, yh_, yi_)
                    torch.cuda.synchronize()
                    per_rank.append(1e3 * (time.perf_counter() - t0) / 3)
                rr[str(G)] = {'ms_per_step_slowest_rank': max(per_rank), 'ms_per_step_ranks': [round(v, 3) for v in per_rank],
                              'series_per_s': N_SERIES / (1e-3 * max(per_rank))}
            res['strong_scaling_rank_by_rank_on_one_gpu'] = {
                'label': 'each rank\'s share of the ONE %d-series panel timed on this GPU, one after the other; a G-GPU '
                         'step = its slowest rank (no collective on the data path); not a multi-GPU measurement' % N_SERIES,
                'gpus': rr}
        except Exception as e:
            res['strong_scaling_rank_by_rank_on_one_gpu'] = {'error': str(e)}
    if world == 1:
        # ONE expectation per leg for the 8-GPU run (round-4 review: two estimates stood side by side): cfg2 -- each
        # rank's share MEASURED on this GPU (above), the slowest rank decides; cfg3 -- the queue model (its shares do
        # not fit a rank-by-rank replay in the default run time).  strong_scaling_expectation keeps the model's curves.
        try:
            rr8 = res.get('strong_scaling_rank_by_rank_on_one_gpu', {}).get('gpus', {}).get('8')
            e8 = {}
            if rr8:
                e8['cfg2'] = {'speedup_vs_1': rr8['series_per_s'] / res['value'],
                              'source': 'each of the 8 shares of the one panel timed on this GPU; slowest share = the step'}
            c3 = res.get('strong_scaling_expectation', {}).get('cfg3', {}).get('gpus', {}).get('8')
            if c3 and c3.get('speedup_vs_1'):
                e8['cfg3'] = {'speedup_vs_1': c3['speedup_vs_1'], 'source': 'queue model calibrated on this run (strong_scaling_expectation.cfg3)'}
            res['expected_8gpu_speedup'] = e8
        except Exception as e:
            res['expected_8gpu_speedup'] = {'error': str(e)}
    # Third arrangement of the same split (round-5 review): ONE process, one host thread + context per visible GPU
    # (forecaster.fit_aligned(devices='all'): series i on device i mod G) -- host pointers, so PCIe is inside
    if world == 1 and not args.timed_only:
        try:
            from time_series_spark_amd import _lib as _l
            G = int(_l.load().tsf_device_count())
            if G > 1:
                fc.fit_aligned(spec, ds_np, y_np, devices='all')
                t0 = time.perf_counter()
                for _ in range(3):
                    rh = fc.fit_aligned(spec, ds_np, y_np, devices='all')
                td = (time.perf_counter() - t0) / 3
                res['in_process_device_split'] = {'devices': G, 'series_per_s': N_SERIES / td, 'ms_per_fit': 1e3 * td,
                                                  'identical_to_one_device': bool(np.array_equal(rh.theta, out.theta.cpu().numpy())),
                                                  'note': 'fit only, host pointers (H2D / D2H inside), one thread per device'}
            else:
                res['in_process_device_split'] = {'devices': G, 'note': 'one visible device: nothing to split over '
                                                                        '(tests/test_gpu_parity.py runs the split on three contexts of one GPU)'}
        except Exception as e:
            res['in_process_device_split'] = {'error': str(e)}
    # the same panel re-fitted with the evaluation counts of the previous fit as scheduling hints
    # (tsf_set_cost_hints: what a job that re-fits its panel regularly can do); never `value` -- the
    # headline has no such knowledge -- but it says how much of the launch is its tail
    if world == 1 and not args.timed_only:
        try:
            o_ = f.alloc_fit_output(N_SERIES)
            yh_ = torch.zeros((N_SERIES, HORIZON), dtype=torch.float64, device=dev)
            yi_ = torch.zeros((N_SERIES, HORIZON), dtype=torch.int32, device=dev)
            hints = n_eval.astype(np.int32)

            def hinted():
                f.set_cost_hints(hints)         # (host-side sort + 40 KB copy: inside the timed region)
                f.fit_aligned(ds, y, o_); f.predict(o_, fut, yh_, yi_)
            hinted()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                hinted()
            torch.cuda.synchronize()
            th_ = (time.perf_counter() - t0) / args.steps
            same = bool(torch.equal(o_.theta, out.theta)) and bool(torch.equal(yh_, yhat))
            res['with_cost_hints'] = {'value': N_SERIES / th_, 'unit': 'series/s', 'ms_per_step': 1e3 * th_,
                                      'bit_identical_to_the_unhinted_fit': same,
                                      'note': 'NOT the headline: the work queue ordered by the evaluation counts of the '
                                              'previous fit of the same panel (tsf_set_cost_hints), longest fits first'}
        except Exception as e:
            res['with_cost_hints'] = {'error': str(e)}
    # host-pointer entry point (what a DataFrame caller uses): the panel crosses PCIe, device
    # buffers are allocated per call; never part of `value`, reported beside it
    if world == 1 and not args.timed_only:
        # first call: the library's device-buffer pool is cold (hipMalloc of every buffer); a job that calls once
        # per partition runs in the steady state, which is what `value_end_to_end_host_pointer` reports
        calls = []
        for _ in range(4):
            t0 = time.perf_counter()
            rh = fc.fit_aligned(spec, ds_np, y_np)
            fc.predict(spec, rh.theta, rh.y_scale, rh.grid, fut_np)
            calls.append(time.perf_counter() - t0)
        th = float(np.median(calls[1:]))
        res['value_end_to_end_host_pointer'] = N_SERIES / th
        res['host_pointer_entry'] = {'ms_per_call': 1e3 * th, 'ms_first_call': 1e3 * calls[0],
                                     'ms_calls': [round(1e3 * c, 3) for c in calls],
                                     'note': 'tsf_fit_aligned + tsf_predict with host (pageable) buffers, median of the '
                                             'calls after the first (which may find the device-buffer pool cold: 21 ms): H2D of the %.0f MB panel over PCIe (1.2 ms at the '
                                             '50 GB/s measured for pageable memory on this box: tools/host_entry_probe.py), '
                                             'D2H of the results' % (y_np.nbytes / 1e6)}
        try:
            res['parity_context'] = parity_context(f, spec, ds, y, fut, yhat)
        except Exception as e:
            res['parity_context'] = {'error': str(e)}
        try:
            res['parity_context']['map_mode'] = map_mode_leg(f, spec, ds, y, fut, out, yhat)
        except Exception as e:
            res['parity_context']['map_mode'] = {'error': str(e)}
        if not args.no_cpu_baseline:
            try:
                res['parity_context']['vs_true_map'] = vs_true_map()
            except Exception as e:
                res['parity_context']['vs_true_map'] = {'error': str(e)}
    # cpu_baseline leg (rank 0, N=1 only): the CPU oracle timed on the host cores, and -- the
    # same leg, the oracle as checker -- the GPU forecasts of the sampled series compared with it
    if world == 1 and not args.no_cpu_baseline and not args.timed_only:
        try:
            res['cpu_baseline'] = cpu_baseline(spec, ds_np, y_np, fut_np, yhat_gpu=yhat[:512].cpu().numpy())
            res['forecast_max_rel_err_vs_oracle'] = res['cpu_baseline'].pop('parity_max_rel_err')
            res['forecast_series_checked_vs_oracle'] = res['cpu_baseline'].pop('parity_series_checked')
            res['parity_note'] = ('oracle = oracle/prophet_canon.c in the SAME evaluation form and arithmetic order '
                                  '(bit-identical by construction: a regression guard); it restates fbprophet 0.5 / '
                                  'Stan 2.19 from recall and is NOT pinned to real fbprophet output (none can be '
                                  'produced here).  parity_context = how far two correct runs differ.')
        except Exception as e:
            res['cpu_baseline'] = {'value': None, 'unit': 'series/s', 'cores': os.cpu_count(),
                                   'kind': 'port', 'sample': 'failed: %s' % e}
    if world == 1 and not args.no_cpu_baseline and not args.timed_only:
        try:
            res['cpu_baseline_python'] = cpu_baseline_python()
        except Exception as e:
            res['cpu_baseline_python'] = {'value': None, 'unit': 'series/s', 'cores': os.cpu_count(), 'kind': 'port',
                                          'sample': 'failed: %s' % e}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
