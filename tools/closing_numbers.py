"""The numbers DESIGN.md section 8 quotes, read from one closing run's files (profiles/<tag>/ or gpurun_out/<tag>/):
    python tools/closing_numbers.py profiles/r06_final"""
import csv
import json
import os
import sys

d = sys.argv[1]
b = json.load(open(os.path.join(d, 'bench.json')))
r = b['roofline']
print('cfg2: %.2f ms per step, %.3f M series/s; kernel %.2f ms (HIP events); %.2f GB/s algorithmic = %.1e of peak; kernel traffic %.1f MB = %.2f x; step %.1f MB = %.2f x'
      % (b['ms_per_step'], b['value'] / 1e6, r['kernel_ms_avg'], r['achieved'], r['frac'], r['traffic'] / 1e6, r['traffic'] / r['algorithmic_bytes_per_launch'],
         r['step_traffic'] / 1e6, r['step_traffic_over_algorithmic']))
for name in ('rocprofv3_kernel_stats.csv', os.path.join('prof_stats', 'stats_kernel_stats.csv')):
    p = os.path.join(d, name)
    if os.path.exists(p):
        for row in csv.DictReader(open(p)):
            if 'fit_quad_kernel' in row['Name']:
                print('rocprofv3: fit_quad_kernel %s calls, average %.3f ms (min %.3f max %.3f)' % (row['Calls'], float(row['AverageNs']) / 1e6, float(row['MinNs']) / 1e6, float(row['MaxNs']) / 1e6))
        break
h = b['with_cost_hints']
print('hints: %.2f ms, %.3f M series/s; host pointers %.0f k series/s' % (h['ms_per_step'], h['value'] / 1e6, b['value_end_to_end_host_pointer'] / 1e3))
m = b['parity_context']['map_mode']
print('MAP: %.2f ms per step, %.2f M series/s, rounds %.1f, solves %.1f (max %d), status %s; continuation %.1f ms; vs true map median %.1e max %.1e (stan-rule median %.1e)'
      % (m['ms_per_step'], m['series_per_s'] / 1e6, m['mean_rounds'], m['mean_cholesky_solves'], m['max_cholesky_solves'], m['status_counts'], m['as_a_continuation']['ms_per_step'],
         m['forecast_max_rel_err_over_horizon_vs_true_map_map_mode']['median'], m['forecast_max_rel_err_over_horizon_vs_true_map_map_mode']['max'],
         m['forecast_max_rel_err_over_horizon_vs_true_map_stan_rule']['median']))
bd = b['boundary']
print('bench boundary: f2f cfg2 %.1f k (best %.1f k), reference %.1f k (best %.1f k); DataFrame %.4f s = %.0f k series/s'
      % (bd['files_to_files_cfg2']['series_per_s'] / 1e3, bd['files_to_files_cfg2']['series_per_s_best_pass'] / 1e3, bd['files_to_files_reference']['series_per_s'] / 1e3,
         bd['files_to_files_reference']['series_per_s_best_pass'] / 1e3, bd['dataframe_boundary']['model_panel_s'] + bd['dataframe_boundary']['forecast_panel_s'],
         bd['dataframe_boundary']['series_per_s'] / 1e3))
for f in ('e2e_cfg2.txt', 'e2e_cfg2_40k.txt', 'e2e_reference.txt', 'e2e_reference_40k.txt'):
    p = os.path.join(d, f)
    if os.path.exists(p):
        e = json.loads(open(p).read().strip().splitlines()[-1])
        print('%s: %.1f k series/s (best %.1f k), modeler %.4f s scorer %.4f s' % (f, e['series_per_s_files_to_files'] / 1e3, e['series_per_s_best_pass'] / 1e3, e['modeler_s'], e['scorer_s']))
o = b['other_baseline_configs']
for k in ('cfg1', 'reference_settings_10k', 'cfg4', 'cfg5', 'cfg5_newton'):
    v = o[k]
    print('%s: %.2f ms per step, %.1f k series/s, max evals %s, longest/launch %s' % (k, v['ms_per_step'], v['series_per_s'] / 1e3, v.get('max_evals'), v.get('longest_fit_ms_over_launch_ms')))
for k in ('irregular_reference_model', 'lattice_reference_model'):
    v = o[k]
    print('%s: %.1f ms, %.1f k series/s, traffic %.1f GB = %.0f x' % (k, min(v['fit_kernel_ms']), v['series_per_s_kernel'] / 1e3, (v.get('traffic') or 0) / 1e9, v.get('traffic_over_algorithmic') or 0))
print('cfg3_sharded: %.1f ms, %.3f M series/s' % (b['cfg3_sharded']['ms_per_step'], b['cfg3_sharded']['value'] / 1e6))
p = os.path.join(d, 'configs.jsonl')
if os.path.exists(p):
    for l in open(p):
        try:
            c = json.loads(l)
        except Exception:
            continue
        print('configs %-10s %.2f ms %.1f k series/s max evals %s' % (c.get('config'), c.get('fit_kernel_ms', -1), c.get('series_per_s', 0) / 1e3, c.get('max_evals')))
print('cpu: C oracle %.2f k (%d threads); python %.0f' % (b['cpu_baseline']['value'] / 1e3, b['cpu_baseline']['cores'], b['cpu_baseline_python']['value']))
p = os.path.join(d, 'pytest_gpu.log')
if os.path.exists(p):
    print('tests:', open(p).read().strip().splitlines()[-1])
for f in ('newton_1m.txt', 'map_direct_timing.txt', 'lattice_probe.txt', 'bench_time.txt'):
    p = os.path.join(d, f)
    if os.path.exists(p):
        print('--', f)
        print(open(p).read()[:1600])
