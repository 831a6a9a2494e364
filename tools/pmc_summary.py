"""Summarise the rocprofv3 counter_collection csv files written by tools/gpu_round.sh: per
kernel name, the mean of each counter over its dispatches (summed over the dimension
instances rocprofv3 emits as separate rows)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in sorted(glob.glob(os.path.join(out, 'prof_*', '**', '*counter_collection.csv'), recursive=True)):
    per = defaultdict(lambda: defaultdict(float))      # (kernel, counter) -> dispatch -> value
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row['Kernel_Name'].split('(')[0][:60]
            per[(k, row['Counter_Name'])][row['Dispatch_Id']] += float(row['Counter_Value'])
    print('--', os.path.relpath(f, out))
    for (k, c), d in sorted(per.items()):
        vals = list(d.values())
        print('%-62s %-22s n=%-4d mean=%.6g' % (k, c, len(vals), sum(vals) / len(vals)))


# profiles/pmc_latest.json: what bench.py reports as roofline.traffic
import json
want = {}
for f in sorted(glob.glob(os.path.join(out, 'prof_*', '**', '*counter_collection.csv'), recursive=True)):
    per = defaultdict(lambda: defaultdict(float))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE', 'SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU',
                                       'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY'):
                k = row['Kernel_Name'].split('(')[0]
                for short in ('fit_quad_kernel', 'fit_kernel'):
                    if short in k and 'gram' not in k:
                        per[(short, row['Counter_Name'])][row['Dispatch_Id']] += float(row['Counter_Value'])
                        break
    for (k, c), d in per.items():
        # HBM counters are in KiB; the SQ counters are plain counts (summed over the shader engines)
        want.setdefault(k, {})[c + ('_KiB' if c in ('FETCH_SIZE', 'WRITE_SIZE') else '')] = sum(d.values()) / len(d)
# every kernel of the step (setup, Gram build, fit, future design, predict): HBM counters per launch -> roofline.step_traffic
step = {}
for f in sorted(glob.glob(os.path.join(out, 'prof_*', '**', '*counter_collection.csv'), recursive=True)):
    per = defaultdict(lambda: defaultdict(float))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                k = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('tsf::', '').split('<')[0].strip()
                # the library's kernels and its workspace memsets; not torch's (bench.py's device-copy probe, its fills)
                if 'tsf::' not in row['Kernel_Name'] and 'fillBufferAligned' not in k:
                    continue
                per[(k, row['Counter_Name'])][row['Dispatch_Id']] += float(row['Counter_Value'])
    for (k, c), d in per.items():
        step.setdefault(k, {})[c + '_KiB'] = sum(d.values()) / len(d)
        step[k]['launches'] = len(d)
if want:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    js = {'kernels': want, 'step_kernels': step, 'kernel_sources_sha16': bench.kernel_sources_digest(),
          'series_per_launch': int(os.environ.get('BENCH_N', '10000')),
          'points': int(os.environ.get('BENCH_T', '730')),
          'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around bench.py, '
                    'tools/gpu_round.sh %s' % os.path.basename(os.path.normpath(out))}
    with open(os.path.join(out, 'pmc_latest.json'), 'w') as fh:
        json.dump(js, fh, indent=1)
    print('wrote', os.path.join(out, 'pmc_latest.json'))
