"""Copy the judged summaries of one tools/gpu_round.sh run from gpurun_out/<tag>/ (scratch) into
profiles/<tag>/ (tracked) and refresh profiles/pmc_latest.json (read by bench.py).
usage: python tools/save_profiles.py r01_quad2"""
import csv
import collections
import os
import shutil
import sys

tag = sys.argv[1]
src = os.path.join('gpurun_out', tag)
dst = os.path.join('profiles', tag)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, 'bench.json'), os.path.join(dst, 'bench.json'))
shutil.copy(os.path.join(src, 'prof_stats', 'stats_kernel_stats.csv'),
            os.path.join(dst, 'rocprofv3_kernel_stats.csv'))
with open(os.path.join(src, 'summary.txt')) as fh:
    keep = [l for l in fh if not l.startswith(('__amd', 'void at::'))]
with open(os.path.join(dst, 'summary.txt'), 'w') as fh:
    fh.writelines(keep)
for k in ('fetch', 'write', 'sq'):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    path = os.path.join(src, 'prof_' + k, k + '_counter_collection.csv')
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        per[(r['Kernel_Name'].split('(')[0], r['Counter_Name'])][r['Dispatch_Id']] += float(r['Counter_Value'])
    with open(os.path.join(dst, 'pmc_%s.csv' % k), 'w') as f:
        f.write('kernel,counter,dispatches,mean_per_dispatch\n')
        for (kn, c), d in sorted(per.items()):
            if kn.startswith(('tsf::', 'void tsf::')):
                f.write('"%s",%s,%d,%.6g\n' % (kn, c, len(d), sum(d.values()) / len(d)))
if os.path.exists(os.path.join(src, 'pmc_latest.json')):
    shutil.copy(os.path.join(src, 'pmc_latest.json'), os.path.join('profiles', 'pmc_latest.json'))
    shutil.copy(os.path.join(src, 'pmc_latest.json'), os.path.join(dst, 'pmc_latest.json'))
print('saved', dst)
# round 6 (tools/gpu_round6.sh): the other outputs of the closing run, as they are
for name in ('configs.jsonl', 'ragged.txt', 'irregular.jsonl', 'irregular_pmc_latest.json', 'e2e_cfg2.txt', 'e2e_cfg2_40k.txt',
             'e2e_reference.txt', 'e2e_reference_40k.txt', 'map_probe.txt', 'map_direct_timing.txt', 'lattice_probe.txt',
             'newton_1m.txt', 'pytest_gpu.log', 'bench_time.txt', 'pmc_summary.txt'):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, name))
if os.path.exists(os.path.join(src, 'irregular_pmc_latest.json')):
    shutil.copy(os.path.join(src, 'irregular_pmc_latest.json'), os.path.join('profiles', 'irregular_pmc_latest.json'))
