"""profiles/irregular_pmc_latest.json from a rocprofv3 --pmc FETCH_SIZE pass around tools/bench_irregular.py
(tools/gpu_round5.sh): FETCH_SIZE (KiB, median over the launches) of the residual-form fit kernel, with the digest of the
kernel sources it was collected on -- what bench.py's `irregular_reference_model.traffic` reports.
    python tools/irregular_pmc_summary.py <dir with *counter_collection.csv> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if r['Counter_Name'] == 'FETCH_SIZE':
            per[k][r['Dispatch_Id']] += float(r['Counter_Value'])
out = {'kernel_sources_sha16': bench.kernel_sources_digest(),
       'source': 'rocprofv3 --pmc FETCH_SIZE --kernel-trace around tools/bench_irregular.py (median over the launches; KiB; '
                 'bytes = 2 x the counter: profiles/r05_fetch_calib)', 'kernels': {}}
for k, d in per.items():
    v = sorted(d.values())
    out['kernels'][k] = {'FETCH_SIZE_KiB_median': v[len(v) // 2], 'launches': len(v)}
    # fit_kernel<KP, GROWTH, MODE, PPL, XIDX, GNTR, SPARSE, HARM, PF>: XIDX = true is the lattice panel's kernel (round 6)
    targs = [t.strip() for t in k.split('<', 1)[1].rsplit('>', 1)[0].split(',')] if 'fit_kernel<' in k else []
    if targs and targs[4] == 'false' and 'fit_kernel_FETCH_SIZE_KiB' not in out:
        out['fit_kernel_FETCH_SIZE_KiB'] = v[len(v) // 2]
        out['fit_kernel'] = k
    if targs and targs[4] == 'true' and 'lattice_fit_kernel_FETCH_SIZE_KiB' not in out:
        out['lattice_fit_kernel_FETCH_SIZE_KiB'] = v[len(v) // 2]
        out['lattice_fit_kernel'] = k
json.dump(out, open(dst, 'w'), indent=1)
print(json.dumps({k: out[k] for k in out if k != 'kernels'}))
