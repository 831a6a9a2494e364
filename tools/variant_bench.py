"""dev tool: times one configuration of tools/bench_configs.py on every library variant under
tools/variants/ (built by tools/build_variant.sh), each in its own process, and prints the fit
kernel time and a digest of (theta, n_eval, status) -- variants must agree to the bit.

  python tools/variant_bench.py cfg2 [tag ...]
"""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'time_series_spark_amd', 'libtsf_amd.so')

CHILD = r'''
import sys, json, hashlib, time
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + '/tools')
import torch
import bench_configs as bc
from time_series_spark_amd import synth
from time_series_spark_amd.device import DeviceForecaster
name = %(name)r
desc, spec, ds_np, y_np, floor, cap, extra, exf, bps = bc.build(name)
dev = torch.device('cuda', 0)
N, T = y_np.shape
to = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ds, y, fl, cp, ex = to(ds_np), to(y_np), to(floor), to(cap), to(extra)
f = DeviceForecaster(spec, 0)
out = f.alloc_fit_output(N)
f.fit_aligned(ds, y, out, floor=fl, cap=cp, extra=ex)
torch.cuda.synchronize()
f.set_profiling(True)
for _ in range(%(steps)d):
    f.fit_aligned(ds, y, out, floor=fl, cap=cp, extra=ex)
torch.cuda.synchronize()
kms = f.profile_read()
h = hashlib.sha256()
for a in (out.theta, out.n_eval, out.status, out.fval):
    h.update(a.cpu().numpy().tobytes())
ne = out.n_eval.cpu().numpy()
if %(dump)r:
    np.save(%(dump)r, ne)
print(json.dumps({'tag': %(tag)r, 'config': name, 'fit_ms_min': float(np.min(kms)), 'fit_ms_mean': float(np.mean(kms)),
                  'series_per_s': N / (float(np.min(kms)) * 1e-3), 'digest': h.hexdigest()[:16],
                  'mean_evals': float(ne.mean()), 'max_evals': int(ne.max())}))
'''


def main():
    name = sys.argv[1]
    tags = sys.argv[2:]
    libs = sorted(glob.glob(os.path.join(ROOT, 'tools', 'variants', 'libtsf_amd_*.so')))
    if tags:
        libs = [l for l in libs if os.path.basename(l)[len('libtsf_amd_'):-3] in tags]
    keep = LIB + '.keep'
    os.replace(LIB, keep)
    try:
        for i, lib in enumerate([keep] + libs):
            tag = 'current' if lib == keep else os.path.basename(lib)[len('libtsf_amd_'):-3]
            subprocess.check_call(['cp', lib, LIB])
            dump = os.path.join(ROOT, 'gpurun_out', 'n_eval_%s.npy' % name) if i == 0 else ''
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            code = CHILD % {'root': ROOT, 'name': name, 'steps': 4, 'tag': tag, 'dump': dump}
            r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
            print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else 'FAILED %s: %s' % (tag, r.stderr[-800:]), flush=True)
    finally:
        os.replace(keep, LIB)


if __name__ == '__main__':
    main()
