"""dev: the native reader (walk + load + parse of 10 000 partition directories) under a few settings of its environment
knobs, each in a fresh process (the knobs are read once): pool threads, transparent huge pages for the file arenas,
background unmapping; and the chunked walk (tsf_csv_root_load) against the whole-tree walk."""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

if len(sys.argv) > 1:
    from time_series_spark_amd.jobs import prophet_modeler as pm
    root = sys.argv[1]
    best, bestc = 1e9, 1e9
    for rep in range(6):
        t0 = time.perf_counter()
        cols = pm.read_model_input_dir(root)
        best = min(best, time.perf_counter() - t0)
        del cols
    for rep in range(4):
        r = pm._CsvRoot(root)
        t0 = time.perf_counter()
        k = 4
        for i in range(k):
            cols = r.read(r.n_children * i // k, r.n_children * (i + 1) // k - r.n_children * i // k)
            del cols
        bestc = min(bestc, time.perf_counter() - t0)
        r.close()
    print('%-60s whole %.1f ms   4 chunks one after the other %.1f ms' % (os.environ.get('PROBE_TAG'), best * 1e3, bestc * 1e3), flush=True)
    sys.exit(0)

from time_series_spark_amd import synth  # noqa: E402
import e2e_bench  # noqa: E402
ds, y = synth.make_panel(10000, 730, 'linear', seed=2)
work = tempfile.mkdtemp(prefix='tsf_rd_')
root = os.path.join(work, 'model-input')
e2e_bench.write_input(root, ds, y)
for env in ({}, {'TSF_HOST_CACHE_MB': '0'}, {'TSF_POOL_THREADS': '16'}, {'TSF_POOL_THREADS': '64'},
            {}):
    e = dict(os.environ, PROBE_TAG=str(env), **env)
    subprocess.call([sys.executable, __file__, root], env=e)
shutil.rmtree(work, ignore_errors=True)
