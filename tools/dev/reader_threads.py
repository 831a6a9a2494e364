"""dev: the native model-input reader (discover + load + parse of 10 000 partition directories) against its thread count."""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from time_series_spark_amd import synth  # noqa: E402
from time_series_spark_amd.jobs import prophet_modeler as pm  # noqa: E402
import e2e_bench  # noqa: E402

ds, y = synth.make_panel(10000, 730, 'linear', seed=2)
work = tempfile.mkdtemp(prefix='tsf_rd_')
root = os.path.join(work, 'model-input')
e2e_bench.write_input(root, ds, y)
for nt in (0, 16, 32, 64, 128, 256, 0):
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        cols = pm.read_model_input_dir(root, n_threads=nt)
        best = min(best, time.perf_counter() - t0)
        del cols
    print('n_threads', nt, 'best of 4: %.1f ms' % (1e3 * best), flush=True)
shutil.rmtree(work, ignore_errors=True)
