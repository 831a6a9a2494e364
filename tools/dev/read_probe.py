"""dev: where read_input_columns spends its time on 10 000 partition files (discovery, native read + parse, fetch)."""
import os, sys, time, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from time_series_spark_amd import synth
from time_series_spark_amd.jobs import prophet_modeler as pm
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import e2e_bench
ds, y = synth.make_panel(10000, 730, 'linear', seed=2)
work = tempfile.mkdtemp(prefix='tsf_rp_')
root = os.path.join(work, 'model-input')
e2e_bench.write_input(root, ds, y)
for rep in range(3):
    t0 = time.perf_counter(); files, part = pm.find_model_input(root); t1 = time.perf_counter()
    cols = pm.read_model_input(files, root, part_sid=part); t2 = time.perf_counter()
    print('find %.3f s  read+parse+fetch %.3f s  rows %d  cpu %d' % (t1 - t0, t2 - t1, len(cols[3]), os.cpu_count()), flush=True)
for rep in range(3):
    t1 = time.perf_counter(); cols = pm.read_model_input_dir(root); t2 = time.perf_counter()
    print('native walk + read in place: %.3f s rows %d' % (t2 - t1, len(cols[3])), flush=True)
import gc
del cols; gc.collect()
for rep in range(3):
    t1 = time.perf_counter(); cols = pm.read_model_input_dir(root); t2 = time.perf_counter(); del cols; gc.collect(); t3 = time.perf_counter()
    print('native (table freed before the next call): read %.3f s, free %.3f s' % (t2 - t1, t3 - t2), flush=True)
for nt in ():
    t1 = time.perf_counter(); cols = pm.read_model_input(files, root, part_sid=part, n_threads=nt); t2 = time.perf_counter()
    print('threads %d: %.3f s' % (nt, t2 - t1), flush=True)
shutil.rmtree(work, ignore_errors=True)
