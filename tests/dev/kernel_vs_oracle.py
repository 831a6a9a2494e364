"""Dev tool (uses the oracle as checker, hence under tests/): any fit kernel vs the oracle,
increasing max_iter; `python tests/dev/kernel_vs_oracle.py <case>`.  Run under a short
`timeout` on the GPU box."""
import os, sys, time, faulthandler
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.dump_traceback_later(int(os.environ.get('DBG_DUMP_AFTER', '40')), exit=True)
from tests import helpers
from time_series_spark_amd import forecaster as fc
from oracle import canon_lib as cl

def bits(a): return np.ascontiguousarray(a, dtype=np.float64).view(np.int64)

case = sys.argv[1] if len(sys.argv) > 1 else 'cfg2_linear_additive'
spec0, ds, y, floor, cap, extra, fut, exf = helpers.make_case(case)
for mi in (1, 2, 5, 20, 100, None):
    lb = dict(spec0.lbfgs)
    if mi is not None:
        lb['max_iter'] = mi
    spec = fc.ModelSpec.from_dict(dict(spec0.to_dict(), lbfgs=lb))
    csp = helpers.oracle_spec(spec)
    t0 = time.time()
    print('max_iter', mi, 'launching', flush=True)
    r = fc.fit_aligned(spec, ds, y, floor=floor, cap=cap, extra=extra)
    print('  gpu done %.2fs' % (time.time() - t0), 'status', r.status, 'iters', r.n_iter, 'evals', r.n_eval, flush=True)
    for n in range(y.shape[0]):
        o = cl.fit(csp, ds, y[n], floor[n], cap[n], extra)
        S = o['info'].S
        th = np.concatenate([o['theta'][:3 + S], np.zeros(spec.n_changepoints - S), o['theta'][3 + S:]])
        nd = int(np.sum(bits(r.theta[n]) != bits(th)))
        print('   n=%d oracle status %d iters %d evals %d resid %d | theta bit mismatches %d  maxabs %.3e  f %s' % (
            n, o['status'], o['n_iter'], o['n_eval'], o.get('n_resid', 0), nd, np.max(np.abs(r.theta[n] - th)),
            'same' if bits(r.fval[n]) == bits(o['f']) else '%r vs %r' % (r.fval[n], o['f'])), flush=True)
print('done')
