"""How far apart are Stan's two optimisers on short series?  fbprophet 0.5 picks
`algorithm='Newton' if T < 100 else 'LBFGS'` (UPSTREAM-RECALL forecaster.py);
this library runs L-BFGS for every T.  The script restates Stan 2.19's Newton
(optimization/newton.hpp: finite-difference Hessian from 4 gradient evaluations per parameter,
eigenvalues made negative, step halving from 1, stop when |d lp| < 1e-8) on top of the CPU
oracle's objective, in numpy, and compares optimum and 90-day forecasts with the oracle's L-BFGS
on cfg5-shaped series (T = 90, weekly seasonality, 25 changepoints).  Two spellings of the
Hessian scaling are tried because the recalled source multiplies by epsilon/2 where the
finite-difference formula divides by epsilon (which only lengthens the step-halving loop in
exact arithmetic) -- the point of the table is that on these ill-conditioned problems
(25 Laplace-penalised slope changes on 72 points) the stopping point moves the forecast by
per cent, whichever optimiser or spelling is used.  CPU only; test infrastructure.

    python tests/dev/newton_vs_lbfgs.py [n_series]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import canon_lib as cl  # noqa: E402
from tests import helpers  # noqa: E402
from time_series_spark_amd import forecaster as fc, synth  # noqa: E402


def newton(csp, ds, yv, recalled_scaling=True, max_iter=400):
    r = cl.fit(csp, ds, yv)                       # L-BFGS optimum (and S, K)
    info = r['info']
    P = 3 + info.S + info.K
    d = cl.design(csp, ds, yv)
    th = np.zeros(P)
    th[0], th[1] = d['k0'], d['m0']
    n_eval = [0]

    def lp_grad(x):
        n_eval[0] += 1
        f, g, rc = cl.eval_at(csp, ds, yv, x)
        if rc or not np.isfinite(f) or not np.isfinite(g).all():
            return -1e100, -g
        return -f, -g

    eps = 1e-3
    pert = [-2 * eps, -eps, eps, 2 * eps]
    coef = [1 / 12, -2 / 3, 2 / 3, -1 / 12]
    scale = 0.5 * eps if recalled_scaling else 0.5 / eps
    lp, _ = lp_grad(th)
    iters = halvings = 0
    for m in range(max_iter):
        last = lp
        f0, g = lp_grad(th)
        A = np.zeros((P, P))
        for dd in range(P):
            for i in range(4):
                x = th.copy()
                x[dd] += pert[i]
                A[dd] += scale * coef[i] * lp_grad(x)[1]
        w, V = np.linalg.eigh(A + A.T)
        step = V @ (-(V.T @ g) / np.abs(w))
        ss, f1, new = 2.0, -1e100, th
        while f1 < f0:
            ss *= 0.5
            halvings += 1
            if ss < 1e-50:
                break
            new = th - ss * step
            f1, _ = lp_grad(new)
        if ss >= 1e-50:
            th, lp = new, f1
        else:
            lp = f0
        iters += 1
        if m > 0 and abs(lp - last) < 1e-8:
            break
    return th, lp, iters, n_eval[0], halvings, r


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    ds, y = synth.make_panel(n, 90, 'linear', seed=751)
    spec = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds))
    csp = helpers.oracle_spec(spec)
    csp.eval_mode = 0
    fut = ds[-1] + synth.DAY_NS * np.arange(1, 91)
    print('series | L-BFGS lp, iters, evals | Newton lp, iters, evals, halvings | forecast rel diff '
          'median (max) Newton vs L-BFGS | the two Newton spellings')
    for i in range(n):
        res = {}
        for flag in (True, False):
            th, lp, it, ne, nh, r = newton(csp, ds, y[i], flag)
            r2 = dict(r)
            r2['theta'] = th
            res[flag] = (lp, it, ne, nh, cl.predict(csp, r2, fut)[0])
        base = cl.predict(csp, r, fut)[0]
        lp, it, ne, nh, yh = res[True]
        rel = np.abs(yh - base) / np.abs(base)
        rel2 = np.abs(res[False][4] - yh) / np.abs(yh)
        print('%4d | %.4f %4d %5d | %.4f %4d %6d %5d | %.1e (%.1e) | %.1e (%.1e)'
              % (i, -r['f'], r['n_iter'], r['n_eval'], lp, it, ne, nh, np.median(rel), rel.max(),
                 np.median(rel2), rel2.max()))


if __name__ == '__main__':
    main()
