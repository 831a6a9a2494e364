"""The TRUE MAP of the restated Prophet posterior -- TEST INFRASTRUCTURE ONLY (oracle/ rules apply).

Stan's L-BFGS, at Stan's tolerances, stops on the kinks of the Laplace prior |delta_j| / tau far from the optimum
(DESIGN.md section 3), and so does any plain quasi-Newton method started on the same function (scipy's L-BFGS-B on the
literal log-posterior stalls 1e-2 .. 0.5 above it: tests/test_oracle.py::test_map_estimate_against_an_independent_optimiser).
The kinks go away with the classical split

    delta = dp - dm,   dp >= 0, dm >= 0,   |delta| -> dp + dm     (at the optimum dp dm = 0, so the two agree)

which turns the prior into a LINEAR term and the problem into a smooth one with simple bounds -- what L-BFGS-B is made
for.  `solve` runs it on the literal model's data dict (oracle/fbprophet_restated.ProphetOracle.stan_data) with the plain
C restatement of prophet.stan (oracle/stan_lbfgs.c: oracle_fg) as the function, to a projected-gradient norm of ~1e-8 of
the objective's scale, and reports the KKT residual it reached.  Nothing here shares an operation order with the kernels.
PARITY UNPINNED w.r.t. real fbprophet / pystan like everything under oracle/.
"""
import numpy as np

from . import oracle_lib


def _smooth_fg(dat, packed, theta):
    """-log posterior WITHOUT the Laplace term, and its gradient (C restatement minus sum |delta| / tau)."""
    S = int(dat['S'])
    f, g, rc = oracle_lib.neg_log_prob_grad_packed(packed, theta)
    d = theta[3:3 + S]
    tau = float(dat['tau'])
    f -= np.abs(d).sum() / tau
    g = g.copy()
    g[3:3 + S] -= np.sign(d) / tau
    return f, g, rc


def solve(dat, theta_start, maxiter=200000, m=40):
    """-> (theta_map, info): minimiser of the literal -log posterior from `theta_start` (fbprophet's initial values or
    a stopped fit: same basin), info = {f, f_start, kkt, n_fun, success}.  kkt: infinity norm of the projected gradient
    of the split problem (0 at a KKT point)."""
    from scipy.optimize import minimize
    S, K = int(dat['S']), int(dat['K'])
    tau = float(dat['tau'])
    packed = oracle_lib.pack(dat)
    th0 = np.asarray(theta_start, dtype=np.float64)
    d0 = th0[3:3 + S]
    z0 = np.concatenate([th0[:3], np.maximum(d0, 0.0), np.maximum(-d0, 0.0), th0[3 + S:]])
    nz = z0.size
    lo = np.full(nz, -np.inf)
    lo[3:3 + 2 * S] = 0.0
    bounds = list(zip(lo, [np.inf] * nz))

    def unsplit(z):
        return np.concatenate([z[:3], z[3:3 + S] - z[3 + S:3 + 2 * S], z[3 + 2 * S:]])

    def fg(z):
        th = unsplit(z)
        f, g, rc = _smooth_fg(dat, packed, th)
        if rc != 0 or not np.isfinite(f):
            return 1e300, np.zeros(nz)
        f = f + (z[3:3 + S].sum() + z[3 + S:3 + 2 * S].sum()) / tau
        gz = np.concatenate([g[:3], g[3:3 + S] + 1.0 / tau, -g[3:3 + S] + 1.0 / tau, g[3 + S:]])
        return f, gz

    def proj_grad(z):
        f, gz = fg(z)
        return f, np.where((z <= lo) & (gz > 0), 0.0, gz)     # active lower bounds with g > 0 are optimal

    # L-BFGS-B gives up now and then in its line search (the valley is flat: condition numbers of 1e8 and more);
    # restarted from where it stopped, with an empty memory, it carries on.  Stop at a projected gradient of 1e-7 (the
    # objective is O(1e3), its gradient at fbprophet's start O(1e3)) or when two restarts in a row gain < 1e-13.
    z, n_fun, res = z0, 0, None
    f_prev = np.inf
    stall = 0
    for attempt in range(60):
        res = minimize(fg, z, jac=True, method='L-BFGS-B', bounds=bounds,
                       options=dict(maxiter=maxiter, maxfun=4 * maxiter, ftol=1e-17, gtol=1e-11, maxcor=m, maxls=60))
        z = res.x
        n_fun += int(res.nfev)
        f, pg = proj_grad(z)
        if np.max(np.abs(pg)) <= 1e-7:
            break
        stall = stall + 1 if f_prev - f < 1e-13 else 0
        if stall >= 2:
            # one plain projected-gradient step with a tiny step length shakes the memory-less restart out of a corner
            step = 1e-6 / max(1.0, np.max(np.abs(pg)))
            z = np.maximum(z - step * pg, lo)
            if stall >= 6:
                break
        f_prev = f
    f, pg = proj_grad(z)
    th = unsplit(z)
    f_lit, _, _ = oracle_lib.neg_log_prob_grad_packed(packed, th)
    f_start, _, _ = oracle_lib.neg_log_prob_grad_packed(packed, th0)
    return th, {'f': float(f_lit), 'f_start': float(f_start), 'kkt': float(np.max(np.abs(pg))), 'n_fun': n_fun,
                'success': bool(res.success), 'message': str(res.message)}
