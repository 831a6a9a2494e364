"""dev (CPU, uses the oracle): can a first fit know which series will run long?  VERDICT round 3, item 3.

The headline launch (10 000 series on 3 072 wave slots, random order) ends ~35 % after the time its work
would take on a perfectly balanced machine; with the series handed out longest-first (tsf_set_cost_hints fed the
counts of a previous fit) most of that tail disappears.  A FIRST fit has no such counts.  This probe measures what
could stand in for them, on the oracle's own trajectories (bit-identical to the kernel's):

  * static features of a series (level, slope, OLS residual level and slope changes from one shared ridge solve);
  * a PILOT: every series runs its first 8 / 16 / 32 / 64 L-BFGS iterations (max_iter-capped fits: the same
    trajectory, truncated), and the decrease of the objective over the last stretches is the feature;

and what a queue ordered by the best predictor would gain (greedy list scheduling over the evaluation counts,
one time per evaluation, slots / series as in the headline launch).

Result (3 000 series of the bench panel, round 4): rank correlation with the evaluation count <= 0.22 for every
static feature, 0.42 / 0.50 / 0.61 for the objective decrease over iterations 16-32 / 32-64 / 64-end, 0.55 for
a log-linear fit of the pilot features (hold-out half).  Launch in evaluation-times: random order 2 252, true
longest-first 1 660, pilot of 64 iterations + predicted-longest-first 2 074 (2 172 with a barrier between the
phases), pilot of 32 iterations 1 995: 8-11 % of the launch before the cost of suspending and resuming 10 000
fits, against the 26 % of the true order -- under the 8.0 ms the verdict asked for (9.3 ms x 0.89 = 8.3).
Not built.  (profiles/r04_quad/pilot_schedule.txt)

    python tests/dev/pilot_schedule_probe.py [n_series]
"""
import heapq
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import canon_lib as cl  # noqa: E402
from time_series_spark_amd import synth  # noqa: E402

SEAS = [(365.25, 10, 'additive', 10.0), (7, 3, 'additive', 10.0)]
CAPS = (10000, 8, 16, 32, 64)
_G = {}


def _work(n):
    out = []
    for mi in CAPS:
        sp = cl.make_spec(seasonalities=SEAS, eval_mode=1, max_iter=mi)
        o = cl.fit(sp, _G['ds'], _G['y'][n])
        out += [o['n_eval'], o['n_iter'], o['f'], o['status']]
    return out


def listsched(jobs, slots, t0=None):
    h = [0.0] * slots if t0 is None else list(t0)
    heapq.heapify(h)
    for j in jobs:
        heapq.heappush(h, heapq.heappop(h) + j)
    return h


def main():
    from scipy.stats import spearmanr
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    ds, y = synth.make_panel(10000, 730, 'linear', seed=751)
    _G.update(ds=ds, y=y[:N])
    cl.lib()
    with mp.get_context('fork').Pool(os.cpu_count()) as p:
        R = np.array(p.map(_work, range(N), chunksize=16), dtype=np.float64)
    ne, f = R[:, 0], R[:, 2]
    f8, f16, f32, f64 = R[:, 6], R[:, 10], R[:, 14], R[:, 18]
    e32, e64 = R[:, 12], R[:, 16]
    print('evaluations: mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f' % (ne.mean(), np.median(ne), np.percentile(ne, 90), np.percentile(ne, 99), ne.max()))
    # static features: one ridge solve shared by every series
    sp = cl.make_spec(seasonalities=SEAS, eval_mode=1)
    d = cl.design(sp, ds, y[0])
    Z = np.column_stack([d['t'], np.ones_like(d['t'])] + [np.maximum(d['t'] - s, 0) for s in d['t_change']] + [d['X']])
    ys = y[:N] / np.abs(y[:N]).max(axis=1, keepdims=True)
    th = (ys @ Z) @ np.linalg.inv(Z.T @ Z + 1e-2 * np.eye(Z.shape[1]))
    sig = (ys - th @ Z.T).std(axis=1)
    feats = {'OLS residual level': sig, 'OLS |delta|_1 / residual level': np.abs(th[:, 2:27]).sum(axis=1) / sig,
             '|y_T - y_0|': np.abs(ys[:, -1] - ys[:, 0]), 'mean level': ys.mean(axis=1),
             'pilot: f(16) - f(32)': f16 - f32, 'pilot: f(32) - f(64)': f32 - f64, '(unknowable) f(64) - f(end)': f64 - f}
    for k, v in feats.items():
        print('  %-34s rank correlation with the evaluation count %+.3f' % (k, spearmanr(v, ne).correlation))
    X = np.column_stack([np.log(np.abs(f32 - f64) + 1e-12), np.log(np.abs(f16 - f32) + 1e-12), np.log(np.abs(f8 - f16) + 1e-12), e64,
                         np.log(np.abs(f64)), np.ones(N)])
    h = N // 2
    w = np.linalg.lstsq(X[:h], np.log(ne[:h]), rcond=None)[0]
    print('  log-linear fit of the pilot features, hold-out half: %+.3f' % spearmanr(X[h:] @ w, ne[h:]).correlation)
    pred = np.exp(X @ np.linalg.lstsq(X, np.log(ne), rcond=None)[0])
    S = int(round(3072 * N / 10000))
    rng = np.random.default_rng(0)
    r = [max(listsched(ne[rng.permutation(N)], S)) for _ in range(20)]
    print('launch in evaluation-times on %d slots: work / slots %.0f, longest fit %.0f' % (S, ne.sum() / S, ne.max()))
    print('  random order               %.0f (%.0f .. %.0f)' % (np.mean(r), min(r), max(r)))
    print('  true longest-first         %.0f' % max(listsched(np.sort(ne)[::-1], S)))
    hq = listsched(e64, S)
    rem = np.maximum(ne - e64, 0)
    print('  pilot 64 it + barrier + predicted-longest-first   %.0f' % max(listsched(rem[np.argsort(-pred)], S, [max(hq)] * S)))
    print('  pilot 64 it, no barrier, predicted-longest-first  %.0f' % max(listsched(rem[np.argsort(-pred)], S, hq)))
    print('  pilot 64 it, TRUE longest-first of the remainder  %.0f' % max(listsched(rem[np.argsort(-rem)], S, hq)))
    X2 = np.column_stack([np.log(np.abs(f16 - f32) + 1e-12), np.log(np.abs(f8 - f16) + 1e-12), e32, np.log(np.abs(f32)), np.ones(N)])
    p2 = np.exp(X2 @ np.linalg.lstsq(X2, np.log(ne), rcond=None)[0])
    hq = listsched(e32, S)
    rem = np.maximum(ne - e32, 0)
    print('  pilot 32 it, no barrier, predicted-longest-first  %.0f' % max(listsched(rem[np.argsort(-p2)], S, hq)))


if __name__ == '__main__':
    main()
