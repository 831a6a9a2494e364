"""dev (round 6): is there a cheap static predictor of a cfg2 fit's length?  Rank correlation of n_eval with features of y
that one pass over the rows could compute, and what an order by the best of them would be worth in a queue simulation
(3 072 slots, longest-first)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from time_series_spark_amd import forecaster as fc, synth
from scipy.stats import spearmanr

N, T = 10000, 730
ds, y = synth.make_panel(N, T, 'linear', seed=751)
spec = fc.ModelSpec(growth='linear', seasonalities=fc.ModelSpec.auto_seasonalities(ds, yearly=True))
r = fc.fit_aligned(spec, ds, y)
ne = r.n_eval.astype(np.float64)
ys = y / np.abs(y).max(axis=1, keepdims=True)
t = np.linspace(0, 1, T)
feats = {}
feats['std'] = ys.std(axis=1)
feats['mean'] = ys.mean(axis=1)
feats['cv'] = ys.std(axis=1) / ys.mean(axis=1)
d1 = np.diff(ys, axis=1)
feats['rough'] = d1.std(axis=1)
feats['rough/std'] = d1.std(axis=1) / ys.std(axis=1)
feats['slope'] = np.abs(((ys - ys.mean(axis=1, keepdims=True)) * (t - 0.5)).sum(axis=1))
w = 73
blocks = ys[:, :w * 10].reshape(N, 10, w).mean(axis=2)
feats['block_curv'] = np.abs(np.diff(blocks, 2, axis=1)).max(axis=1)
feats['block_range'] = blocks.max(axis=1) - blocks.min(axis=1)
feats['first_last'] = np.abs(blocks[:, -1] - blocks[:, 0])
feats['min'] = ys.min(axis=1)
d7 = ys[:, 7:] - ys[:, :-7]
feats['d7std'] = d7.std(axis=1)
feats['noise_ratio'] = d7.std(axis=1) / ys.std(axis=1)
print('n_eval: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f' % (ne.mean(), *np.percentile(ne, [50, 90, 99]), ne.max()))
best = None
for k, v in feats.items():
    rho = spearmanr(v, ne).correlation
    print('%-12s spearman %+.3f' % (k, rho))
    if best is None or abs(rho) > abs(best[1]):
        best = (k, rho)

def sim(order, slots=3072):
    import heapq
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for i in order:
        s = heapq.heappop(h)
        e = s + ne[i]
        end = max(end, e)
        heapq.heappush(h, e)
    return end
base = sim(np.arange(N))
print('queue simulation (evaluations on the longest slot): series order %.0f, true longest-first %.0f (%.3f), by %s %.0f (%.3f)'
      % (base, sim(np.argsort(-ne)), sim(np.argsort(-ne)) / base, best[0],
         sim(np.argsort(-np.sign(best[1]) * feats[best[0]])), sim(np.argsort(-np.sign(best[1]) * feats[best[0]])) / base))
# a least-squares combination of the features (in-sample: an upper bound of what a fixed linear rule could do)
X = np.column_stack([np.ones(N)] + [np.log(np.abs(v) + 1e-9) for v in feats.values()])
coef, *_ = np.linalg.lstsq(X, np.log(ne), rcond=None)
pred = X @ coef
print('log-linear combination: spearman %+.3f, simulated %.3f' % (spearmanr(pred, ne).correlation, sim(np.argsort(-pred)) / base))
