"""The oracle's weighted 1-D k-means (parity with kmeans1d unpinned, see oracle/ganq_oracle.c)
is checked against brute-force enumeration of all contiguous partitions.  CPU only."""
import itertools

import numpy as np

from oracle import c_oracle


def brute_force(x, w, V):
    order = np.argsort(x, kind="stable")
    xs, ws = x[order].astype(np.float64), w[order]
    n = len(xs)
    best = (np.inf, None)
    for cuts in itertools.combinations(range(1, n), V - 1):
        bounds = (0,) + cuts + (n,)
        cost, cents = 0.0, []
        for a, b in zip(bounds[:-1], bounds[1:]):
            sw = ws[a:b].sum()
            mu = (ws[a:b] * xs[a:b]).sum() / sw
            cost += (ws[a:b] * (xs[a:b] - mu) ** 2).sum()
            cents.append(mu)
        if cost < best[0] - 1e-15:
            best = (cost, cents)
    return best


def test_kmeans_matches_brute_force():
    rng = np.random.default_rng(3)
    for trial in range(6):
        n, V = 12, 4
        x = rng.standard_normal(n).astype(np.float32)
        w = rng.uniform(0.1, 10.0, n) ** 2
        T0 = c_oracle.kmeans_init(x[None], w, V)[0]
        cost, cents = brute_force(x, w, V)
        assert np.allclose(T0, np.array(cents, dtype=np.float32), rtol=1e-6, atol=1e-7), (trial, T0, cents)
        assert np.all(np.diff(T0) > 0)


def test_kmeans_unweighted_and_large():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((3, 2000)).astype(np.float32)
    T0 = c_oracle.kmeans_init(x, None, 16)
    assert T0.shape == (3, 16) and np.all(np.diff(T0, axis=1) > 0)
    # Lloyd fixed point: every centroid is the mean of the points nearest to it
    for i in range(3):
        lab = np.abs(x[i][:, None] - T0[i][None]).argmin(1)
        means = np.array([x[i][lab == k].mean() for k in range(16)])
        assert np.allclose(means, T0[i], atol=1e-5)
