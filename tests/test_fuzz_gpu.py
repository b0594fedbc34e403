"""GPU fuzz: random shapes / k / metrics / data regimes through the index boundary vs the CPU oracle, bit-exact.
Covers ragged tails (N, nq, d not multiples of the tile sizes), tiny and multi-tile query batches, both search
paths, duplicated rows (ties) and sorted data (worst case for threshold pruning)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 63, 64, 65, 255, 256, 257, 1000, 4097, 12345, 30000]))
    d = int(rng.choice([1, 3, 16, 17, 64, 100, 128, 257]))
    nq = int(rng.choice([1, 2, 19, 20, 255, 256, 257, 600]))
    k = int(rng.choice([1, 7, 100, 128]))
    metric = int(rng.integers(0, 2))
    regime = rng.choice(["normal", "lattice", "dups", "sorted", "mixed_scale"])
    if regime == "lattice":
        X = rng.integers(-8, 9, (n, d)).astype(np.float32)
        Q = rng.integers(-8, 9, (nq, d)).astype(np.float32)
    else:
        X = rng.standard_normal((n, d), dtype=np.float32)
        Q = rng.standard_normal((nq, d), dtype=np.float32)
        if regime == "dups" and n > 4:
            X[rng.integers(0, n, n // 2)] = X[rng.integers(0, n, n // 2)]
        if regime == "sorted":
            X = X[np.argsort(X @ Q[0])]
        if regime == "mixed_scale":
            X *= np.exp(rng.standard_normal((n, 1)).astype(np.float32) * 2)
    return X, Q, k, metric, regime


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_against_oracle(seed):
    from oracle import knn as ok
    from viquae_amd.index import MI355XFlatIndex
    X, Q, k, metric, regime = _case(seed)
    Do, Io = ok.knn(X, Q, k, metric=metric)
    for screen in (True, False):
        idx = MI355XFlatIndex(string_factory="Flat", metric_type=metric, screen=screen)
        idx.add_vectors(X)
        D, I = idx.search_batch(Q, k)
        assert np.array_equal(I, Io), (seed, regime, X.shape, Q.shape, k, metric, screen)
        assert np.array_equal(D, Do), (seed, regime, X.shape, Q.shape, k, metric, screen)
