"""k beyond the screen's own range (224): P = ceil(k / 112) contiguous row ranges are searched for their exact top-224 by the
screened pipeline, merged, and the merge proves the result or flags the query tile for the exact rounds (csrc/knn.hip,
partition_merge_kernel).  Bar: bit-identical scores and ids with the exact fp32 scan, whichever way a tile was served."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(X, metric, factory="Flat", tie_order=None):
    from viquae_amd.index import MI355XFlatIndex
    a = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=True, tie_order=tie_order)
    a.add(X)
    b = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=False, tie_order=tie_order)
    b.add(X)
    return a, b


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("k", [225, 256, 500, 1000, 1792, 1793])
def test_big_k_equals_the_exact_scan(metric, k):
    g = torch.Generator(device="cuda").manual_seed(k + metric)
    X = torch.randn((300_000, 64), generator=g, device="cuda")
    Q = torch.randn((300, 64), generator=g, device="cuda")
    a, b = _pair(X, metric)
    D, I = a.search_device(Q, k)
    De, Ie = b.search_device(Q, k)
    assert torch.equal(I, Ie) and torch.equal(D, De)
    assert a.scan_kind(300, k) == ("tile" if k <= 1792 else "none")


@pytest.mark.parametrize("tie_order", ["id_asc", "id_desc"])
def test_ties_across_the_ranges(tie_order):
    """Integer data: thousands of exactly equal scores spread over all ranges -- membership at the k-th place and the order inside
    the answer follow the tie rule across range boundaries."""
    g = torch.Generator(device="cuda").manual_seed(7)
    X = torch.randint(-2, 3, (200_000, 16), generator=g, device="cuda").float()
    Q = torch.randint(-2, 3, (70, 16), generator=g, device="cuda").float()
    for metric in (0, 1):
        a, b = _pair(X, metric, tie_order=tie_order)
        D, I = a.search_device(Q, 400)
        De, Ie = b.search_device(Q, 400)
        assert torch.equal(I, Ie) and torch.equal(D, De)


def test_a_range_that_holds_more_than_it_can_deliver_is_recomputed():
    """Rows sorted by their score for the first queries: the whole top-k sits in ONE range, the merge cannot prove the answer
    and the tile goes through the exact rounds; other tiles of the same call are served by the ranges."""
    g = torch.Generator(device="cuda").manual_seed(11)
    X = torch.randn((262_144, 32), generator=g, device="cuda")
    Q = torch.randn((600, 32), generator=g, device="cuda")
    X = X[torch.argsort(X @ Q[0])]
    Q[:256] = Q[0] + 0.01 * torch.randn((256, 32), generator=g, device="cuda")   # the first tile: all of it in the last range
    a, b = _pair(X, 0)
    D, I = a.search_device(Q, 512)
    De, Ie = b.search_device(Q, 512)
    assert int((I[:256] >= 262_144 - 40_000).all())          # the planted structure is there
    assert torch.equal(I, Ie) and torch.equal(D, De)


def test_l2norm_small_index_and_numpy_boundary():
    """'L2norm,' queries are normalised once per range call; an index too small for ranges keeps the exact rounds; search_batch."""
    rng = np.random.default_rng(3)
    X = rng.standard_normal((150_000, 48), dtype=np.float32)
    Q = rng.standard_normal((33, 48), dtype=np.float32)
    a, b = _pair(torch.from_numpy(X).cuda(), 0, factory="L2norm,Flat")
    Da, Ia = a.search_batch(Q, 300)
    Db, Ib = b.search_batch(Q, 300)
    assert np.array_equal(Ia, Ib) and np.array_equal(Da, Db)
    small, small_e = _pair(torch.from_numpy(X[:20_000]).cuda(), 1)
    assert small.scan_kind(33, 300) == "none"
    Ds, Is = small.search_batch(Q, 300)
    De, Ie = small_e.search_batch(Q, 300)
    assert np.array_equal(Is, Ie) and np.array_equal(Ds, De)
