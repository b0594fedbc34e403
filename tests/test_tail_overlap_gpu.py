"""A screened search of several query chunks runs what follows each chunk's scan on a second stream under the next chunk's
scan (MQ_KNN_FLAG_PHASE_*, MI355XFlatIndex._search_chunks_pipelined, DistributedFlatIndex.search_device).  Same kernels per
chunk, so the bar is bit-identity with the serial chunk loop (MQ_KNN_TAIL_OVERLAP=0) and with the exact fp32 scan."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(monkeypatch, fn):
    monkeypatch.setenv("MQ_KNN_TAIL_OVERLAP", "0")
    serial = fn()
    monkeypatch.setenv("MQ_KNN_TAIL_OVERLAP", "1")
    return serial, fn()


@pytest.mark.parametrize("metric,factory,form", [(0, "Flat", "numpy"), (1, "Flat", "numpy"), (0, "L2norm,Flat", "faiss"),
                                                 (1, "L2norm,Flat", "numpy")])
@pytest.mark.parametrize("nq", [4097, 8192 + 57, 4096 * 3 + 5])  # 4097 and 3 * 4096 + 5: a short last chunk (FAISS's small-batch L2 form)
def test_chunked_search_is_bit_identical_to_the_serial_loop(monkeypatch, metric, factory, form, nq):
    import torch
    from viquae_amd.index import MI355XFlatIndex
    g = torch.Generator(device="cuda").manual_seed(nq + metric)
    X = torch.randn((30011, 96), generator=g, device="cuda")
    Q = torch.randn((nq, 96), generator=g, device="cuda")
    idx = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=True, l2norm_form=form)
    idx.add(X)
    exact = MI355XFlatIndex(string_factory=factory, metric_type=metric, screen=False, l2norm_form=form)
    exact.add(X)
    (D0, I0), (D1, I1) = _both(monkeypatch, lambda: tuple(t.clone() for t in idx.search_device(Q, 20)))
    assert torch.equal(I0, I1) and torch.equal(D0, D1)
    De, Ie = exact.search_device(Q, 20)
    assert torch.equal(I1, Ie) and torch.equal(D1, De)
    # twice in a row (the workspaces and the second stream are reused) and on a side stream of the caller
    D2, I2 = idx.search_device(Q, 20)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        D3, I3 = idx.search_device(Q, 20)
    side.synchronize()
    assert torch.equal(I2, Ie) and torch.equal(D2, De) and torch.equal(I3, Ie) and torch.equal(D3, De)


def test_chunks_the_screen_does_not_serve(monkeypatch):
    """k beyond the screen's range: the FRONT call is the whole (exact-rounds) search, the TAIL call a no-op."""
    import torch
    from viquae_amd.index import MI355XFlatIndex
    g = torch.Generator(device="cuda").manual_seed(3)
    X = torch.randn((9000, 64), generator=g, device="cuda")
    Q = torch.randn((4096 + 300, 64), generator=g, device="cuda")
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
    idx.add(X)
    (D0, I0), (D1, I1) = _both(monkeypatch, lambda: tuple(t.clone() for t in idx.search_device(Q, 300)))
    assert torch.equal(I0, I1) and torch.equal(D0, D1)
    S = Q @ X.T
    assert torch.equal(I1[:, 0], S.argmax(dim=1))


def test_flagged_tiles_are_recomputed_on_the_second_stream(monkeypatch):
    """Rows inside every query's margin by the tens of thousands overflow the candidate buffers: the exact-scan recomputation
    (last step of the second half) then does real work next to the following chunk's scan."""
    import torch
    from viquae_amd.index import MI355XFlatIndex
    g = torch.Generator(device="cuda").manual_seed(5)
    c = torch.randn((1, 128), generator=g, device="cuda") * 4
    X = c + 1e-4 * torch.randn((70000, 128), generator=g, device="cuda")
    Q = c + 0.05 * torch.randn((4096 * 2 + 100, 128), generator=g, device="cuda")
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
    idx.add(X)
    exact = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=False)
    exact.add(X)
    (D0, I0), (D1, I1) = _both(monkeypatch, lambda: tuple(t.clone() for t in idx.search_device(Q, 100)))
    De, Ie = exact.search_device(Q, 100)
    assert idx.screen_stats(100, 100)[0] > 0  # the last chunk had flagged tiles
    assert torch.equal(I0, Ie) and torch.equal(D0, De) and torch.equal(I1, Ie) and torch.equal(D1, De)


def test_phase_flags_through_the_c_abi():
    """FRONT then TAIL on ONE stream = the one-call search; both bits = the one-call search; unknown bits are refused."""
    import torch
    from viquae_amd import _lib
    from viquae_amd.index import FLAG_PHASE_FRONT, FLAG_PHASE_TAIL, MI355XFlatIndex
    g = torch.Generator(device="cuda").manual_seed(9)
    X = torch.randn((20000, 80), generator=g, device="cuda")
    Q = torch.randn((700, 80), generator=g, device="cuda")
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
    idx.add(X)
    D, I = idx.search_device(Q, 10)
    spaces, tail = idx.pipeline_workspaces(700, 10)
    main = torch.cuda.current_stream()
    for phases in ((FLAG_PHASE_FRONT, FLAG_PHASE_TAIL), (FLAG_PHASE_FRONT | FLAG_PHASE_TAIL,)):
        out = (torch.zeros_like(D), torch.zeros_like(I))
        for ph in phases:
            idx.search_phase(Q, 10, out, spaces[1], ph, main)
        assert torch.equal(out[0], D) and torch.equal(out[1], I)
    lib = _lib.load()
    rc = lib.mq_knn_search_screened_f32(None, idx._sqnorm.data_ptr(), idx._rowmajor.data_ptr(), idx._bf16.data_ptr(),
                                        idx._xmax2.data_ptr(), idx.ntotal, idx.d, Q.data_ptr(), 700, 10, 0, 32, 0, D.data_ptr(),
                                        I.data_ptr(), spaces[0].data_ptr(), spaces[0].numel(), main.cuda_stream, None, None)
    assert rc != 0


@pytest.mark.parametrize("front,tail", [(8, 4), (4, 8)])
def test_streaming_variant_switched_between_the_two_phases(front, tail):
    """The two streaming scans leave their pools in different layouts (knn_screen.inc: POOL_LAYOUT_*); the scan records which in
    the workspace.  A TAIL call whose host-side choice disagrees with what the FRONT call's scan left (MQ_KNN_OPT_SMALL_WAVES
    changed in between) must notice and hand the tile to the exact scan -- same answer, never a walk of the wrong layout."""
    import torch
    from viquae_amd import _lib
    from viquae_amd.index import FLAG_PHASE_FRONT, FLAG_PHASE_TAIL, MI355XFlatIndex
    g = torch.Generator(device="cuda").manual_seed(front)
    X = torch.randn((70000, 96), generator=g, device="cuda")
    Q = torch.randn((200, 96), generator=g, device="cuda")
    idx = MI355XFlatIndex(string_factory="Flat", metric_type=0, screen=True)
    idx.add(X)
    assert idx.scan_kind(200, 100) == "stream"
    D, I = idx.search_device(Q, 100)
    assert idx.screen_stats(200, 100)[0] == 0
    spaces, _ = idx.pipeline_workspaces(200, 100)
    main = torch.cuda.current_stream()
    out = (torch.zeros_like(D), torch.zeros_like(I))
    with _lib.knn_option(_lib.KNN_OPT_SMALL_WAVES, front):
        idx.search_phase(Q, 100, out, spaces[1], FLAG_PHASE_FRONT, main)
    with _lib.knn_option(_lib.KNN_OPT_SMALL_WAVES, tail):
        idx.search_phase(Q, 100, out, spaces[1], FLAG_PHASE_TAIL, main)
    torch.cuda.synchronize()
    assert torch.equal(out[0], D) and torch.equal(out[1], I)
    idx._last_ws = spaces[1]
    idx._last_call_nq = 200
    assert idx.screen_stats(200, 100)[0] == 1  # recomputed by the exact scan
