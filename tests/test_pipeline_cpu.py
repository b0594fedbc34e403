"""CPU: the host logic of the encode pipeline (viquae_amd/pipeline.py) -- the tokenizer fast path against the tokenizer's own
output, and the lookahead scheduler (order, back-pressure, worker failures).  No GPU work."""
import io
import threading
import time

import numpy as np
import pytest
import torch


def _tokenizer(tmp_path, n_words=600):
    from transformers import BertTokenizer
    rng = np.random.default_rng(0)
    words = sorted({"".join(rng.choice(list("abcdefghijklmnopqrstuvwxyz"), rng.integers(2, 8))) for _ in range(n_words)})
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + words + ["##" + w for w in words[:50]] + [",", ".", "é", "##s"]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n")
    return BertTokenizer(str(tmp_path / "vocab.txt")), words


def _texts(words, n, rng):
    out = []
    for i in range(n):
        k = int(rng.integers(0, 60))
        t = " ".join(rng.choice(words, k)) if k else ""
        if i % 7 == 0:
            t += " Unknownword, with PUNCTUATION. and accents é"
        out.append(t)
    return out


@pytest.mark.parametrize("kwargs", [
    dict(return_tensors="pt", padding="max_length", truncation=True, max_length=32),    # the shipped DPR config's shape
    dict(return_tensors="pt", padding="max_length", truncation=True, max_length=256),
    dict(return_tensors="pt", padding=True, truncation=True, max_length=40),            # pad to the longest of the batch
    dict(return_tensors="pt", padding="longest"),
])
def test_fast_batch_tokenizer_equals_the_tokenizer(tmp_path, kwargs):
    from viquae_amd.pipeline import FastBatchTokenizer
    tok, words = _tokenizer(tmp_path)
    rng = np.random.default_rng(1)
    fast = FastBatchTokenizer(tok, kwargs)
    assert fast.ok
    for n in (1, 5, 64):
        texts = _texts(words, n, rng)
        want = tok(texts, **kwargs)
        got, lens = fast(texts)
        assert list(got.keys()) == list(want.keys())
        for k in want:
            assert got[k].dtype == want[k].dtype == torch.int64 and torch.equal(got[k], want[k]), (k, n)
        assert np.array_equal(lens, want["attention_mask"].sum(1).numpy())
    assert fast.check(_texts(words, 16, rng))


def test_fast_batch_tokenizer_left_padding_and_refusals(tmp_path):
    from viquae_amd.pipeline import FastBatchTokenizer
    tok, words = _tokenizer(tmp_path)
    rng = np.random.default_rng(2)
    texts = _texts(words, 9, rng)
    tok.padding_side = "left"
    kw = dict(return_tensors="pt", padding="max_length", truncation=True, max_length=48)
    fast = FastBatchTokenizer(tok, kw)
    want = tok(texts, **kw)
    got, _ = fast(texts)
    assert all(torch.equal(got[k], want[k]) for k in want)
    tok.padding_side = "right"
    # anything the fast path does not understand is left to the tokenizer itself
    assert not FastBatchTokenizer(tok, dict(return_tensors="np", padding="max_length", max_length=8)).ok
    assert not FastBatchTokenizer(tok, dict(return_tensors="pt", padding="max_length", max_length=8, return_offsets_mapping=True)).ok
    assert not FastBatchTokenizer(tok, dict(return_tensors="pt")).ok            # no padding: ragged, tensors impossible
    assert not FastBatchTokenizer(object(), dict(return_tensors="pt", padding=True)).ok
    # a fast path that disagrees with the tokenizer disables itself
    bad = FastBatchTokenizer(tok, kw)
    bad.pad_id = 3
    assert not bad.check(texts) and not bad.ok


def test_lookahead_runs_one_batch_ahead_in_order():
    from viquae_amd.pipeline import Lookahead
    log, lock = [], threading.Lock()

    class H:
        def __init__(self, j):
            self.j = j

        def result(self):
            with lock:
                log.append(("result", self.j))
            return self.j * 10, None

    def prepare(j):
        time.sleep(0.01)
        with lock:
            log.append(("prepare", j))
        return j

    def launch(j):
        with lock:
            log.append(("launch", j))
        return H(j)

    look = Lookahead(5, prepare, launch, depth=2)
    outs = [look.step(i)[0] for i in range(5)]
    look.close()
    assert outs == [0, 10, 20, 30, 40]
    order = [e for e in log if e[0] != "prepare"]
    # batch i + 1 is launched BEFORE batch i's result is awaited; the last batch has nothing to run ahead of
    assert order == [("launch", 0), ("launch", 1), ("result", 0), ("launch", 2), ("result", 1), ("launch", 3), ("result", 2),
                     ("launch", 4), ("result", 3), ("result", 4)]
    with pytest.raises(RuntimeError):
        look.step(2)


def test_lookahead_back_pressure_and_worker_failure():
    from viquae_amd.pipeline import Lookahead
    prepared = []

    def prepare(j):
        prepared.append(j)
        if j == 3:
            raise ValueError("decoding batch 3 failed")
        return j

    class H:
        def __init__(self, j):
            self.j = j

        def result(self):
            return self.j, None

    look = Lookahead(6, prepare, H, depth=1)
    time.sleep(0.3)
    assert len(prepared) <= 2            # queue of one + the item the worker holds: it does not run away from the consumer
    assert look.step(0)[0] == 0 and look.step(1)[0] == 1
    with pytest.raises(ValueError, match="batch 3"):
        look.step(2)                     # launching batch 3 ahead surfaces the worker's exception in the caller
    look.close()


def test_pipelines_decline_what_they_cannot_prefetch(tmp_path):
    import datasets
    from viquae_amd import pipeline as P
    ds = datasets.Dataset.from_dict({"passage": ["a b", "c"], "n": [1, 2]})
    tok, _ = _tokenizer(tmp_path)
    lin = torch.nn.Linear(2, 2)          # a CPU model: there is no GPU here, and no CPU path either
    assert P.text_pipeline_or_none(ds, {}, model=lin, tokenizer=tok, key="passage") is None
    assert P.text_pipeline_or_none(ds, {}, model=lin, tokenizer=tok, key="passage", run=object()) is None
    assert P.image_pipeline_or_none(ds, {}, model=lin, transform=object(), image_key="passage") is None
    assert not P._plain_dataset(ds.select([1, 0]), {}) and not P._plain_dataset(ds, {"num_proc": 2}) and P._plain_dataset(ds, {"batch_size": 7})
    assert P._arrow_strings(ds, "n") is None and P._arrow_strings(ds, "passage") is not None


def test_decode_pool_writes_rgb_bytes_into_the_shared_slots(tmp_path):
    """The image pipeline's decode workers (forked processes, two phases): sizes with load_image's error handling, then the
    RGB bytes of every readable file at the byte offsets the parent planned -- equal to np.asarray(Image.open().convert('RGB')),
    for RGB / greyscale / palette / RGBA files, a missing file, a file that opens but cannot be decoded, and across both slots."""
    from PIL import Image
    from viquae_amd.image.decode_pool import DecodePool
    rng = np.random.default_rng(0)
    paths, want = [], []
    for i, (mode, ext) in enumerate([("RGB", "png"), ("L", "png"), ("RGB", "jpg"), ("P", "png"), ("RGBA", "png"), ("RGB", "bmp")] * 3):
        h, w = int(rng.integers(5, 60)), int(rng.integers(5, 60))
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        im = Image.fromarray(a).convert(mode) if mode != "RGBA" else Image.fromarray(np.dstack([a, a[..., :1]]))
        p = str(tmp_path / f"{i}.{ext}")
        im.save(p)
        paths.append(p)
        want.append(np.asarray(Image.open(p).convert("RGB")))
    paths.insert(4, str(tmp_path / "missing.png"))
    want.insert(4, None)
    trunc = tmp_path / "trunc.jpg"
    big = tmp_path / "big.jpg"
    Image.fromarray(rng.integers(0, 256, (200, 200, 3), dtype=np.uint8)).save(str(big), quality=95)
    data = open(str(big), "rb").read()
    trunc.write_bytes(data[: len(data) // 2])                    # the header parses, the data does not
    paths.insert(9, str(trunc))
    want.insert(9, "fails")
    pool = DecodePool(3, 1 << 20, n_slots=2)
    try:
        for rep in range(3):                                      # slots are reused round-robin
            with pytest.warns(UserWarning, match="missing.png"):
                sizes = pool.sizes(paths)
            assert [s is None for s in sizes] == [w is None for w in want]
            kept = [i for i, s in enumerate(sizes) if s is not None]
            offs, o = {}, 64 * rep
            for i in kept:
                offs[i] = o
                o += -(-(sizes[i][0] * sizes[i][1] * 3) // 16) * 16
            slot = pool.take_slot()
            assert slot == rep % 2
            with pytest.warns(UserWarning, match="trunc.jpg"):
                failed = pool.decode(slot, offs, staged=set())    # every file as RGB bytes (the JPEG files too: Pillow)
            assert failed == {9}
            buf = pool.tensors[slot].numpy()
            for i in kept:
                if i == 9:
                    continue
                h, w = sizes[i]
                assert want[i].shape == (h, w, 3)
                assert np.array_equal(buf[offs[i]:offs[i] + h * w * 3].reshape(h, w, 3), want[i]), i
    finally:
        pool.close()


def test_decode_pool_stages_jpeg_scans_for_the_device(tmp_path):
    """JPEG files the split decoder covers leave the workers as staging areas (header + quantised coefficients, csrc/jpeg.hip) at
    the offsets of `viquae_amd.image.jpeg.plan_layout`; the oracle's inverse DCT / upsampling / colour conversion on those areas
    gives Pillow's pixels (a progressive file among them).  A PNG goes the Pillow way (RGB bytes); a JPEG whose scan is damaged but which
    Pillow still decodes is stored as RGB inside its staging area; a truncated one fails like before."""
    from PIL import Image, ImageFile
    from oracle import jpeg as oj
    from viquae_amd.image import jpeg as dj
    from viquae_amd.image.decode_pool import DecodePool
    rng = np.random.default_rng(1)
    paths = []
    for i, kw in enumerate([dict(subsampling=2), dict(subsampling=0, quality=95), dict(subsampling=1, optimize=True), dict(progressive=True),
                            dict(quality=30, restart_marker_blocks=2), None, dict(), dict()]):
        h, w = int(rng.integers(20, 90)), int(rng.integers(20, 90))
        a = np.clip(rng.normal(128, 50, (h, w, 3)), 0, 255).astype(np.uint8)
        p = str(tmp_path / (f"{i}.jpg" if kw is not None else f"{i}.png"))
        im = Image.fromarray(a)
        if i == 6:
            im = im.convert("L")
        im.save(p, **(kw or {}))
        paths.append(p)
    data = open(paths[7], "rb").read()
    (tmp_path / "cut.jpg").write_bytes(data[: len(data) * 2 // 3])     # fails in Pillow as well
    paths.append(str(tmp_path / "cut.jpg"))
    sos = data.find(b"\xff\xda")
    damaged = None
    for pos in range(sos + 20, len(data) - 2):                          # a damaged scan that Pillow still decodes (with a warning)
        d = bytearray(data)
        d[pos:pos + 2] = b"\xff\xd3"                                     # a restart marker where none belongs
        p = dj.probe(bytes(d))
        st = np.zeros(p[4], dtype=np.uint8)
        if not dj.stage(bytes(d), st.ctypes.data, st.size):
            try:
                Image.open(io.BytesIO(bytes(d))).convert("RGB")
                damaged = bytes(d)
                break
            except OSError:
                continue
    assert damaged is not None
    (tmp_path / "damaged.jpg").write_bytes(damaged)
    paths.append(str(tmp_path / "damaged.jpg"))
    want = [None if p.endswith("cut.jpg") else np.asarray(Image.open(p).convert("RGB")) for p in paths]
    pool = DecodePool(2, 1 << 21, n_slots=2)
    try:
        sizes = pool.sizes(paths)
        assert all(s is not None for s in sizes)
        assert set(pool.last_jpeg) == {0, 1, 2, 3, 4, 6, 7, 8, 9}       # not the PNG
        geom = np.zeros((len(paths), 12), dtype=np.int64)
        geom[:, 1:3] = sizes
        totals = np.zeros(5, dtype=np.int64)
        layout = dj.plan_layout(geom, totals, dict(pool.last_jpeg))
        assert layout["h2d_bytes"] <= pool.slot_bytes and totals[0] > layout["h2d_bytes"] and (geom[:, 0] % 16 == 0).all()
        slot = pool.take_slot()
        with pytest.warns(UserWarning, match="cut.jpg"):
            failed = pool.decode(slot, {i: int(layout["staging"].get(i, geom[i, 0])) for i in range(len(paths))})
        assert failed == {8}
        buf = pool.tensors[slot].numpy()
        for i, (h, w) in enumerate(sizes):
            if i == 8:
                continue
            if i in layout["staging"]:
                o = layout["staging"][i]
                got = oj.decode_staging(buf[o:o + pool.last_jpeg[i][1]])
                magic = int(buf[o:o + 4].view(np.int32)[0])
                assert magic == (dj.MAGIC_RGB if i == 9 else 0x4745504A)
            else:
                got = buf[geom[i, 0]:geom[i, 0] + h * w * 3].reshape(h, w, 3)
            assert np.array_equal(got, want[i]), i
        # a batch whose staging areas do not fit: the caller demotes the JPEG files to Pillow's RGB bytes
        sizes = pool.sizes(paths[:3])
        offs, o = {}, 0
        for i, (h, w) in enumerate(sizes):
            offs[i] = o
            o += -(-(h * w * 3) // 16) * 16
        slot = pool.take_slot()
        assert pool.decode(slot, offs, staged=set()) == set()
        buf = pool.tensors[slot].numpy()
        for i, (h, w) in enumerate(sizes):
            assert np.array_equal(buf[offs[i]:offs[i] + h * w * 3].reshape(h, w, 3), want[i])
    finally:
        pool.close()


def test_job_fingerprint_is_deterministic_and_sees_every_part():
    """viquae_amd.utils.job_fingerprint: the deterministic `new_fingerprint` of the pipelined embedding jobs."""
    import datasets
    import torch
    from viquae_amd.utils import job_fingerprint
    ds = datasets.Dataset.from_dict({"passage": ["a", "b"]})
    torch.manual_seed(0)
    m1 = torch.nn.Linear(8, 4)
    m2 = torch.nn.Linear(8, 4)
    m2.load_state_dict(m1.state_dict())
    kw = dict(key="passage", save_as="emb", tokenization_kwargs={"max_length": 16, "padding": "max_length"}, layers=None)
    f1 = job_fingerprint(ds, "job", model=m1, **kw)
    assert f1 == job_fingerprint(ds, "job", model=m2, **kw) and len(f1) == 16     # same weights in another object: same job
    with torch.no_grad():
        m2.weight[1, 2] += 1e-3
    assert job_fingerprint(ds, "job", model=m2, **kw) != f1                       # one weight
    with torch.no_grad():                                                          # two weights swapped: same sum, other order
        m2.load_state_dict(m1.state_dict())
        a, b = m2.weight[0, 0].clone(), m2.weight[0, 1].clone()
        m2.weight[0, 0], m2.weight[0, 1] = b, a
    assert job_fingerprint(ds, "job", model=m2, **kw) != f1
    assert job_fingerprint(ds, "other job", model=m1, **kw) != f1
    assert job_fingerprint(ds, "job", model=m1, **dict(kw, save_as="emb2")) != f1
    assert job_fingerprint(ds, "job", model=m1, **dict(kw, tokenization_kwargs={"max_length": 32, "padding": "max_length"})) != f1
    assert job_fingerprint(datasets.Dataset.from_dict({"passage": ["a", "c"]}), "job", model=m1, **kw) != f1


def test_job_fingerprint_takes_every_kind_of_model_config():
    import datasets
    import torch
    from viquae_amd.utils import job_fingerprint
    ds = datasets.Dataset.from_dict({"passage": ["a"]})

    class M(torch.nn.Module):
        def __init__(self, config):
            super().__init__()
            self.config = config
            self.register_buffer("w", torch.arange(6.0).reshape(2, 3), persistent=False)   # the HIP encoders' weights are such buffers

    class Cfg:
        def __init__(self):
            self.hidden = 8

    fps = {job_fingerprint(ds, "job", model=M(c)) for c in ({"hidden_size": 8}, {"hidden_size": 16}, Cfg(), None, 3)}
    assert len(fps) == 5
    m = M({"hidden_size": 8})
    m.w[0, 0] = 1.0
    assert job_fingerprint(ds, "job", model=m) not in fps     # non-persistent buffers are part of the model's identity


def test_job_fingerprint_follows_the_code_that_computes_the_embeddings(monkeypatch):
    """the deterministic fingerprint must change when the library (or the pipeline source) changes, otherwise a
    rebuilt libmeerqat_hip.so serves the stale cached column."""
    from viquae_amd import utils

    class DS:
        _fingerprint = "abc"
    a = utils.job_fingerprint(DS(), "job", key="passage")
    assert a == utils.job_fingerprint(DS(), "job", key="passage")
    assert "meerqat_hip" in utils.code_fingerprint()
    monkeypatch.setattr(utils, "_CODE_FINGERPRINT", "meerqat_hip 9.9 (gfx950):0123456789abcdef")
    assert utils.job_fingerprint(DS(), "job", key="passage") != a


def test_decode_pool_runs_small_host_functions_in_the_workers(tmp_path):
    """DecodePool.call_start / call_finish: a named function over per-image arguments, spread over the workers like the files
    (the face job's alignment matrices); replies in request order with a decode in flight; a failure is raised in the parent."""
    from PIL import Image
    from viquae_amd.image import face_recognition as fr
    from viquae_amd.image.decode_pool import DecodePool
    rng = np.random.default_rng(0)
    paths = []
    for i in range(7):
        p = str(tmp_path / f"{i}.png")
        Image.fromarray(rng.integers(0, 256, (20 + i, 30, 3), dtype=np.uint8)).save(p)
        paths.append(p)
    lms = [[(fr.SRC * (0.8 + 0.1 * f) + np.array([10.0 * i, 5.0 * f], np.float32)).tolist() for f in range(1 + i % 3)] for i in range(7)]
    pool = DecodePool(3, 1 << 20, n_slots=2)
    try:
        sizes = pool.sizes(paths)
        pool.call_start("viquae_amd.image.face_recognition", "face_matrices", {i: (lms[i], 2) for i in range(7) if i != 3})
        offs, o = {}, 0
        for i, (h, w) in enumerate(sizes):
            offs[i] = o
            o += -(-(h * w * 3) // 16) * 16
        pool.decode_start(pool.take_slot(), offs)
        got = pool.call_finish()
        assert pool.decode_finish() == set() and sorted(got) == [0, 1, 2, 4, 5, 6]
        for i, (n, mats) in got.items():
            want = fr.face_matrices((lms[i], 2))
            assert n == want[0] == min(2, len(lms[i])) and all(np.array_equal(a, b) for a, b in zip(mats, want[1]))
        pool.sizes(paths)
        pool.call_start("viquae_amd.image.face_recognition", "no_such_function", {0: 1})
        with pytest.raises(RuntimeError, match="no_such_function"):
            pool.call_finish()
    finally:
        pool.close()
