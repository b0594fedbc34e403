"""The split JPEG decoder (csrc/jpeg.hip: Huffman scan on the host, inverse DCT / upsampling / colour on the GPU) against
Pillow itself and against oracle/jpeg.py, bit for bit -- what `load_image` (meerqat/data/loading.py:108-124) returns."""
import io
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
torch = pytest.importorskip("torch")
Image = pytest.importorskip("PIL.Image")

pytestmark = pytest.mark.gpu


def _pillow(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def _files(seed, n, max_side=320):
    import jpeg_pillow_parity as jp
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        h, w = jp.sizes(rng)
        data, kw = jp.encode(rng, jp.picture(rng, min(h, max_side), min(w, max_side), grey=(i % 8 == 3)))
        out.append(data)
    return out


def test_batch_of_files_decodes_like_pillow_and_the_oracle():
    from oracle import jpeg as oj
    from viquae_amd.image import jpeg as dj
    files = _files(11, 96)
    got = dj.decode_files(files)
    for n, (d, g) in enumerate(zip(files, got)):
        ref = _pillow(d)
        assert np.array_equal(g.cpu().numpy(), ref), n
        if n % 8 == 0:
            assert np.array_equal(ref, oj.decode(d)), n


@pytest.mark.parametrize("subsampling", [0, 1, 2])
def test_small_and_odd_sizes(subsampling):
    from viquae_amd.image import jpeg as dj
    rng = np.random.default_rng(5 + subsampling)
    files = []
    for h, w in [(1, 1), (1, 7), (2, 2), (3, 5), (5, 3), (8, 8), (9, 17), (16, 16), (17, 33), (31, 2), (33, 4), (4, 6), (250, 3), (3, 250)]:
        buf = io.BytesIO()
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB").save(buf, "JPEG", quality=90, subsampling=subsampling)
        files.append(buf.getvalue())
    for d, g in zip(files, dj.decode_files(files)):
        assert np.array_equal(g.cpu().numpy(), _pillow(d))


def test_h1v2_file_restart_intervals_extreme_qualities():
    import jpeg_pillow_parity as jp
    from viquae_amd.image import jpeg as dj
    rng = np.random.default_rng(3)
    buf = io.BytesIO()
    jp.picture(rng, 48, 64).save(buf, "JPEG", quality=85, subsampling=1)
    data = bytearray(buf.getvalue())     # a 4:2:2 stream re-labelled 4:4:0, see tests/test_jpeg_oracle_cpu.py
    at = data.find(b"\xff\xc0")
    data[at + 5:at + 9] = bytes([0, 96, 0, 32])
    data[at + 11] = 0x12
    files = [bytes(data)]
    im = jp.picture(rng, 70, 90)
    for kw in (dict(restart_marker_blocks=1), dict(restart_marker_rows=1), dict(optimize=True), dict(quality=100), dict(quality=1)):
        buf = io.BytesIO()
        im.save(buf, "JPEG", **kw)
        files.append(buf.getvalue())
    for d, g in zip(files, dj.decode_files(files)):
        assert np.array_equal(g.cpu().numpy(), _pillow(d))


def test_one_large_photo_sized_file():
    import jpeg_pillow_parity as jp
    from viquae_amd.image import jpeg as dj
    buf = io.BytesIO()
    jp.picture(np.random.default_rng(9), 1201, 1603).save(buf, "JPEG", quality=80)
    (g,) = dj.decode_files([buf.getvalue()])
    assert np.array_equal(g.cpu().numpy(), _pillow(buf.getvalue()))


def test_host_decoded_image_is_copied_through():
    """The staging form of a file Pillow had to decode (MQ_JPEG_MAGIC_RGB): the device moves its bytes."""
    from viquae_amd.image import jpeg as dj
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    host = np.zeros(dj.HEADER + 37 * 53 * 3 + 64, dtype=np.uint8)
    dj.stage_rgb(rgb, host, 0)
    n = len(host) + ((37 * 53 * 3 + 15) & ~15)
    buf = torch.zeros(n, dtype=torch.uint8, device="cuda")
    buf[:len(host)] = torch.from_numpy(host).cuda()
    dj.decode_staged(buf, np.array([[0, len(host)]], dtype=np.int64), 0, dj.strips(37, 53))
    torch.cuda.synchronize()
    assert np.array_equal(buf[len(host):len(host) + 37 * 53 * 3].cpu().numpy().reshape(37, 53, 3), rgb)


def test_progressive_files():
    import jpeg_pillow_parity as jp
    from viquae_amd.image import jpeg as dj
    rng = np.random.default_rng(21)
    files = []
    for i in range(24):
        h, w = jp.sizes(rng)
        im = jp.picture(rng, min(h, 260), min(w, 260), grey=(i % 8 == 5))
        kw = dict(quality=int(rng.choice([20, 50, 75, 90, 100])), progressive=True)
        if im.mode == "RGB":
            kw["subsampling"] = i % 3
        buf = io.BytesIO()
        im.save(buf, "JPEG", **kw)
        files.append(buf.getvalue())
    for n, (d, g) in enumerate(zip(files, dj.decode_files(files))):
        assert np.array_equal(g.cpu().numpy(), _pillow(d)), n


@pytest.mark.parametrize("mode", ["420", "422", "444", "grey", "420p"])
def test_every_small_size(mode):
    """All sizes 1 x 1 ... 34 x 34 (every position of the image edge inside a block, an MCU and the colour kernel's 8 x 2 strips;
    row starts at all four byte alignments), one batch per sampling."""
    from viquae_amd.image import jpeg as dj
    rng = np.random.default_rng(len(mode))
    base = rng.integers(0, 256, (34, 34, 3), dtype=np.uint8)
    files = []
    for h in range(1, 35):
        for w in range(1, 35):
            im = Image.fromarray(base[:h, :w].copy(), "RGB")
            kw = dict(quality=92)
            if mode == "grey":
                im = im.convert("L")
            else:
                kw["subsampling"] = {"444": 0, "422": 1, "420": 2, "420p": 2}[mode]
            if mode == "420p":
                kw["progressive"] = True
            buf = io.BytesIO()
            im.save(buf, "JPEG", **kw)
            files.append(buf.getvalue())
    got = dj.decode_files(files)
    bad = [(n // 34 + 1, n % 34 + 1) for n, (d, g) in enumerate(zip(files, got)) if not np.array_equal(g.cpu().numpy(), _pillow(d))]
    assert not bad, bad[:10]
