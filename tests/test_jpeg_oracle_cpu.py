"""oracle/jpeg.py (the restatement of libjpeg's baseline decoding: Huffman scans, ISLOW IDCT, fancy upsampling, YCbCr tables)
against Pillow itself, bit for bit -- what ``load_image`` hands the image encoders (meerqat/data/loading.py:108-124).
Files are written by Pillow's encoder here; tools/jpeg_pillow_parity.py runs the same comparison over 1000 files."""
import io
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))

Image = pytest.importorskip("PIL.Image")
from oracle import jpeg as oj  # noqa: E402
import jpeg_pillow_parity as jp  # noqa: E402


def _pillow(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def test_random_files_decode_like_pillow():
    rng = np.random.default_rng(42)
    kinds = set()
    for i in range(40):
        h, w = jp.sizes(rng)
        h, w = min(h, 200), min(w, 200)
        data, kw = jp.encode(rng, jp.picture(rng, h, w, grey=(i % 9 == 0)))
        f = oj.read_coefficients(data)
        kinds.add(tuple((c["h"], c["v"]) for c in f["components"]))
        assert np.array_equal(oj.decode(data), _pillow(data)), (i, h, w, kw)
    assert {((1, 1),), ((1, 1),) * 3, ((2, 1), (1, 1), (1, 1)), ((2, 2), (1, 1), (1, 1))} <= kinds


@pytest.mark.parametrize("h,w", [(1, 1), (1, 7), (2, 2), (3, 5), (5, 3), (8, 8), (9, 17), (16, 16), (17, 33), (31, 2), (33, 4), (4, 6)])
@pytest.mark.parametrize("subsampling", [0, 1, 2])
def test_small_and_odd_sizes(h, w, subsampling):
    """Edge handling: components no wider than two samples take the replicating upsamplers; odd sizes end inside an MCU."""
    rng = np.random.default_rng(h * 100 + w)
    im = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB")
    buf = io.BytesIO()
    im.save(buf, "JPEG", quality=90, subsampling=subsampling)
    assert np.array_equal(oj.decode(buf.getvalue()), _pillow(buf.getvalue()))


def test_h1v2_sampling_file():
    """4:4:0 (luma 1 x 2): Pillow's encoder does not write it, but a 4:2:2 stream IS a 4:4:0 stream of another shape -- two luma
    blocks + Cb + Cr per MCU either way -- so the frame header of a 64 x 48 4:2:2 file (4 x 6 MCUs of 16 x 8) is rewritten to
    32 x 96 with luma 1 x 2 (4 x 6 MCUs of 8 x 16).  The picture is scrambled; both decoders read the same valid file."""
    rng = np.random.default_rng(3)
    buf = io.BytesIO()
    jp.picture(rng, 48, 64).save(buf, "JPEG", quality=85, subsampling=1)
    data = bytearray(buf.getvalue())
    at = data.find(b"\xff\xc0")
    assert at > 0 and data[at + 9] == 3 and data[at + 11] == 0x21
    data[at + 5:at + 9] = bytes([0, 96, 0, 32])   # height 96, width 32
    data[at + 11] = 0x12                          # luma h = 1, v = 2
    data = bytes(data)
    f = oj.read_coefficients(data)
    assert [(c["h"], c["v"]) for c in f["components"]] == [(1, 2), (1, 1), (1, 1)]
    assert np.array_equal(oj.decode(data), _pillow(data))


def test_restart_intervals_and_optimised_tables():
    rng = np.random.default_rng(5)
    im = jp.picture(rng, 70, 90)
    for kw in (dict(restart_marker_blocks=1), dict(restart_marker_rows=1), dict(optimize=True), dict(quality=100), dict(quality=1)):
        buf = io.BytesIO()
        im.save(buf, "JPEG", **kw)
        assert np.array_equal(oj.decode(buf.getvalue()), _pillow(buf.getvalue())), kw


def test_progressive_is_declined():
    buf = io.BytesIO()
    jp.picture(np.random.default_rng(0), 40, 40).save(buf, "JPEG", progressive=True)
    with pytest.raises(oj.Unsupported):
        oj.decode(buf.getvalue())


def test_idct_of_a_dc_only_block_is_flat():
    c = np.zeros((3, 8, 8), dtype=np.int64)
    c[:, 0, 0] = [-1024, 8, 1016]
    out = oj.idct_islow(c)
    assert [int(o[0, 0]) for o in out] == [0, 129, 255] and all((o == o[0, 0]).all() for o in out)


# ---- the host half of the split decoder (csrc/jpeg.hip: mq_jpeg_probe / mq_jpeg_read_coefficients; no GPU needed) ----------
def _staged(data):
    from viquae_amd.image import jpeg as dj
    p = dj.probe(data)
    assert p is not None
    st = np.zeros(p[4], dtype=np.uint8)
    assert dj.stage(data, st.ctypes.data, st.size)
    return p, st


def test_host_coefficients_equal_the_oracles():
    rng = np.random.default_rng(7)
    for i in range(40):
        h, w = jp.sizes(rng)
        data, kw = jp.encode(rng, jp.picture(rng, min(h, 200), min(w, 200), grey=(i % 7 == 0)))
        (hh, ww, ncomp, blocks, nbytes), st = _staged(data)
        f = oj.read_coefficients(data)
        hw = st[:128].view(np.int32)
        assert (hh, ww, ncomp) == (h if h <= 200 else 200, w if w <= 200 else 200, len(f["components"])) and hw[23] == blocks
        coef = st[512:512 + blocks * 128].view(np.int16).reshape(-1, 64)
        for c, comp in enumerate(f["components"]):
            first, bw, bh = hw[20 + c], hw[14 + c], hw[17 + c]
            assert np.array_equal(coef[first:first + bw * bh].reshape(bh, bw, 64), comp["coef"][:bh, :bw]), (i, c, kw)
            assert np.array_equal(st[128 + 128 * c:256 + 128 * c].view(np.uint16).astype(np.int64), f["qt"][comp["tq"]])
        # the oracle's arithmetic on the library's coefficients = Pillow's pixels
        assert nbytes % 16 == 0 and nbytes >= 512 + max(blocks * 128, hh * ww * 3)


def test_host_decoder_declines_what_it_does_not_cover():
    from viquae_amd.image import jpeg as dj
    im = jp.picture(np.random.default_rng(0), 40, 56)
    buf = io.BytesIO()
    im.convert("CMYK").save(buf, "JPEG")
    assert dj.probe(buf.getvalue()) is None
    buf = io.BytesIO()
    im.save(buf, "PNG")
    assert dj.probe(buf.getvalue()) is None
    assert dj.probe(b"") is None and dj.probe(b"\xff\xd8\xff") is None


def test_blocks_outside_the_16_bit_domain_are_left_to_pillow():
    """libjpeg-turbo's vector inverse DCT wraps / saturates in 16-bit lanes where the C code (and the device kernel) carry 32 bits:
    a block whose coefficients could leave 16 bits in the column pass (damaged data) is declined.  Made here by raising a
    quantisation table entry of a good file: the scan is untouched, the dequantised values are 255 x larger."""
    from viquae_amd.image import jpeg as dj
    buf = io.BytesIO()
    jp.picture(np.random.default_rng(4), 48, 48).save(buf, "JPEG", quality=50)
    data = bytearray(buf.getvalue())
    p = dj.probe(bytes(data))
    st = np.zeros(p[4], dtype=np.uint8)
    assert dj.stage(bytes(data), st.ctypes.data, st.size)
    at = data.find(b"\xff\xdb")
    data[at + 5] = 255   # the DC entry of the first table
    assert not dj.stage(bytes(data), st.ctypes.data, st.size)


def test_irregular_scans_are_left_to_pillow():
    """A truncated file, a missing end-of-image marker, a flipped bit that derails the Huffman stream, a marker in the middle of
    the scan: ``stage`` reports failure (the caller falls back to Pillow, whose behaviour is the reference's)."""
    from viquae_amd.image import jpeg as dj
    buf = io.BytesIO()
    jp.picture(np.random.default_rng(2), 64, 80).save(buf, "JPEG", quality=85)
    good = buf.getvalue()
    p = dj.probe(good)
    st = np.zeros(p[4], dtype=np.uint8)
    assert dj.stage(good, st.ctypes.data, st.size)
    for bad in (good[:len(good) // 2], good[:-2], good[:-2] + b"\xff\xd0", good[:-300] + b"\xff\xd9"):
        assert not dj.stage(bad, st.ctypes.data, st.size)
    derailed = 0
    for k in range(40):
        b = bytearray(good)
        b[len(good) - 400 + 9 * k] ^= 0x10
        if not dj.stage(bytes(b), st.ctypes.data, st.size):
            derailed += 1
        else:   # a flip that only changes some coefficients (Huffman streams resynchronise): a regular scan, the same pixels as Pillow's
            assert np.array_equal(oj.decode(bytes(b)), _pillow(bytes(b))), k
    assert 5 <= derailed < 40
    assert not dj.stage(good, st.ctypes.data, 100)   # a staging area that is too small


def test_progressive_files_through_the_host_decoder():
    """Progressive files (SOF2: DC / AC first and refinement scans, end-of-band runs, successive approximation): the library's
    final coefficients through the oracle's inverse DCT / upsampling / colour = Pillow's pixels."""
    from viquae_amd.image import jpeg as dj
    rng = np.random.default_rng(11)
    for i in range(36):
        h, w = jp.sizes(rng)
        im = jp.picture(rng, min(h, 160), min(w, 160), grey=(i % 9 == 4))
        kw = dict(quality=int(rng.choice([10, 30, 50, 75, 85, 95, 100])), progressive=True)
        if im.mode == "RGB":
            kw["subsampling"] = i % 3
        if i % 5 == 0:
            kw["restart_marker_blocks"] = 1 + i % 3
        buf = io.BytesIO()
        im.save(buf, "JPEG", **kw)
        data = buf.getvalue()
        assert b"\xff\xc2" in data
        _, st = _staged(data)
        assert np.array_equal(oj.decode_staging(st), _pillow(data)), (i, kw)
    # an INCOMPLETE progression (the last scans cut off, an end-of-image marker put in their place) decodes in Pillow -- with
    # libjpeg's inter-block smoothing -- and is declined here
    sos = [k for k in range(len(data) - 1) if data[k] == 0xFF and data[k + 1] == 0xDA]
    assert len(sos) >= 6
    cut = data[:sos[-2]] + b"\xff\xd9"
    p = dj.probe(cut)
    st = np.zeros(p[4], dtype=np.uint8)
    assert _pillow(cut).shape == _pillow(data).shape and not dj.stage(cut, st.ctypes.data, st.size)


def test_grey_file_with_sampling_factors_other_than_one():
    """A one-component frame is never interleaved: its scan codes ceil(W / 8) x ceil(H / 8) blocks whatever sampling factors the
    header gives the component (libjpeg; made here by rewriting the factors of a grey file to 2 x 2)."""
    rng = np.random.default_rng(0)
    for h, w in ((37, 53), (64, 64), (9, 100)):
        buf = io.BytesIO()
        jp.picture(rng, h, w, grey=True).save(buf, "JPEG", quality=80)
        d = bytearray(buf.getvalue())
        at = d.find(b"\xff\xc0")
        assert d[at + 9] == 1 and d[at + 11] == 0x11
        d[at + 11] = 0x22
        d = bytes(d)
        _, st = _staged(d)
        assert np.array_equal(oj.decode_staging(st), _pillow(d)) and np.array_equal(oj.decode(d), _pillow(d))


def test_fill_bytes_in_front_of_markers_are_legal():
    """Any number of 0xFF fill bytes may precede a marker (the restart markers inside a scan, the end-of-image marker): still a
    regular file, decoded here, same pixels."""
    from viquae_amd.image import jpeg as dj
    buf = io.BytesIO()
    jp.picture(np.random.default_rng(0), 50, 70).save(buf, "JPEG", quality=80, restart_marker_blocks=3)
    d = buf.getvalue()
    d2 = d[:-2] + b"\xff\xff\xff" + d[-2:]
    for k in range(8):
        d2 = d2.replace(bytes([0xFF, 0xD0 + k]), b"\xff\xff" + bytes([0xFF, 0xD0 + k]))
    assert len(d2) > len(d) + 6
    _, st = _staged(d2)
    assert np.array_equal(oj.decode_staging(st), _pillow(d2)) and np.array_equal(_pillow(d2), _pillow(d))
